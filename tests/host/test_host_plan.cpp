// Host-side logic of the C ABI (radio-mapper_amd/csrc/host_plan.hpp) under AddressSanitizer + UBSan.
// Built and run by tests/test_host_plan_sanitized.py:  g++ -std=c++17 -O1 -g -fsanitize=address,undefined
// -fno-sanitize-recover=all -o ... tests/host/test_host_plan.cpp ; exit code 0 = every check passed and neither
// sanitizer reported anything.
#include <cassert>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <thread>

#include "../../radio-mapper_amd/csrc/host_plan.hpp"

using namespace rmx;
using namespace rmx::host;

#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #x, __LINE__); std::exit(1); } } while (0)

static void test_options() {
    std::string why;
    clear_default_options();
    CHECK(set_default_option("wscr", 2, &why) == 0);
    CHECK(set_default_option("wscr", 3, &why) == -1 && why.find("not in 0..2") != std::string::npos);
    CHECK(set_default_option("no_such_key", 1, &why) == -1 && why.find("unknown option") != std::string::npos);
    CHECK(set_default_option(nullptr, 1, &why) == -1);
    CHECK(set_default_option("stag", 5, &why) == 0 && set_default_option("stag", 6, &why) == -1);
    Knobs snap = snapshot_default_options();
    long v = -1;
    CHECK(snap.get("wscr", &v) && v == 2 && snap.get_or("stag", 1) == 5 && snap.get_or("fused", 7) == 7);
    CHECK(set_default_option("wscr", LONG_MIN, &why) == 0);                 // removes the key ...
    CHECK(!snapshot_default_options().get("wscr", &v) && snap.get("wscr", &v));   // ... but not from an earlier snapshot
    // concurrent setters and snapshots (the store is shared by every thread that creates engines)
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
        th.emplace_back([t] {
            std::string w;
            for (int i = 0; i < 2000; ++i) {
                set_default_option(t & 1 ? "fused" : "wfused", i & 1, &w);
                Knobs k = snapshot_default_options();
                (void)k.get_or("fused", 0);
            }
        });
    for (auto& x : th) x.join();
    clear_default_options();
    CHECK(snapshot_default_options().v.empty());
    for (const OptionSpec& s : option_specs()) CHECK(s.lo <= s.hi && s.key && s.what);
}

static void test_create_args() {
    std::string why;
    CHECK(check_create_args(8, 4096, 1, &why) == 0);
    CHECK(check_create_args(1, 4096, 1, &why) == -1 && why.find("n_buoys") != std::string::npos);
    CHECK(check_create_args(4097, 4096, 1, &why) == -1);
    CHECK(check_create_args(8, 4095, 1, &why) == -1 && check_create_args(8, 8, 1, &why) == -1);
    CHECK(check_create_args(8, 1 << 22, 1, &why) == 0 && check_create_args(8, 1 << 23, 1, &why) == -1);
    CHECK(check_create_args(8, 4096, 0, &why) == -1 && check_create_args(8, INT_MIN, 1, &why) == -1);
    CHECK(is_pow2(1) && is_pow2(1L << 40) && !is_pow2(0) && !is_pow2(-8) && !is_pow2(12));
}

static void test_pair_plan() {
    std::string why;
    for (int B = 2; B <= 33; ++B) {                                          // default plan of every buoy count
        for (int ppb : {1, 7, 0, 1000}) {
            PairPlan p;
            const int P = B * (B - 1) / 2;
            CHECK(make_pair_plan(B, nullptr, P, ppb, &p, &why) == 0);
            CHECK(p.all_pairs && (int)p.items.size() == P && (int)p.pairs.size() == 2 * P);
            CHECK(p.part_begin.front() == 0 && p.part_begin.back() == P && (int)p.part_begin.size() == p.n_parts + 1);
            const int eff = ppb > 0 ? ppb : 7;
            for (int k = 0; k < p.n_parts; ++k) CHECK(p.part_begin[k + 1] > p.part_begin[k] && p.part_begin[k + 1] - p.part_begin[k] <= eff);
            int q = 0;
            for (int i = 0; i < B; ++i)
                for (int j = i + 1; j < B; ++j, ++q) {
                    CHECK(p.items[q].i == i && p.items[q].j == j && p.items[q].out == q);
                    CHECK(p.items[q].run == B - j);                              // pairs left in anchor i's run
                }
        }
    }
    PairPlan p;
    CHECK(make_pair_plan(8, nullptr, 27, 7, &p, &why) == -1 && why.find("pairs == NULL") != std::string::npos);
    CHECK(make_pair_plan(8, nullptr, -1, 7, &p, &why) == -1);
    const int32_t bad1[] = {0, 1, 2, 8}, bad2[] = {-1, 0}, ok0[] = {7, 0, 0, 1, 1, 1, 0, 1};
    CHECK(make_pair_plan(8, bad1, 2, 7, &p, &why) == -1 && why.find("pair 1 = (2,8)") != std::string::npos);
    CHECK(make_pair_plan(8, bad2, 1, 7, &p, &why) == -1);
    CHECK(make_pair_plan(8, ok0, 4, 7, &p, &why) == 0 && !p.all_pairs && p.n_parts == 1);   // reversed, repeated, autocorrelation
    CHECK(p.items[0].run == 1 && p.items[1].run == 1 && p.items[2].run == 1 && p.items[3].run == 1);
    CHECK(make_pair_plan(8, ok0, 0, 7, &p, &why) == 0 && p.n_parts == 0 && p.items.empty() && p.part_begin.size() == 1);
    // the default list passed explicitly is recognised as the default plan; one swap is not
    std::vector<int32_t> pl;
    for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) { pl.push_back(i); pl.push_back(j); }
    CHECK(make_pair_plan(5, pl.data(), 10, 7, &p, &why) == 0 && p.all_pairs);
    std::swap(pl[2], pl[3]);
    CHECK(make_pair_plan(5, pl.data(), 10, 7, &p, &why) == 0 && !p.all_pairs);
    // random lists: runs are consistent, parts cover everything once
    std::mt19937 rng(7);
    for (int trial = 0; trial < 200; ++trial) {
        const int B = 2 + rng() % 30, P = rng() % 200, ppb = 1 + rng() % 12;
        std::vector<int32_t> r(2 * (size_t)P + 2);                             // (+2: a non-NULL pointer also for P = 0)
        for (auto& x : r) x = (int32_t)(rng() % B);
        CHECK(make_pair_plan(B, r.data(), P, ppb, &p, &why) == 0);
        for (int q = 0; q + 1 < P; ++q)
            CHECK(p.items[q].run == (p.items[q].i == p.items[q + 1].i ? p.items[q + 1].run + 1 : 1));
        if (P) CHECK(p.items[P - 1].run == 1 && p.part_begin.back() == P);
    }
}

static void test_chunks() {
    // cfg2: 3 buoys, L = 2^21, 64 windows -> one chunk; the cap and the 4096 limit; never below one window
    {   // the N = 4096 dispatch model: the measured crossovers of tools/exp_small4096.py, and no way to break it
        int q = 0;
        CHECK(split_cost4096(256, 8, 28, 1, 0, &q) < fused_cost4096(256, 8, 28, 1) && q == 1);        // one group: per-transform, 1 pair per block
        CHECK(split_cost4096(256, 16, 120, 1, 0, &q) < fused_cost4096(256, 16, 120, 1) && q == 1);
        CHECK(split_cost4096(256, 8, 28, 64, 0, &q) < fused_cost4096(256, 8, 28, 64) && q == 7);      // 64 windows: 7 pairs per block
        CHECK(split_cost4096(256, 8, 28, 256, 0, &q) > fused_cost4096(256, 8, 28, 256));              // a full round: fused
        CHECK(split_cost4096(256, 3, 3, 128, 0, &q) > fused_cost4096(256, 3, 3, 128));
        CHECK(split_cost4096(256, 3, 3, 32, 0, &q) < fused_cost4096(256, 3, 3, 32));
        CHECK(split_cost4096(256, 8, 28, 4, 5, &q) > 0 && q == 5);                                    // the caller's block size is kept
        for (int cus : {0, 1, 7, 256, 4096})
            for (long w : {0L, 1L, 255L, 256L, 257L, 1L << 20})
                for (int b : {2, 3, 64})
                    for (int fixed : {0, 1, 1 << 20}) {
                        const int pairs = b * (b - 1) / 2;
                        const double t = split_cost4096(cus, b, pairs, w, fixed, &q), f = fused_cost4096(cus, b, pairs, w);
                        CHECK(t > 0 && t < 1e30 && f > 0 && q >= 1 && (fixed ? q == fixed : q <= 7));
                    }
        CHECK(split_cost4096(256, 4, 0, 3, 0, &q) > 0 && q >= 1);                                     // an empty pair list divides by nothing
    }
    CHECK(generic_chunk_windows(3, 1L << 21, 64, 32L << 30, 0) == 64);
    CHECK(generic_chunk_windows(3, 1L << 21, 64, 32L << 30, 16) == 16);
    CHECK(generic_chunk_windows(32, 1L << 19, 64, 32L << 30, 0) == (32L << 30) / ((32 + 496) * (1L << 19) * 8));
    CHECK(generic_chunk_windows(8, 512, 1 << 20, 32L << 30, 0) == 4096);
    CHECK(generic_chunk_windows(4096, 1L << 23, 5, 1 << 20, 0) == 1);
    CHECK(generic_chunk_windows(2, 32, 1, 32L << 30, 4096) == 1);
}

int main() {
    test_options();
    test_create_args();
    test_pair_plan();
    test_chunks();
    std::puts("host_plan: all checks passed");
    return 0;
}
