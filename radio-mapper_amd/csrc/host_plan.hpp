// host_plan.hpp -- the host-only logic of the C ABI: kernel-selection options, argument checks, the pair plan and the
// chunk arithmetic.  No HIP in here: rmx_hip.hip includes it, and tests/host/test_host_plan.cpp compiles it with
// g++ -fsanitize=address,undefined (tests/test_host_plan_sanitized.py), which is the sanitizer target SURVEY.md
// section 5 asks for on the CPU side of the boundary.
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace rmx {

// one pair of the plan, in the order results are written: run = pairs left in this anchor run (same i), this one included
struct PairItem {
    int i, j, out, run;
};

namespace host {

// ---- kernel-selection options ------------------------------------------------------------------------------------
// Process-wide defaults that rmx_create() copies into the new ctx (a ctx never sees later changes: the kernels, block
// sizes and LDS sizes it derived from them at creation stay consistent).  Set through rmx_set_default_option(); the
// library itself never reads the environment.  They exist for tests and A/B measurements (tools/README.md lists them).
struct OptionSpec {
    const char* key;
    long lo, hi;
    const char* what;
};
inline const std::vector<OptionSpec>& option_specs() {
    static const std::vector<OptionSpec> s = {
        {"stag", 0, 5, "N = 4096 fused kernel: which waves run the two halves between barriers in the opposite order (default 1)"},
        {"ncus", 1, 4096, "persistent workgroups of the fused N = 4096 kernel (default: one per CU)"},
        {"chunk_windows", 1, 1 << 20, "windows per chunk of the unfused N = 4096 path"},
        {"small4096", 0, 1, "0: N = 4096 through the fused kernel however few the windows (default 1: few windows through the per-transform kernels)"},
        {"generic4096", 0, 1, "1: N = 4096 through the generic whole-window kernels instead of k_win"},
        {"small_maxl", 256, 16384, "largest zero-padded length that runs as one LDS transform per workgroup (default 8192)"},
        {"logl1", 4, 10, "four-step: log2 of the column length"},
        {"wfused", 0, 1, "0: never g_win_fused (<= 4 buoys, N = 256 ... 2048)"},
        {"wscr", 0, 2, "g_win_scr: 0 never, 2 also for batches that do not fill the chip"},
        {"wscr14", 0, 1, "0: N = 8192 with 1024 threads x one butterfly instead of 512 x two"},
        {"wscr_per_cu", 1, 8, "persistent g_win_scr workgroups per CU"},
        {"rows_anchor", 0, 64, "four-step inverse rows: 0 never g_rows_anchor, n = from n buoys on (default 6)"},
        {"fused", 0, 2, "four-step, <= 4 buoys: 0 always the two-kernel row passes, 2 g_rows_fused for any batch size"},
        {"fused_def", 0, 1, "0: the run-time pair loop of g_rows_fused also for the default plan"},
        {"gen_chunk", 1, 4096, "generic path: at most n windows per chunk of kernel launches"},
        {"rows_tpr", 1, 1024, "four-step: threads per row (selects the run-time-length row kernels)"},
        {"cols_threads", 64, 1024, "four-step: threads per column tile"},
        {"col_logt", 3, 5, "four-step: log2 columns per tile"},
        {"kwin16k", 0, 2, "N = 16384: 1 (default) k16_fwd + k16_pairs (four quarter transforms on the fused N = 4096 network) unless wscr = 2, 2 always, 0 g_win_eo15 / the four-step kernels"},
        {"k16_min_windows", 1, 1 << 30, "kwin16k = 1: smallest batch that takes k16_fwd / k16_pairs (default: 100 / (buoys + pairs), rounded up)"},
        {"kwin8k", 0, 2, "N = 8192, batches that fill the chip: 1 (default) k_win8kl (the bin-parity halves on the fused N = 4096 network, one anchor half in LDS), 0 g_win_scr14, 2 k_win8k (-DRMX_EXPERIMENTS builds only)"},
#ifdef RMX_EXPERIMENTS
        {"pk", 0, 1, "1: k_winp (packed fp32 build of the fused kernel; -DRMX_EXPERIMENTS builds only)"},
        {"win8", 0, 1, "1: k_win8 (8 points x 1024 threads; -DRMX_EXPERIMENTS builds only)"},
#endif
    };
    return s;
}

struct Knobs {
    std::map<std::string, long> v;
    bool get(const char* key, long* out) const {
        auto it = v.find(key);
        if (it == v.end()) return false;
        *out = it->second;
        return true;
    }
    long get_or(const char* key, long dflt) const {
        long x;
        return get(key, &x) ? x : dflt;
    }
};
inline std::mutex& knob_mutex() { static std::mutex m; return m; }
inline Knobs& default_knobs() { static Knobs k; return k; }

// 0, or -1 with *err set (unknown key / value out of range).  value == LONG_MIN removes the key.
inline int set_default_option(const char* key, long value, std::string* err) {
    if (!key) { if (err) *err = "NULL key"; return -1; }
    for (const OptionSpec& s : option_specs()) {
        if (std::strcmp(s.key, key) != 0) continue;
        std::lock_guard<std::mutex> g(knob_mutex());
        if (value == LONG_MIN) { default_knobs().v.erase(key); return 0; }
        if (value < s.lo || value > s.hi) {
            if (err) { char b[160]; std::snprintf(b, sizeof b, "option '%s': %ld not in %ld..%ld", key, value, s.lo, s.hi); *err = b; }
            return -1;
        }
        default_knobs().v[key] = value;
        return 0;
    }
    if (err) *err = std::string("unknown option '") + key + "'";
    return -1;
}
inline void clear_default_options() {
    std::lock_guard<std::mutex> g(knob_mutex());
    default_knobs().v.clear();
}
inline Knobs snapshot_default_options() {
    std::lock_guard<std::mutex> g(knob_mutex());
    return default_knobs();
}

// ---- argument checks ---------------------------------------------------------------------------------------------
inline bool is_pow2(long n) { return n > 0 && (n & (n - 1)) == 0; }

// rmx_create: 0 or -1 with *err
inline int check_create_args(int n_buoys, int n_samples, int max_windows, std::string* err) {
    char b[160];
    if (n_buoys < 2 || n_buoys > 4096) { std::snprintf(b, sizeof b, "n_buoys %d not in 2..4096", n_buoys); *err = b; return -1; }
    if (!is_pow2(n_samples) || n_samples < 16 || n_samples > (1 << 22)) {
        std::snprintf(b, sizeof b, "n_samples %d must be a power of two in 16..4194304", n_samples); *err = b; return -1;
    }
    if (max_windows < 1) { std::snprintf(b, sizeof b, "max_windows %d < 1", max_windows); *err = b; return -1; }
    return 0;
}

// ---- the pair plan ------------------------------------------------------------------------------------------------
struct PairPlan {
    std::vector<int32_t> pairs;     // [P][2]
    std::vector<PairItem> items;    // [P]
    std::vector<int> part_begin;    // [n_parts + 1]
    int n_parts = 0;
    bool all_pairs = false;         // the default list: all i < j in nested-loop order (tdoa_processor.py:156-157)
};

// pairs == nullptr: the default list (n_pairs must then be P = B(B-1)/2).  0, or -1 with *err.
inline int make_pair_plan(int n_buoys, const int32_t* pairs, int n_pairs, int pairs_per_block, PairPlan* out, std::string* err) {
    char b[200];
    const long all = (long)n_buoys * (n_buoys - 1) / 2;
    if (n_pairs < 0) { std::snprintf(b, sizeof b, "n_pairs %d < 0", n_pairs); *err = b; return -1; }
    std::vector<int32_t> pl;
    if (pairs) {
        pl.assign(pairs, pairs + 2 * (size_t)n_pairs);
    } else {
        if (n_pairs != all) { std::snprintf(b, sizeof b, "pairs == NULL needs n_pairs == %ld, got %d", all, n_pairs); *err = b; return -1; }
        pl.reserve(2 * (size_t)all);
        for (int i = 0; i < n_buoys; ++i)
            for (int j = i + 1; j < n_buoys; ++j) { pl.push_back(i); pl.push_back(j); }
    }
    for (int q = 0; q < n_pairs; ++q) {
        const int i = pl[2 * q], j = pl[2 * q + 1];
        if (i < 0 || j < 0 || i >= n_buoys || j >= n_buoys) {
            std::snprintf(b, sizeof b, "pair %d = (%d,%d) out of range for %d buoys", q, i, j, n_buoys);
            *err = b;
            return -1;
        }
    }
    out->items.resize(n_pairs);
    for (int q = 0; q < n_pairs; ++q) out->items[q] = PairItem{pl[2 * q], pl[2 * q + 1], q, 1};
    for (int q = n_pairs - 2; q >= 0; --q)
        if (out->items[q].i == out->items[q + 1].i) out->items[q].run = out->items[q + 1].run + 1;
    const int ppb = pairs_per_block > 0 ? pairs_per_block : 7;
    const int n_parts = n_pairs > 0 ? (n_pairs + ppb - 1) / ppb : 0;
    out->part_begin.assign(n_parts + 1, 0);
    for (int k = 0; k <= n_parts && n_parts > 0; ++k) out->part_begin[k] = (int)((long)k * n_pairs / n_parts);
    out->n_parts = n_parts;
    bool def = n_pairs == all;
    int q = 0;
    for (int i = 0; def && i < n_buoys; ++i)
        for (int j = i + 1; j < n_buoys; ++j, ++q)
            if (pl[2 * q] != i || pl[2 * q + 1] != j) { def = false; break; }
    out->all_pairs = def;
    out->pairs.swap(pl);
    return 0;
}

// ---- chunk arithmetic of the generic path ----------------------------------------------------------------------------
// windows per chunk of kernel launches: spectra (B*L) + products (P*L), 8 bytes each, under `budget_bytes` (32 GiB of
// the 288: cfg2's 64 windows of 2^20 samples are one chunk of 6.4 GB), at most 4096 and at most max_windows
inline long generic_chunk_windows(int n_buoys, long L, int max_windows, long budget_bytes, long cap) {
    const long all_pairs = (long)n_buoys * (n_buoys - 1) / 2;
    const long per_win = (long)(n_buoys + all_pairs) * L * 8;
    long chunk = per_win > 0 ? budget_bytes / per_win : 1;
    if (chunk < 1) chunk = 1;
    if (chunk > max_windows) chunk = max_windows;
    if (chunk > 4096) chunk = 4096;
    if (cap >= 1 && cap < chunk) chunk = cap;
    return chunk;
}

// ---- N = 4096: fused kernel or per-transform kernels, and how many pairs per workgroup -----------------------------
// Estimated microseconds, fitted to tools/exp_small4096.py on MI355X: 2.6 per transform of the
// fused kernel (one workgroup per window, a round = one window on every CU); 3.5 per round of forward workgroups (two per
// CU); 4.0 per pair workgroup + 3.3 per pair in it (one per CU); a round that fills the chip runs up to 45 % slower than
// a lone workgroup, less so when many rounds follow each other out of step.
inline double fused_cost4096(int n_cus, int n_buoys, int n_pairs, long n_windows) {
    const long cus = n_cus > 0 ? n_cus : 1;
    return (double)((n_windows + cus - 1) / cus) * (n_buoys + n_pairs) * 2.6 + 3.0;
}
// the per-transform estimate and the pairs per workgroup (1 ... 7, or `fixed_ppb` when the caller set them) that minimise it
inline double split_cost4096(int n_cus, int n_buoys, int n_pairs, long n_windows, int fixed_ppb, int* ppb) {
    const long cus = n_cus > 0 ? n_cus : 1;
    auto rounds = [](long blocks, long per_round) { return (double)((blocks + per_round - 1) / per_round); };
    const double t_fwd = rounds(n_windows * n_buoys, 2 * cus) * 3.5 + 2.0;
    double best = 1e30;
    *ppb = fixed_ppb > 0 ? fixed_ppb : 7;
    for (int q = 7; q >= 1; --q) {
        const int qq = fixed_ppb > 0 ? fixed_ppb : q;
        const long parts = n_pairs > 0 ? (n_pairs + qq - 1) / qq : 1;
        const long blocks = n_windows * parts;
        const double r = blocks > 0 ? rounds(blocks, cus) : 1.0;
        const double fill = blocks >= cus ? 1.0 : (double)blocks / (double)cus;
        const double t = (t_fwd + r * (4.0 + 3.3 * (double)((n_pairs + parts - 1) / parts))) * (1.15 + 0.3 * fill / std::sqrt(r));
        if (t < best) { best = t; *ppb = qq; }
        if (fixed_ppb > 0) break;
    }
    return best;
}

}  // namespace host
}  // namespace rmx
