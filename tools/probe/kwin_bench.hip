// Standalone A/B harness for the fused window kernel (GPU box): compiles kwin.hpp (+ experimental variants) without the
// rest of the library (seconds instead of a minute), runs each variant on the cfg3 shape (4096 windows x 8 buoys x
// 4096 samples, synthetic: a common random source with integer delays per buoy + noise), checks every integer lag
// against the generator's delays and compares the three output arrays of every variant bit for bit with variant 0.
// Not a parity test (that is tests/ through the C ABI against the oracle): a timing instrument whose kernels must
// at least agree with each other.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false \
//        [-DKWB_VARIANTS=...] -o kwin_bench kwin_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../radio-mapper_amd/csrc/kwin.hpp"
#if !defined(KWB_NO_KWIN2)
#include "kwin2.hpp"
#define KWB_HAVE_KWIN2 1
#endif

using namespace rmx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__host__ __device__ inline unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ inline int delay_of(int w, int b) { return (int)(hash32(0x9e3779b9u * (unsigned)(w * 64 + b) + 12345u) % 201u) - 100; }
__device__ inline float unif(unsigned h) { return (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f; }

__global__ void k_gen(float2* iq, int B, int N) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over W*B*N
    const int n = (int)(idx % N);
    const long wb = idx / N;
    const int b = (int)(wb % B), w = (int)(wb / B);
    const int m = n - delay_of(w, b);                                // x_b[n] = s[n - d_b]
    const unsigned hs = hash32((unsigned)w * 0x85ebca6bu + (unsigned)(m + 4096) * 0xc2b2ae35u + 1u);
    const unsigned hn = hash32((unsigned)idx * 0x27d4eb2fu + 77u);
    const float sr = unif(hs), si = unif(hash32(hs ^ 0xdeadbeefu));
    const float nr = unif(hn), ni = unif(hash32(hn ^ 0x1234567u));
    iq[idx] = make_float2(70.0f * sr + 25.0f * nr, 70.0f * si + 25.0f * ni);
}

struct Bufs {
    float2* iq; float4* spec; float4* tw1; float2* tw2; int* li; float* lf; float* pk;
    int W, B; float out_scale;
};

typedef void (*launch_fn)(const Bufs&, hipStream_t);

static void launch_base(const Bufs& b, hipStream_t s) {
    hipLaunchKernelGGL(k_win<false>, dim3(256), dim3(kThreads), kLdsWinBytes, s, (const void*)b.iq, b.spec, b.tw1, b.tw2, b.B, 0L,
                       b.out_scale, b.li, b.lf, b.pk, b.W, 0, 1);
}
#define KWB_STAG(NAME, ST)                                                                                               \
    static void NAME(const Bufs& b, hipStream_t s) {                                                                    \
        hipLaunchKernelGGL(k_win<false>, dim3(256), dim3(kThreads), kLdsWinBytes, s, (const void*)b.iq, b.spec, b.tw1, b.tw2, \
                           b.B, 0L, b.out_scale, b.li, b.lf, b.pk, b.W, 0, ST);                                        \
    }
KWB_STAG(launch_stag2, 2)
KWB_STAG(launch_stag3, 3)
KWB_STAG(launch_stag4, 4)
static void launch_stag0(const Bufs& b, hipStream_t s) {
    hipLaunchKernelGGL(k_win<false>, dim3(256), dim3(kThreads), kLdsWinBytes, s, (const void*)b.iq, b.spec, b.tw1, b.tw2, b.B, 0L,
                       b.out_scale, b.li, b.lf, b.pk, b.W, 0, 0);
}
#ifdef RMX_ABLATE
#define KWB_ABL(NAME, DBG)                                                                                              \
    static void NAME(const Bufs& b, hipStream_t s) {                                                                    \
        hipLaunchKernelGGL(k_win<false>, dim3(256), dim3(kThreads), kLdsWinBytes, s, (const void*)b.iq, b.spec, b.tw1, b.tw2, \
                           b.B, 0L, b.out_scale, b.li, b.lf, b.pk, b.W, DBG, 1);                                       \
    }
KWB_ABL(launch_d128, 128)
KWB_ABL(launch_d64, 64)
KWB_ABL(launch_d192, 192)
KWB_ABL(launch_d2, 2)
KWB_ABL(launch_d4, 4)
KWB_ABL(launch_d8, 8)
KWB_ABL(launch_d12, 12)
KWB_ABL(launch_d206, 206)
KWB_ABL(launch_d256, 256)
KWB_ABL(launch_d512, 512)
#endif
#ifdef KWB_HAVE_KWIN2
#define KWB_V2(NAME, ...)                                                                                               \
    static void NAME(const Bufs& b, hipStream_t s) {                                                                    \
        hipLaunchKernelGGL((k_win2<false, __VA_ARGS__>), dim3(256), dim3(kThreads), kLdsWin2Bytes, s, (const void*)b.iq, \
                           b.spec, b.tw1, b.tw2, b.B, 0L, b.out_scale, b.li, b.lf, b.pk, b.W);                          \
    }
KWB_V2(launch_2a, 1)
KWB_V2(launch_2a_s0, 0)
KWB_V2(launch_2a_l, 1, 0, true)
KWB_V2(launch_a1, 1, 1)
KWB_V2(launch_a4, 1, 4)
KWB_V2(launch_a8, 1, 8)
KWB_V2(launch_a15, 1, 15)
#define KWB_KWIN2_TABLE \
    {"k_win2 two anchors, STAG 1", launch_2a, (const void*)k_win2<false, 1>, kLdsWin2Bytes}, \
    {"k_win2 two anchors, STAG 0", launch_2a_s0, (const void*)k_win2<false, 0>, kLdsWin2Bytes}, \
    {"k_win2 two anchors, STAG 1, TW2 row from LDS", launch_2a_l, (const void*)k_win2<false, 1, 0, true>, kLdsWin2Bytes}, \
    {"k_win2 ABL 1: no spectrum loads (phase 2)", launch_a1, (const void*)k_win2<false, 1, 1>, kLdsWin2Bytes}, \
    {"k_win2 ABL 4: no spectrum stores", launch_a4, (const void*)k_win2<false, 1, 4>, kLdsWin2Bytes}, \
    {"k_win2 ABL 8: no sample prefetch", launch_a8, (const void*)k_win2<false, 1, 8>, kLdsWin2Bytes}, \
    {"k_win2 ABL 15: all", launch_a15, (const void*)k_win2<false, 1, 15>, kLdsWin2Bytes},
#endif

struct Variant { const char* name; launch_fn fn; const void* kfn; int lds; };

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 4096, B = argc > 4 ? atoi(argv[4]) : 8, N = kM;
    const int reps = argc > 2 ? atoi(argv[2]) : 200, warm = argc > 3 ? atoi(argv[3]) : 150;
    const int P = B * (B - 1) / 2;
    Bufs b{};
    b.W = W; b.B = B;
    {
        int logl = 0;
        while ((1 << logl) < kL) ++logl;
        b.out_scale = std::ldexp(1.0f, 3 * kTw1ScaleLog2 - logl);
    }
    CK(hipMalloc(&b.iq, (size_t)W * B * N * 8));
    CK(hipMalloc(&b.spec, (size_t)256 * B * 8 * kThreads * 16));
    CK(hipMalloc(&b.li, (size_t)W * P * 4)); CK(hipMalloc(&b.lf, (size_t)W * P * 4)); CK(hipMalloc(&b.pk, (size_t)W * P * 4));
    std::vector<float4> tw1; std::vector<float2> tw2;
    build_tables(tw1, tw2);
    CK(hipMalloc(&b.tw1, tw1.size() * 16)); CK(hipMalloc(&b.tw2, tw2.size() * 8));
    CK(hipMemcpy(b.tw1, tw1.data(), tw1.size() * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.tw2, tw2.data(), tw2.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gen, dim3((unsigned)((long)W * B * N / 256)), dim3(256), 0, 0, b.iq, B, N);
    CK(hipDeviceSynchronize());

    std::vector<Variant> vars = {
        {"k_win stag=1 (production)", launch_base, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win stag=0", launch_stag0, (const void*)k_win<false>, kLdsWinBytes},
#ifdef KWB_ALL_STAGS
        {"k_win stag=2 (odd waves late)", launch_stag2, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win stag=3 (waves 4-7 late)", launch_stag3, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win stag=4 (one of every SIMD, alternating)", launch_stag4, (const void*)k_win<false>, kLdsWinBytes},
#endif
#ifdef KWB_HAVE_KWIN2
        KWB_KWIN2_TABLE
#endif
#ifdef RMX_ABLATE
        {"k_win dbg 128: no spectrum requests", launch_d128, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 64: no sample requests", launch_d64, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 192: neither", launch_d192, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 2: no peak search behind |r|^2", launch_d2, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 4: no wave-local exchange", launch_d4, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 8: no barrier exchange traffic", launch_d8, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 12: no LDS exchange traffic", launch_d12, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 206: none of the above", launch_d206, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 256: no resolve_batch", launch_d256, (const void*)k_win<false>, kLdsWinBytes},
        {"k_win dbg 512: no halo stores", launch_d512, (const void*)k_win<false>, kLdsWinBytes},
#endif
    };
    for (auto& v : vars) CK(hipFuncSetAttribute(v.kfn, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<int> li0, li((size_t)W * P);
    std::vector<float> lf0, pk0, lf((size_t)W * P), pk((size_t)W * P);
    const double alg_bytes = (double)W * P * (16.0 * N + 12.0);
    for (int round = 0; round < 2; ++round)
        for (size_t vi = 0; vi < vars.size(); ++vi) {
            auto& v = vars[vi];
            CK(hipMemset(b.li, 0xff, (size_t)W * P * 4)); CK(hipMemset(b.lf, 0xff, (size_t)W * P * 4)); CK(hipMemset(b.pk, 0xff, (size_t)W * P * 4));
            for (int i = 0; i < (round ? 20 : warm); ++i) v.fn(b, 0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) v.fn(b, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            CK(hipMemcpy(li.data(), b.li, li.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(lf.data(), b.lf, lf.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(pk.data(), b.pk, pk.size() * 4, hipMemcpyDeviceToHost));
            long bad_truth = 0, diff_li = 0, diff_lf = 0, diff_pk = 0;
            for (int w = 0; w < W; ++w) {
                int o = 0;
                for (int i = 0; i < B; ++i)
                    for (int j = i + 1; j < B; ++j, ++o)
                        if (li[(size_t)w * P + o] != delay_of(w, j) - delay_of(w, i)) ++bad_truth;
            }
            if (vi == 0 && round == 0) { li0 = li; lf0 = lf; pk0 = pk; }
            else
                for (size_t k = 0; k < li.size(); ++k) {
                    diff_li += li[k] != li0[k];
                    diff_lf += memcmp(&lf[k], &lf0[k], 4) != 0;
                    diff_pk += memcmp(&pk[k], &pk0[k], 4) != 0;
                }
#ifdef RMX_KWIN_STAMPS
            if (true) {
                std::vector<long long> st(256 * 64 * 4);
                CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(rmx_stamps), st.size() * 8));
                double p1 = 0, p2 = 0, tot = 0; int n = 0;
                for (int wg = 0; wg < 256; ++wg)
                    for (int k = 0; k + 1 < W / 256 && k + 1 < 64; ++k) {
                        const long long* a = &st[(wg * 64 + k) * 4];
                        p1 += (double)(a[1] - a[0]); p2 += (double)(a[2] - a[1]); tot += (double)(a[4] - a[0]); ++n;
                    }
                std::vector<int> vm(256 * 64 * 8);
                CK(hipMemcpyFromSymbol(vm.data(), HIP_SYMBOL(rmx_stamps_vm), vm.size() * 4));
                double wv[8] = {0};
                for (int wg = 0; wg < 256; ++wg) for (int k = 0; k + 1 < W / 256 && k + 1 < 64; ++k) for (int w8 = 0; w8 < 8; ++w8) wv[w8] += vm[(wg * 64 + k) * 8 + w8];
                printf("    vmcnt wait at the head of h1, ticks per window, waves 0..7:");
                for (int w8 = 0; w8 < 8; ++w8) printf(" %.0f", wv[w8] / n);
                printf("\n");
                std::vector<int> br(256 * 64 * 8 * 2);
                CK(hipMemcpyFromSymbol(br.data(), HIP_SYMBOL(rmx_stamps_bar), br.size() * 4));
                double dr[8] = {0}, bw[8] = {0};
                for (int wg = 0; wg < 256; ++wg) for (int k = 0; k + 1 < W / 256 && k + 1 < 64; ++k) for (int w8 = 0; w8 < 8; ++w8) {
                    dr[w8] += br[((wg * 64 + k) * 8 + w8) * 2]; bw[w8] += br[((wg * 64 + k) * 8 + w8) * 2 + 1]; }
                printf("    LDS drain before the barriers, ticks per window, waves 0..7:");
                for (int w8 = 0; w8 < 8; ++w8) printf(" %.0f", dr[w8] / n);
                printf("\n    waiting at the barriers, ticks per window, waves 0..7:        ");
                for (int w8 = 0; w8 < 8; ++w8) printf(" %.0f", bw[w8] / n);
                printf("\n");
                std::vector<int> pc(256 * 64 * 8 * 2);
                CK(hipMemcpyFromSymbol(pc.data(), HIP_SYMBOL(rmx_stamps_pc), pc.size() * 4));
                double c1[8] = {0}, c2[8] = {0};
                for (int wg = 0; wg < 256; ++wg) for (int k = 0; k + 1 < W / 256 && k + 1 < 64; ++k) for (int w8 = 0; w8 < 8; ++w8) {
                    c1[w8] += pc[((wg * 64 + k) * 8 + w8) * 2]; c2[w8] += pc[((wg * 64 + k) * 8 + w8) * 2 + 1]; }
                printf("    anchor loop, first piece of the interval, ticks per window, waves 0..7: ");
                for (int w8 = 0; w8 < 8; ++w8) printf(" %.0f", c1[w8] / n);
                printf("\n    anchor loop, second piece, ticks per window, waves 0..7:                ");
                for (int w8 = 0; w8 < 8; ++w8) printf(" %.0f", c2[w8] / n);
                printf("\n");
                std::vector<int> l1(256 * 64 * 8 * 8);
                CK(hipMemcpyFromSymbol(l1.data(), HIP_SYMBOL(rmx_stamps_p1), l1.size() * 4));
                static const char* lapname[8] = {"forward role A (+ load wait, cvt)", "forward barrier", "forward roles B + C", "spectrum store",
                                                 "h1 of (0, e)", "pair barrier", "h2 of (0, e)", "vmcnt wait at the head of forward"};
                for (int k7 = 0; k7 < 8; ++k7) {
                    double a8[8] = {0};
                    for (int wg = 0; wg < 256; ++wg) for (int k = 0; k + 1 < W / 256 && k + 1 < 64; ++k) for (int w8 = 0; w8 < 8; ++w8)
                        a8[w8] += l1[((wg * 64 + k) * 8 + w8) * 8 + k7];
                    printf("    phase 1, %-34s ticks per window, waves 0..7:", lapname[k7]);
                    for (int w8 = 0; w8 < 8; ++w8) printf(" %.0f", a8[w8] / n);
                    printf("\n");
                }
                printf("    stamps (s_memtime ticks, mean over %d windows): phase 1 %.0f  phase 2 %.0f  whole window %.0f\n", n, p1 / n, p2 / n, tot / n);
            }
#endif
            // FNV-1a over the three output arrays: binaries built with different -D switches are compared through it
            unsigned long long hsum = 1469598103934665603ULL;
            auto fnv = [&](const void* ptr, size_t nb) {
                const unsigned char* c = (const unsigned char*)ptr;
                for (size_t k = 0; k < nb; ++k) { hsum ^= c[k]; hsum *= 1099511628211ULL; }
            };
            fnv(li.data(), li.size() * 4); fnv(lf.data(), lf.size() * 4); fnv(pk.data(), pk.size() * 4);
            printf("[%d] %-44s %.4f ms  frac %.4f | lags != generator %ld, vs variant 0: lag_int %ld lag_frac %ld peak %ld differ | fnv %016llx\n", round,
                   v.name, ms, alg_bytes / (ms * 1e-3) / 8e12, bad_truth, diff_li, diff_lf, diff_pk, hsum);
            fflush(stdout);
        }
    return 0;
}
