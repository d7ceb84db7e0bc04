// Standalone timing harness for k16_fwd / k16_pairs (GPU box): compiles kwin16k.hpp alone, runs B buoys x W windows of
// N = 16384 synthetic samples (common random source, integer delay per buoy, noise) in chunks of C windows, checks every
// integer lag against the generator and prints ms per batch + fraction of the 8 TB/s algorithmic roofline.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false -Wno-inline-asm \
//          -I../../radio-mapper_amd/csrc -o k16_bench k16_bench.hip
//   run:   k16_bench [B=8] [W=256] [reps=50] [chunk=64] [pair grid per XCD S=32] [warm-up launches=20] [overlap: pair workgroups per XCD, 0 = off]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kwin16k.hpp"

using namespace rmx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__host__ __device__ inline unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ inline int delay_of(int w, int b) { return (int)(hash32(0x9e3779b9u * (unsigned)(w * 64 + b) + 12345u) % 201u) - 100; }
__device__ inline float unif(unsigned h) { return (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f; }

__global__ void k_gen(float2* iq, int B, int N) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = (int)(idx % N);
    const long wb = idx / N;
    const int b = (int)(wb % B), w = (int)(wb / B);
    const int m = n - delay_of(w, b);
    const unsigned hs = hash32((unsigned)w * 0x85ebca6bu + (unsigned)(m + 4096) * 0xc2b2ae35u + 1u);
    const unsigned hn = hash32((unsigned)idx * 0x27d4eb2fu + 77u);
    const float sr = unif(hs), si = unif(hash32(hs ^ 0xdeadbeefu));
    const float nr = unif(hn), ni = unif(hash32(hn ^ 0x1234567u));
    iq[idx] = make_float2(70.0f * sr + 25.0f * nr, 70.0f * si + 25.0f * ni);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, W = argc > 2 ? atoi(argv[2]) : 256, reps = argc > 3 ? atoi(argv[3]) : 50;
    int chunk = argc > 4 ? atoi(argv[4]) : 64;
    const int S = argc > 5 ? atoi(argv[5]) : 32;
    const int warm = argc > 6 ? atoi(argv[6]) : 20;      // (counter runs: 1)
    const int Sp = argc > 7 ? atoi(argv[7]) : 0;         // > 0: pairs(c) on Sp workgroups per XCD CONCURRENTLY with fwd(c + 1) on the other CUs (two streams)
    if (chunk > W) chunk = W;
    const int N = k16::kN16, P = B * (B - 1) / 2;
    float2* iq; float4 *spec, *tw1; float2 *tw2, *tws, *gq; int* li; float *lf, *pk; k16::Pair2* prs;
    CK(hipMalloc(&iq, (size_t)W * B * N * 8));
    CK(hipMalloc(&spec, (size_t)2 * chunk * B * 4 * k16::kQuarterBytes));      // (two chunks: the overlap mode double-buffers)
    CK(hipMalloc(&li, (size_t)W * P * 4)); CK(hipMalloc(&lf, (size_t)W * P * 4)); CK(hipMalloc(&pk, (size_t)W * P * 4));
    std::vector<float4> t1, t1_4096; std::vector<float2> t2, ts, tg;
    build_tables(t1_4096, t2);
    k16::build_tables16k(t1, tg, ts);
    CK(hipMalloc(&tw1, t1.size() * 16)); CK(hipMalloc(&tw2, t2.size() * 8)); CK(hipMalloc(&tws, ts.size() * 8)); CK(hipMalloc(&gq, tg.size() * 8));
    CK(hipMemcpy(gq, tg.data(), tg.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tw1, t1.data(), t1.size() * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(tw2, t2.data(), t2.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tws, ts.data(), ts.size() * 8, hipMemcpyHostToDevice));
    std::vector<k16::Pair2> hp;
    for (int i = 0; i < B; ++i) for (int j = i + 1; j < B; ++j) hp.push_back({i, j});
    CK(hipMalloc(&prs, hp.size() * 8));
    CK(hipMemcpy(prs, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gen, dim3((unsigned)((long)W * B * N / 256)), dim3(256), 0, 0, iq, B, N);
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute((const void*)k16::k16_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, k16::kLdsFwdBytes));
    CK(hipFuncSetAttribute((const void*)k16::k16_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, k16::kLdsPairBytes));
    const float out_scale = std::ldexp(1.0f, 3 * kTw1ScaleLog2 - 15);
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    float fwd_ms = 0.0f; int fwd_n = 0;
    auto launch = [&](bool time_fwd) {
        for (int w0 = 0; w0 < W; w0 += chunk) {
            const int wc = W - w0 < chunk ? W - w0 : chunk;
            const int items = wc * B;
            if (time_fwd) CK(hipEventRecord(f0));
            hipLaunchKernelGGL(k16::k16_fwd<false>, dim3(items < 256 ? items : 256), dim3(kThreads), k16::kLdsFwdBytes, 0, (const void*)iq,
                               spec, tw1, gq, tw2, tws, (long)w0 * B, items);
            if (time_fwd) { CK(hipEventRecord(f1)); CK(hipEventSynchronize(f1)); float ms; CK(hipEventElapsedTime(&ms, f0, f1)); fwd_ms += ms; ++fwd_n; }
            long per_xcd = (long)((wc + 7) / 8) * P;
            const int s = per_xcd < S ? (int)per_xcd : S;
            hipLaunchKernelGGL(k16::k16_pairs, dim3(8 * s), dim3(kThreads), k16::kLdsPairBytes, 0, spec, tw1, gq, tw2, tws, B, prs, P,
                               (long)w0 * P, wc, 0, out_scale, li, lf, pk);
        }
    };
    hipStream_t sF, sP;
    CK(hipStreamCreateWithFlags(&sF, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sP, hipStreamNonBlocking));
    const int nch = (W + chunk - 1) / chunk;
    std::vector<hipEvent_t> evF(nch), evP(nch);
    for (int c = 0; c < nch; ++c) { CK(hipEventCreateWithFlags(&evF[c], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evP[c], hipEventDisableTiming)); }
    hipEvent_t evDone;
    CK(hipEventCreateWithFlags(&evDone, hipEventDisableTiming));
    // overlap mode: fwd(0) on the whole chip; then pairs(c) on Sp workgroups per XCD beside fwd(c + 1) on the rest; the last pairs alone
    auto launch_overlap = [&]() {
        CK(hipStreamWaitEvent(sF, evDone, 0));                 // (the previous batch has read both buffers)
        for (int c = 0; c < nch; ++c) {
            const int w0 = c * chunk, wc = W - w0 < chunk ? W - w0 : chunk, items = wc * B;
            float4* buf = spec + (size_t)(c & 1) * chunk * B * 4 * (k16::kQuarterBytes / 16);
            if (c == 0) {
                hipLaunchKernelGGL(k16::k16_fwd<false>, dim3(items < 256 ? items : 256), dim3(kThreads), k16::kLdsFwdBytes, sF, (const void*)iq, buf,
                                   tw1, gq, tw2, tws, (long)w0 * B, items);
                CK(hipEventRecord(evF[0], sF));
            }
            CK(hipStreamWaitEvent(sP, evF[c], 0));
            const bool last = c + 1 == nch;
            long per_xcd = (long)((wc + 7) / 8) * P;
            const int sp = last ? S : Sp;
            const int s = per_xcd < sp ? (int)per_xcd : sp;
            hipLaunchKernelGGL(k16::k16_pairs, dim3(8 * s), dim3(kThreads), k16::kLdsPairBytes, sP, buf, tw1, gq, tw2, tws, B, prs, P,
                               (long)w0 * P, wc, 0, out_scale, li, lf, pk);
            CK(hipEventRecord(evP[c], sP));
            if (!last) {
                const int w1 = (c + 1) * chunk, wc1 = W - w1 < chunk ? W - w1 : chunk, items1 = wc1 * B;
                float4* buf1 = spec + (size_t)((c + 1) & 1) * chunk * B * 4 * (k16::kQuarterBytes / 16);
                if (c >= 1) CK(hipStreamWaitEvent(sF, evP[c - 1], 0));      // buffer (c + 1) & 1 was read by pairs(c - 1)
                int nfw = 256 - 8 * Sp;
                if (nfw > items1) nfw = items1;
                hipLaunchKernelGGL(k16::k16_fwd<false>, dim3(nfw), dim3(kThreads), k16::kLdsFwdBytes, sF, (const void*)iq, buf1,
                                   tw1, gq, tw2, tws, (long)w1 * B, items1);
                CK(hipEventRecord(evF[c + 1], sF));
            }
        }
        CK(hipEventRecord(evDone, sP));
    };
    if (Sp > 0) {
        CK(hipEventRecord(evDone, sP));
        CK(hipMemset(li, 0xff, (size_t)W * P * 4));
        for (int i = 0; i < warm; ++i) launch_overlap();
        CK(hipStreamSynchronize(sP)); CK(hipStreamSynchronize(sF));
        CK(hipGetLastError());
        hipEvent_t t0, t1;
        CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        CK(hipEventRecord(t0, sP));
        CK(hipStreamWaitEvent(sF, t0, 0));
        for (int i = 0; i < reps; ++i) launch_overlap();
        CK(hipStreamWaitEvent(sP, evDone, 0));
        CK(hipEventRecord(t1, sP));
        CK(hipEventSynchronize(t1));
        CK(hipStreamSynchronize(sF));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        ms /= reps;
        std::vector<int> h((size_t)W * P);
        CK(hipMemcpy(h.data(), li, h.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int w = 0; w < W; ++w) {
            int o = 0;
            for (int i = 0; i < B; ++i)
                for (int j = i + 1; j < B; ++j, ++o) bad += h[(size_t)w * P + o] != delay_of(w, j) - delay_of(w, i);
        }
        const double alg = (double)W * P * (16.0 * N + 12.0);
        printf("[overlap] k16 B=%d W=%d chunk=%d pair workgroups per XCD %d (forward %d)  %.4f ms  frac %.4f | lags != generator %ld of %ld\n", B, W, chunk, Sp,
               256 - 8 * Sp, ms, alg / (ms * 1e-3) / 8e12, bad, (long)W * P);
        fflush(stdout);
    }
    for (int round = 0; round < 2; ++round) {
        CK(hipMemset(li, 0xff, (size_t)W * P * 4));
        for (int i = 0; i < (round ? (warm < 5 ? warm : 5) : warm); ++i) launch(false);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(false);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        fwd_ms = 0.0f; fwd_n = 0;
        for (int i = 0; i < (warm < 5 ? 1 : 5); ++i) launch(true);
        CK(hipDeviceSynchronize());
        std::vector<int> h((size_t)W * P);
        CK(hipMemcpy(h.data(), li, h.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0; int shown = 0;
        for (int w = 0; w < W; ++w) {
            int o = 0;
            for (int i = 0; i < B; ++i)
                for (int j = i + 1; j < B; ++j, ++o) {
                    const int want = delay_of(w, j) - delay_of(w, i);
                    if (h[(size_t)w * P + o] != want) {
                        ++bad;
                        if (shown < 6) { printf("   w=%d pair (%d,%d): got %d want %d\n", w, i, j, h[(size_t)w * P + o], want); ++shown; }
                    }
                }
        }
        std::vector<float> hf((size_t)W * P), hpk((size_t)W * P);
        CK(hipMemcpy(hf.data(), lf, hf.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hpk.data(), pk, hpk.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long hsum = 1469598103934665603ULL;
        auto fnv = [&](const void* ptr, size_t nb) {
            const unsigned char* c = (const unsigned char*)ptr;
            for (size_t k = 0; k < nb; ++k) { hsum ^= c[k]; hsum *= 1099511628211ULL; }
        };
        fnv(h.data(), h.size() * 4); fnv(hf.data(), hf.size() * 4); fnv(hpk.data(), hpk.size() * 4);
        const double alg = (double)W * P * (16.0 * N + 12.0);
        printf("[%d] k16 B=%d W=%d chunk=%d S=%d  %.4f ms  frac %.4f  (forward kernels %.4f ms of it) | us per quarter transform and CU %.3f | "
               "lags != generator %ld of %ld | frac[0] %.5f peak[0] %.3f | fnv %016llx\n",
               round, B, W, chunk, S, ms, alg / (ms * 1e-3) / 8e12, fwd_ms / (warm < 5 ? 1.0f : 5.0f), ms * 1e3 * 256.0 / ((double)W * 4.0 * (B + P)), bad, (long)W * P,
               hf[0], hpk[0], hsum);
        fflush(stdout);
    }
    return 0;
}
