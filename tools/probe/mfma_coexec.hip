// mfma_coexec.hip -- can the SIMD's matrix pipe take one radix-16 pass off k_win's VALU?  (VERDICT r03 item 1, step a)
//
// k_win is bound by fp32 VALU issue at two waves per SIMD.  A 16-point complex DFT is a real
// 32 x 32 matrix; on 64 columns per wave that is 65 536 MACs = 32 v_mfma_f32_32x32x1_2b_f32 / 32x32x2_f32 (64 cycles
// each) or 64 16x16x4_f32 / 16x16x1_4b_f32 (32 cycles each) = 2048 cycles of matrix pipe per wave and pass, against
// ~150-200 VALU instructions of the butterfly network.  The question is what the PARTNER wave's VALU stream gets while
// a wave sits in such a pass, and what a wave pays for having the pass in its own stream.
//
// One workgroup of 512 threads per CU (waves w and w + 4 share SIMD w); every wave stamps its own start / end with
// s_memtime; the host prints the kernel's wall time per iteration and the cycles per iteration of waves 0 and 4.
//   mode 0  VALU only, NV instructions per iteration in both waves (NV = 780: a k_win transform; 640: one pass gone)
//   mode 1  MFMA only, waves 0-3 (waves 4-7 leave at once): the pass alone
//   mode 2  MFMA only, all eight waves: two passes competing for one pipe
//   mode 3  waves 0-3 MFMA only, waves 4-7 VALU only (NV per iteration): what each keeps beside the other
//   mode 4  both waves: the pass spread evenly through the VALU (one MFMA per NV/NP instructions)
//   mode 5  both waves: [pass][NV VALU] per iteration, waves 4-7 in the opposite order (anti-phase)
//   mode 6  both waves: [pass][NV VALU], same order (in phase)
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_coexec mfma_coexec.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));

// ---- VALU filler: the butterflies' mix (fmac, fma with 2.0, add, sub) on eight independent accumulators ----
template <int N>
__device__ __forceinline__ void valu(float (&a)[8], float b, float c) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int r = i & 7;
        if ((i & 3) == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
        if ((i & 3) == 1) asm volatile("v_fma_f32 %0, %0, 2.0, -%1" : "+v"(a[r]) : "v"(c));
        if ((i & 3) == 2) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[r]) : "v"(c));
        if ((i & 3) == 3) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(c));
    }
}

// ---- one DFT16 pass worth of matrix work per wave (65 536 MACs), in NP<SHAPE> instructions, issued G at a time ----
template <int SHAPE> struct Pass;
template <> struct Pass<0> {   // v_mfma_f32_32x32x1_2b_f32: 32 K-steps into one 32-register accumulator
    static constexpr int NP = 32;
    static constexpr const char* name = "32x32x1_2b";
    v32f acc;
    __device__ void init(float s) { for (int i = 0; i < 32; ++i) acc[i] = s + i; }
    __device__ __forceinline__ void step(int i, const float (&d)[32], const float (&f)[32]) {
        acc = __builtin_amdgcn_mfma_f32_32x32x1f32(f[i & 31], d[i & 31], acc, 0, 0, 0);
    }
    __device__ float sum() { float s = 0; for (int i = 0; i < 32; ++i) s += acc[i]; return s; }
};
template <> struct Pass<1> {   // v_mfma_f32_32x32x2_f32: 16 K-steps x 2 column blocks
    static constexpr int NP = 32;
    static constexpr const char* name = "32x32x2";
    v16f acc[2];
    __device__ void init(float s) { for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = s + i + j; }
    __device__ __forceinline__ void step(int i, const float (&d)[32], const float (&f)[32]) {
        const int j = i & 1;   // alternate the two accumulators
        if (j == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[(i >> 1) & 15], d[i & 31], acc[0], 0, 0, 0);
        else acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[(i >> 1) & 15], d[i & 31], acc[1], 0, 0, 0);
    }
    __device__ float sum() { float s = 0; for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i]; return s; }
};
template <> struct Pass<2> {   // v_mfma_f32_16x16x4_f32: 8 K-steps x (4 column blocks x re / im)
    static constexpr int NP = 64;
    static constexpr const char* name = "16x16x4";
    v4f acc[8];
    __device__ void init(float s) { for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = s + i + j; }
    __device__ __forceinline__ void step(int i, const float (&d)[32], const float (&f)[32]) {
#define RMX_S2(J) if ((i & 7) == J) acc[J] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[i & 31], f[(i >> 2) & 15], acc[J], 0, 0, 0);
        RMX_S2(0) RMX_S2(1) RMX_S2(2) RMX_S2(3) RMX_S2(4) RMX_S2(5) RMX_S2(6) RMX_S2(7)
#undef RMX_S2
    }
    __device__ float sum() { float s = 0; for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i]; return s; }
};
template <> struct Pass<3> {   // v_mfma_f32_16x16x1_4b_f32: 32 K-steps x 2 row blocks
    static constexpr int NP = 64;
    static constexpr const char* name = "16x16x1_4b";
    v16f acc[2];
    __device__ void init(float s) { for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = s + i + j; }
    __device__ __forceinline__ void step(int i, const float (&d)[32], const float (&f)[32]) {
        if ((i & 1) == 0) acc[0] = __builtin_amdgcn_mfma_f32_16x16x1f32(f[(i >> 1) & 31], d[(i >> 1) & 31], acc[0], 0, 0, 0);
        else acc[1] = __builtin_amdgcn_mfma_f32_16x16x1f32(f[(i >> 1) & 31], d[(i >> 1) & 31], acc[1], 0, 0, 0);
    }
    __device__ float sum() { float s = 0; for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i]; return s; }
};

template <> struct Pass<4> {   // contrast: v_mfma_f32_32x32x16_bf16 (a real second pipe), 64 of them = the same 2048 cycles
    static constexpr int NP = 64;
    static constexpr const char* name = "bf16 32x32x16";
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    v16f acc[2];
    __device__ void init(float s) { for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = s + i + j; }
    __device__ __forceinline__ void step(int i, const float (&d)[32], const float (&f)[32]) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 da = {d[(4 * i) & 31], d[(4 * i + 1) & 31], d[(4 * i + 2) & 31], d[(4 * i + 3) & 31]};
        const f4 fa = {f[(4 * i) & 31], f[(4 * i + 1) & 31], f[(4 * i + 2) & 31], f[(4 * i + 3) & 31]};
        const bf8 a = __builtin_bit_cast(bf8, da), b = __builtin_bit_cast(bf8, fa);
        if ((i & 1) == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
        else acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
    }
    __device__ float sum() { float s = 0; for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i]; return s; }
};

template <int SHAPE, int G, int I0>
__device__ __forceinline__ void mf_group(Pass<SHAPE>& ps, const float (&d)[32], const float (&f)[32]) {
#pragma unroll
    for (int i = 0; i < G; ++i) ps.step(I0 + i, d, f);
    __builtin_amdgcn_sched_barrier(0);
}

template <int SHAPE, int MODE, int NV>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* stamps, int iters, float bb, float cc) {
    extern __shared__ char pad[];   // 100 KiB: one workgroup per CU
    constexpr int NP = Pass<SHAPE>::NP;
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;
    float b = bb + threadIdx.x * 1e-9f, c = cc;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    float d[32], f[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { d[i] = 1e-3f * (threadIdx.x + i); f[i] = 1e-3f * (i - 7); }
    Pass<SHAPE> ps;
    ps.init(bb);
    if (MODE == 1 && second) return;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { valu<NV>(a, b, c); }
        if (MODE == 1 || MODE == 2) { mf_group<SHAPE, NP, 0>(ps, d, f); }
        if (MODE == 3) {
            if (second) valu<NV>(a, b, c);
            else mf_group<SHAPE, NP, 0>(ps, d, f);
        }
        if (MODE == 4) {
            // NP groups of (1 MFMA + NV / NP VALU); for the 64-instruction shapes two MFMAs per group of NV / 32
            constexpr int GM = NP / 32, GV = NV / 32;
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                if (g == 0) mf_group<SHAPE, GM, 0>(ps, d, f);
#define RMX_G(K) if (g == K) mf_group<SHAPE, GM, K * GM>(ps, d, f);
                RMX_G(1) RMX_G(2) RMX_G(3) RMX_G(4) RMX_G(5) RMX_G(6) RMX_G(7) RMX_G(8) RMX_G(9) RMX_G(10) RMX_G(11) RMX_G(12)
                RMX_G(13) RMX_G(14) RMX_G(15) RMX_G(16) RMX_G(17) RMX_G(18) RMX_G(19) RMX_G(20) RMX_G(21) RMX_G(22) RMX_G(23)
                RMX_G(24) RMX_G(25) RMX_G(26) RMX_G(27) RMX_G(28) RMX_G(29) RMX_G(30) RMX_G(31)
#undef RMX_G
                valu<GV>(a, b, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 5 || MODE == 6) {
            if (MODE == 5 && second) {
                valu<NV>(a, b, c);
                __builtin_amdgcn_sched_barrier(0);
                mf_group<SHAPE, NP, 0>(ps, d, f);
            } else {
                mf_group<SHAPE, NP, 0>(ps, d, f);
                valu<NV>(a, b, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = ps.sum();
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* g_out;
static long long* g_st;

template <int SHAPE, int MODE, int NV>
void run(const char* what) {
    const int iters = 2000;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, MODE, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CHECK(hipMemset(g_st, 0, 256 * 8 * 8));
    hipLaunchKernelGGL((k<SHAPE, MODE, NV>), dim3(256), dim3(512), 100 * 1024, 0, g_out, g_st, 50, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<SHAPE, MODE, NV>), dim3(256), dim3(512), 100 * 1024, 0, g_out, g_st, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<long long> st(256 * 8);
    CHECK(hipMemcpy(st.data(), g_st, st.size() * 8, hipMemcpyDeviceToHost));
    double c0 = 0, c4 = 0;
    for (int bl = 0; bl < 256; ++bl) { c0 += st[bl * 8 + 0]; c4 += st[bl * 8 + 4]; }
    c0 /= 256.0 * iters; c4 /= 256.0 * iters;
    std::printf("%-11s mode %d NV %3d  %-44s %8.3f us/iter   wave0 %7.0f  wave4 %7.0f cycles/iter\n", Pass<SHAPE>::name, MODE, NV,
                what, best * 1e3 / iters, c0, c4);
}

template <int SHAPE>
void shape() {
    run<SHAPE, 1, 0>("pass alone (waves 0-3)");
    run<SHAPE, 2, 0>("pass in all 8 waves");
    run<SHAPE, 3, 512>("waves 0-3 pass | waves 4-7 512 VALU");
    run<SHAPE, 3, 640>("waves 0-3 pass | waves 4-7 640 VALU");
    run<SHAPE, 4, 640>("both: pass spread through 640 VALU");
    run<SHAPE, 5, 640>("both: [pass][640 VALU], anti-phase");
    run<SHAPE, 6, 640>("both: [pass][640 VALU], in phase");
    run<SHAPE, 4, 576>("both: pass spread through 576 VALU");
    run<SHAPE, 5, 576>("both: [pass][576 VALU], anti-phase");
}

int main(int argc, char** argv) {
    CHECK(hipMalloc(&g_out, sizeof(float) * 512 * 256));
    CHECK(hipMalloc(&g_st, 256 * 8 * 8));
    const int reps = argc > 1 ? std::atoi(argv[1]) : 2;
    for (int rep = 0; rep < reps; ++rep) {
        run<0, 0, 768>("VALU only: 768 per wave (a k_win transform)");
        run<0, 0, 640>("VALU only: 640 per wave");
        run<0, 0, 576>("VALU only: 576 per wave");
        shape<0>();
        shape<1>();
        shape<2>();
        shape<3>();
        shape<4>();
    }
    return 0;
}
