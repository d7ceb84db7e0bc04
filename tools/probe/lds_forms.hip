// Probe (GPU box): what do the LDS exchange forms cost inside a k_win-like transform loop?
//   part 1  semantics of ds_write_addtid_b32 (address = M0 + offset + 4*lane): how many bits of M0 count, and does
//           M0 + offset reach beyond 128 KiB?
//   part 2  8 waves x 16 complex points, three radix-16 passes + two exchanges per transform (one behind a workgroup
//           barrier, one wave-local), persistent workgroup per CU, the real butterfly code of fft_r16.hpp:
//             mode 0  the exchange primitives k_win uses (ds_write_b64 / ds_read_b64 both ways)
//             mode 1  barrier exchange as 32 ds_write_addtid_b32 + 8 ds_read_b128 (planar image, wave blocks
//                     shifted by 16 B each), local exchange as in mode 0
//             mode 2  both exchanges that way (local rows 272 B apart, S / S^c rows interleaved)
//             mode 3  no LDS traffic at all (VALU + barrier only)
//           Timing only: modes 1-3 do not compute a transform (the data stay finite and mixed).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o lds_forms lds_forms.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../radio-mapper_amd/csrc/fft_r16.hpp"

using namespace rmx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- part 1 ------------------------------------------------------------------------------------
__global__ void k_sem(int* out, int m0v, int n_words) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) lds[i] = 0.0f;
    __syncthreads();
    if (threadIdx.x < 64) {
        float x = 1000.0f + threadIdx.x;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:65280\n\ts_waitcnt lgkmcnt(0)" ::"v"(x), "s"(m0v) : "memory", "m0");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int first = -1, cnt = 0;
        for (int i = 0; i < n_words; ++i)
            if (lds[i] != 0.0f) { if (first < 0) first = i; ++cnt; }
        out[0] = first;
        out[1] = cnt;
        out[2] = first >= 0 ? (int)lds[first] : 0;
    }
}

// ---- part 2 ------------------------------------------------------------------------------------
constexpr int kImgBytes = 69632;                 // mode 0 image (two of them); modes 1, 2: one planar image + local area
constexpr int kWaveBlk = 8192 + 16;              // planar image: [wave][plane][row][64 lanes] + 16 B shift per wave
constexpr int kLocRow = 272, kLocPlane = 16 * kLocRow, kLocBlk = 2 * kLocPlane;   // local area per wave

template <int O0, int O1, int O2, int O3, int O4, int O5, int O6, int O7>
__device__ __forceinline__ void addtid8(int m0v, float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    asm volatile(
        "s_mov_b32 m0, %8\n\t"
        "ds_write_addtid_b32 %0 offset:%9\n\t"
        "ds_write_addtid_b32 %1 offset:%10\n\t"
        "ds_write_addtid_b32 %2 offset:%11\n\t"
        "ds_write_addtid_b32 %3 offset:%12\n\t"
        "ds_write_addtid_b32 %4 offset:%13\n\t"
        "ds_write_addtid_b32 %5 offset:%14\n\t"
        "ds_write_addtid_b32 %6 offset:%15\n\t"
        "ds_write_addtid_b32 %7 offset:%16"
        ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(m0v), "n"(O0), "n"(O1), "n"(O2), "n"(O3),
          "n"(O4), "n"(O5), "n"(O6), "n"(O7)
        : "memory", "m0");
}
// row position of slot q in the local area: rows of S = {0-3, 12-15} on even positions, the others on odd ones
constexpr int loc_pos(int q) { return q < 4 ? 2 * q : (q >= 12 ? 2 * (q - 8) : 2 * (q - 4) + 1); }

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_t(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float2 x[16], tw[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        x[q] = make_float2(in[(blockIdx.x * 512 + t) * 32 + 2 * q], in[(blockIdx.x * 512 + t) * 32 + 2 * q + 1]);
        float s, c;
        __sincosf(0.37f * (t + 1) * q, &s, &c);
        tw[q] = make_float2(0.25f * c, 0.25f * s);
    }
    // mode 1/2 addresses
    const int d = lane & 15, b4 = (lane >> 4) & 1, b5 = lane >> 5;
    char* img_pl = smem;                                           // planar image (65664 B)
    char* loc = smem + 8 * kWaveBlk + wave * kLocBlk;              // this wave's local area
    const int m0_img = __builtin_amdgcn_readfirstlane(wave * kWaveBlk);
    const int m0_loc = __builtin_amdgcn_readfirstlane(8 * kWaveBlk + wave * kLocBlk);
    // barrier exchange reader: row = 2*wave + b4, source wave = d & 7, source group = b5 + 2*(d >> 3)
    const char* rd_img = img_pl + (d & 7) * kWaveBlk + (2 * wave + b4) * 256 + (b5 + 2 * (d >> 3)) * 64;
    // local reader: row position of slot d, group 2*b4 + b5
    const int lp = d < 4 ? 2 * d : (d >= 12 ? 2 * (d - 8) : 2 * (d - 4) + 1);
    const char* rd_loc = loc + lp * kLocRow + (2 * b4 + b5) * 64;

    for (int it = 0; it < iters; ++it) {
        float2* img = reinterpret_cast<float2*>(smem + (MODE == 0 ? (it & 1) * kImgBytes : 0));
        dft16_tw<false>(x, tw);
        if (MODE == 0) xchg_a_write(img, x, t);
        if (MODE == 1 || MODE == 2) {
#define A8(Q) addtid8<(Q) * 256, 4096 + (Q) * 256, (Q + 1) * 256, 4096 + (Q + 1) * 256, (Q + 2) * 256, 4096 + (Q + 2) * 256, (Q + 3) * 256, 4096 + (Q + 3) * 256>( \
        m0_img, x[Q].x, x[Q].y, x[Q + 1].x, x[Q + 1].y, x[Q + 2].x, x[Q + 2].y, x[Q + 3].x, x[Q + 3].y)
            A8(0); A8(4); A8(8); A8(12);
#undef A8
        }
        __syncthreads();
        if (MODE == 0) xchg_b_read(img, x, t);
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 re = *reinterpret_cast<const float4*>(rd_img + 16 * j);
                const float4 im = *reinterpret_cast<const float4*>(rd_img + 4096 + 16 * j);
                x[4 * j] = make_float2(re.x, im.x); x[4 * j + 1] = make_float2(re.y, im.y);
                x[4 * j + 2] = make_float2(re.z, im.z); x[4 * j + 3] = make_float2(re.w, im.w);
            }
        }
        dft16_tw<false>(x, tw);
        if (MODE == 0) {
            xchg_bc_write_b(img, x, t);
            wave_lds_order();
            xchg_bc_read_c(img, x, t);
        }
        if (MODE == 1) {   // production local exchange in this wave's local area (two half-wave regions of 4352 B)
            float2* l2 = reinterpret_cast<float2*>(smem + 8 * kWaveBlk) - 0;
            xchg_bc_write_b(l2, x, t);
            wave_lds_order();
            xchg_bc_read_c(l2, x, t);
        }
        if (MODE == 2) {
#define L8(Q) addtid8<loc_pos(Q) * kLocRow, kLocPlane + loc_pos(Q) * kLocRow, loc_pos(Q + 1) * kLocRow, kLocPlane + loc_pos(Q + 1) * kLocRow, \
                      loc_pos(Q + 2) * kLocRow, kLocPlane + loc_pos(Q + 2) * kLocRow, loc_pos(Q + 3) * kLocRow, kLocPlane + loc_pos(Q + 3) * kLocRow>( \
        m0_loc, x[Q].x, x[Q].y, x[Q + 1].x, x[Q + 1].y, x[Q + 2].x, x[Q + 2].y, x[Q + 3].x, x[Q + 3].y)
            L8(0); L8(4); L8(8); L8(12);
#undef L8
            wave_lds_order();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 re = *reinterpret_cast<const float4*>(rd_loc + 16 * j);
                const float4 im = *reinterpret_cast<const float4*>(rd_loc + kLocPlane + 16 * j);
                x[4 * j] = make_float2(re.x, im.x); x[4 * j + 1] = make_float2(re.y, im.y);
                x[4 * j + 2] = make_float2(re.z, im.z); x[4 * j + 3] = make_float2(re.w, im.w);
            }
        }
        dft16_tw<false>(x, tw);
        dft16(x);
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = make_float2(0.25f * x[q].x, 0.25f * x[q].y);
    }
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += x[q].x + x[q].y;
    out[blockIdx.x * 512 + t] = s;
}

template <int MODE>
static void run(const char* name, float* d_out, const float* d_in, int iters) {
    const size_t lds = 2 * kImgBytes;   // 139264 B: one workgroup per CU in every mode
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_t<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_t<MODE>, dim3(256), dim3(512), lds, 0, d_out, d_in, iters);
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_t<MODE>, dim3(256), dim3(512), lds, 0, d_out, d_in, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    std::vector<float> h(16);
    CK(hipMemcpy(h.data(), d_out, 64, hipMemcpyDeviceToHost));
    printf("%-34s %8.4f ms mean %8.4f ms best  = %.3f us per transform  (out[0]=%g)\n", name, sum / reps, best,
           1e3 * (sum / reps) / iters, h[0]);
}

int main() {
    int* d_i; CK(hipMalloc(&d_i, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sem), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int words = 160 * 1024 / 4;
    for (int m0v : {0, 4096, 65280, 65536, 65536 + 4096, 98304}) {
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(256), 160 * 1024, 0, d_i, m0v, words);
        CK(hipDeviceSynchronize());
        int h[3]; CK(hipMemcpy(h, d_i, 12, hipMemcpyDeviceToHost));
        printf("addtid: M0=%6d offset=65280 -> first nonzero byte address %d (expected %d), %d words written, first value %d\n",
               m0v, h[0] * 4, m0v + 65280, h[1], h[2]);
    }
    const int iters = 576;   // transforms per workgroup in a cfg3 launch (16 windows x 36)
    float *d_in, *d_out;
    std::vector<float> h_in(256 * 512 * 32);
    srand(7);
    for (auto& v : h_in) v = (rand() / (float)RAND_MAX - 0.5f) * 200.0f;
    CK(hipMalloc(&d_in, h_in.size() * 4)); CK(hipMalloc(&d_out, 256 * 512 * 4));
    CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("mode 0: b64 writes / b64 reads", d_out, d_in, iters);
        run<1>("mode 1: barrier xchg addtid+b128", d_out, d_in, iters);
        run<2>("mode 2: both xchg addtid+b128", d_out, d_in, iters);
        run<3>("mode 3: no LDS", d_out, d_in, iters);
    }
    return 0;
}
