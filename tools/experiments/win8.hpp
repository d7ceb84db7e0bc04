// win8.hpp -- k_win8: the fused window kernel (N = 4096, all pairs) with 8 complex points per thread on
// 1024 threads, i.e. 16 waves = 4 per SIMD at <= 128 VGPRs, where k_win (16 points x 512 threads, ~200
// VGPRs) runs 2 per SIMD.  Same definition, same schedule per window, same spectrum scratch size.
//
// Decomposition of one length-L = 2M transform, M = 4096 = 8^4 (tools/model_win8.py executes exactly
// these index maps on the CPU and checks them against numpy's FFT):
//     time  n = 512 n3 + 64 n2 + 8 n1 + n0        bin  k = c0 + 8 c1 + 64 d0 + 512 d1   (L-bin 2k + p)
//     role 1  T = 2 t1 + p, t1 = n0 + 8 n1 + 64 n2  slots n3 <-> c0   time side; the bin parity p sits on
//                                                                   lane bit 0 (last radix-2 = one DPP op)
//     role 2  T = 512 p + 64 c0 + 8 n1 + n0         slots n2 <-> c1
//     role 3  T = 512 p + 64 c0 + 8 c1 + n0         slots n1 <-> d0
//     role 4  T = 512 p + 64 c0 + 8 c1 + d0         slots n0 <-> d1   frequency side (spectrum layout)
//   four radix-8 passes in registers; exchange 1<->2 crosses waves (the transform's only workgroup
//   barrier), exchanges 2<->3 and 3<->4 stay inside one wave (its own 576-complex region of the image).
//   Twiddles: W_M^(c0 t1) [* W_L^t1 on odd lanes] per thread in registers (TW1, 4096 distinct values) at
//   the barrier exchange; W_512^(c1 (8 n1 + n0)) and W_64^(d0 n0) from LDS rows (TB: one row per lane,
//   TC: one row per lane & 7), merged into the first butterfly layer of the pass that follows them.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../radio-mapper_amd/csrc/fft_r16.hpp"
#include "../../radio-mapper_amd/csrc/fft_r8.hpp"   // dft8 and its helpers (shared with the generic path)

namespace rmx {
namespace w8 {

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
using u32x2 = unsigned int __attribute__((ext_vector_type(2)));

constexpr int kT8 = 1024;                    // threads per workgroup
constexpr int kS8 = 8;                       // complex points per thread
constexpr int kReg = 576;                    // complex per wave region of an exchange image (512 + padding)
constexpr int kImgF2 = 16 * kReg;            // complex per image
constexpr int kLdsImg = kImgF2 * 8;          // 73728 B, two images
constexpr int kRowF2 = 10;                   // twiddle row: 7 complex in use order + pad = 80 B (conflict-free b128)
constexpr int kLdsTbOff = 2 * kLdsImg;
constexpr int kLdsTb = 64 * kRowF2 * 8;
constexpr int kLdsTcOff = kLdsTbOff + kLdsTb;
constexpr int kLdsTc = 8 * kRowF2 * 8;
constexpr int kRes8 = 4;                     // record ring; winners are resolved in batches of kBatch8 pairs
constexpr int kBatch8 = 3;
constexpr int kLdsHaloOff = kLdsTcOff + kLdsTc;                    // [slots][16 waves][4 rows][8] float
constexpr int kLdsRedOff = kLdsHaloOff + kRes8 * 16 * 4 * 8 * 4;   // [slots][16] float4
constexpr int kLdsOidxOff = kLdsRedOff + kRes8 * 16 * 16;          // [slots] int
constexpr int kLdsWin8Bytes = kLdsOidxOff + kRes8 * 4;
static_assert(kLdsWin8Bytes <= 160 * 1024, "k_win8 LDS");

struct C8 {
    float re[8], im[8];
    __device__ __forceinline__ float2 get(int q) const { return make_float2(re[q], im[q]); }
    __device__ __forceinline__ void set(int q, float x, float y) { re[q] = x; im[q] = y; }
};

// first layer with the pre-twiddles w[q] merged (v[q] w[q] +- v[q+4] w[q+4]: 10 instructions per pair, 6
// when w[q] == 1)
template <bool W0_IS_ONE>
__device__ __forceinline__ void dft8_stage_a_tw(float2 (&v)[8], float2 w0, float2 w1, float2 w2, float2 w3, float2 w4,
                                                float2 w5, float2 w6, float2 w7) {
    const float2 a0 = W0_IS_ONE ? v[0] : cmul(v[0], w0);
    v[0] = cfma(a0, v[4], w4);
    v[4] = twice_minus(a0, v[0]);
    const float2 a1 = cmul(v[1], w1);
    v[1] = cfma(a1, v[5], w5);
    v[5] = twice_minus(a1, v[1]);
    const float2 a2 = cmul(v[2], w2);
    v[2] = cfma(a2, v[6], w6);
    v[6] = twice_minus(a2, v[2]);
    const float2 a3 = cmul(v[3], w3);
    v[3] = cfma(a3, v[7], w7);
    v[7] = twice_minus(a3, v[3]);
}
// Second half of every pass, handing the four even outputs X[0], X[2], X[4], X[6] and then the four odd
// ones to `emit` as soon as they exist (the caller's LDS stores / prefetch requests spread over the
// arithmetic instead of one burst); v is left in half order (not natural).
template <class F>
__device__ __forceinline__ void dft8_finish_emit(float2 (&v)[8], F&& emit) {
    dft4(v[0], v[1], v[2], v[3]);
    emit(std::integral_constant<int, 0>{}, v[0], v[1], v[2], v[3]);
    __builtin_amdgcn_sched_barrier(0);
    dft4_w8(v[4], v[5], v[6], v[7]);
    emit(std::integral_constant<int, 1>{}, v[4], v[5], v[6], v[7]);
    __builtin_amdgcn_sched_barrier(0);
}
// twiddle row in LDS: (w4, w1, w5, w2, w6, w3, w7, pad) -- the order stage A consumes it; w0 = 1 is not stored
__device__ __forceinline__ void dft8_stage_a_row(float2 (&v)[8], const float4* row, float4 f0) {
    const float4 f1 = row[1], f2 = row[2], f3 = row[3];
    dft8_stage_a_tw<true>(v, make_float2(1.0f, 0.0f), make_float2(f0.z, f0.w), make_float2(f1.z, f1.w),
                          make_float2(f2.z, f2.w), make_float2(f0.x, f0.y), make_float2(f1.x, f1.y),
                          make_float2(f2.x, f2.y), make_float2(f3.x, f3.y));
}
// v[k] *= w[k], k = 1..7, from such a row (post-twiddle of the forward passes)
__device__ __forceinline__ void mul_row(float2 (&v)[8], const float4* row) {
    const float4 f0 = row[0], f1 = row[1], f2 = row[2], f3 = row[3];
    v[4] = cmul(v[4], make_float2(f0.x, f0.y));
    v[1] = cmul(v[1], make_float2(f0.z, f0.w));
    v[5] = cmul(v[5], make_float2(f1.x, f1.y));
    v[2] = cmul(v[2], make_float2(f1.z, f1.w));
    v[6] = cmul(v[6], make_float2(f2.x, f2.y));
    v[3] = cmul(v[3], make_float2(f2.z, f2.w));
    v[7] = cmul(v[7], make_float2(f3.x, f3.y));
}

// W16^q, q = 0..7: the per-slot part W_L^(512 q) of the odd sub-transform's W_L^n
__device__ __forceinline__ float2 w16c(int q) {
    constexpr float c[8] = {1.0f, RMX_C1, RMX_RH, RMX_S1, 0.0f, -RMX_S1, -RMX_RH, -RMX_C1};
    constexpr float s[8] = {0.0f, RMX_S1, RMX_RH, RMX_C1, 1.0f, RMX_C1, RMX_RH, RMX_S1};
    return make_float2(c[q], -s[q]);
}

__device__ __forceinline__ void k_to_owner8(int kk, int& tt, int& q) {
    const int par = (kk >= kM - 1) ? 0 : 1;
    const int n = par ? (kk + 1) : (kk - (kM - 1));
    tt = 2 * (n & 511) + par;
    q = n >> 9;
}

// One whole wave resolves up to kBatch8 pairs after a barrier that published their records: lane =
// 16 g + r looks at wave r's record of the g-th pair (see resolve_batch of k_win for the protocol).
__device__ __forceinline__ void resolve_batch8(int lane, const float4* red, const float* halo, const int* oidx,
                                               int first, int cnt, long obase, float out_scale,
                                               int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                               float* __restrict__ peak) {
    const int g = lane >> 4, r = lane & 15;
    const bool act = g < cnt;
    const int slot = (first + g) & (kRes8 - 1);
    const float* rf = reinterpret_cast<const float*>(red) + 4 * (slot * 16 + r);
    const int* ri = reinterpret_cast<const int*>(rf);
    const float ex = act ? rf[0] : -3.0f;
    const int k = act ? ri[1] : 0x7fffffff;
    const float tm = rf[2], tp = rf[3];
    const int out = oidx[slot];
    float gmax = ex;                                     // max over the 16 lanes of the group (one DPP row)
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0xB1>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x4E>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x141>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x140>(__builtin_bit_cast(int, gmax))));
    int kstar = (ex == gmax) ? k : 0x7fffffff;
    kstar = min(kstar, dpp_i<0xB1>(kstar));
    kstar = min(kstar, dpp_i<0x4E>(kstar));
    kstar = min(kstar, dpp_i<0x141>(kstar));
    kstar = min(kstar, dpp_i<0x140>(kstar));
    const bool win = act && ex == gmax && k == kstar;     // exactly one lane per active group
    auto halo_tap = [&](int kk) -> float {
        kk = kk < 0 ? 0 : (kk > 2 * kM - 2 ? 2 * kM - 2 : kk);
        int tt, q;
        k_to_owner8(kk, tt, q);
        const int ln = tt & 63;
        const int row = ln < 2 ? ln : (ln >= 62 ? ln - 60 : 0);
        return halo[(((slot * 16 + (tt >> 6)) * 4) + row) * 8 + q];
    };
    const int kc = win ? k : (kM - 1);
    const float hm = halo_tap(kc - 1), hp = halo_tap(kc + 1);
    const float b = sqrtf(fmaxf(ex, 0.0f)) * out_scale;
    const float a = sqrtf(tm >= 0.0f ? tm : hm) * out_scale;
    const float c = sqrtf(tp >= 0.0f ? tp : hp) * out_scale;
    const double den = (double)a - 2.0 * (double)b + (double)c;
    float frac = 0.0f;
    if (kc > 0 && kc < 2 * kM - 2 && den != 0.0) frac = (float)(0.5 * ((double)a - (double)c) / den);
    if (win) {
        lag_int[obase + out] = kc - (kM - 1);
        lag_frac[obase + out] = frac;
        peak[obase + out] = b;
    }
}

// spec scratch: [workgroup][b][j = 0..3][T = 0..1023] float4 = slots (2j, 2j+1) of thread T in role 4
// tw1_g: [j = 0..3][T] float4 = TW1 slots (2j, 2j+1) of thread T (role 1); tb_g [64][10], tc_g [8][10] complex
template <bool U8>
__global__ __launch_bounds__(kT8, 4) void k_win8(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                 const float4* __restrict__ tw1_g, const float2* __restrict__ tb_g,
                                                 const float2* __restrict__ tc_g, int n_buoys, long first_window,
                                                 float out_scale, int* __restrict__ lag_int,
                                                 float* __restrict__ lag_frac, float* __restrict__ peak, int n_win,
                                                 int stag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* img0 = reinterpret_cast<float2*>(smem);
    float2* img1 = reinterpret_cast<float2*>(smem + kLdsImg);
    float2* tb_lds = reinterpret_cast<float2*>(smem + kLdsTbOff);
    float2* tc_lds = reinterpret_cast<float2*>(smem + kLdsTcOff);
    float* halo = reinterpret_cast<float*>(smem + kLdsHaloOff);
    float4* red = reinterpret_cast<float4*>(smem + kLdsRedOff);
    int* oidx = reinterpret_cast<int*>(smem + kLdsOidxOff);

    const int T = threadIdx.x;
    const int lane = T & 63, wave = T >> 6;
    const int p1 = T & 1, t1 = T >> 1;            // role 1
    const int pt = T >> 9, lo = lane & 7, hi = lane >> 3;   // roles 2..4
    const int B = n_buoys;
    const int n_pairs = B * (B - 1) / 2;

    for (int i = T; i < 64 * kRowF2; i += kT8) tb_lds[i] = tb_g[i];
    if (T < 8 * kRowF2) tc_lds[T] = tc_g[T];
    float2 tw1[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 w = tw1_g[j * kT8 + T];
        tw1[2 * j] = make_float2(w.x, w.y);
        tw1[2 * j + 1] = make_float2(w.z, w.w);
    }
    const float4* tbrow = reinterpret_cast<const float4*>(tb_lds + lane * kRowF2);
    const float4* tcrow = reinterpret_cast<const float4*>(tc_lds + lo * kRowF2);
    // LDS addresses (complex index inside an image; tools/model_win8.py: a43, a32, a21)
    const int reg = wave * kReg;
    const int o43_4 = reg + hi * 72 + lo;            // role 4: slot stride 9
    const int o43_3 = reg + hi * 72 + lo * 9;        // role 3: slot stride 1
    const int o32_3 = reg + lane;                    // role 3: slot stride 72
    const int o32_2 = reg + hi * 72 + lo;            // role 2: slot stride 8
    const int o21_2 = reg + 16 * pt + (lane ^ (8 * pt));                   // role 2: slot stride 64
    const int o21_1 = 8 * p1 * kReg + 16 * p1 + (t1 ^ (8 * p1));           // role 1: slot stride kReg
    const float sgn = p1 ? -1.0f : 1.0f;
    const int kbase = p1 ? (t1 - 1) : (t1 + kM - 1);
    const int hl = lane < 2 ? lane : lane - 60;      // halo row of lanes 0,1,62,63
    const bool is_halo = lane < 2 || lane >= 62;
    bool late_h2 = false;                            // staggered order of the two halves between barriers
    if (stag == 1) late_h2 = (wave >> 1) & 1;
    else if (stag == 2) late_h2 = (wave >> 2) & 1;
    else if (stag == 3) late_h2 = (wave >> 3) & 1;
    else if (stag == 4) late_h2 = wave & 1;
    else if (stag == 5) late_h2 = true;
    __syncthreads();

    for (int wl = blockIdx.x; wl < n_win; wl += gridDim.x) {
    C8 sa, sb;   // anchor spectrum X_i and the streamed X_j
    const long wbase = (long)blockIdx.x * B;
    const long obase = (first_window + wl) * (long)n_pairs;
    int seq = 0;         // transform counter: selects the exchange image
    int npair = 0;       // pair counter: selects the record slot
    int npend = 0;       // pairs whose records await a resolve

    auto barrier_hook = [&](bool flush) __attribute__((always_inline)) {
        __syncthreads();
        if (npend == kBatch8 || (flush && npend > 0)) {
            if (wave == (seq & 15))
                resolve_batch8(lane, red, halo, oidx, (npair - npend) & (kRes8 - 1), npend, obase, out_scale, lag_int,
                               lag_frac, peak);
            npend = 0;
        }
    };
    // odd lanes: v[q] *= W16^q (in place, exec-masked)
    auto mul_w16_odd = [&](float2 (&v)[8]) __attribute__((always_inline)) {
        if (p1) {
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float2 w = w16c(q);
                float x = v[q].x, y = v[q].y;
                cmul_inplace(x, y, w.x, w.y);
                v[q].x = x;
                v[q].y = y;
            }
            float x0 = v[4].x, y0 = v[4].y, x1 = v[5].x, y1 = v[5].y;
            float x2 = v[6].x, y2 = v[6].y, x3 = v[7].x, y3 = v[7].y;
            cmul4_inplace(x0, y0, x1, y1, x2, y2, x3, y3, w16c(4), w16c(5), w16c(6), w16c(7));
            v[4].x = x0; v[4].y = y0; v[5].x = x1; v[5].y = y1;
            v[6].x = x2; v[6].y = y2; v[7].x = x3; v[7].y = y3;
        }
    };
    const int samp_bytes = U8 ? 2 : 8;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(iq_v)) + (first_window + wl) * (long)B * kM * samp_bytes, 0,
        B * kM * samp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(spec) + wbase * (long)(4 * kT8 * 16), 0, B * (4 * kT8 * 16), 0x00020000);
    const int xoff = t1 * samp_bytes, soff = T * 16;
    // quarter G of the window samples of buoy b (slots 2G, 2G+1; uint8 pairs stay packed until cvt_x)
    auto load_x_part = [&](C8& d, int b, auto part) __attribute__((always_inline)) {
        constexpr int G = decltype(part)::value;
        if constexpr (U8) {
#pragma unroll
            for (int q = 2 * G; q < 2 * G + 2; ++q)
                d.re[q] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(xs, xoff, (b * kM + q * 512) * 2, 0));
        } else {
#pragma unroll
            for (int q = 2 * G; q < 2 * G + 2; ++q) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(xs, xoff, (b * kM + q * 512) * 8, 0);
                d.set(q, __uint_as_float(r.x), __uint_as_float(r.y));
            }
        }
    };
    auto load_x = [&](C8& d, int b) __attribute__((always_inline)) {
        load_x_part(d, b, std::integral_constant<int, 0>{});
        load_x_part(d, b, std::integral_constant<int, 1>{});
        load_x_part(d, b, std::integral_constant<int, 2>{});
        load_x_part(d, b, std::integral_constant<int, 3>{});
    };
    auto cvt_x = [&](C8& d) __attribute__((always_inline)) {
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned r = __float_as_uint(d.re[q]);
                d.set(q, (float)(r & 0xffu) - 127.5f, (float)(r >> 8) - 127.5f);
            }
        }
    };
    auto load_spec_part = [&](C8& d, int b, auto part) __attribute__((always_inline)) {
        constexpr int G = decltype(part)::value;
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(ss, soff, (b * 4 + G) * (kT8 * 16), 0);
        d.set(2 * G, __uint_as_float(w.x), __uint_as_float(w.y));
        d.set(2 * G + 1, __uint_as_float(w.z), __uint_as_float(w.w));
    };
    auto store_spec = [&](const C8& d, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e0 = d.re[2 * j], e1 = d.im[2 * j], e2 = d.re[2 * j + 1], e3 = d.im[2 * j + 1];
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
            const u32x4 w = {__float_as_uint(e0), __float_as_uint(e1), __float_as_uint(e2), __float_as_uint(e3)};
            // whole byte offset in the VGPR offset (see store_spec of k_win)
            __builtin_amdgcn_raw_buffer_store_b128(w, ss, soff + (b * 4 + j) * (kT8 * 16), 0, 0);
        }
    };
    // forward spectrum of the samples in xc, in place (carries the 2^-6 of the TW1 table)
    auto fwd = [&](C8& xc) __attribute__((always_inline)) {
        float2* img = (seq & 1) ? img1 : img0;
        float2 x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = xc.get(q);
        mul_w16_odd(x);            // odd sub-transform input x W_L^(512 n3) (W_L^t1 is folded into tw1)
        dft8(x);                   // n3 -> c0
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x[k] = cmul(x[k], tw1[k]);
            img[o21_1 + k * kReg] = make_float2(x[k].x, x[k].y);
        }
        barrier_hook(false);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = img[o21_2 + 64 * k];
        dft8(x);                   // n2 -> c1
        mul_row(x, tbrow);         // W_512^(c1 (8 n1 + n0))
#pragma unroll
        for (int k = 0; k < 8; ++k) img[o32_2 + 8 * k] = make_float2(x[k].x, x[k].y);
        wave_lds_order();
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = img[o32_3 + 72 * k];
        dft8(x);                   // n1 -> d0
        mul_row(x, tcrow);         // W_64^(d0 n0)
#pragma unroll
        for (int k = 0; k < 8; ++k) img[o43_3 + k] = make_float2(x[k].x, x[k].y);
        wave_lds_order();
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = img[o43_4 + 9 * k];
        dft8(x);                   // n0 -> d1
#pragma unroll
        for (int q = 0; q < 8; ++q) xc.set(q, x[q].x, x[q].y);
        ++seq;
    };
    // One pair = two halves around its only workgroup barrier (see k_win):
    //   h1  conj-multiply merged into the role-4 pass, two wave-local exchanges with the role-3 and role-2
    //       passes, stores into this wave's region of exchange image `tr & 1`; `prefetch(part)` is called
    //       four times after the last read of a and s
    //   h2  reads image `tr & 1` across all regions, role-1 pass, last radix-2, |.|^2, peak records
    auto pair_h1 = [&](const C8& a, const C8& s, int tr, auto prefetch) __attribute__((always_inline)) {
        float2* img = (tr & 1) ? img1 : img0;
        float2 v[8];
        // R = X_j conj(X_i), (im,re)-swapped == swap(X_j) * X_i: the pre-twiddle of the first pass
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = make_float2(s.im[q], s.re[q]);
        dft8_stage_a_tw<false>(v, a.get(0), a.get(1), a.get(2), a.get(3), a.get(4), a.get(5), a.get(6), a.get(7));
#pragma unroll
        for (int q = 0; q < 8; q += 4)           // pin: the requests below must follow the reads above
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q + 1].x), "+v"(v[q + 1].y), "+v"(v[q + 2].x),
                         "+v"(v[q + 2].y), "+v"(v[q + 3].x), "+v"(v[q + 3].y));
        __builtin_amdgcn_sched_barrier(0);
        dft8_finish_emit(v, [&](auto hc, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                __attribute__((always_inline)) {       // d1 -> n0: slots n0 = h, h+2, h+4, h+6
            constexpr int h = decltype(hc)::value;
            img[o43_4 + 9 * h] = make_float2(x0.x, x0.y);
            img[o43_4 + 9 * (h + 2)] = make_float2(x1.x, x1.y);
            img[o43_4 + 9 * (h + 4)] = make_float2(x2.x, x2.y);
            img[o43_4 + 9 * (h + 6)] = make_float2(x3.x, x3.y);
            prefetch(std::integral_constant<int, 2 * h>{});
            prefetch(std::integral_constant<int, 2 * h + 1>{});
        });
        const float4 c0 = tcrow[0];                    // ahead of the exchange reads (LDS returns in issue order)
        wave_lds_order();
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = img[o43_3 + k];
        dft8_stage_a_row(v, tcrow, c0);                // W_64^(d0 n0), d0 -> n1
        dft8_finish_emit(v, [&](auto hc, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
            img[o32_3 + 72 * h] = make_float2(x0.x, x0.y);
            img[o32_3 + 72 * (h + 2)] = make_float2(x1.x, x1.y);
            img[o32_3 + 72 * (h + 4)] = make_float2(x2.x, x2.y);
            img[o32_3 + 72 * (h + 6)] = make_float2(x3.x, x3.y);
        });
        const float4 b0 = tbrow[0];
        wave_lds_order();
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = img[o32_2 + 8 * k];
        dft8_stage_a_row(v, tbrow, b0);                // W_512^(c1 (8 n1 + n0)), c1 -> n2
        dft8_finish_emit(v, [&](auto hc, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
            img[o21_2 + 64 * h] = make_float2(x0.x, x0.y);
            img[o21_2 + 64 * (h + 2)] = make_float2(x1.x, x1.y);
            img[o21_2 + 64 * (h + 4)] = make_float2(x2.x, x2.y);
            img[o21_2 + 64 * (h + 6)] = make_float2(x3.x, x3.y);
        });
    };
    auto pair_h2 = [&](int tr, int out_idx) __attribute__((always_inline)) {
        const float2* img = (tr & 1) ? img1 : img0;
        const int rb = npair & (kRes8 - 1);
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = img[o21_1 + k * kReg];
        dft8_stage_a_tw<false>(v, tw1[0], tw1[1], tw1[2], tw1[3], tw1[4], tw1[5], tw1[6], tw1[7]);   // W_M^(c0 t1) [* W_L^t1 odd]
        dft4(v[0], v[1], v[2], v[3]);
        dft4_w8(v[4], v[5], v[6], v[7]);
        dft8_unshuffle(v);                       // c0 -> n3: even lanes e[n], odd lanes o[n] W_L^t1, n = 512 q + t1
        mul_w16_odd(v);                          // odd lanes: * W16^q
        // last radix-2 stage across the lane pair, up to a sign that |.| does not see
        pair_fmac8(v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y, sgn);
        pair_fmac8(v[4].x, v[4].y, v[5].x, v[5].y, v[6].x, v[6].y, v[7].x, v[7].y, sgn);
        float mag[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) mag[q] = fmaf(v[q].x, v[q].x, v[q].y * v[q].y);
        if (p1 && t1 == 0) mag[0] = -1.0f;       // lag -M is not part of the 'full' output
        if (is_halo) {
            float4* hp = reinterpret_cast<float4*>(halo + ((rb * 16 + wave) * 4 + hl) * 8);
            hp[0] = make_float4(mag[0], mag[1], mag[2], mag[3]);
            hp[1] = make_float4(mag[4], mag[5], mag[6], mag[7]);
        }
        float tmax = mag[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) tmax = fmaxf(tmax, mag[q]);
        int qa = 8, qb = 8, qc = 8, qd = 8;      // lowest slot holding the max: descending, lower slots win
        argsel4<4>(qa, qb, qc, qd, mag[4], mag[5], mag[6], mag[7], tmax);
        argsel4<0>(qa, qb, qc, qd, mag[0], mag[1], mag[2], mag[3], tmax);
        const int qsel = min(min(qa, qb), min(qc, qd));
        const int kq = kbase + qsel * 512;
        const float wmax = wave_max_f32(tmax);
        const int kw = wave_min_i32(tmax == wmax ? kq : 0x7fffffff);
        // the winner's neighbours k*-1, k*+1 live in lanes l*-2, l*+2 (same slot) when those exist
        int ts, qs;
        k_to_owner8(kw, ts, qs);
        const int ls = ts & 63;
        typedef float f8v __attribute__((ext_vector_type(8)));
        const f8v mv = {mag[0], mag[1], mag[2], mag[3], mag[4], mag[5], mag[6], mag[7]};
        const float sel = mv[__builtin_amdgcn_readfirstlane(qs)];
        const int seli = __builtin_bit_cast(int, sel);
        const float tapm = ls >= 2 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(seli, ls >= 2 ? ls - 2 : 0)) : -2.0f;
        const float tapp = ls <= 61 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(seli, ls <= 61 ? ls + 2 : 63)) : -2.0f;
        if (lane == 0) {
            const u32x4 rec = {__float_as_uint(wmax), (unsigned)kw, __float_as_uint(tapm), __float_as_uint(tapp)};
            *reinterpret_cast<u32x4*>(red + rb * 16 + wave) = rec;
            if (wave == 0) oidx[rb] = out_idx;
        }
        ++npend;
        ++npair;
    };
    auto pair = [&](const C8& a, const C8& s, int out_idx, auto prefetch) __attribute__((always_inline)) {
        pair_h1(a, s, seq, prefetch);
        barrier_hook(false);                     // the pair's only barrier
        pair_h2(seq, out_idx);
        ++seq;
    };
    auto out_of = [&](int i, int j) -> int { return i * B - (i * (i + 1)) / 2 + (j - i - 1); };

    // ---- anchor 0 (schedule and comments: k_win)
    load_x(sa, 0);
    if (B > 1) load_x(sb, 1);
    cvt_x(sa);
    fwd(sa);
    for (int e = 1; e + 1 < B; ++e) {
        cvt_x(sb);
        fwd(sb);
        store_spec(sb, e);
        pair(sa, sb, out_of(0, e), [&](auto part) __attribute__((always_inline)) { load_x_part(sb, e + 1, part); });
    }
    if (B > 1) {
        cvt_x(sb);
        fwd(sb);
        store_spec(sb, B - 1);
        pair(sa, sb, out_of(0, B - 1), [&](auto part) __attribute__((always_inline)) {
            if (B > 2) {
                load_spec_part(sa, 1, part);
                load_spec_part(sb, B - 1, part);
            }
        });
    }
    // ---- anchors 1..B-2: the stream direction alternates; h2 of pair m and h1 of pair m+1 sit between the
    // same two barriers and are independent, so some waves run them in the opposite order (`stag`)
    {
        const int M2 = (B - 1) * (B - 2) / 2;
        auto j_of = [&](int i, int s) -> int { return (i & 1) ? (B - 1 - s) : (i + 1 + s); };
        int ci = 1, cs = 0;
        int ni = 1, ns = 1;
        if (ns >= B - 1 - ni) { ++ni; ns = 0; }
        auto h1_of = [&](int hi_, int hs, int tr) __attribute__((always_inline)) {
            int pi = hi_, ps = hs + 1;
            if (ps >= B - 1 - pi) { ++pi; ps = 0; }
            const bool valid = pi + 1 < B;
            const bool new_anchor = pi != hi_;
            const int pj = j_of(pi, ps);
            pair_h1(sa, sb, tr, [&](auto part) __attribute__((always_inline)) {
                if (valid) {
                    if (new_anchor) load_spec_part(sa, pi, part);
                    load_spec_part(sb, pj, part);
                }
            });
        };
        if (M2 > 0) h1_of(ci, cs, seq);
        for (int m = 0; m < M2; ++m) {
            barrier_hook(false);
            const bool has_next = m + 1 < M2;
            const int out_idx = out_of(ci, j_of(ci, cs));
            if (late_h2) {
                if (has_next) h1_of(ni, ns, seq + 1);
                pair_h2(seq, out_idx);
            } else {
                pair_h2(seq, out_idx);
                if (has_next) h1_of(ni, ns, seq + 1);
            }
            ++seq;
            ci = ni; cs = ns;
            ++ns;
            if (ns >= B - 1 - ni) { ++ni; ns = 0; }
        }
    }
    seq = 0;   // any wave may resolve the last pairs; take wave 0
    barrier_hook(true);
    }   // next window of this workgroup
}

}  // namespace w8
}  // namespace rmx
