// rmx_hip.hip -- gfx950 kernels and the C ABI (include/rmx.h) of the TDoA cross-correlation engine.
//
// Path (conventions of tdoa_processor.py:20,51,156-157,166-170):
//   per window w, pair (i<j):  r = IFFT_L( FFT_L(x_j) * conj(FFT_L(x_i)) ), L = 2N (zero padded)
//                               m = |r|, k = argmax over the 2N-1 'full' lags, parabolic offset.
// Two kernel families, launched back to back per chunk of windows on one stream:
//   k_fwd   one workgroup per (window, buoy): forward spectrum -> HBM/L2 scratch, stored in the
//           *register layout* of the pair kernel (every thread later reads back exactly the 16
//           bins it wrote: 16-B-per-lane coalesced both ways, no reordering pass);
//   k_pair  one workgroup per (window, group of pairs): loads the two spectra, multiplies in
//           registers, inverse transform (3 radix-16 passes, 2 LDS exchanges), |.|^2, workgroup
//           argmax with numpy tie-breaking, 3-tap parabola, 12 bytes out per pair-window.
// No MFMA (not a contraction), no library FFT: everything below is hand written for wave64 / LDS.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/rmx.h"
#include "host_plan.hpp"
#include "fft_r16.hpp"
#include "kwin.hpp"
#include "fft_r8.hpp"
#ifdef RMX_EXPERIMENTS   // k_win8 / k_winp: two other builds of the fused kernel, measured slower (tools/experiments/README.md)
#include "../../tools/experiments/win8.hpp"
#include "../../tools/experiments/winpk.hpp"
#endif
#include "generic_path.hpp"
#include "win_eo.hpp"
#include "kwin8k.hpp"                           // N = 8192 on k_win's network, one anchor half resident in LDS (k_win8kl)
#include "kwin16k.hpp"                          // N = 16384 on the same network: four quarter transforms, two kernels (k16_fwd, k16_pairs)
#ifdef RMX_EXPERIMENTS
#include "../../tools/experiments/kwin8k.hpp"   // its two predecessors: parity-green, not faster (LABNOTES.md R4.6)
#endif
#include "detect_path.hpp"

namespace rmx {




// ------------------------------------------------------------------------------------------------
// Forward spectra.  grid = n_items workgroups of 512; item = wl * B + b inside the chunk.
//   spec layout: [item][j = 0..7][t = 0..511] float4 = bins of slots (2j, 2j+1) of thread t,
//   scaled by `scale` (a power of two; the pair kernel's product then carries 1/L exactly).
template <bool U8>
__global__ __launch_bounds__(kThreads, 4) void k_fwd(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                     const float4* __restrict__ tw1_g,
                                                     const float2* __restrict__ tw2_g, long first_item,
                                                     float scale, const float2* __restrict__ rot, int wrap_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* xl = reinterpret_cast<float2*>(smem);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLdsXchg);
    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    // wrap_items > 0 (rmx_caf_batch, all hypotheses in one launch): workgroup b transforms real item b % wrap_items
    // de-rotated by hypothesis b / wrap_items; its spectrum still goes to slot b
    const long item = first_item + (wrap_items > 0 ? (long)(blockIdx.x % (unsigned)wrap_items) : (long)blockIdx.x);
    if (wrap_items > 0 && rot) rot += (size_t)(blockIdx.x / (unsigned)wrap_items) * kM;

    load_tw2_to_lds(tw2_lds, tw2_g, t);
    float2 tw1[16];
    load_tw1(tw1, tw1_g, t);

    float2 v[16];
    if constexpr (U8) {
        const uchar2* x = reinterpret_cast<const uchar2*>(iq_v) + item * kM;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uchar2 b = x[q * 256 + u];
            v[q] = make_float2((float)b.x - 127.5f, (float)b.y - 127.5f);
        }
    } else {
        const float2* x = reinterpret_cast<const float2*>(iq_v) + item * kM;
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = x[q * 256 + u];
    }
    if (rot) {   // rmx_caf_batch: the window de-rotated by this Doppler hypothesis (rounded as numpy rounds it)
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = gen::rot_mul(v[q], rot[q * 256 + u]);
    }
    // odd sub-transform: x[n] * W_L^n = x * W32^q * W_L^u ; W_L^u is folded into tw1 (odd lanes)
    if (p) {
#pragma unroll
        for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], w32(q));
    }
    dft16(v);        // n2 -> k0
    mul_tw1(v, tw1); // W_M^(u*k0) [* W_L^u on odd lanes]
    xchg_a_write(xl, v, t);
    __syncthreads();
    xchg_b_read(xl, v, t);
    dft16(v);                         // n1 -> k1
    mul_tw2(v, tw2_lds, u & 15);      // W_256^(n0*k1)
    xchg_bc_write_b(xl, v, t);        // own half-wave region: no barrier
    wave_lds_fence();
    xchg_bc_read_c(xl, v, t);
    dft16(v);                         // n0 -> k2
    float4* out = spec + (long)blockIdx.x * (8 * kThreads);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        out[j * kThreads + t] = make_float4(v[2 * j].x * scale, v[2 * j].y * scale,
                                            v[2 * j + 1].x * scale, v[2 * j + 1].y * scale);
}

// ------------------------------------------------------------------------------------------------
// Pair kernel.  grid = n_windows_in_chunk * n_parts; each workgroup walks items[part_begin..end).
//   out arrays are indexed [(first_window + wl) * n_pairs + item.out].
// Streaming variant (k_pair_str, option "resident" = 0): two workgroups per CU (<= 128 VGPRs); TW1,
// X_i and X_j are re-read every pair.  The resident variant is k_pair_res below.
__device__ __forceinline__ void pair_body(const float4* __restrict__ spec, const float4* __restrict__ spec_j,
                                          const float4* __restrict__ tw1_g,
                                          const float2* __restrict__ tw2_g, const PairItem* __restrict__ items,
                                          const int* __restrict__ part_begin, int n_parts, int n_buoys,
                                          int n_pairs, int xcd_map, long first_window, float out_scale,
                                          int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                          float* __restrict__ peak, int i_wrap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* xl = reinterpret_cast<float2*>(smem);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLdsXchg);
    float* red_max = reinterpret_cast<float*>(smem + kLdsXchg + kLdsTw2);      // [8]
    int* red_k = reinterpret_cast<int*>(smem + kLdsXchg + kLdsTw2 + 32);       // [1]
    float* red_tap = reinterpret_cast<float*>(smem + kLdsXchg + kLdsTw2 + 48); // [3]

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;

    // blockIdx -> (window, part).  Workgroups of one window share its spectra through the XCD's
    // L2, so keep them on one XCD (blocks b and b+8 share an XCD) and adjacent in dispatch order.
    // Placement only changes speed, never results.
    int wl, part;
    {
        const int b = blockIdx.x;
        if (xcd_map) {
            const int xcd = b & 7, s = b >> 3;
            wl = (s / n_parts) * 8 + xcd;
            part = s % n_parts;
        } else {
            wl = b / n_parts;
            part = b % n_parts;
        }
    }

    load_tw2_to_lds(tw2_lds, tw2_g, t);
    __syncthreads();

    const float sgn = p ? -1.0f : 1.0f;
    const int it_begin = part_begin[part];
    const int it_end = part_begin[part + 1];
    const long wbase = (long)wl * n_buoys;
    const long wbase_i = (long)(i_wrap > 0 ? wl % i_wrap : wl) * n_buoys;   // (rmx_caf_batch: wl = hypothesis * windows + window)
    float4 sa[8], sb[8];   // X_i (anchor) and X_j of the pair about to be processed
    PairItem pi = items[it_begin < it_end ? it_begin : 0];
    if (it_begin < it_end) {
        const float4* xi = spec + (wbase_i + pi.i) * (8 * kThreads);
        const float4* xj = spec_j + (wbase + pi.j) * (8 * kThreads);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sa[j] = xi[j * kThreads + t];
            sb[j] = xj[j * kThreads + t];
        }
    }
    for (int it = it_begin; it < it_end; ++it) {
        const int out_idx = pi.out;
        float2 v[16];
        // R = X_j * conj(X_i), written (im, re)-swapped: the forward blocks below then compute the
        // inverse transform (swap o F o swap = conj F).
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = sa[j];
            const float4 b = sb[j];
            v[2 * j] = make_float2(b.y * a.x - b.x * a.y, b.x * a.x + b.y * a.y);
            v[2 * j + 1] = make_float2(b.w * a.z - b.z * a.w, b.z * a.z + b.w * a.w);
        }
        dft16(v);                     // k2 -> n0   (role C)
        mul_tw2(v, tw2_lds, u & 15);  // W_256^(k1*n0)
        xchg_bc_write_c(xl, v, t);
        wave_lds_fence();
        xchg_bc_read_b(xl, v, t);
        dft16(v);                     // k1 -> n1   (role B)
        float2 tw1s[16];
        {
            // TW1 re-read every pair (64 KiB per workgroup from L2); the pointer is laundered so
            // that the loads stay here, in flight across the exchange.
            const float4* twp = tw1_g;
            asm volatile("" : "+s"(twp));
            load_tw1(tw1s, twp, t);
        }
        xchg_b_write(xl, v, t);       // into this half wave's own region: no barrier needed before
        __syncthreads();
        xchg_a_read(xl, v, t);
        mul_tw1(v, tw1s);             // W_M^(u*k0) [* W_L^u odd]
        dft16(v);                     // k0 -> n2   (role A): lane holds e[n] (p=0) or o[n]*W_L^u (p=1)
        if (p) {
#pragma unroll
            for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], w32(q));
        }
        // last radix-2 stage across the lane pair: even lane r[n] = e + o', odd lane r[n+M] = e - o'
        float mag[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float rx = sgn * v[q].x + dpp_xor1(v[q].x);
            const float ry = sgn * v[q].y + dpp_xor1(v[q].y);
            mag[q] = rx * rx + ry * ry;
        }
        {
            if (it + 1 < it_end) {
                pi = items[it + 1];
                const float4* xi = spec + (wbase_i + pi.i) * (8 * kThreads);
                const float4* xj = spec_j + (wbase + pi.j) * (8 * kThreads);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sa[j] = xi[j * kThreads + t];
                    sb[j] = xj[j * kThreads + t];
                }
            }
        }
        // 'full' order index of slot q: even lanes lag tau = n >= 0 -> k = n + M - 1;
        // odd lanes tau = n - M -> k = n - 1 (n = 0, i.e. tau = -M, is not part of 'full').
        if (p && u == 0) mag[0] = -1.0f;
        float tmax = mag[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) tmax = fmaxf(tmax, mag[q]);
        float wmax = tmax;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
        if (lane == 0) red_max[wave] = wmax;
        if (t == 0) *red_k = 0x7fffffff;
        __syncthreads();
        float gmax = red_max[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) gmax = fmaxf(gmax, red_max[w]);
        const int kbase = p ? (u - 1) : (u + kM - 1);
        if (tmax == gmax) {
            int kmin = 0x7fffffff;
#pragma unroll
            for (int q = 15; q >= 0; --q)
                if (mag[q] == gmax) kmin = kbase + q * 256;
            atomicMin(red_k, kmin);
        }
        __syncthreads();
        const int kstar = *red_k;
        // owners of taps k*-1, k*, k*+1 publish |r| (scipy scaling)
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
            const int kk = kstar + d;
            if (kk >= 0 && kk <= 2 * kM - 2) {
                const int par = (kk >= kM - 1) ? 0 : 1;
                const int n = par ? (kk + 1) : (kk - (kM - 1));
                if (p == par && u == (n & 255)) {
                    const int qo = n >> 8;
                    float val = 0.0f;
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (q == qo) val = mag[q];
                    red_tap[d + 1] = sqrtf(val) * out_scale;
                }
            }
        }
        __syncthreads();
        if (t == 0) {
            const double b = (double)red_tap[1];
            double frac = 0.0;
            if (kstar > 0 && kstar < 2 * kM - 2) {
                const double a = (double)red_tap[0], c = (double)red_tap[2];
                const double den = a - 2.0 * b + c;
                if (den != 0.0) frac = 0.5 * (a - c) / den;
            }
            const long o = (first_window + wl) * (long)n_pairs + out_idx;
            lag_int[o] = kstar - (kM - 1);
            lag_frac[o] = (float)frac;
            peak[o] = (float)b;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Resident pair kernel (one workgroup per CU, <= 256 VGPRs).  One workgroup barrier per pair:
//   pair n:  product -> [request X_j of pair n+1] -> DFT16 -> TW2 -> wave-local exchange -> DFT16 ->
//            write A<->B image -> BARRIER -> [resolve pair n-1] -> read image -> TW1 -> DFT16 -> W32 ->
//            lane-pair butterfly, |.|^2 -> publish: all |.|^2 to an LDS tap buffer, wave winner
//            (max, lowest 'full' index) to an LDS slot.
// The cross-wave part of the argmax and the 3-tap parabola of pair n are "resolved" by one lane
// after the barrier of pair n+1 (both LDS buffers are double buffered by pair parity), so the
// reduction's latency chain hides behind the next pair's arithmetic.
constexpr int kLdsMag = kL * 4;                         // one |.|^2 image: [q4][t] float4
constexpr int kLdsResOff = kLdsXchg + kLdsTw2;
constexpr int kLdsResBytes = kLdsXchg + kLdsTw2 + 2 * kLdsMag + 256;

__device__ __forceinline__ void resolve_pair(const float* magbuf, const float2* red, long out_pos, float out_scale,
                                             int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                             float* __restrict__ peak) {
    // red[w] = (wave max of |r|^2 as float bits, lowest 'full' index attaining it), w = 0..7
    float gmax = -2.0f;
    int kstar = 0x7fffffff;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float2 e = red[w];
        const float m = e.x;
        const float ey = e.y;
        const int k = __builtin_bit_cast(int, ey);
        if (m > gmax || (m == gmax && k < kstar)) { gmax = m; kstar = k; }
    }
    auto tap = [&](int kk) -> float {
        const int par = (kk >= kM - 1) ? 0 : 1;
        const int n = par ? (kk + 1) : (kk - (kM - 1));
        const int tt = 2 * (n & 255) + par, q = n >> 8;
        return sqrtf(magbuf[((q >> 2) * kThreads + tt) * 4 + (q & 3)]) * out_scale;
    };
    const float b = sqrtf(gmax) * out_scale;
    float frac = 0.0f;
    if (kstar > 0 && kstar < 2 * kM - 2) {
        const float a = tap(kstar - 1), c = tap(kstar + 1);
        const double den = (double)a - 2.0 * (double)b + (double)c;
        if (den != 0.0) frac = (float)(0.5 * ((double)a - (double)c) / den);
    }
    lag_int[out_pos] = kstar - (kM - 1);
    lag_frac[out_pos] = frac;
    peak[out_pos] = b;
}

__global__ __launch_bounds__(kThreads, 2) void k_pair_res(
    const float4* __restrict__ spec, const float4* __restrict__ spec_j, const float4* __restrict__ tw1_g, const float2* __restrict__ tw2_g,
    const PairItem* __restrict__ items, const int* __restrict__ part_begin, int n_parts, int n_buoys, int n_pairs,
    int xcd_map, long first_window, float out_scale, int* __restrict__ lag_int, float* __restrict__ lag_frac,
    float* __restrict__ peak, int dbg_rt, int i_wrap) {
#ifdef RMX_ABLATE
    const int dbg = dbg_rt;   // timing-only ablation build (wrong results), as in k_win
#else
    constexpr int dbg = 0;
    (void)dbg_rt;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* xl = reinterpret_cast<float2*>(smem);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLdsXchg);
    float4* magbuf = reinterpret_cast<float4*>(smem + kLdsResOff);                 // [2][4][512] float4
    float2* red = reinterpret_cast<float2*>(smem + kLdsResOff + 2 * kLdsMag);      // [2][8]

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;
    int wl, part;
    {
        const int b = blockIdx.x;
        if (xcd_map) {
            const int xcd = b & 7, s = b >> 3;
            wl = (s / n_parts) * 8 + xcd;
            part = s % n_parts;
        } else {
            wl = b / n_parts;
            part = b % n_parts;
        }
    }
    load_tw2_to_lds(tw2_lds, tw2_g, t);
    float2 tw1[16];
    load_tw1(tw1, tw1_g, t);
    __syncthreads();

    const float sgn = p ? -1.0f : 1.0f;
    const int kbase = p ? (u - 1) : (u + kM - 1);
    const int it_begin = part_begin[part];
    const int it_end = part_begin[part + 1];
    if (it_begin >= it_end) return;
    const long wbase = (long)wl * n_buoys;
    const long wbase_i = (long)(i_wrap > 0 ? wl % i_wrap : wl) * n_buoys;   // (rmx_caf_batch: wl = hypothesis * windows + window)
    const long obase = (first_window + wl) * (long)n_pairs;
    float4 sa[8], sb[8];
    PairItem pi = items[it_begin];
    int cur_i = pi.i;
    {
        const float4* xi = spec + (wbase_i + pi.i) * (8 * kThreads);
        const float4* xj = spec_j + (wbase + pi.j) * (8 * kThreads);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sa[j] = xi[j * kThreads + t];
            sb[j] = xj[j * kThreads + t];
        }
    }
    int prev_out = -1;
    for (int it = it_begin; it < it_end; ++it) {
        const int out_idx = pi.out;
        const int buf = (it - it_begin) & 1;
        float2 v[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = sa[j];
            const float4 b = sb[j];
            v[2 * j] = make_float2(b.y * a.x - b.x * a.y, b.x * a.x + b.y * a.y);
            v[2 * j + 1] = make_float2(b.w * a.z - b.z * a.w, b.z * a.z + b.w * a.w);
        }
        if (it + 1 < it_end && !(dbg & 16)) {   // request the next pair's spectra: a whole pair of compute hides it
            pi = items[it + 1];
            const float4* xj = spec_j + (wbase + pi.j) * (8 * kThreads);
#pragma unroll
            for (int j = 0; j < 8; ++j) sb[j] = xj[j * kThreads + t];
            if (pi.i != cur_i) {
                cur_i = pi.i;
                const float4* xi = spec + (wbase_i + pi.i) * (8 * kThreads);
#pragma unroll
                for (int j = 0; j < 8; ++j) sa[j] = xi[j * kThreads + t];
            }
        }
        dft16(v);                     // k2 -> n0   (role C)
        mul_tw2(v, tw2_lds, u & 15);  // W_256^(k1*n0)
        if (!(dbg & 4)) {
        xchg_bc_write_c(xl, v, t);
        wave_lds_fence();
        xchg_bc_read_b(xl, v, t);
        }
        dft16(v);                     // k1 -> n1   (role B)
        if (!(dbg & 8)) xchg_b_write(xl, v, t);       // into this half wave's own region
        if (!(dbg & 1)) __syncthreads();              // the pair's only barrier; also publishes pair it-1's winners
        if (!(dbg & 2) && prev_out >= 0 && t == ((it - it_begin) & 7) * 64)
            resolve_pair(reinterpret_cast<const float*>(magbuf + (buf ^ 1) * (4 * kThreads)), red + (buf ^ 1) * 8,
                         obase + prev_out, out_scale, lag_int, lag_frac, peak);
        if (!(dbg & 8)) xchg_a_read(xl, v, t);
        mul_tw1(v, tw1);              // W_M^(u*k0) [* W_L^u on odd lanes]
        dft16(v);                     // k0 -> n2   (role A): e[n] (even lanes) / o[n]*W_L^u (odd lanes)
        if (p) {
#pragma unroll
            for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], w32(q));
        }
        float mag[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float rx = sgn * v[q].x + dpp_xor1(v[q].x);
            const float ry = sgn * v[q].y + dpp_xor1(v[q].y);
            mag[q] = rx * rx + ry * ry;
        }
        if (p && u == 0) mag[0] = -1.0f;   // lag -M is not part of the 'full' output
        if (dbg & 2) { float s = 0; 
#pragma unroll
            for (int q = 0; q < 16; ++q) s += mag[q];
            if (s == 12345.678f) lag_int[0] = 1; prev_out = out_idx; continue; }
        float4* mb = magbuf + buf * (4 * kThreads) + t;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            mb[q4 * kThreads] = make_float4(mag[4 * q4], mag[4 * q4 + 1], mag[4 * q4 + 2], mag[4 * q4 + 3]);
        float tmax = mag[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) tmax = fmaxf(tmax, mag[q]);
        int kq = 0;
#pragma unroll
        for (int q = 15; q >= 0; --q)
            if (mag[q] == tmax) kq = kbase + q * 256;   // lowest 'full' index of this lane's maximum
        const float wmax = wave_max_f32(tmax);
        const int kw = wave_min_i32(tmax == wmax ? kq : 0x7fffffff);
        if (lane == 0) red[buf * 8 + wave] = make_float2(wmax, __builtin_bit_cast(float, kw));
        prev_out = out_idx;
    }
    __syncthreads();
    if (t == 0) {
        const int buf = (it_end - 1 - it_begin) & 1;
        resolve_pair(reinterpret_cast<const float*>(magbuf + buf * (4 * kThreads)), red + buf * 8, obase + prev_out,
                     out_scale, lag_int, lag_frac, peak);
    }
}


#ifdef RMX_EXPERIMENTS
// k_win on packed fp32 (winpk.hpp): same protocol, same resolve routine
template <bool U8>
__global__ __launch_bounds__(kThreads, 2) void k_winp(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                      const float4* __restrict__ tw1_g,
                                                      const float2* __restrict__ tw2_g, int n_buoys,
                                                      long first_window, float out_scale, int* __restrict__ lag_int,
                                                      float* __restrict__ lag_frac, float* __restrict__ peak, int n_win) {
    pk::winp_body<U8>(iq_v, spec, tw1_g, tw2_g, n_buoys, first_window, out_scale, lag_int, lag_frac, peak, n_win,
                      [](int lane, const float4* red, const float* halo, const int* oidx, int first, int cnt, long obase,
                         float osc, int* li, float* lf, float* pk_) __attribute__((always_inline)) {
                          resolve_batch(lane, red, halo, oidx, first, cnt, obase, osc, li, lf, pk_);
                      });
}
#endif

#define RMX_PAIR_ARGS                                                                                         \
    const float4 *__restrict__ spec, const float4 *__restrict__ spec_j, const float4 *__restrict__ tw1_g,     \
        const float2 *__restrict__ tw2_g,                                                                     \
        const PairItem *__restrict__ items, const int *__restrict__ part_begin, int n_parts, int n_buoys,     \
        int n_pairs, int xcd_map, long first_window, float out_scale, int *__restrict__ lag_int,              \
        float *__restrict__ lag_frac, float *__restrict__ peak, int i_wrap
#define RMX_PAIR_PASS                                                                                         \
    spec, spec_j, tw1_g, tw2_g, items, part_begin, n_parts, n_buoys, n_pairs, xcd_map, first_window, out_scale,      \
        lag_int, lag_frac, peak, i_wrap

__global__ __launch_bounds__(kThreads, 4) void k_pair_str(RMX_PAIR_ARGS) { pair_body(RMX_PAIR_PASS); }


// ---- CAF helper: running best hypothesis per pair-window -------------------------------------------
__global__ void k_caf_select(int d, long first, long n, const int* __restrict__ lag_d, const float* __restrict__ frac_d,
                             const float* __restrict__ peak_d, int* __restrict__ dop, int* __restrict__ lag,
                             float* __restrict__ frac, float* __restrict__ peak) {
    const long i = first + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + n) return;
    if (d == 0 || peak_d[i] > peak[i]) {   // strict: ties keep the lowest d
        dop[i] = d; lag[i] = lag_d[i]; frac[i] = frac_d[i]; peak[i] = peak_d[i];
    }
}


// the same over all hypotheses of one launch: arrays [n_dop][n], d-major first maximum (strict >: ties keep the lowest d)
__global__ void k_caf_select_all(int n_dop, long n, const int* __restrict__ lag_d, const float* __restrict__ frac_d,
                                 const float* __restrict__ peak_d, int* __restrict__ dop, int* __restrict__ lag,
                                 float* __restrict__ frac, float* __restrict__ peak) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bd = 0;
    float bp = peak_d[i];
    for (int d = 1; d < n_dop; ++d) {
        const float pd = peak_d[(long)d * n + i];
        if (pd > bp) { bp = pd; bd = d; }
    }
    dop[i] = bd; lag[i] = lag_d[(long)bd * n + i]; frac[i] = frac_d[(long)bd * n + i]; peak[i] = bp;
}


// ---- batched hyperbolic position solve (rmx_solve_batch) --------------------------------------------
// One thread per window: 3 unknowns, P residuals, everything in float64 registers; the buoy table and
// the pair list sit in LDS.  The lag arrays are read transposed-strided (lane = window), which is a
// few hundred bytes per window and iteration: the kernel is latency-bound by its dependent
// sqrt/divide chains, not by memory.
constexpr int kSolveMaxBuoys = 64;
constexpr int kSolveMaxPairs = kSolveMaxBuoys * (kSolveMaxBuoys - 1) / 2;
__global__ __launch_bounds__(64) void k_solve(const double* __restrict__ buoy_xyz, int n_buoys,
                                              const int* __restrict__ pairs, int n_pairs,
                                              const int* __restrict__ lag_int, const float* __restrict__ lag_frac,
                                              const float* __restrict__ weight, double metres_per_sample,
                                              int n_windows, int max_iter, double* __restrict__ pos,
                                              double* __restrict__ cost, int* __restrict__ iters) {
#pragma clang fp contract(off)   // same roundings as the numpy restatement wherever the order is the same
    __shared__ double sb[kSolveMaxBuoys * 3];
    __shared__ short spair[kSolveMaxPairs * 2];
    for (int i = threadIdx.x; i < n_buoys * 3; i += blockDim.x) sb[i] = buoy_xyz[i];
    for (int i = threadIdx.x; i < n_pairs * 2; i += blockDim.x) spair[i] = (short)pairs[i];
    __syncthreads();
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_windows) return;
    const int* li = lag_int + (long)w * n_pairs;
    const float* lf = lag_frac + (long)w * n_pairs;
    const float* wg = weight ? weight + (long)w * n_pairs : nullptr;
    double px = 0, py = 0, pz = 0;
    for (int b = 0; b < n_buoys; ++b) { px += sb[3 * b]; py += sb[3 * b + 1]; pz += sb[3 * b + 2]; }
    px /= n_buoys; py /= n_buoys; pz /= n_buoys;
    auto f_at = [&](double x, double y, double z) -> double {
        double f = 0;
        for (int q = 0; q < n_pairs; ++q) {
            const double* b1 = sb + 3 * spair[2 * q];
            const double* b2 = sb + 3 * spair[2 * q + 1];
            const double n1 = sqrt((x - b1[0]) * (x - b1[0]) + (y - b1[1]) * (y - b1[1]) + (z - b1[2]) * (z - b1[2]));
            const double n2 = sqrt((x - b2[0]) * (x - b2[0]) + (y - b2[1]) * (y - b2[1]) + (z - b2[2]) * (z - b2[2]));
            const double d = ((double)li[q] + (double)lf[q]) * metres_per_sample;
            const double r = n2 - n1 - d;
            f += (wg ? (double)wg[q] : 1.0) * r * r;
        }
        return f;
    };
    double lam = 1e-3;
    double f = f_at(px, py, pz);
    int it = 0;
    while (it < max_iter) {
        ++it;
        double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, g0 = 0, g1 = 0, g2 = 0;
        for (int q = 0; q < n_pairs; ++q) {
            const double* b1 = sb + 3 * spair[2 * q];
            const double* b2 = sb + 3 * spair[2 * q + 1];
            const double v1x = px - b1[0], v1y = py - b1[1], v1z = pz - b1[2];
            const double v2x = px - b2[0], v2y = py - b2[1], v2z = pz - b2[2];
            const double n1 = sqrt(v1x * v1x + v1y * v1y + v1z * v1z);
            const double n2 = sqrt(v2x * v2x + v2y * v2y + v2z * v2z);
            const double d = ((double)li[q] + (double)lf[q]) * metres_per_sample;
            const double r = n2 - n1 - d;
            const double jx = v2x / n2 - v1x / n1, jy = v2y / n2 - v1y / n1, jz = v2z / n2 - v1z / n1;
            const double ww = wg ? (double)wg[q] : 1.0;
            a00 += ww * jx * jx; a01 += ww * jx * jy; a02 += ww * jx * jz;
            a11 += ww * jy * jy; a12 += ww * jy * jz; a22 += ww * jz * jz;
            g0 += ww * jx * r; g1 += ww * jy * r; g2 += ww * jz * r;
        }
        // (A + lam diag A) delta = -g by Cholesky; a failed factorisation counts as a rejected step
        const double d00 = a00 * (1.0 + lam), d11 = a11 * (1.0 + lam), d22 = a22 * (1.0 + lam);
        bool ok = d00 > 0.0;
        const double l00 = sqrt(ok ? d00 : 1.0);
        const double l10 = a01 / l00, l20 = a02 / l00;
        const double t11 = d11 - l10 * l10;
        ok = ok && t11 > 0.0;
        const double l11 = sqrt(ok ? t11 : 1.0);
        const double l21 = (a12 - l20 * l10) / l11;
        const double t22 = d22 - l20 * l20 - l21 * l21;
        ok = ok && t22 > 0.0;
        const double l22 = sqrt(ok ? t22 : 1.0);
        bool accepted = false;
        double dn = 0.0;
        if (ok) {
            const double y0 = -g0 / l00;
            const double y1 = (-g1 - l10 * y0) / l11;
            const double y2 = (-g2 - l20 * y0 - l21 * y1) / l22;
            const double dz = y2 / l22;
            const double dy = (y1 - l21 * dz) / l11;
            const double dx = (y0 - l10 * dy - l20 * dz) / l00;
            const double fn = f_at(px + dx, py + dy, pz + dz);
            if (fn < f) {
                px += dx; py += dy; pz += dz;
                f = fn;
                accepted = true;
                dn = sqrt(dx * dx + dy * dy + dz * dz);
            }
        }
        if (accepted) {
            lam = lam / 3.0 > 1e-12 ? lam / 3.0 : 1e-12;
            if (dn < 1e-4) break;
        } else {
            lam *= 4.0;
            if (lam > 1e12) break;
        }
    }
    pos[3 * (long)w] = px; pos[3 * (long)w + 1] = py; pos[3 * (long)w + 2] = pz;
    cost[w] = f;
    iters[w] = it;
}

// ================================================================================================
// host side
// ================================================================================================
static thread_local std::string g_create_error;

}  // namespace rmx

struct rmx_ctx {
    rmx::host::Knobs knobs;   // kernel-selection options as they were at rmx_create (host_plan.hpp)
    int device = 0;
    int n_cus = 256;   // compute units of the device (persistent-grid size of the fused kernel)
    int n_buoys = 0, n_samples = 0, max_windows = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int chunk_windows = 0;
    int pairs_per_block = 7;
    bool ppb_user = false;  // set through rmx_set_option: the small-batch rule then leaves it alone
    bool small_batch = true; // option small4096: few windows of N = 4096 through the per-transform kernels (see rmx_xcorr_batch)
    bool timing = false;
    int dbg = 0;
    bool resident = true;
    int pk = 0;             // fused path: 1 = k_winp (k_win on packed fp32)
    int win8 = 0;           // fused path: 1 = k_win8 (8 points x 1024 threads, 4 waves/SIMD), 0 = k_win
    int stag = 1;           // k_win8: which waves run the two halves between barriers in the opposite order
    float4* d_tw1_8 = nullptr;
    float2* d_tb8 = nullptr;
    float2* d_tc8 = nullptr;
    bool fused = true;      // one kernel per chunk: forward spectra + pairs in one workgroup per window   // pair kernel variant: 1 workgroup/CU with resident tables
    // device buffers
    float4* d_spec = nullptr;
    float4* d_tw1 = nullptr;
    float2* d_tw2 = nullptr;
    rmx::PairItem* d_items = nullptr;
    int* d_part_begin = nullptr;
    void* d_in = nullptr;      size_t d_in_bytes = 0;
    int* d_lag = nullptr;      float* d_frac = nullptr;  float* d_peak = nullptr;  size_t d_out_elems = 0;   // ONE block: lag | frac | peak
    int* d_dop = nullptr;
    void* h_out = nullptr;     size_t h_out_bytes = 0;   // pinned staging of small host-pointer results (fetch_out)
    size_t spec_bytes = 0, scratch_bytes = 0;
    // generic path (n_samples != 4096): see generic_path.hpp
    bool generic = false;
    int g_logL = 0, g_logL1 = 0, g_logL2 = 0, g_lo_bits = 0, g_chunk = 0;
    size_t g_dyn_bytes = 0, g_pair_bytes = 0;   // chunk-sized buffers of generic_ensure (all of them / the pair-dependent ones)
    bool g_fused = false;      // four-step: both row passes in g_rows_fused (n_buoys <= 4, plain batches)
    bool g_fused_always = false;
    const void* g_fused_fn = nullptr;
    const void* g_fused_def_fn = nullptr;  // the same for the default plan (pair loop unrolled)
    bool g_wfused = false;     // LDS-resident lengths, n_buoys <= 4: whole windows in g_win_fused (no spectra in HBM)
    const void* g_wf_fn[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [default plan][u8]
    size_t g_wf_lds[2] = {0, 0};           // [default plan]
    bool g_wscr = false;       // 512 <= L <= 16384, any buoy count: whole windows in g_win_scr (spectra in a cache-resident scratch)
    bool g_wscr_always = false;
    const void* g_ws_fn[2] = {nullptr, nullptr};   // [u8]
    size_t g_ws_lds = 0;
    int g_ws_thr = 0, g_ws_upw = 0, g_ws_grid = 0;
    float4* g_ws_scratch = nullptr;
    float2* g_tw_win = nullptr;            // W_L half table of that kernel
    float2* g_tw_l = nullptr;              // g_win_eo15 (N = 16384): W_32768^i, i < 1024
    bool g_k8 = false;                     // N = 8192: k_win8kl (kwin8k.hpp) instead of g_win_scr14 for batches that fill the chip
    int g_k8_kind = 1;                     // 1 = k_win8kl (LDS-resident anchor half), 2 = k_win8k (experiments build)
    float4* g_k8_tw1 = nullptr;            // its TW1 tables (both halves)
    float2* g_k8_tw2 = nullptr;            // k_win's TW2 table
    int g_k16 = 0;                         // N = 16384: k16_fwd + k16_pairs (kwin16k.hpp); 1 = from the measured batch size on, 2 = always
    float4* g_k16_tw1 = nullptr;           // slot factors of TW1
    float2 *g_k16_gq = nullptr, *g_k16_tw2 = nullptr, *g_k16_tws = nullptr;   // per-quarter thread factors, k_win's TW2, W_128 rows
    const void* g_cols_inv_fn = nullptr;
    const void* g_cols_fwd_fn[2] = {nullptr, nullptr};   // [u8]
    const void* g_rows_inv_fn = nullptr;
    const void* g_rows_anchor_fn = nullptr;   // inverse rows with a resident anchor (default pair list), or null
    int g_rows_anchor_min_b = 6;              // ... from this many buoys on (RMX_ROWS_ANCHOR=n sets it, 0 = never)
    const void* g_rows_fwd_fn = nullptr;   // g_rows<inverse> compiled for this row length (or the run-time one)
    float2 *g_tw = nullptr, *g_tw1 = nullptr, *g_tw2 = nullptr, *g_thi = nullptr, *g_tlo = nullptr;
    float2 *g_spec = nullptr, *g_spec_r = nullptr, *g_prod = nullptr;   // spectra, de-rotated spectra (CAF), products
    rmx::gen::GTile* g_rec = nullptr;   // per column tile: partial argmax + taps
    float* g_halo = nullptr;            // per column tile: |r|^2 of its two edge columns
    rmx::gen::GPair* g_pairs = nullptr;
    long g_slots_alloc = 0;
    int g_pairs_n = -1;
    std::vector<int32_t> g_pairs_plan;
    // host-pointer input of the fused path: copies pipelined against the kernels on a second stream
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // rmx_detect_batch: twiddle table of the last window length, dB spectra, staging
    float2* dt_tw = nullptr;  int dt_logn = 0;  float* dt_pdb = nullptr;  size_t dt_pdb_bytes = 0;
    void* dt_in = nullptr;  size_t dt_in_bytes = 0;  void* dt_out = nullptr;  size_t dt_out_bytes = 0;
    // rmx_solve_batch work buffers
    double* sv_buoys = nullptr;  int* sv_pairs = nullptr;  size_t sv_pairs_cap = 0;
    void* sv_in = nullptr;  size_t sv_in_bytes = 0;  void* sv_out = nullptr;  size_t sv_out_bytes = 0;
    // CAF (rmx_caf_batch): de-rotated spectra of the N = 4096 path, phasor table, per-hypothesis results
    float4* d_spec_r = nullptr;  size_t spec_r_bytes = 0;
    float2* caf_rot = nullptr;  size_t caf_rot_elems = 0;
    int* caf_lag = nullptr;  float* caf_frac = nullptr;  float* caf_peak = nullptr;  int* caf_dop = nullptr;
    size_t caf_out_elems = 0;
    std::vector<double> caf_grid;
    // cached pair plan
    std::vector<int32_t> plan_pairs;
    int plan_n_pairs = -1, plan_n_parts = 0, plan_ppb = 0;
    bool plan_all_pairs = false;   // the plan is the default list: all i<j in nested-loop order
    // timing
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_kind;  // 0 fwd, 1 pair (per launch: ev[2k], ev[2k+1])
    size_t ev_used = 0;
    float t_fwd = 0, t_pair = 0;
    int n_fwd = 0, n_pair = 0;
    std::string err;
};

namespace rmx {

static int fail(rmx_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define RMX_HIP(c, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), RMX_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                      \
    } while (0)

using host::is_pow2;


#ifdef RMX_EXPERIMENTS
// tables of k_win8 (win8.hpp; the same values tools/model_win8.py checks against numpy's FFT)
static void build_tables8(std::vector<float4>& tw1, std::vector<float2>& tb, std::vector<float2>& tc) {
    const double two_pi = 6.283185307179586476925286766559;
    tw1.resize(4 * w8::kT8);
    for (int T = 0; T < w8::kT8; ++T) {
        const int p = T & 1, t1 = T >> 1;
        float2 w[8];
        for (int c0 = 0; c0 < 8; ++c0) {
            double ang = -two_pi * (double)((t1 * c0) % kM) / (double)kM;      // W_M^(c0 t1)
            if (p) ang += -two_pi * (double)t1 / (double)kL;                    // * W_L^t1 on odd lanes
            w[c0] = make_float2((float)(std::cos(ang) * kTw1Scale), (float)(std::sin(ang) * kTw1Scale));
        }
        for (int j = 0; j < 4; ++j) tw1[j * w8::kT8 + T] = make_float4(w[2 * j].x, w[2 * j].y, w[2 * j + 1].x, w[2 * j + 1].y);
    }
    // rows in the order the first butterfly layer consumes them: w4, w1, w5, w2, w6, w3, w7 (w0 = 1 not stored)
    const int order[7] = {4, 1, 5, 2, 6, 3, 7};
    auto fill = [&](std::vector<float2>& t, int rows, int n) {
        t.assign((size_t)rows * w8::kRowF2, make_float2(0.0f, 0.0f));
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < 7; ++k) {
                const double ang = -two_pi * (double)((r * order[k]) % n) / (double)n;
                t[(size_t)r * w8::kRowF2 + k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
            }
    };
    fill(tb, 64, 512);   // W_512^(c1 * lane), lane = n0 + 8 n1
    fill(tc, 8, 64);     // W_64^(d0 * n0)
}
#endif

static int ensure_events(rmx_ctx* c, size_t n) {
    while (c->ev.size() < n) {
        hipEvent_t e;
        RMX_HIP(c, hipEventCreate(&e));
        c->ev.push_back(e);
    }
    return RMX_OK;
}

// Per-launch timing (option "timing"): every kernel launch of a call is bracketed by two HIP events on the launch stream
// and tagged with its kernel family; rmx_last_timing / rmx_last_timing_kind read them back.  The event pool grows in
// front of every record (a budget computed ahead of the chunk loop was too small once every chunk could take
// the partial-round route), so no call pattern can index past it.
enum TimeKind {
    kTkFwd4096 = 0,      // k_fwd
    kTkPair4096 = 1,     // k_win (fused forward + pairs) or k_pair_res / k_pair_str
    kTkColsFwd = 2,      // g_cols_fwd
    kTkRowsFwd = 3,      // g_rows (forward)
    kTkRowsFused = 4,    // g_rows_fused (forward rows, products, inverse rows)
    kTkRowsAnchor = 5,   // g_rows_anchor
    kTkRowsInv = 6,      // g_rows (product, inverse)
    kTkColsInv = 7,      // g_cols_inv
    kTkFinal = 8,        // g_final
    kTkWindow = 9,       // g_win_fused / g_win_scr / g_win_scr14 / g_win_eo15: whole windows in one kernel
    kTkFwdSmall = 10,    // g_fwd_small
    kTkPairSmall = 11,   // g_pair_small
    kTkCafSelect = 12,   // k_caf_select / k_caf_select_all
    kTkFwd16k = 13,      // k16_fwd
    kTkPairs16k = 14,    // k16_pairs
    kTkCount = 15
};
static const char* const kTimeKindName[kTkCount] = {
    "k_fwd", "k_win|k_pair", "g_cols_fwd", "g_rows_fwd", "g_rows_fused", "g_rows_anchor", "g_rows_inv", "g_cols_inv",
    "g_final", "g_win_*", "g_fwd_small", "g_pair_small", "k_caf_select", "k16_fwd", "k16_pairs"};
static int tm_begin(rmx_ctx* c) {
    if (!c->timing) return RMX_OK;
    const int rc = ensure_events(c, c->ev_used + 2);
    if (rc != RMX_OK) return rc;
    RMX_HIP(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    return RMX_OK;
}
static int tm_end(rmx_ctx* c, int kind) {
    if (!c->timing) return RMX_OK;
    RMX_HIP(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
    c->ev_used += 2;
    c->ev_kind.push_back(kind);
    return RMX_OK;
}
static void tm_reset(rmx_ctx* c) {
    c->ev_used = 0;
    c->ev_kind.clear();
}
#define RMX_TM_BEGIN(c) do { const int rc_tm_ = tm_begin(c); if (rc_tm_ != RMX_OK) return rc_tm_; } while (0)
#define RMX_TM_END(c, kind) do { const int rc_tm_ = tm_end((c), (kind)); if (rc_tm_ != RMX_OK) return rc_tm_; } while (0)

// pair plan (host_plan.hpp: validation, anchor runs, parts of <= pairs_per_block consecutive items) -> device copies
static int build_plan(rmx_ctx* c, const int32_t* pairs, int n_pairs) {
    host::PairPlan pp;
    std::string why;
    if (host::make_pair_plan(c->n_buoys, pairs, n_pairs, c->pairs_per_block, &pp, &why) != 0) return fail(c, RMX_E_INVAL, "%s", why.c_str());
    if (c->plan_n_pairs == n_pairs && c->plan_ppb == c->pairs_per_block && pp.pairs == c->plan_pairs) return RMX_OK;
    if (c->d_items) { (void)hipFree(c->d_items); c->d_items = nullptr; }
    if (c->d_part_begin) { (void)hipFree(c->d_part_begin); c->d_part_begin = nullptr; }
    RMX_HIP(c, hipMalloc((void**)&c->d_items, sizeof(PairItem) * (size_t)n_pairs));
    RMX_HIP(c, hipMalloc((void**)&c->d_part_begin, sizeof(int) * (size_t)(pp.n_parts + 1)));
    RMX_HIP(c, hipMemcpy(c->d_items, pp.items.data(), sizeof(PairItem) * (size_t)n_pairs, hipMemcpyHostToDevice));
    RMX_HIP(c, hipMemcpy(c->d_part_begin, pp.part_begin.data(), sizeof(int) * (size_t)(pp.n_parts + 1), hipMemcpyHostToDevice));
    c->plan_all_pairs = pp.all_pairs;
    c->plan_pairs.swap(pp.pairs);
    c->plan_n_pairs = n_pairs;
    c->plan_n_parts = pp.n_parts;
    c->plan_ppb = c->pairs_per_block;
    return RMX_OK;
}

}  // namespace rmx

namespace rmx {

static int upload(rmx_ctx* c, float2** dst, const std::vector<float2>& v) {
    RMX_HIP(c, hipMalloc((void**)dst, v.size() * sizeof(float2)));
    RMX_HIP(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice));
    c->scratch_bytes += v.size() * sizeof(float2);
    return RMX_OK;
}

// windows up to this zero-padded length run with the whole transform in LDS (128 KiB of the 160)
// (L = 16384 -- N = 8192, the reference's iq_stream_client capture length -- is better off in the four-step path: one
// 128 KiB transform per CU leaves nothing to overlap with; 0.77 -> 0.60 ms for 3 buoys x 1024 windows)
static long gen_small_max_l(const rmx_ctx* c) { return c->knobs.get_or("small_maxl", 8192L); }
#define kGenSmallMaxL gen_small_max_l(c)
static int gen_small_threads(long L) { const long t = L >> 4; return t >= 1024 ? 1024 : (t < 64 ? 64 : (int)t); }   // one radix-16 group per thread and pass
// dynamic LDS of the four-step kernels
static int gen_rows_tpr(const rmx_ctx* c, int R) {
    int t = gen::rows_tpr(R);
    long v;
    if (c->knobs.get("rows_tpr", &v) && v >= 1 && v <= gen::kGThreads && (v & (v - 1)) == 0 && v <= R) t = (int)v;
    return t;
}
static size_t gen_rows_lds(const rmx_ctx* c, int R) {   // rows + per-row twiddle tables (TW passes)
    const int tpr = gen_rows_tpr(c, R);
    int logR = 0;
    while ((1 << logR) < R) ++logR;
    const int a = logR >> 1;
    return ((size_t)(gen::kGThreads / tpr) * ((size_t)gen::lp(R) + (1 << a) + (R >> a)) + (size_t)(R >> 1) + (R < 4096 ? 240 : 0)) * 8;   // + W_R table + the q = 16 pass's own
}
// column kernels with the column length compiled in (16-column tiles, the default thread count), else the run-time ones
static const void* cols_inv_fn(int l1, int lt, int thr) {
    using namespace gen;
    if (lt == 3) return (const void*)g_cols_inv<3>;
    if (lt == 5) {          // 32-column tiles (256-byte row segments): option col_logt = 5, an A/B switch (VERDICT r04 #3a)
        if (thr == cols_threads(l1, 5) && l1 == 8) return (const void*)g_cols_inv<5, 8>;
        if (thr == cols_threads(l1, 5) && l1 == 9) return (const void*)g_cols_inv<5, 9>;
        return (const void*)g_cols_inv<5>;
    }
    if (thr == cols_threads(l1, 4)) switch (l1) {
        case 5: return (const void*)g_cols_inv<4, 5>;
        case 6: return (const void*)g_cols_inv<4, 6>;
        case 7: return (const void*)g_cols_inv<4, 7>;
        case 8: return (const void*)g_cols_inv<4, 8>;
        case 9: return (const void*)g_cols_inv<4, 9>;
        default: break;
    }
    return (const void*)g_cols_inv<4>;
}
template <bool U8>
static const void* cols_fwd_fn(int l1, int lt, int thr) {
    using namespace gen;
    if (lt == 3) return (const void*)g_cols_fwd<U8, 3>;
    if (lt == 5) {
        if (thr == cols_threads(l1, 5) && l1 == 8) return (const void*)g_cols_fwd<U8, 5, 8>;
        if (thr == cols_threads(l1, 5) && l1 == 9) return (const void*)g_cols_fwd<U8, 5, 9>;
        return (const void*)g_cols_fwd<U8, 5>;
    }
    if (thr == cols_threads(l1, 4)) switch (l1) {
        case 5: return (const void*)g_cols_fwd<U8, 4, 5>;
        case 6: return (const void*)g_cols_fwd<U8, 4, 6>;
        case 7: return (const void*)g_cols_fwd<U8, 4, 7>;
        case 8: return (const void*)g_cols_fwd<U8, 4, 8>;
        case 9: return (const void*)g_cols_fwd<U8, 4, 9>;
        default: break;
    }
    return (const void*)g_cols_fwd<U8, 4>;
}
static const void* rows_fwd_fn(int logR, int tpr) {
    using namespace gen;
    if (tpr != rows_tpr(1 << logR)) return (const void*)g_rows<true, false, false>;
    switch (logR) {
        case 9: return (const void*)g_rows<true, false, false, 9>;
        case 10: return (const void*)g_rows<true, false, false, 10>;
        case 11: return (const void*)g_rows<true, false, false, 11>;
        case 12: return (const void*)g_rows<true, false, false, 12>;
        case 13: return (const void*)g_rows<true, false, false, 13>;
        default: return (const void*)g_rows<true, false, false>;
    }
}
static const void* rows_inv_fn(int logR, int tpr) {       // the inverse row kernel, row length compiled in where we have it
    using namespace gen;
    if (tpr != rows_tpr(1 << logR)) return (const void*)g_rows<false, true, true>;      // RMX_ROWS_TPR override
    switch (logR) {
        case 9: return (const void*)g_rows<false, true, true, 9>;
        case 10: return (const void*)g_rows<false, true, true, 10>;
        case 11: return (const void*)g_rows<false, true, true, 11>;
        case 12: return (const void*)g_rows<false, true, true, 12>;
        case 13: return (const void*)g_rows<false, true, true, 13>;
        default: return (const void*)g_rows<false, true, true>;
    }
}
static const void* rows_anchor_fn(int logR, int tpr) {
    using namespace gen;
    if (tpr != rows_tpr(1 << logR)) return nullptr;
    switch (logR) {
        case 9: return (const void*)g_rows_anchor<9>;
        case 10: return (const void*)g_rows_anchor<10>;
        case 11: return (const void*)g_rows_anchor<11>;
        case 12: return (const void*)g_rows_anchor<12>;
        case 13: return (const void*)g_rows_anchor<13>;
        default: return nullptr;
    }
}
static const void* fused_fn(int nb, int logR, bool def) {   // g_rows_fused<n_buoys, log2 row length, default plan>
#define RMX_FF(NB, D) (logR == 9 ? (const void*)gen::g_rows_fused<NB, 9, D> : logR == 10 ? (const void*)gen::g_rows_fused<NB, 10, D> : \
                       logR == 11 ? (const void*)gen::g_rows_fused<NB, 11, D> : (const void*)gen::g_rows_fused<NB, 12, D>)
    if (def) return nb == 2 ? RMX_FF(2, true) : nb == 3 ? RMX_FF(3, true) : RMX_FF(4, true);
    return nb == 2 ? RMX_FF(2, false) : nb == 3 ? RMX_FF(3, false) : RMX_FF(4, false);
#undef RMX_FF
}
template <bool DEF, bool U8>
static const void* wfused_fn(int nb, int logR) {            // g_win_fused<n_buoys, log2 L, default plan, uint8 input>
#define RMX_WF(NB) (logR == 9 ? (const void*)gen::g_win_fused<NB, 9, DEF, U8> : logR == 10 ? (const void*)gen::g_win_fused<NB, 10, DEF, U8> : \
                    logR == 11 ? (const void*)gen::g_win_fused<NB, 11, DEF, U8> : (const void*)gen::g_win_fused<NB, 12, DEF, U8>)
    return nb == 2 ? RMX_WF(2) : nb == 3 ? RMX_WF(3) : RMX_WF(4);
#undef RMX_WF
}
static size_t gen_wfused_lds(int logR, bool tw_regs) {     // transform buffers + the passes' twiddle tables + argmax words
    const int tpr = (1 << logR) >> 4, upw = gen::kGThreads / tpr;
    const int buf = logR == 9 ? gen::FusedPlan<9>::buf : logR == 10 ? gen::FusedPlan<10>::buf : logR == 11 ? gen::FusedPlan<11>::buf : gen::FusedPlan<12>::buf;
    return ((size_t)upw * buf + (size_t)(tw_regs ? 0 : gen::fused_tab_total(logR)) + 8) * 8;
}
template <bool U8>
static const void* wscr_fn(int logR) {                      // g_win_scr<log2 L, uint8 input>
    switch (logR) {
        case 9: return (const void*)gen::g_win_scr<9, U8>;
        case 10: return (const void*)gen::g_win_scr<10, U8>;
        case 11: return (const void*)gen::g_win_scr<11, U8>;
        case 12: return (const void*)gen::g_win_scr<12, U8>;
        case 13: return (const void*)gen::g_win_scr<13, U8>;
        default: return (const void*)gen::g_win_scr<14, U8>;
    }
}
static void wscr_shape(int logR, int* thr, int* upw, size_t* lds) {
#define RMX_WS(L) case L: *thr = gen::WinPlan<L>::thr; *upw = gen::WinPlan<L>::upw; *lds = gen::WinPlan<L>::lds_bytes; break;
    switch (logR) { RMX_WS(9) RMX_WS(10) RMX_WS(11) RMX_WS(12) RMX_WS(13) default: RMX_WS(14) }
#undef RMX_WS
}
static size_t gen_fused_lds(int R, bool tw_regs = false) {   // g_rows_fused: R/16 threads per row (tw_regs: no twiddle tables)
    int logR = 0;
    while ((1 << logR) < R) ++logR;
    const int a = logR >> 1, tpr = R >> 4, upw = gen::kGThreads / (tpr > 0 ? tpr : 1);
    const int buf = logR == 9 ? gen::FusedPlan<9>::buf : logR == 10 ? gen::FusedPlan<10>::buf : logR == 11 ? gen::FusedPlan<11>::buf : gen::FusedPlan<12>::buf;
    return ((size_t)upw * ((size_t)buf + (1 << a) + (R >> a)) + (size_t)(tw_regs ? 0 : gen::fused_tab_total(logR))) * 8;   // + the passes' twiddle tables
}
static int host_col_log_t(const rmx_ctx* c, int l1) {   // gen::col_log_t, or the "col_logt" option (3 | 4) for experiments
    return (int)c->knobs.get_or("col_logt", gen::col_log_t(l1));
}
static size_t gen_cols_lds(const rmx_ctx* c, int l1) {  // [L1][T] tile + T per-column twiddle tables
    const int a = l1 >> 1, T = 1 << host_col_log_t(c, l1);
    return ((size_t)gen::lp((long)(1 << l1) * T) + (size_t)T * ((1 << a) + ((1 << l1) >> a) + 1) + (size_t)((1 << l1) >> 1)) * 8;   // + W_L1 table
}
static int gen_cols_threads(const rmx_ctx* c, int l1) { // one radix-16 work item per thread and pass, <= 1024
    const long work = (((long)1 << l1) << host_col_log_t(c, l1)) / 16;
    long v;
    if (c->knobs.get("cols_threads", &v)) return (int)v;
    return work >= 1024 ? 1024 : (work < 64 ? 64 : (int)work);
}

static int generic_init(rmx_ctx* c) {
    using namespace gen;
    c->generic = true;
    int logn = 0;
    while ((1 << logn) < c->n_samples) ++logn;
    c->g_logL = logn + 1;
    const long L = 1L << c->g_logL;
    const int all_pairs = c->n_buoys * (c->n_buoys - 1) / 2;
    std::vector<float2> t;
    if (L <= kGenSmallMaxL) {
        make_row_table(t, (int)L);
        int rc = upload(c, &c->g_tw, t);
        if (rc) return rc;
        const int slds = (int)(gen::lp(L) * 8 + 1024 * 8);
        RMX_HIP(c, hipFuncSetAttribute((const void*)g_pair_small, hipFuncAttributeMaxDynamicSharedMemorySize, slds));
        RMX_HIP(c, hipFuncSetAttribute((const void*)g_fwd_small<false>, hipFuncAttributeMaxDynamicSharedMemorySize, slds));
        RMX_HIP(c, hipFuncSetAttribute((const void*)g_fwd_small<true>, hipFuncAttributeMaxDynamicSharedMemorySize, slds));
        // at most four buoys, 512 <= L <= 4096: whole windows in one kernel, spectra in registers (RMX_WFUSED=0: the two
        // kernels above, as for custom Doppler searches)
        c->g_wfused = c->n_buoys >= 2 && c->n_buoys <= 4 && c->g_logL >= 9 && c->g_logL <= 12;
        c->g_wfused = c->g_wfused && c->knobs.get_or("wfused", 1) != 0;
        if (c->g_wfused) {
            c->g_wf_fn[0][0] = wfused_fn<false, false>(c->n_buoys, c->g_logL);
            c->g_wf_fn[0][1] = wfused_fn<false, true>(c->n_buoys, c->g_logL);
            c->g_wf_fn[1][0] = wfused_fn<true, false>(c->n_buoys, c->g_logL);
            c->g_wf_fn[1][1] = wfused_fn<true, true>(c->n_buoys, c->g_logL);
            c->g_wf_lds[0] = gen_wfused_lds(c->g_logL, false);
            c->g_wf_lds[1] = gen_wfused_lds(c->g_logL, gen::fused_tw_regs(c->n_buoys, c->g_logL, true));
            for (int d = 0; d < 2; ++d)
                for (int u = 0; u < 2; ++u)
                    RMX_HIP(c, hipFuncSetAttribute(c->g_wf_fn[d][u], hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->g_wf_lds[d]));
        }
    } else {
        // columns of length L1 <= 1024 (a tile of 16 columns is L1*128 bytes of LDS), rows of L2 = L/L1 <= 8192
        // columns a little shorter than rows (measured: 512 x 4096 beats 1024 x 2048 at L = 2^21, 256 x 2048 beats
        // 512 x 1024 at 2^19), rows at most 8192 (64 KiB of LDS)
        c->g_logL1 = (c->g_logL - 3) / 2;
        if (c->g_logL == 18) c->g_logL1 = 8;   // (measured with this round's kernels: 256 x 1024 beats 128 x 2048 by 5 %)
        if (c->g_logL1 < c->g_logL - 13) c->g_logL1 = c->g_logL - 13;
        if (c->g_logL1 > 10) c->g_logL1 = 10;
        { long v; if (c->knobs.get("logl1", &v) && v >= 4 && v <= 10 && c->g_logL - v <= 13 && c->g_logL - v >= 4) c->g_logL1 = (int)v; }
        c->g_logL2 = c->g_logL - c->g_logL1;
        c->g_lo_bits = (c->g_logL + 1) / 2;
        make_row_table(t, 1 << c->g_logL1);
        int rc = upload(c, &c->g_tw1, t);
        if (rc) return rc;
        make_row_table(t, 1 << c->g_logL2);
        rc = upload(c, &c->g_tw2, t);
        if (rc) return rc;
        std::vector<float2> thi, tlo;
        make_big_tables(thi, tlo, L, c->g_lo_bits);
        rc = upload(c, &c->g_thi, thi);
        if (rc) return rc;
        rc = upload(c, &c->g_tlo, tlo);
        if (rc) return rc;
        const int cols_lds = (int)gen_cols_lds(c, c->g_logL1), rows_lds = (int)gen_rows_lds(c, 1 << c->g_logL2);
        {
            const int lt = host_col_log_t(c, c->g_logL1), thr = gen_cols_threads(c, c->g_logL1);
            c->g_cols_inv_fn = cols_inv_fn(c->g_logL1, lt, thr);
            c->g_cols_fwd_fn[0] = cols_fwd_fn<false>(c->g_logL1, lt, thr);
            c->g_cols_fwd_fn[1] = cols_fwd_fn<true>(c->g_logL1, lt, thr);
            RMX_HIP(c, hipFuncSetAttribute(c->g_cols_inv_fn, hipFuncAttributeMaxDynamicSharedMemorySize, cols_lds));
            RMX_HIP(c, hipFuncSetAttribute(c->g_cols_fwd_fn[0], hipFuncAttributeMaxDynamicSharedMemorySize, cols_lds));
            RMX_HIP(c, hipFuncSetAttribute(c->g_cols_fwd_fn[1], hipFuncAttributeMaxDynamicSharedMemorySize, cols_lds));
        }
        RMX_HIP(c, hipFuncSetAttribute((const void*)(g_rows<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds));
        RMX_HIP(c, hipFuncSetAttribute((const void*)(g_rows<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds));
        c->g_rows_inv_fn = rows_inv_fn(c->g_logL2, gen_rows_tpr(c, 1 << c->g_logL2));
        c->g_rows_fwd_fn = rows_fwd_fn(c->g_logL2, gen_rows_tpr(c, 1 << c->g_logL2));
        RMX_HIP(c, hipFuncSetAttribute(c->g_rows_fwd_fn, hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds));
        RMX_HIP(c, hipFuncSetAttribute(c->g_rows_inv_fn, hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds));
        c->g_rows_anchor_fn = rows_anchor_fn(c->g_logL2, gen_rows_tpr(c, 1 << c->g_logL2));
        { long v; if (c->knobs.get("rows_anchor", &v)) { if (v == 0) c->g_rows_anchor_fn = nullptr; else if (v >= 2) c->g_rows_anchor_min_b = (int)v; } }
        if (c->g_rows_anchor_fn)
            RMX_HIP(c, hipFuncSetAttribute(c->g_rows_anchor_fn, hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds));
        // both row passes in one kernel when the buoys' spectra rows fit the register file (generic_path.hpp)
        c->g_fused = c->n_buoys <= 4 && c->g_logL2 >= 9 && c->g_logL2 <= 12;
        { long v; if (c->knobs.get("fused", &v)) {      // 0: never, 2: also for batches too small to fill the chip (tests)
            c->g_fused = c->g_fused && v != 0;
            c->g_fused_always = v == 2;
        } }
        if (c->g_fused) {
            const int flds = (int)gen_fused_lds(1 << c->g_logL2);
            const void* fn = fused_fn(c->n_buoys, c->g_logL2, false);
            c->g_fused_fn = fn;
            c->g_fused_def_fn = fused_fn(c->n_buoys, c->g_logL2, true);
            if (c->knobs.get_or("fused_def", 1) == 0) c->g_fused_def_fn = nullptr;
            if (c->g_fused_def_fn) {
                const int dlds = (int)gen_fused_lds(1 << c->g_logL2, gen::fused_tw_regs(c->n_buoys, c->g_logL2, true));
                RMX_HIP(c, hipFuncSetAttribute(c->g_fused_def_fn, hipFuncAttributeMaxDynamicSharedMemorySize, dlds));
            }
            RMX_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, flds));
        }
    }
    // any buoy count, 512 <= L <= 16384: whole windows in one persistent kernel, spectra in a per-workgroup scratch
    // (RMX_WSCR=0: the two-kernel LDS path / the four-step path, which the Doppler search uses in any case)
    c->g_wscr = !c->g_wfused && c->g_logL >= 9 && c->g_logL <= 15;   // (15: the even / odd-bin halves, win_eo.hpp)
    { long v; if (c->knobs.get("wscr", &v)) {          // 0: never, 2: also for batches that do not fill the chip (tests)
        c->g_wscr = c->g_wscr && v != 0;
        c->g_wscr_always = v == 2;
    } }
    if (c->g_wscr) {
        make_row_table(t, c->g_logL == 15 ? 16384 : (int)L);       // (N = 16384: two 16384-point halves)
        int rc = upload(c, &c->g_tw_win, t);
        if (rc) return rc;
        c->g_ws_fn[0] = wscr_fn<false>(c->g_logL);
        c->g_ws_fn[1] = wscr_fn<true>(c->g_logL);
        wscr_shape(c->g_logL, &c->g_ws_thr, &c->g_ws_upw, &c->g_ws_lds);
        if (c->g_logL == 15) {
            const double two_pi = 6.283185307179586476925286766559;
            std::vector<float2> wl(1024);
            for (int i = 0; i < 1024; ++i) wl[i] = make_float2((float)std::cos(two_pi * i / 32768.0), (float)-std::sin(two_pi * i / 32768.0));
            rc = upload(c, &c->g_tw_l, wl);
            if (rc) return rc;
            c->g_ws_fn[0] = (const void*)gen::g_win_eo15<false>;
            c->g_ws_fn[1] = (const void*)gen::g_win_eo15<true>;
            c->g_ws_thr = 512;
            c->g_ws_upw = 1;
            c->g_ws_lds = gen::kWinEoLds;
        }
        // L = 16384: 512 threads x two butterflies (g_win_scr14) unless RMX_WSCR14=0 (1024 threads x one: g_win_scr<14>)
        bool two = c->g_logL == 14;
        two = two && c->knobs.get_or("wscr14", 1) != 0;
        if (two) {
            c->g_ws_fn[0] = (const void*)gen::g_win_scr14<false>;
            c->g_ws_fn[1] = (const void*)gen::g_win_scr14<true>;
            c->g_ws_thr = 512;
            c->g_ws_lds = ((size_t)gen::lp(16384) + 15 * 64 + 15 * 4 + 16) * 8;
        }
        int per_cu = 1;
        for (int u = 0; u < 2; ++u)
            RMX_HIP(c, hipFuncSetAttribute(c->g_ws_fn[u], hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->g_ws_lds));
        RMX_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, c->g_ws_fn[0], c->g_ws_thr, c->g_ws_lds));
        if (per_cu < 1) per_cu = 1;
        { long v; if (c->knobs.get("wscr_per_cu", &v) && v >= 1 && v <= per_cu) per_cu = (int)v; }
        long grid = (long)c->n_cus * per_cu;
        const long need = ((long)c->max_windows + c->g_ws_upw - 1) / c->g_ws_upw;
        if (grid > need) grid = need;
        c->g_ws_grid = (int)grid;
        const size_t sbytes = (size_t)grid * c->g_ws_upw * c->n_buoys * L * 8;
        RMX_HIP(c, hipMalloc((void**)&c->g_ws_scratch, sbytes));
        c->scratch_bytes += sbytes;
        // N = 8192: the two bin-parity halves on the fused N = 4096 kernel's network (kwin8k.hpp), same scratch size
        // (B x 2 x 64 KiB per persistent workgroup), one workgroup per CU.  Option kwin8k: 1 (default) = k_win8kl, one anchor
        // half resident in LDS -- 0.51 against g_win_scr14's 0.73 ms at 8 buoys x 512 windows, 1.00 against 1.42 at 16 x 256,
        // 0.19 against 0.285 at 3 x 1024 --; 0 = g_win_scr14; 2 = k_win8k (no resident anchor: -DRMX_EXPERIMENTS builds only)
        if (c->g_logL == 14 && c->knobs.get_or("kwin8k", 1) != 0) {
            std::vector<float4> t1;
            std::vector<float4> t1_4096;
            std::vector<float2> t2;
            build_tables(t1_4096, t2);
#ifdef RMX_EXPERIMENTS
            if (c->knobs.get_or("kwin8k", 1) == 2) k8::build_tables8k(t1);      // (k_win8k: a full TW1 table per half)
            else
#endif
            k8::build_tables8kl(t1);
            RMX_HIP(c, hipMalloc((void**)&c->g_k8_tw1, t1.size() * sizeof(float4)));
            RMX_HIP(c, hipMemcpy(c->g_k8_tw1, t1.data(), t1.size() * sizeof(float4), hipMemcpyHostToDevice));
            RMX_HIP(c, hipMalloc((void**)&c->g_k8_tw2, t2.size() * sizeof(float2)));
            RMX_HIP(c, hipMemcpy(c->g_k8_tw2, t2.data(), t2.size() * sizeof(float2), hipMemcpyHostToDevice));
            c->scratch_bytes += t1.size() * sizeof(float4) + t2.size() * sizeof(float2);
            RMX_HIP(c, hipFuncSetAttribute((const void*)k8::k_win8kl<false>, hipFuncAttributeMaxDynamicSharedMemorySize, k8::kLdsLBytes));
            RMX_HIP(c, hipFuncSetAttribute((const void*)k8::k_win8kl<true>, hipFuncAttributeMaxDynamicSharedMemorySize, k8::kLdsLBytes));
            c->g_k8 = c->g_ws_upw == 1 && grid <= c->n_cus;      // (its grid is the scratch's: one window slot per workgroup)
            c->g_k8_kind = (int)c->knobs.get_or("kwin8k", 1);
#ifdef RMX_EXPERIMENTS
            RMX_HIP(c, hipFuncSetAttribute((const void*)k8::k_win8k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, k8::kLds8Bytes));
            RMX_HIP(c, hipFuncSetAttribute((const void*)k8::k_win8k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, k8::kLds8Bytes));
#else
            if (c->g_k8_kind == 2) return fail(c, RMX_E_UNSUPPORTED, "option kwin8k = 2 (k_win8k) exists only in a -DRMX_EXPERIMENTS build");
#endif
        }
        // N = 16384: four quarter transforms on the same network, forward and pair kernels per chunk of g_ws_grid windows
        // (kwin16k.hpp; the spectra of a chunk take the SAME B x 256 KiB per window as g_win_eo15's per-workgroup scratch, so
        // the buffer is shared).  Option kwin16k: 1 (default) = from the measured batch size on, 2 = every batch, 0 = g_win_eo15.
        if (c->g_logL == 15 && c->knobs.get_or("kwin16k", 1) != 0 && c->g_ws_upw == 1) {
            std::vector<float4> t1, t1_4096;
            std::vector<float2> t2, gq, tws;
            build_tables(t1_4096, t2);
            k16::build_tables16k(t1, gq, tws);
            RMX_HIP(c, hipMalloc((void**)&c->g_k16_tw1, t1.size() * sizeof(float4)));
            RMX_HIP(c, hipMemcpy(c->g_k16_tw1, t1.data(), t1.size() * sizeof(float4), hipMemcpyHostToDevice));
            rc = upload(c, &c->g_k16_gq, gq);
            if (rc) return rc;
            rc = upload(c, &c->g_k16_tw2, t2);
            if (rc) return rc;
            rc = upload(c, &c->g_k16_tws, tws);
            if (rc) return rc;
            c->scratch_bytes += t1.size() * sizeof(float4);
            RMX_HIP(c, hipFuncSetAttribute((const void*)k16::k16_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, k16::kLdsFwdBytes));
            RMX_HIP(c, hipFuncSetAttribute((const void*)k16::k16_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, k16::kLdsFwdBytes));
            RMX_HIP(c, hipFuncSetAttribute((const void*)k16::k16_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, k16::kLdsPairBytes));
            c->g_k16 = (int)c->knobs.get_or("kwin16k", 1);
        }
    }
    // windows per chunk (host_plan.hpp: spectra + products under 32 GiB of the 288, "gen_chunk" caps it for experiments)
    const long chunk = host::generic_chunk_windows(c->n_buoys, L, c->max_windows, 32L << 30, c->knobs.get_or("gen_chunk", 0));
    c->g_chunk = (int)chunk;
    return RMX_OK;
}

// Device buffers of the generic path, allocated on first need:
//   need_spec      the per-window spectra (two-kernel LDS path, four-step path, Doppler search)
//   need_pairbufs  products / tile records / halo of the FOUR-STEP pair kernels -- not when the whole-window kernels
//                  run (g_win_fused, g_win_scr never touch them: an 8-buoy, 4096-window engine of 8192-sample windows
//                  would otherwise pin 15 GB it never uses)
// A failed hipMalloc halves the chunk and retries (down to one window) instead of giving up.
static int generic_ensure(rmx_ctx* c, int n_pairs, bool need_spec = true, bool need_pairbufs = true) {
    using namespace gen;
    const long L = 1L << c->g_logL;
    const bool four_step = L > kGenSmallMaxL;
    auto release = [&]() {
        for (void** p : {(void**)&c->g_spec, (void**)&c->g_spec_r, (void**)&c->g_prod, (void**)&c->g_rec, (void**)&c->g_halo})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        c->scratch_bytes -= c->g_dyn_bytes;
        c->g_dyn_bytes = 0;
        c->g_slots_alloc = 0;
    };
    for (;;) {
        const long items = (long)c->g_chunk * c->n_buoys, slots = (long)c->g_chunk * n_pairs;
        bool ok = true;
        auto grab = [&](void** p, size_t bytes) {
            if (!ok) return;
            if (hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; ok = false; return; }
            c->scratch_bytes += bytes;
            c->g_dyn_bytes += bytes;
        };
        if (!c->g_spec && need_spec) grab((void**)&c->g_spec, (size_t)items * L * 8);
        if (four_step && need_pairbufs && slots > c->g_slots_alloc) {
            // (re)allocate the pair-dependent buffers when the pair count grows
            for (void** p : {(void**)&c->g_prod, (void**)&c->g_rec, (void**)&c->g_halo})
                if (*p) { (void)hipFree(*p); *p = nullptr; }
            c->scratch_bytes -= c->g_pair_bytes;
            c->g_dyn_bytes -= c->g_pair_bytes;
            c->g_pair_bytes = 0;
            c->g_slots_alloc = 0;
            const size_t before = c->g_dyn_bytes;
            const long parts = (1L << c->g_logL2) >> host_col_log_t(c, c->g_logL1);   // one record per column tile
            grab((void**)&c->g_prod, (size_t)slots * L * 8);
            grab((void**)&c->g_rec, (size_t)slots * parts * sizeof(GTile));
            grab((void**)&c->g_halo, (size_t)slots * parts * 2 * sizeof(float) << c->g_logL1);
            c->g_pair_bytes = c->g_dyn_bytes - before;
            if (ok) c->g_slots_alloc = slots;
        }
        if (ok) break;
        if (c->g_chunk <= 1) { release(); return fail(c, RMX_E_NOMEM, "device allocation failed even for one window per chunk"); }
        RMX_HIP(c, hipStreamSynchronize(c->stream));
        release();                                    // every chunk-sized buffer goes; they come back at half the size
        c->g_chunk = (c->g_chunk + 1) / 2;
    }
    if (c->g_pairs_n == n_pairs && c->g_pairs_plan == c->plan_pairs) return RMX_OK;
    if (c->g_pairs) { (void)hipFree(c->g_pairs); c->g_pairs = nullptr; }
    std::vector<GPair> gp(n_pairs);
    for (int q = 0; q < n_pairs; ++q) gp[q] = GPair{c->plan_pairs[2 * q], c->plan_pairs[2 * q + 1]};
    RMX_HIP(c, hipMalloc((void**)&c->g_pairs, sizeof(GPair) * (size_t)n_pairs));
    RMX_HIP(c, hipMemcpy(c->g_pairs, gp.data(), sizeof(GPair) * (size_t)n_pairs, hipMemcpyHostToDevice));
    c->g_pairs_n = n_pairs;
    c->g_pairs_plan = c->plan_pairs;
    return RMX_OK;
}

// forward spectra of windows [w0, w0 + wc) of d_iq into g_spec (rot == nullptr) or, de-rotated by the phasor
// table rot[N], into g_spec_r (rmx_caf_batch)
static int generic_forward(rmx_ctx* c, const void* d_iq, int w0, int wc, bool u8, const float2* rot,
                           bool cols_only = false) {
    using namespace gen;
    const int N = c->n_samples, logL = c->g_logL, B = c->n_buoys;
    const long L = 1L << logL;
    const float fwd_scale = std::ldexp(1.0f, -(logL / 2));
    hipStream_t st = c->stream;
    const int items = wc * B;
    const long first_item = (long)w0 * B;
    if (rot && !c->g_spec_r) {
        RMX_HIP(c, hipMalloc((void**)&c->g_spec_r, (size_t)c->g_chunk * B * L * 8));
        c->scratch_bytes += (size_t)c->g_chunk * B * L * 8;
        c->g_dyn_bytes += (size_t)c->g_chunk * B * L * 8;
    }
    float2* dst = rot ? c->g_spec_r : c->g_spec;
    if (L <= kGenSmallMaxL) {
        const int sthr = gen_small_threads(L);
        RMX_TM_BEGIN(c);
        if (u8)
            hipLaunchKernelGGL(g_fwd_small<true>, dim3(items), dim3(sthr), (size_t)gen::lp(L) * 8, st, d_iq, dst, c->g_tw, N, logL,
                               first_item, fwd_scale, rot);
        else
            hipLaunchKernelGGL(g_fwd_small<false>, dim3(items), dim3(sthr), (size_t)gen::lp(L) * 8, st, d_iq, dst, c->g_tw, N, logL,
                               first_item, fwd_scale, rot);
        RMX_HIP(c, hipGetLastError());
        RMX_TM_END(c, kTkFwdSmall);
        return RMX_OK;
    }
    const int l1 = c->g_logL1, l2 = c->g_logL2, L1 = 1 << l1, L2 = 1 << l2;
    const int cthr = gen_cols_threads(c, l1), tpr = gen_rows_tpr(c, L2);
    const size_t clds = gen_cols_lds(c, l1), rlds = gen_rows_lds(c, L2);
    const int lt = host_col_log_t(c, l1), ntiles = L2 >> lt;
    // column pass (zero-padded window -> [k1'][n2] * W_L^(n2 k1)), row pass in place (-> [k1'][k2'])
    {
        const void* a_iq = d_iq;
        float2* a_out = dst;
        const float2 *a_tw = c->g_tw1, *a_thi = c->g_thi, *a_tlo = c->g_tlo, *a_rot = rot;
        int a_l1 = l1, a_l2 = l2, a_lo = c->g_lo_bits;
        long a_first = first_item;
        void* args[] = {&a_iq, &a_out, &a_tw, &a_l1, &a_l2, &a_first, &a_lo, &a_thi, &a_tlo, &a_rot};
        RMX_TM_BEGIN(c);
        RMX_HIP(c, hipLaunchKernel(c->g_cols_fwd_fn[u8 ? 1 : 0], dim3(ntiles, items), dim3(cthr), args, clds, st));
        RMX_TM_END(c, kTkColsFwd);
    }
    if (cols_only) {                      // g_rows_fused does the rows
        RMX_HIP(c, hipGetLastError());
        return RMX_OK;
    }
    const long rows = (long)items * L1;
    const int rpw = kGThreads / tpr;
    {
        float2* a_data = dst;
        const float2 *a_tw = c->g_tw2, *a_thi = c->g_thi, *a_tlo = c->g_tlo, *a_null = nullptr;
        const GPair* a_pairs = nullptr;
        int a_l2 = l2, a_L1 = L1, a_l1 = l1, a_lo = c->g_lo_bits, a_zero = 0, a_tpr = tpr;
        long a_L = L, a_rows = rows;
        float a_scale = fwd_scale;
        void* args[] = {&a_data, &a_tw, &a_l2, &a_L1, &a_l1, &a_L, &a_lo, &a_thi, &a_tlo, &a_scale, &a_rows, &a_null, &a_null,
                        &a_pairs, &a_zero, &a_zero, &a_tpr};
        RMX_TM_BEGIN(c);
        RMX_HIP(c, hipLaunchKernel(c->g_rows_fwd_fn, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(kGThreads), args, rlds, st));
        RMX_TM_END(c, kTkRowsFwd);
    }
    RMX_HIP(c, hipGetLastError());
    return RMX_OK;
}

// pair kernels of that chunk: X_i from g_spec, X_j from g_spec (use_rot false) or g_spec_r; results at
// [(w0 + wl) * n_pairs + q] of the three output arrays
static int generic_pairs(rmx_ctx* c, int w0, int wc, int n_pairs, int* d_lag, float* d_frac, float* d_peak,
                         bool use_rot, bool fused = false) {
    using namespace gen;
    const int N = c->n_samples, logL = c->g_logL, B = c->n_buoys;
    const long L = 1L << logL;
    const int hs = logL / 2;
    const float out_scale = std::ldexp(1.0f, -(logL - 2 * hs));
    hipStream_t st = c->stream;
    const int slots = wc * n_pairs;
    const float2* spec_j = use_rot ? c->g_spec_r : c->g_spec;
    if (L <= kGenSmallMaxL) {
        const int sthr = gen_small_threads(L);
        RMX_TM_BEGIN(c);
        hipLaunchKernelGGL(g_pair_small, dim3(slots), dim3(sthr), (size_t)gen::lp(L) * 8 + (size_t)sthr * 8, st, c->g_spec,
                           spec_j, c->g_tw, c->g_pairs, n_pairs, B, N, logL, (long)w0, out_scale, d_lag, d_frac, d_peak);
        RMX_HIP(c, hipGetLastError());
        RMX_TM_END(c, kTkPairSmall);
        return RMX_OK;
    }
    const int l1 = c->g_logL1, l2 = c->g_logL2, L1 = 1 << l1, L2 = 1 << l2;
    const int cthr = gen_cols_threads(c, l1), tpr = gen_rows_tpr(c, L2);
    const size_t clds = gen_cols_lds(c, l1), rlds = gen_rows_lds(c, L2);
    const int lt = host_col_log_t(c, l1), ntiles = L2 >> lt;
    const long rows = (long)slots * L1;
    const int rpw = kGThreads / tpr;
    // row pass ([product on load] rows(L2)^-1 * conj W_L^(n2 k1)), column pass (-> tile records + halo), reduction
    if (fused) {
        // g_spec holds the column pass's output: forward rows, products and inverse rows in one kernel
        const long units = (long)wc * L1;
        const int upw = kGThreads / (L2 >> 4 > 0 ? L2 >> 4 : 1);
        const long blocks = (units + upw - 1) / upw;
        const bool def_plan = c->g_fused_def_fn && c->plan_all_pairs && n_pairs == B * (B - 1) / 2;
        const dim3 grid((unsigned)blocks);
        const float fs = std::ldexp(1.0f, -(logL / 2));
        const float2* colsp = c->g_spec;
        float2* prodp = c->g_prod;
        const float2 *twp = c->g_tw2, *thip = c->g_thi, *tlop = c->g_tlo;
        const GPair* pp = c->g_pairs;
        int a_L1 = L1, a_l1 = l1, a_lo = c->g_lo_bits, a_np = n_pairs;
        long a_L = L, a_units = units;
        float a_scale = fs * fs;
        void* args[] = {&colsp, &prodp, &twp, &a_L1, &a_l1, &a_L, &a_lo, &thip, &tlop, &a_scale, &a_units, &pp, &a_np};
        RMX_TM_BEGIN(c);
        RMX_HIP(c, hipLaunchKernel(def_plan ? c->g_fused_def_fn : c->g_fused_fn, grid, dim3(kGThreads), args,
                                   gen_fused_lds(L2, def_plan && gen::fused_tw_regs(B, l2, true)), st));
        RMX_TM_END(c, kTkRowsFused);
    } else if (c->g_rows_anchor_fn && c->plan_all_pairs && n_pairs == B * (B - 1) / 2 && B >= c->g_rows_anchor_min_b) {
        // default pair list: the anchor's row stays in registers over its run of pairs (from 6 buoys on: with 5 the runs are
        // too short for the per-workgroup set-up, 0.649 -> 0.692 ms at N = 16384; with 6 / 7 / 8 buoys 0.89 -> 0.84, 1.16 -> 1.10,
        // 1.54 -> 1.39 ms, cfg5 250 -> 203 ms)
        float2* a_data = c->g_prod;
        const float2 *a_tw = c->g_tw2, *a_thi = c->g_thi, *a_tlo = c->g_tlo, *a_spec = c->g_spec, *a_specj = spec_j;
        int a_L1 = L1, a_l1 = l1, a_lo = c->g_lo_bits, a_np = n_pairs, a_B = B;
        long a_L = L;
        float a_scale = 1.0f;
        void* args[] = {&a_data, &a_tw, &a_L1, &a_l1, &a_L, &a_lo, &a_thi, &a_tlo, &a_scale, &a_spec, &a_specj, &a_np, &a_B};
        RMX_TM_BEGIN(c);
        RMX_HIP(c, hipLaunchKernel(c->g_rows_anchor_fn, dim3((unsigned)((L1 / rpw) * (B - 1)), (unsigned)wc), dim3(kGThreads), args, rlds, st));
        RMX_TM_END(c, kTkRowsAnchor);
    } else
    {
        float2* a_data = c->g_prod;
        const float2 *a_tw = c->g_tw2, *a_thi = c->g_thi, *a_tlo = c->g_tlo, *a_spec = c->g_spec, *a_specj = spec_j;
        const GPair* a_pairs = c->g_pairs;
        int a_l2 = l2, a_L1 = L1, a_l1 = l1, a_lo = c->g_lo_bits, a_np = n_pairs, a_B = B, a_tpr = tpr;
        long a_L = L, a_rows = rows;
        float a_scale = 1.0f;
        void* args[] = {&a_data, &a_tw, &a_l2, &a_L1, &a_l1, &a_L, &a_lo, &a_thi, &a_tlo, &a_scale, &a_rows, &a_spec, &a_specj,
                        &a_pairs, &a_np, &a_B, &a_tpr};
        RMX_TM_BEGIN(c);
        RMX_HIP(c, hipLaunchKernel(c->g_rows_inv_fn, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(kGThreads), args, rlds, st));
        RMX_TM_END(c, kTkRowsInv);
    }
    {
        const float2 *a_in = c->g_prod, *a_tw = c->g_tw1;
        int a_l1 = l1, a_l2 = l2;
        GTile* a_rec = c->g_rec;
        float* a_halo = c->g_halo;
        void* args[] = {&a_in, &a_tw, &a_l1, &a_l2, &a_rec, &a_halo};
        RMX_TM_BEGIN(c);
        RMX_HIP(c, hipLaunchKernel(c->g_cols_inv_fn, dim3(ntiles, slots), dim3(cthr), args, clds, st));
        RMX_TM_END(c, kTkColsInv);
    }
    RMX_TM_BEGIN(c);
    hipLaunchKernelGGL(g_final, dim3(slots), dim3(64), 0, st, N, l1, l2, lt, c->g_rec, c->g_halo, ntiles, slots,
                       (long)w0 * n_pairs, out_scale, d_lag, d_frac, d_peak);
    RMX_HIP(c, hipGetLastError());
    RMX_TM_END(c, kTkFinal);
    return RMX_OK;
}

// windows [w0, w0 + n) of a batch through the four-step kernels (the partial last round behind a whole-window kernel's full
// rounds, same stream): buffers on first need, chunks as in generic_batch's own loop
static int four_step_windows(rmx_ctx* c, const void* d_iq, int w_first, int n, int n_pairs, int* d_lag, float* d_frac,
                             float* d_peak, bool u8) {
    int rc = generic_ensure(c, n_pairs, true, true);
    if (rc) return rc;
    for (int w0 = w_first; w0 < w_first + n; w0 += c->g_chunk) {
        const int wc = w_first + n - w0 < c->g_chunk ? w_first + n - w0 : c->g_chunk;
        const long fused_blocks = c->g_fused ? (long)wc * (1L << c->g_logL1) / (gen::kGThreads / ((1 << c->g_logL2) >> 4)) : 0;
        const bool fused = c->g_fused && (fused_blocks >= 2L * c->n_cus || c->g_fused_always);
        rc = generic_forward(c, d_iq, w0, wc, u8, nullptr, fused);
        if (rc) return rc;
        rc = generic_pairs(c, w0, wc, n_pairs, d_lag, d_frac, d_peak, false, fused);
        if (rc) return rc;
    }
    return RMX_OK;
}

// N = 16384 through k16_fwd / k16_pairs (kwin16k.hpp): per chunk of g_ws_grid windows (the spectrum scratch holds that many)
// one forward launch -- (window, buoy) items over at most one workgroup per CU -- and one pair launch of 8 S workgroups,
// S = workgroups per XCD.  No per-window state survives the chunk.
static int k16_batch(rmx_ctx* c, const void* d_iq, int n_windows, int n_pairs, int* d_lag, float* d_frac, float* d_peak, bool u8) {
    const int B = c->n_buoys;
    const float out_scale = std::ldexp(1.0f, 3 * kTw1ScaleLog2 - 15);
    const int chunk = c->g_ws_grid;
    const k16::Pair2* prs = reinterpret_cast<const k16::Pair2*>(c->g_pairs);
    for (int w0 = 0; w0 < n_windows; w0 += chunk) {
        const int wc = n_windows - w0 < chunk ? n_windows - w0 : chunk;
        const int items = wc * B;
        const unsigned fgrid = (unsigned)(items < c->n_cus ? items : c->n_cus);
        RMX_TM_BEGIN(c);
        if (u8)
            hipLaunchKernelGGL(k16::k16_fwd<true>, dim3(fgrid), dim3(kThreads), k16::kLdsFwdBytes, c->stream, d_iq, c->g_ws_scratch,
                               c->g_k16_tw1, c->g_k16_gq, c->g_k16_tw2, c->g_k16_tws, (long)w0 * B, items);
        else
            hipLaunchKernelGGL(k16::k16_fwd<false>, dim3(fgrid), dim3(kThreads), k16::kLdsFwdBytes, c->stream, d_iq, c->g_ws_scratch,
                               c->g_k16_tw1, c->g_k16_gq, c->g_k16_tw2, c->g_k16_tws, (long)w0 * B, items);
        RMX_HIP(c, hipGetLastError());
        RMX_TM_END(c, kTkFwd16k);
        // XCD-aware item order (the pairs of a window on one XCD) unless the windows do not spread evenly over eight XCDs
        // and are few: 9 windows would give XCD 0 two windows' work and leave the batch waiting for it
        const int flat = (wc < 32 && wc % 8 != 0) || c->n_cus < 8 ? 1 : 0;
        const long per_xcd = (long)((wc + 7) / 8) * n_pairs;       // items of the busiest XCD
        long S = c->n_cus / 8 > 0 ? c->n_cus / 8 : 1;
        if (per_xcd < S) S = per_xcd;
        long pgrid = 8 * S;
        if (flat) pgrid = (long)wc * n_pairs < c->n_cus ? (long)wc * n_pairs : c->n_cus;
        RMX_TM_BEGIN(c);
        hipLaunchKernelGGL(k16::k16_pairs, dim3((unsigned)pgrid), dim3(kThreads), k16::kLdsPairBytes, c->stream, c->g_ws_scratch,
                           c->g_k16_tw1, c->g_k16_gq, c->g_k16_tw2, c->g_k16_tws, B, prs, n_pairs, (long)w0 * n_pairs, wc, flat,
                           out_scale, d_lag, d_frac, d_peak);
        RMX_HIP(c, hipGetLastError());
        RMX_TM_END(c, kTkPairs16k);
    }
    return RMX_OK;
}

static int generic_batch(rmx_ctx* c, const void* d_iq, int n_windows, int n_pairs, int* d_lag, float* d_frac,
                         float* d_peak, bool u8) {
    // N = 16384 on k_win's network (kwin16k.hpp): fine-grained items, so no partial last round and a low minimum batch -- its
    // two launches cost about 40 us whatever the size, the four-step kernels 26 us for one window: from about 100 transforms
    // (windows x (buoys + pairs)) on it wins (tools/exp_k16_sweep.py: 3 buoys from 16 windows, 5 from 8, 8 from 1).  wscr = 2
    // (tests: "the whole-window kernel, whatever the batch") keeps g_win_eo15, as do pair lists beyond the kernel's LDS copy
    if (c->g_k16 && c->g_logL == 15 && n_pairs <= k16::kMaxPairs16 &&
        (c->g_k16 == 2 || (!c->g_wscr_always &&
                           n_windows >= (int)c->knobs.get_or("k16_min_windows", (100 + c->n_buoys + n_pairs - 1) / (c->n_buoys + n_pairs))))) {
        const int rc16 = generic_ensure(c, n_pairs, false, false);
        if (rc16) return rc16;
        return k16_batch(c, d_iq, n_windows, n_pairs, d_lag, d_frac, d_peak, u8);
    }
    // g_win_scr runs a window's B + P transforms one after the other in one workgroup: batches that leave most of the chip
    // without a workgroup are better off in the per-transform kernels below (8 buoys x 8 windows of 8192: 0.32 vs 0.05 ms).
    // Measured crossovers (tools/smallw.sh): 0.43 workgroups per CU at L = 16384 (8 buoys; 0.63 with 3), 0.66 .. 0.68 below
    const long ws_blocks = ((long)n_windows + (c->g_ws_upw > 0 ? c->g_ws_upw : 1) - 1) / (c->g_ws_upw > 0 ? c->g_ws_upw : 1);
    // (L = 32768, g_win_eo15: it wins from about 0.7 workgroups per CU on for 2 ... 16 buoys -- 0.67-0.79 x the four-step's time
    // at 256 windows, 0.85-0.94 x at 192, 0.9-1.25 x at 128: profiles/r03_weo_sweep.log)
    // (L = 16384 through k_win8kl: 242 us for anything up to one window per CU at 8 buoys, 50 us at 3, against 194 / 50 us at
    // 64 windows and 290 / 64 us at 96 through the per-transform kernels -- tools/exp_k8_small.py: from 5/16 of a workgroup per CU)
    const bool use_wscr = c->g_wscr && (c->g_wscr_always || ws_blocks >= (c->g_logL == 14 ? (c->g_k8 ? 5L : 7L) : 11L) * c->n_cus / 16);
    const bool whole_window = c->g_wfused || use_wscr;   // those kernels keep no per-window state in HBM
    int rc = generic_ensure(c, n_pairs, !whole_window, !whole_window);
    if (rc) return rc;
    if (use_wscr) {
        const int logL = c->g_logL, hs = logL / 2;
        const void* a_iq = d_iq;
        float4* a_scr = c->g_ws_scratch;
        const float2* a_tw = c->g_tw_win;
        int a_nb = c->n_buoys, a_np = n_pairs;
        long a_nw = n_windows, a_first = 0;
        float a_fs = std::ldexp(1.0f, -hs), a_os = std::ldexp(1.0f, -(logL - 2 * hs));
        const gen::GPair* a_pairs = c->g_pairs;
        long grid = ((long)n_windows + c->g_ws_upw - 1) / c->g_ws_upw;
        if (grid > c->g_ws_grid) grid = c->g_ws_grid;
        const bool def_list = c->plan_all_pairs && n_pairs == c->n_buoys * (c->n_buoys - 1) / 2;
        if (c->g_k8 && (def_list || n_pairs <= k8::kMaxPairs8)) {   // N = 8192 on k_win's network (kwin8k.hpp)
            const float out_scale = std::ldexp(1.0f, 3 * kTw1ScaleLog2 - 14);
            const k8::Pair2* prs = def_list ? nullptr : reinterpret_cast<const k8::Pair2*>(c->g_pairs);
            const int stag = (int)c->knobs.get_or("stag", 1);
            // The batch's last, partial round (W mod CUs windows behind at least one full round) costs this kernel a whole
            // round whatever its size -- 3.5 us per half transform: 245 us at 8 buoys, 50 at 3 --, the four-step kernels about
            // 30 us + 0.07 us per window and spectrum-or-pair (tools/exp_k8_small.py: 8 buoys 104 us for 32 windows, 140 for 48;
            // 3 buoys 43 / 47) plus four more launches.  The cheaper one runs, on the same stream behind the full rounds:
            // 8 buoys x 300 windows 507 -> 444 us, x 560 762 -> 710; 3 buoys never split.  Not with wscr = 2 (tests force the kernel).
            int n_tail = 0;
            if (!c->g_wscr_always && n_windows > c->g_ws_grid) {
                const int r = n_windows % c->g_ws_grid;
                const double round_us = 3.5 * (2.0 * c->n_buoys + 2.0 * n_pairs);
                const double four_us = 40.0 + 0.07 * r * (double)(c->n_buoys + n_pairs);
                if (r > 0 && (long)r * 16 < 5L * c->n_cus && four_us < 0.7 * round_us) n_tail = r;   // (5 buoys x 300: 234 split, 228 whole)
            }
            const int n_head = n_windows - n_tail;
            n_windows = n_head;                 // (the launches below take the full rounds)
            RMX_TM_BEGIN(c);
#ifdef RMX_EXPERIMENTS
            if (c->g_k8_kind == 2) {
                if (u8)
                    hipLaunchKernelGGL(k8::k_win8k<true>, dim3((unsigned)grid), dim3(kThreads), k8::kLds8Bytes, c->stream, d_iq,
                                       c->g_ws_scratch, c->g_k8_tw1, c->g_k8_tw2, c->n_buoys, prs, n_pairs, 0L, out_scale, d_lag,
                                       d_frac, d_peak, n_windows, stag);
                else
                    hipLaunchKernelGGL(k8::k_win8k<false>, dim3((unsigned)grid), dim3(kThreads), k8::kLds8Bytes, c->stream, d_iq,
                                       c->g_ws_scratch, c->g_k8_tw1, c->g_k8_tw2, c->n_buoys, prs, n_pairs, 0L, out_scale, d_lag,
                                       d_frac, d_peak, n_windows, stag);
            } else
#endif
            if (u8)
                hipLaunchKernelGGL(k8::k_win8kl<true>, dim3((unsigned)grid), dim3(kThreads), k8::kLdsLBytes, c->stream, d_iq,
                                   c->g_ws_scratch, c->g_k8_tw1, c->g_k8_tw2, c->n_buoys, prs, n_pairs, 0L, out_scale, d_lag,
                                   d_frac, d_peak, n_windows, stag);
            else
                hipLaunchKernelGGL(k8::k_win8kl<false>, dim3((unsigned)grid), dim3(kThreads), k8::kLdsLBytes, c->stream, d_iq,
                                   c->g_ws_scratch, c->g_k8_tw1, c->g_k8_tw2, c->n_buoys, prs, n_pairs, 0L, out_scale, d_lag,
                                   d_frac, d_peak, n_windows, stag);
            RMX_HIP(c, hipGetLastError());
            RMX_TM_END(c, kTkWindow);
            if (n_tail) return four_step_windows(c, d_iq, n_head, n_tail, n_pairs, d_lag, d_frac, d_peak, u8);
            return RMX_OK;
        }
        if (logL == 15) {                          // g_win_eo15: one more table
            // (the partial last round as above: a round costs this kernel 12.6 us per half transform -- 907 us at 8 buoys --, the
            // four-step kernels about 0.14 us per window and spectrum-or-pair at this length)
            int n_tail = 0;
            if (!c->g_wscr_always && n_windows > c->g_ws_grid && c->g_ws_upw == 1) {
                const int r = n_windows % c->g_ws_grid;
                const double round_us = 12.6 * (2.0 * c->n_buoys + 2.0 * n_pairs);
                const double four_us = 60.0 + 0.14 * r * (double)(c->n_buoys + n_pairs);
                if (r > 0 && (long)r * 16 < 11L * c->n_cus && four_us < 0.85 * round_us) n_tail = r;
            }
            const int n_head = n_windows - n_tail;
            a_nw = n_head;
            const float2* a_twl = c->g_tw_l;
            void* args[] = {&a_iq, &a_scr, &a_tw, &a_twl, &a_nb, &a_nw, &a_first, &a_fs, &a_os, &a_pairs, &a_np, &d_lag, &d_frac, &d_peak};
            RMX_TM_BEGIN(c);
            RMX_HIP(c, hipLaunchKernel(c->g_ws_fn[u8 ? 1 : 0], dim3((unsigned)grid), dim3(c->g_ws_thr), args, c->g_ws_lds, c->stream));
            RMX_TM_END(c, kTkWindow);
            if (n_tail) return four_step_windows(c, d_iq, n_head, n_tail, n_pairs, d_lag, d_frac, d_peak, u8);
            return RMX_OK;
        }
        void* args[] = {&a_iq, &a_scr, &a_tw, &a_nb, &a_nw, &a_first, &a_fs, &a_os, &a_pairs, &a_np, &d_lag, &d_frac, &d_peak};
        RMX_TM_BEGIN(c);
        RMX_HIP(c, hipLaunchKernel(c->g_ws_fn[u8 ? 1 : 0], dim3((unsigned)grid), dim3(c->g_ws_thr), args, c->g_ws_lds, c->stream));
        RMX_TM_END(c, kTkWindow);
        return RMX_OK;
    }
    if (c->g_wfused) {
        using namespace gen;
        const int logL = c->g_logL, B = c->n_buoys, hs = logL / 2;
        const bool def_plan = c->plan_all_pairs && n_pairs == B * (B - 1) / 2;
        const int upw = kGThreads / ((1 << logL) >> 4);
        const void* a_iq = d_iq;
        const float2* a_tw = c->g_tw;
        long a_nw = n_windows, a_first = 0;
        float a_fs = std::ldexp(1.0f, -hs), a_os = std::ldexp(1.0f, -(logL - 2 * hs));
        const GPair* a_pairs = c->g_pairs;
        int a_np = n_pairs;
        void* args[] = {&a_iq, &a_tw, &a_nw, &a_first, &a_fs, &a_os, &a_pairs, &a_np, &d_lag, &d_frac, &d_peak};
        RMX_TM_BEGIN(c);
        RMX_HIP(c, hipLaunchKernel(c->g_wf_fn[def_plan ? 1 : 0][u8 ? 1 : 0], dim3((unsigned)((n_windows + upw - 1) / upw)),
                                   dim3(kGThreads), args, c->g_wf_lds[def_plan ? 1 : 0], c->stream));
        RMX_TM_END(c, kTkWindow);
        return RMX_OK;
    }
    for (int w0 = 0; w0 < n_windows; w0 += c->g_chunk) {
        const int wc = n_windows - w0 < c->g_chunk ? n_windows - w0 : c->g_chunk;
        // the fused row kernel when its (window, row block) units fill the chip at least twice: below that (cfg1's single
        // window: 128 workgroups) its long serial chain per unit loses to the two-kernel passes' wider grids (57 vs 47 us)
        const long fused_blocks = c->g_fused ? (long)wc * (1L << c->g_logL1) / (gen::kGThreads / ((1 << c->g_logL2) >> 4)) : 0;
        const bool fused = c->g_fused && (1L << c->g_logL) > kGenSmallMaxL && (fused_blocks >= 2L * c->n_cus || c->g_fused_always);
        rc = generic_forward(c, d_iq, w0, wc, u8, nullptr, fused);
        if (rc) return rc;
        rc = generic_pairs(c, w0, wc, n_pairs, d_lag, d_frac, d_peak, false, fused);
        if (rc) return rc;
    }
    return RMX_OK;
}

}  // namespace rmx

using namespace rmx;

static constexpr int kHostSubChunk = 512;   // windows per copy/kernel step of the pipelined host-pointer path

extern "C" {

int rmx_version(void) { return RMX_VERSION; }

int rmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* rmx_last_error(const rmx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rmx_create(rmx_ctx** out, int device_id, int n_buoys, int n_samples, int max_windows, unsigned flags) {
    if (!out) return fail(nullptr, RMX_E_INVAL, "out is NULL");
    *out = nullptr;
    (void)flags;
    {
        std::string why;
        if (host::check_create_args(n_buoys, n_samples, max_windows, &why) != 0) return fail(nullptr, RMX_E_INVAL, "%s", why.c_str());
    }
    int ndev = rmx_device_count();
    if (ndev <= 0) return fail(nullptr, RMX_E_NODEV, "no HIP device visible");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, RMX_E_INVAL, "device_id %d not in 0..%d", device_id, ndev - 1);
    rmx_ctx* c = new (std::nothrow) rmx_ctx();
    if (!c) return fail(nullptr, RMX_E_NOMEM, "host allocation failed");
    c->knobs = host::snapshot_default_options();   // options set later do not reach this ctx
    c->device = device_id;
    c->n_buoys = n_buoys;
    c->n_samples = n_samples;
    c->max_windows = max_windows;
    int rc = RMX_OK;
    auto bail = [&](int code) {
        g_create_error = c->err;
        rmx_destroy(c);
        return code;
    };
    auto init = [&]() -> int {
        RMX_HIP(c, hipSetDevice(device_id));
        {
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && ncu > 0)
                c->n_cus = ncu;
            // experiment knob: fewer persistent workgroups than CUs (per-CU time without chip-wide contention)
            { long v; if (c->knobs.get("ncus", &v) && v > 0 && v < c->n_cus) c->n_cus = (int)v; }
        }
        RMX_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
        // (option generic4096 = 1: N = 4096 through the generic kernels too -- g_win_scr<13> against k_win, an experiment)
        if (n_samples != kM || c->knobs.get_or("generic4096", 0) != 0) return rmx::generic_init(c);
        int chunk = (int)c->knobs.get_or("chunk_windows", 4096);
        if (chunk < 8) chunk = 8;
        chunk = (chunk + 7) & ~7;
        if (chunk > max_windows) chunk = max_windows;
        {   // spectra scratch = chunk * B * 64 KiB: keep it under 8 GiB
            const long per_win = (long)n_buoys * (8 * kThreads) * (long)sizeof(float4);
            const long cap = (8L << 30) / per_win;
            if (chunk > cap) chunk = (int)(cap > 8 ? (cap & ~7L) : 8);
        }
        c->chunk_windows = chunk;
        // the fused kernel keeps its spectra per persistent workgroup (n_cus * B * 64 KiB); the
        // per-window scratch of the unfused path (custom pair lists) is allocated on first use
        {
            const int wg = chunk < c->n_cus ? chunk : c->n_cus;
            c->spec_bytes = (size_t)wg * n_buoys * (8 * kThreads) * sizeof(float4);
        }
        RMX_HIP(c, hipMalloc((void**)&c->d_spec, c->spec_bytes));
        std::vector<float4> tw1;
        std::vector<float2> tw2;
        build_tables(tw1, tw2);
        RMX_HIP(c, hipMalloc((void**)&c->d_tw1, tw1.size() * sizeof(float4)));
        RMX_HIP(c, hipMalloc((void**)&c->d_tw2, tw2.size() * sizeof(float2)));
        RMX_HIP(c, hipMemcpy(c->d_tw1, tw1.data(), tw1.size() * sizeof(float4), hipMemcpyHostToDevice));
        RMX_HIP(c, hipMemcpy(c->d_tw2, tw2.data(), tw2.size() * sizeof(float2), hipMemcpyHostToDevice));
        c->scratch_bytes = c->spec_bytes + tw1.size() * sizeof(float4) + tw2.size() * sizeof(float2);
        c->stag = (int)c->knobs.get_or("stag", 1);
        c->small_batch = c->knobs.get_or("small4096", 1) != 0;
#ifdef RMX_EXPERIMENTS
        {   // the two other builds of the fused kernel: measured slower, not in the default build
            std::vector<float4> t1;
            std::vector<float2> tb, tc;
            build_tables8(t1, tb, tc);
            RMX_HIP(c, hipMalloc((void**)&c->d_tw1_8, t1.size() * sizeof(float4)));
            RMX_HIP(c, hipMalloc((void**)&c->d_tb8, tb.size() * sizeof(float2)));
            RMX_HIP(c, hipMalloc((void**)&c->d_tc8, tc.size() * sizeof(float2)));
            RMX_HIP(c, hipMemcpy(c->d_tw1_8, t1.data(), t1.size() * sizeof(float4), hipMemcpyHostToDevice));
            RMX_HIP(c, hipMemcpy(c->d_tb8, tb.data(), tb.size() * sizeof(float2), hipMemcpyHostToDevice));
            RMX_HIP(c, hipMemcpy(c->d_tc8, tc.data(), tc.size() * sizeof(float2), hipMemcpyHostToDevice));
            c->scratch_bytes += t1.size() * sizeof(float4) + (tb.size() + tc.size()) * sizeof(float2);
            RMX_HIP(c, hipFuncSetAttribute((const void*)w8::k_win8<false>, hipFuncAttributeMaxDynamicSharedMemorySize, w8::kLdsWin8Bytes));
            RMX_HIP(c, hipFuncSetAttribute((const void*)w8::k_win8<true>, hipFuncAttributeMaxDynamicSharedMemorySize, w8::kLdsWin8Bytes));
            RMX_HIP(c, hipFuncSetAttribute((const void*)k_winp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, pk::kLdsWinpBytes));
            RMX_HIP(c, hipFuncSetAttribute((const void*)k_winp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, pk::kLdsWinpBytes));
            c->pk = (int)c->knobs.get_or("pk", 0);
            c->win8 = (int)c->knobs.get_or("win8", 0);
        }
#endif
        RMX_HIP(c, hipFuncSetAttribute((const void*)k_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        RMX_HIP(c, hipFuncSetAttribute((const void*)k_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        RMX_HIP(c, hipFuncSetAttribute((const void*)k_pair_res, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsResBytes));
        RMX_HIP(c, hipFuncSetAttribute((const void*)k_win<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWinBytes));
        RMX_HIP(c, hipFuncSetAttribute((const void*)k_win<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWinBytes));
        RMX_HIP(c, hipFuncSetAttribute((const void*)k_pair_str, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        return RMX_OK;
    };
    rc = init();
    if (rc != RMX_OK) return bail(rc);
    *out = c;
    return RMX_OK;
}

void rmx_destroy(rmx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (void* p : {(void*)c->g_tw, (void*)c->g_tw1, (void*)c->g_tw2, (void*)c->g_thi, (void*)c->g_tlo, (void*)c->g_spec,
                    (void*)c->g_spec_r, (void*)c->g_prod, (void*)c->g_rec, (void*)c->g_halo, (void*)c->g_pairs,
                    (void*)c->g_ws_scratch, (void*)c->g_tw_win, (void*)c->g_tw_l, (void*)c->g_k8_tw1, (void*)c->g_k8_tw2,
                    (void*)c->g_k16_tw1, (void*)c->g_k16_gq, (void*)c->g_k16_tw2, (void*)c->g_k16_tws})
        if (p) (void)hipFree(p);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (hipEvent_t e : c->copy_ev)
        if (e) (void)hipEventDestroy(e);
    for (void* p : {(void*)c->dt_tw, (void*)c->dt_pdb, c->dt_in, c->dt_out})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)c->sv_buoys, (void*)c->sv_pairs, c->sv_in, c->sv_out})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)c->d_spec_r, (void*)c->caf_rot, (void*)c->caf_lag, (void*)c->caf_frac, (void*)c->caf_peak,
                    (void*)c->caf_dop})
        if (p) (void)hipFree(p);
    if (c->d_spec) (void)hipFree(c->d_spec);
    if (c->d_tw1) (void)hipFree(c->d_tw1);
    if (c->d_tw2) (void)hipFree(c->d_tw2);
    for (void* p : {(void*)c->d_tw1_8, (void*)c->d_tb8, (void*)c->d_tc8})
        if (p) (void)hipFree(p);
    if (c->d_items) (void)hipFree(c->d_items);
    if (c->d_part_begin) (void)hipFree(c->d_part_begin);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_lag) (void)hipFree(c->d_lag);            // (d_frac / d_peak point into the same block)
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rmx_set_stream(rmx_ctx* c, void* hip_stream) {
    if (!c) return RMX_E_INVAL;
    RMX_HIP(c, hipSetDevice(c->device));
    if (c->stream) RMX_HIP(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return RMX_OK;
}

int rmx_set_default_option(const char* key, long value) {
    std::string why;
    if (host::set_default_option(key, value, &why) != 0) return fail(nullptr, RMX_E_INVAL, "%s", why.c_str());
    return RMX_OK;
}

void rmx_clear_default_options(void) { host::clear_default_options(); }

int rmx_set_option(rmx_ctx* c, const char* key, long value) {
    if (!c || !key) return RMX_E_INVAL;
    if (!strcmp(key, "chunk_windows")) {
        if (c->generic) return RMX_OK;   // the generic path sizes its own chunks
        long chunk = value < 8 ? 8 : value;
        chunk = (chunk + 7) & ~7L;
        if (chunk > c->max_windows) chunk = c->max_windows;
        {   // same 8 GiB cap on the unfused path's scratch as at creation
            const long per_win = (long)c->n_buoys * (8 * kThreads) * (long)sizeof(float4);
            const long cap = (8L << 30) / per_win;
            if (chunk > cap) chunk = cap > 8 ? (cap & ~7L) : 8;
        }
        c->chunk_windows = (int)chunk;   // (scratch follows on the next call: ensure_spec)
        return RMX_OK;
    }
    if (!strcmp(key, "pairs_per_block")) {
        if (value < 1 || value > 1 << 20) return fail(c, RMX_E_INVAL, "pairs_per_block %ld out of range", value);
        c->pairs_per_block = (int)value;
        c->ppb_user = true;
        return RMX_OK;
    }
    if (!strcmp(key, "dbg")) {
#ifdef RMX_ABLATE
        c->dbg = (int)value;
        return RMX_OK;
#else
        return fail(c, RMX_E_UNSUPPORTED, "option 'dbg' exists only in the -DRMX_ABLATE timing build");
#endif
    }
    if (!strcmp(key, "pk") || !strcmp(key, "win8")) {
#ifdef RMX_EXPERIMENTS
        (key[0] == 'p' ? c->pk : c->win8) = value != 0;
        return RMX_OK;
#else
        return fail(c, RMX_E_UNSUPPORTED, "option '%s' exists only in the -DRMX_EXPERIMENTS build (k_win8 / k_winp)", key);
#endif
    }
    if (!strcmp(key, "stag")) {
        if (value < 0 || value > 5) return fail(c, RMX_E_INVAL, "stag %ld not in 0..5", value);
        c->stag = (int)value;
        return RMX_OK;
    }
    if (!strcmp(key, "small4096")) {
        c->small_batch = value != 0;
        return RMX_OK;
    }
    if (!strcmp(key, "fused")) {
        c->fused = value != 0;
        return RMX_OK;
    }
    if (!strcmp(key, "resident")) {
        c->resident = value != 0;
        return RMX_OK;
    }
    if (!strcmp(key, "timing")) {
        c->timing = value != 0;
        return RMX_OK;
    }
    return fail(c, RMX_E_INVAL, "unknown option '%s'", key);
}

// spectrum scratch of at least `windows` window slots (B * 64 KiB each)
static int ensure_spec(rmx_ctx* c, long windows) {
    const size_t nb = (size_t)windows * c->n_buoys * (8 * kThreads) * sizeof(float4);
    if (nb <= c->spec_bytes) return RMX_OK;
    RMX_HIP(c, hipStreamSynchronize(c->stream));
    float4* nspec = nullptr;
    RMX_HIP(c, hipMalloc((void**)&nspec, nb));
    (void)hipFree(c->d_spec);
    c->d_spec = nspec;
    c->scratch_bytes += nb - c->spec_bytes;
    c->spec_bytes = nb;
    return RMX_OK;
}

// N = 4096, unfused path: forward spectra of windows [w0, w0 + wc) into d_spec (rot == nullptr) or, de-rotated by
// the phasor table rot[N], into d_spec_r (rmx_caf_batch); then the pair kernels over the plan's pair list
static int fwd4096(rmx_ctx* c, const void* d_iq, int w0, int wc, bool u8, const float2* rot, int n_bins = 1) {
    const long first_item = (long)w0 * c->n_buoys;
    const int wrap = n_bins > 1 ? wc * c->n_buoys : 0;
    const int n_items = wc * c->n_buoys * n_bins;
    if (rot) {
        const size_t nb = n_bins > 1 ? (size_t)n_items * (8 * kThreads) * sizeof(float4)
                                     : (size_t)(wc < c->chunk_windows ? c->chunk_windows : wc) * c->n_buoys * (8 * kThreads) * sizeof(float4);
        if (c->spec_r_bytes < nb) {
            RMX_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_spec_r) (void)hipFree(c->d_spec_r);
            c->d_spec_r = nullptr;
            c->spec_r_bytes = 0;
            RMX_HIP(c, hipMalloc((void**)&c->d_spec_r, nb));
            c->spec_r_bytes = nb;
        }
    }
    float4* dst = rot ? c->d_spec_r : c->d_spec;
    RMX_TM_BEGIN(c);
    if (u8)
        hipLaunchKernelGGL(k_fwd<true>, dim3(n_items), dim3(kThreads), kLdsBytes, c->stream, d_iq, dst, c->d_tw1, c->d_tw2,
                           first_item, 1.0f, rot, wrap);
    else
        hipLaunchKernelGGL(k_fwd<false>, dim3(n_items), dim3(kThreads), kLdsBytes, c->stream, d_iq, dst, c->d_tw1, c->d_tw2,
                           first_item, 1.0f, rot, wrap);
    RMX_HIP(c, hipGetLastError());
    RMX_TM_END(c, kTkFwd4096);
    return RMX_OK;
}
static int pairs4096(rmx_ctx* c, int w0, int wc, int n_pairs, int* d_lag, float* d_frac, float* d_peak, float out_scale,
                     bool use_rot, int n_bins = 1) {
    const int n_parts = c->plan_n_parts;
    const int i_wrap = n_bins > 1 ? wc : 0;     // (all hypotheses in one launch: virtual window = hypothesis * wc + window)
    wc *= n_bins;
    const int xcd_map = (wc % 8 == 0) ? 1 : 0;
    const float4* spec_j = use_rot ? c->d_spec_r : c->d_spec;
    RMX_TM_BEGIN(c);
    if (c->resident)
        hipLaunchKernelGGL(k_pair_res, dim3(wc * n_parts), dim3(kThreads), kLdsResBytes, c->stream, (const float4*)c->d_spec,
                           spec_j, c->d_tw1, c->d_tw2, c->d_items, c->d_part_begin, n_parts, c->n_buoys, n_pairs, xcd_map,
                           (long)w0, out_scale, d_lag, d_frac, d_peak, c->dbg, i_wrap);
    else
        hipLaunchKernelGGL(k_pair_str, dim3(wc * n_parts), dim3(kThreads), kLdsBytes, c->stream, (const float4*)c->d_spec, spec_j,
                           c->d_tw1, c->d_tw2, c->d_items, c->d_part_begin, n_parts, c->n_buoys, n_pairs, xcd_map, (long)w0,
                           out_scale, d_lag, d_frac, d_peak, i_wrap);
    RMX_HIP(c, hipGetLastError());
    RMX_TM_END(c, kTkPair4096);
    return RMX_OK;
}
static float out_scale4096() {
    // power-of-two scaling: the TW1 table carries 2^-6, so the spectra carry 2^-6, the product 2^-12 and
    // the inverse transform (which uses the table once more) 2^-18; the taps get the remaining factor
    int logl = 0;
    while ((1 << logl) < kL) ++logl;
    return std::ldexp(1.0f, 3 * kTw1ScaleLog2 - logl);
}

// host-pointer results: the three arrays live in ONE device block.  Small batches (the seam's one frequency group per
// call) come back as ONE copy into pinned memory + three memcpys instead of three pageable copies (each of which the
// runtime stages and synchronises on its own); large ones keep the three direct copies.
static constexpr size_t kPackedOutMax = 256 * 1024;   // bytes of all three arrays
static int ensure_out(rmx_ctx* c, size_t out_elems) {
    if (c->d_out_elems >= out_elems) return RMX_OK;
    if (c->d_lag) (void)hipFree(c->d_lag);
    c->d_lag = nullptr; c->d_frac = nullptr; c->d_peak = nullptr; c->d_out_elems = 0;
    const size_t stride = (out_elems + 63) & ~(size_t)63;           // 256-byte aligned sub-arrays
    RMX_HIP(c, hipMalloc((void**)&c->d_lag, 4 * stride * sizeof(float)));
    c->d_frac = reinterpret_cast<float*>(c->d_lag) + stride;
    c->d_peak = c->d_frac + stride;
    c->d_dop = reinterpret_cast<int*>(c->d_peak + stride);     // (rmx_caf_batch's fourth array)
    c->d_out_elems = out_elems;
    return RMX_OK;
}
static int fetch_out(rmx_ctx* c, size_t out_elems, int32_t* lag_int, float* lag_frac, float* peak, int32_t* dop = nullptr) {
    const size_t nb = out_elems * sizeof(float);
    const char* last = reinterpret_cast<char*>(dop ? (void*)c->d_dop : (void*)c->d_peak);
    const size_t span = (size_t)(last - reinterpret_cast<char*>(c->d_lag)) + nb;
    if (span <= kPackedOutMax) {
        if (c->h_out_bytes < span) {
            if (c->h_out) (void)hipHostFree(c->h_out);
            c->h_out = nullptr; c->h_out_bytes = 0;
            RMX_HIP(c, hipHostMalloc(&c->h_out, kPackedOutMax, hipHostMallocDefault));
            c->h_out_bytes = kPackedOutMax;
        }
        RMX_HIP(c, hipMemcpyAsync(c->h_out, c->d_lag, span, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipStreamSynchronize(c->stream));
        const char* h = static_cast<const char*>(c->h_out);
        std::memcpy(lag_int, h, nb);
        std::memcpy(lag_frac, h + (reinterpret_cast<char*>(c->d_frac) - reinterpret_cast<char*>(c->d_lag)), nb);
        std::memcpy(peak, h + (reinterpret_cast<char*>(c->d_peak) - reinterpret_cast<char*>(c->d_lag)), nb);
        if (dop) std::memcpy(dop, h + (reinterpret_cast<char*>(c->d_dop) - reinterpret_cast<char*>(c->d_lag)), nb);
        return RMX_OK;
    }
    if (dop) RMX_HIP(c, hipMemcpyAsync(dop, c->d_dop, nb, hipMemcpyDeviceToHost, c->stream));
    RMX_HIP(c, hipMemcpyAsync(lag_int, c->d_lag, nb, hipMemcpyDeviceToHost, c->stream));
    RMX_HIP(c, hipMemcpyAsync(lag_frac, c->d_frac, nb, hipMemcpyDeviceToHost, c->stream));
    RMX_HIP(c, hipMemcpyAsync(peak, c->d_peak, nb, hipMemcpyDeviceToHost, c->stream));
    RMX_HIP(c, hipStreamSynchronize(c->stream));
    return RMX_OK;
}

// N = 4096: the model of host_plan.hpp with this ctx's numbers
static double split_cost4096(const rmx_ctx* c, int n_windows, int n_pairs, int* ppb) {
    return host::split_cost4096(c->n_cus, c->n_buoys, n_pairs, n_windows, c->ppb_user ? c->pairs_per_block : 0, ppb);
}

int rmx_xcorr_batch(rmx_ctx* c, const void* iq, int n_windows, const int32_t* pairs, int n_pairs,
                    int32_t* lag_int, float* lag_frac, float* peak, unsigned flags) {
    if (!c) return RMX_E_INVAL;
    if (!iq || !lag_int || !lag_frac || !peak) return fail(c, RMX_E_INVAL, "NULL buffer");
    if (n_windows < 0 || n_windows > c->max_windows)
        return fail(c, RMX_E_INVAL, "n_windows %d not in 0..max_windows=%d", n_windows, c->max_windows);
    const int all_pairs = c->n_buoys * (c->n_buoys - 1) / 2;
    if (!pairs) {
        if (n_pairs != 0 && n_pairs != all_pairs)
            return fail(c, RMX_E_INVAL, "pairs == NULL needs n_pairs == 0 or %d, got %d", all_pairs, n_pairs);
        n_pairs = all_pairs;
    }
    if (n_pairs < 0) return fail(c, RMX_E_INVAL, "n_pairs %d < 0", n_pairs);
    if (n_windows == 0 || n_pairs == 0) return RMX_OK;
    RMX_HIP(c, hipSetDevice(c->device));
    tm_reset(c);
    // Few windows of N = 4096: the fused kernel is one workgroup per WINDOW -- (B + P) transforms in sequence, 92 us for
    // one window of 8 buoys, 344 us for 16 buoys, whatever the rest of the chip does -- while the per-transform kernels
    // spread a window's spectra and pairs over the CUs: 13-15 us for the same single windows (tools/exp_small4096.py;
    // the crossover sits at 64 / 110 / 140 windows for 3 / 8 / 16 buoys).  This is the shape of the reference's seam:
    // one frequency group per call (tdoa_processor.py:363-377).
    // Both paths are costed with a small model read off that table (us: 2.6 per transform of the fused kernel; 3.5 per
    // round of forward workgroups, two per CU; 4.0 per pair workgroup + 3.3 per pair in it, one workgroup per CU; a
    // round that fills the chip runs up to 45 % slower than a lone workgroup, less so when many rounds follow each other
    // out of step) and the cheaper one runs; the same model picks the pairs per workgroup (1 ... 7) unless the caller
    // set them.  Against the measured table it is within 10 % on every row and picks the faster path on all of them.
    bool small = false;
    int ppb_small = 7;
    if (!c->generic && c->fused && c->small_batch && c->n_buoys >= 3 && n_pairs == all_pairs) {
        small = split_cost4096(c, n_windows, n_pairs, &ppb_small) < host::fused_cost4096(c->n_cus, c->n_buoys, n_pairs, n_windows);
    }
    const bool in_dev = flags & RMX_IN_DEVICE, out_dev = flags & RMX_OUT_DEVICE, u8 = flags & RMX_IN_U8;
    // The same arithmetic for the LAST round of a larger batch: W = k CUs + r windows cost the fused kernel k + 1 rounds
    // (300 windows of 8 buoys: two rounds, 184 us); when the model says the r windows are cheaper through the
    // per-transform kernels than a round of the fused one, they go there (after the k full rounds, on the same stream).
    // The decision is taken per CHUNK of windows (a batch-wide tail larger than the last chunk ran past the
    // chunk and past the spectrum scratch): in the chunk loop below a chunk's own partial round, wc mod CUs, takes that
    // route when the model says so for ITS size, so it never exceeds the chunk, and the scratch of min(chunk, CUs) window
    // slots covers it.  `tail` here is the gate -- the largest remainder of ANY chunk that pays (every chunk but the last
    // has chunk_windows windows; ADVICE r04: it used to look at the last chunk only) -- and sizes the pair workgroups
    // (one plan per call) for that largest partial round.
    int tail = 0;
    const bool tail_ok = !small && !c->generic && c->fused && c->small_batch && c->n_buoys >= 3 && n_pairs == all_pairs &&
                         c->n_cus > 0 && (in_dev || n_windows <= kHostSubChunk);
    auto tail_pays = [&](int r, int* q) -> bool {
        return split_cost4096(c, r, n_pairs, q) < 0.9 * host::fused_cost4096(c->n_cus, c->n_buoys, n_pairs, 1);
    };
    if (tail_ok && n_windows > c->n_cus) {
        const long cw = c->chunk_windows;
        const int last_wc = (int)(n_windows - ((long)(n_windows - 1) / cw) * cw);   // windows of the last chunk
        const int r_last = last_wc % c->n_cus;
        const int r_full = (n_windows > cw) ? (int)(cw % c->n_cus) : 0;             // every earlier chunk's remainder
        for (int r : {r_full > r_last ? r_full : r_last, r_full > r_last ? r_last : r_full}) {
            int q = 7;
            if (r != 0 && tail_pays(r, &q)) {
                tail = r;
                ppb_small = q;
                break;
            }
        }
    }
    if (!c->generic && !c->ppb_user) {
        if (!small && !tail && (pairs != nullptr || !c->fused)) {   // a custom pair list (or fused = 0): the per-transform kernels anyway
            int q = 7;
            (void)split_cost4096(c, n_windows, n_pairs, &q);
            ppb_small = q;
            c->pairs_per_block = q;
        } else {
            c->pairs_per_block = (small || tail) ? ppb_small : 7;
        }
    }
    int rc = build_plan(c, pairs, n_pairs);
    if (rc != RMX_OK) return rc;
    const bool fused_now = c->fused && c->plan_all_pairs && !small;
    if (!fused_now) tail = 0;

    const size_t samp_bytes = u8 ? 2 : 8;
    const size_t in_bytes = (size_t)n_windows * c->n_buoys * c->n_samples * samp_bytes;
    const void* d_iq = iq;
    bool pipelined = false;
    if (!in_dev) {
        if (c->d_in_bytes < in_bytes) {
            if (c->d_in) (void)hipFree(c->d_in);
            c->d_in = nullptr;
            c->d_in_bytes = 0;
            RMX_HIP(c, hipMalloc(&c->d_in, in_bytes));
            c->d_in_bytes = in_bytes;
        }
        // fused path: the copy is cut into sub-chunks issued on a second stream, each followed by its
        // kernel launch, so that copy k+1 travels while kernel k runs (below); otherwise one copy up front
        pipelined = !c->generic && fused_now && n_windows > kHostSubChunk;
        if (pipelined) {
            if (!c->copy_stream) RMX_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            for (hipEvent_t& e : c->copy_ev)
                if (!e) RMX_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            // earlier work of this ctx may still read the staging buffer
            RMX_HIP(c, hipEventRecord(c->copy_ev[3], c->stream));
            RMX_HIP(c, hipStreamWaitEvent(c->copy_stream, c->copy_ev[3], 0));
        } else {
            RMX_HIP(c, hipMemcpyAsync(c->d_in, iq, in_bytes, hipMemcpyHostToDevice, c->stream));
        }
        d_iq = c->d_in;
    }
    int* d_lag = lag_int;
    float* d_frac = lag_frac;
    float* d_peak = peak;
    const size_t out_elems = (size_t)n_windows * n_pairs;
    if (!out_dev) {
        {
            const int rc_out = ensure_out(c, out_elems);
            if (rc_out != RMX_OK) return rc_out;
        }
        d_lag = c->d_lag; d_frac = c->d_frac; d_peak = c->d_peak;
    }

    if (c->generic) {
        rc = rmx::generic_batch(c, d_iq, n_windows, n_pairs, d_lag, d_frac, d_peak, u8);
        if (rc != RMX_OK) return rc;
        if (!out_dev) return fetch_out(c, out_elems, lag_int, lag_frac, peak);
        return RMX_OK;
    }
    const float out_scale = out_scale4096();   // (the forward scale 2^-6 rides on the TW1 table)

    {   // scratch: per workgroup for the fused kernel, per window of a chunk otherwise
        const bool fused_path = fused_now;
        const long chunk_w = n_windows < c->chunk_windows ? n_windows : c->chunk_windows;
        rc = ensure_spec(c, fused_path ? (chunk_w < c->n_cus ? chunk_w : c->n_cus) : chunk_w);
        if (rc != RMX_OK) return rc;
    }
    int n_sub = 0;
    for (int w0 = 0; w0 < n_windows; w0 += c->chunk_windows) {
        const int wc = (n_windows - w0 < c->chunk_windows) ? n_windows - w0 : c->chunk_windows;
        if (fused_now) {
            // this chunk's partial round (see above): only behind at least one full round of this call, and only when the
            // model says so for ITS size -- with the default chunk (a multiple of the CU count) that is the batch's last
            // partial round; 0 <= wtail < CUs and wtail <= wc by construction
            int wtail = 0;
            if (tail && (w0 > 0 || wc > c->n_cus) && wc % c->n_cus != 0) {
                int q_unused = 7;
                if (tail_pays(wc % c->n_cus, &q_unused)) wtail = wc % c->n_cus;
            }
            const int wf = wc - wtail;
            const int sub = pipelined ? kHostSubChunk : (wf > 0 ? wf : 1);
            for (int s0 = 0; s0 < wf; s0 += sub) {
                const int sc = wf - s0 < sub ? wf - s0 : sub;
                const long wfirst = (long)w0 + s0;
                if (pipelined) {
                    const size_t off = (size_t)wfirst * c->n_buoys * c->n_samples * samp_bytes;
                    const size_t nb = (size_t)sc * c->n_buoys * c->n_samples * samp_bytes;
                    hipEvent_t ev = c->copy_ev[n_sub % 3];
                    ++n_sub;
                    RMX_HIP(c, hipMemcpyAsync((char*)c->d_in + off, (const char*)iq + off, nb, hipMemcpyHostToDevice,
                                              c->copy_stream));
                    RMX_HIP(c, hipEventRecord(ev, c->copy_stream));
                    RMX_HIP(c, hipStreamWaitEvent(c->stream, ev, 0));
                }
                // one persistent workgroup per CU (the kernel's LDS footprint allows exactly one)
                const int wgrid = sc < c->n_cus ? sc : c->n_cus;
                RMX_TM_BEGIN(c);
#ifdef RMX_EXPERIMENTS
                if (c->win8) {
                    if (u8)
                        hipLaunchKernelGGL(w8::k_win8<true>, dim3(wgrid), dim3(w8::kT8), w8::kLdsWin8Bytes, c->stream, d_iq,
                                           c->d_spec, c->d_tw1_8, c->d_tb8, c->d_tc8, c->n_buoys, wfirst, out_scale, d_lag,
                                           d_frac, d_peak, sc, c->stag);
                    else
                        hipLaunchKernelGGL(w8::k_win8<false>, dim3(wgrid), dim3(w8::kT8), w8::kLdsWin8Bytes, c->stream, d_iq,
                                           c->d_spec, c->d_tw1_8, c->d_tb8, c->d_tc8, c->n_buoys, wfirst, out_scale, d_lag,
                                           d_frac, d_peak, sc, c->stag);
                } else if (c->pk) {
                    if (u8)
                        hipLaunchKernelGGL(k_winp<true>, dim3(wgrid), dim3(kThreads), pk::kLdsWinpBytes, c->stream, d_iq, c->d_spec,
                                           c->d_tw1, c->d_tw2, c->n_buoys, wfirst, out_scale, d_lag, d_frac, d_peak, sc);
                    else
                        hipLaunchKernelGGL(k_winp<false>, dim3(wgrid), dim3(kThreads), pk::kLdsWinpBytes, c->stream, d_iq, c->d_spec,
                                           c->d_tw1, c->d_tw2, c->n_buoys, wfirst, out_scale, d_lag, d_frac, d_peak, sc);
                } else
#endif
                if (u8)
                    hipLaunchKernelGGL(k_win<true>, dim3(wgrid), dim3(kThreads), kLdsWinBytes, c->stream, d_iq, c->d_spec,
                                       c->d_tw1, c->d_tw2, c->n_buoys, wfirst, out_scale, d_lag, d_frac,
                                       d_peak, sc, c->dbg, c->stag);
                else
                    hipLaunchKernelGGL(k_win<false>, dim3(wgrid), dim3(kThreads), kLdsWinBytes, c->stream, d_iq, c->d_spec,
                                       c->d_tw1, c->d_tw2, c->n_buoys, wfirst, out_scale, d_lag, d_frac,
                                       d_peak, sc, c->dbg, c->stag);
                RMX_HIP(c, hipGetLastError());
                RMX_TM_END(c, kTkPair4096);
            }
            if (wtail) {
                rc = fwd4096(c, d_iq, w0 + wf, wtail, u8, nullptr);
                if (rc != RMX_OK) return rc;
                rc = pairs4096(c, w0 + wf, wtail, n_pairs, d_lag, d_frac, d_peak, out_scale, false);
                if (rc != RMX_OK) return rc;
            }
            continue;
        }
        rc = fwd4096(c, d_iq, w0, wc, u8, nullptr);
        if (rc != RMX_OK) return rc;
        rc = pairs4096(c, w0, wc, n_pairs, d_lag, d_frac, d_peak, out_scale, false);
        if (rc != RMX_OK) return rc;
    }
    if (!out_dev) return fetch_out(c, out_elems, lag_int, lag_frac, peak);
    return RMX_OK;
}

int rmx_caf_batch(rmx_ctx* c, const void* iq, int n_windows, const int32_t* pairs, int n_pairs,
                  const double* doppler_cps, int n_dopplers, int32_t* dop_idx, int32_t* lag_int, float* lag_frac,
                  float* peak, unsigned flags) {
    if (!c) return RMX_E_INVAL;
    if (!iq || !dop_idx || !lag_int || !lag_frac || !peak || !doppler_cps) return fail(c, RMX_E_INVAL, "NULL buffer");
    if (n_dopplers < 1 || n_dopplers > 4096) return fail(c, RMX_E_INVAL, "n_dopplers %d not in 1..4096", n_dopplers);
    if (n_windows < 0 || n_windows > c->max_windows)
        return fail(c, RMX_E_INVAL, "n_windows %d not in 0..max_windows=%d", n_windows, c->max_windows);
    const int B = c->n_buoys, N = c->n_samples;
    const int all_pairs = B * (B - 1) / 2;
    if (!pairs) {
        if (n_pairs != 0 && n_pairs != all_pairs)
            return fail(c, RMX_E_INVAL, "pairs == NULL needs n_pairs == 0 or %d, got %d", all_pairs, n_pairs);
        n_pairs = all_pairs;
    }
    if (n_pairs < 0) return fail(c, RMX_E_INVAL, "n_pairs %d < 0", n_pairs);
    if (n_windows == 0 || n_pairs == 0) return RMX_OK;
    RMX_HIP(c, hipSetDevice(c->device));
    // N = 4096, one chunk: all hypotheses in ONE forward launch and ONE pair launch (virtual window = hypothesis x
    // window) instead of three launches per hypothesis -- 21 hypotheses on one frequency group: 431 -> ~60 us
    const bool batch_bins = !c->generic && n_dopplers > 1 && n_windows <= c->chunk_windows &&
                            (size_t)n_dopplers * n_windows * B * (8 * kThreads) * sizeof(float4) <= ((size_t)1 << 30) &&
                            (long)n_dopplers * n_windows < (1L << 20);
    if (!c->generic && !c->ppb_user) {               // (N = 4096: blocks sized to this batch, not to the previous call's)
        int q = 7;
        (void)split_cost4096(c, batch_bins ? n_windows * n_dopplers : n_windows, n_pairs, &q);
        c->pairs_per_block = q;
    }
    int rc = build_plan(c, pairs, n_pairs);          // validates the pair list
    if (rc != RMX_OK) return rc;
    const bool in_dev = flags & RMX_IN_DEVICE, out_dev = flags & RMX_OUT_DEVICE, u8 = flags & RMX_IN_U8;
    const size_t in_bytes = (size_t)n_windows * B * N * (u8 ? 2 : 8);
    const void* d_iq = iq;
    if (!in_dev) {
        if (c->d_in_bytes < in_bytes) {
            if (c->d_in) (void)hipFree(c->d_in);
            c->d_in = nullptr; c->d_in_bytes = 0;
            RMX_HIP(c, hipMalloc(&c->d_in, in_bytes));
            c->d_in_bytes = in_bytes;
        }
        RMX_HIP(c, hipMemcpyAsync(c->d_in, iq, in_bytes, hipMemcpyHostToDevice, c->stream));
        d_iq = c->d_in;
    }
    // phasor table [D][N]: exp(-2 pi i nu n) in double, rounded once to float
    std::vector<double> grid(doppler_cps, doppler_cps + n_dopplers);
    if (grid != c->caf_grid || c->caf_rot_elems < (size_t)n_dopplers * N) {
        std::vector<float2> rot((size_t)n_dopplers * N);
        for (int d = 0; d < n_dopplers; ++d)
            for (int n = 0; n < N; ++n) {
                const double a = -6.283185307179586476925286766559 * grid[d] * (double)n;
                rot[(size_t)d * N + n] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        RMX_HIP(c, hipStreamSynchronize(c->stream));
        if (c->caf_rot) (void)hipFree(c->caf_rot);
        c->caf_rot = nullptr; c->caf_rot_elems = 0;
        RMX_HIP(c, hipMalloc((void**)&c->caf_rot, rot.size() * sizeof(float2)));
        RMX_HIP(c, hipMemcpy(c->caf_rot, rot.data(), rot.size() * sizeof(float2), hipMemcpyHostToDevice));
        c->caf_rot_elems = rot.size();
        c->caf_grid = grid;
    }
    const size_t out_elems = (size_t)n_windows * n_pairs;
    const size_t caf_elems = batch_bins ? out_elems * (size_t)n_dopplers : out_elems;   // per-hypothesis results: [d][w][pair]
    if (c->caf_out_elems < caf_elems) {
        for (void* p : {(void*)c->caf_lag, (void*)c->caf_frac, (void*)c->caf_peak, (void*)c->caf_dop})
            if (p) (void)hipFree(p);
        c->caf_lag = nullptr; c->caf_frac = nullptr; c->caf_peak = nullptr; c->caf_dop = nullptr; c->caf_out_elems = 0;
        RMX_HIP(c, hipMalloc((void**)&c->caf_lag, caf_elems * sizeof(int)));
        RMX_HIP(c, hipMalloc((void**)&c->caf_frac, caf_elems * sizeof(float)));
        RMX_HIP(c, hipMalloc((void**)&c->caf_peak, caf_elems * sizeof(float)));
        c->caf_out_elems = caf_elems;
    }
    int *b_dop = dop_idx, *b_lag = lag_int;
    float *b_frac = lag_frac, *b_peak = peak;
    if (!out_dev) {
        {
            const int rc_out = ensure_out(c, out_elems);
            if (rc_out != RMX_OK) return rc_out;
        }
        b_dop = c->d_dop; b_lag = c->d_lag; b_frac = c->d_frac; b_peak = c->d_peak;
    }
    // Per chunk of windows: the un-rotated spectra once, then per hypothesis the de-rotated spectra (the rotation
    // is applied as the window is loaded), the pair kernels with X_i un-rotated and X_j de-rotated, and the
    // running d-major first maximum.  No augmented copy of the windows, no second engine.
    tm_reset(c);
    int chunk;
    if (c->generic) {
        rc = rmx::generic_ensure(c, n_pairs);
        if (rc != RMX_OK) return rc;
        chunk = c->g_chunk;
    } else {
        chunk = c->chunk_windows;
        rc = ensure_spec(c, n_windows < chunk ? n_windows : chunk);
        if (rc != RMX_OK) return rc;
    }
    const float osc = out_scale4096();
    if (batch_bins) {
        rc = fwd4096(c, d_iq, 0, n_windows, u8, nullptr);
        if (rc == RMX_OK) rc = fwd4096(c, d_iq, 0, n_windows, u8, c->caf_rot, n_dopplers);
        if (rc == RMX_OK) rc = pairs4096(c, 0, n_windows, n_pairs, c->caf_lag, c->caf_frac, c->caf_peak, osc, true, n_dopplers);
        if (rc == RMX_OK) rc = tm_begin(c);
        if (rc == RMX_OK) {
            hipLaunchKernelGGL(k_caf_select_all, dim3((unsigned)((out_elems + 255) / 256)), dim3(256), 0, c->stream, n_dopplers,
                               (long)out_elems, c->caf_lag, c->caf_frac, c->caf_peak, b_dop, b_lag, b_frac, b_peak);
            if (hipGetLastError() != hipSuccess) rc = fail(c, RMX_E_HIP, "k_caf_select_all launch failed");
        }
        if (rc == RMX_OK) rc = tm_end(c, kTkCafSelect);
    }
    for (int w0 = 0; !batch_bins && w0 < n_windows && rc == RMX_OK; w0 += chunk) {
        const int wc = n_windows - w0 < chunk ? n_windows - w0 : chunk;
        rc = c->generic ? rmx::generic_forward(c, d_iq, w0, wc, u8, nullptr) : fwd4096(c, d_iq, w0, wc, u8, nullptr);
        for (int d = 0; d < n_dopplers && rc == RMX_OK; ++d) {
            const float2* rot = c->caf_rot + (size_t)d * N;
            rc = c->generic ? rmx::generic_forward(c, d_iq, w0, wc, u8, rot) : fwd4096(c, d_iq, w0, wc, u8, rot);
            if (rc != RMX_OK) break;
            rc = c->generic ? rmx::generic_pairs(c, w0, wc, n_pairs, c->caf_lag, c->caf_frac, c->caf_peak, true)
                            : pairs4096(c, w0, wc, n_pairs, c->caf_lag, c->caf_frac, c->caf_peak, osc, true);
            if (rc != RMX_OK) break;
            const long first = (long)w0 * n_pairs, cnt = (long)wc * n_pairs;
            rc = tm_begin(c);
            if (rc != RMX_OK) break;
            hipLaunchKernelGGL(k_caf_select, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, d, first, cnt,
                               c->caf_lag, c->caf_frac, c->caf_peak, b_dop, b_lag, b_frac, b_peak);
            if (hipGetLastError() != hipSuccess) rc = fail(c, RMX_E_HIP, "k_caf_select launch failed");
            if (rc == RMX_OK) rc = tm_end(c, kTkCafSelect);
        }
    }
    if (rc != RMX_OK) return rc;
    if (!out_dev) return fetch_out(c, out_elems, lag_int, lag_frac, peak, dop_idx);
    return RMX_OK;
}

int rmx_solve_batch(rmx_ctx* c, const double* buoy_xyz, int n_buoys, const int32_t* pairs, int n_pairs,
                    const int32_t* lag_int, const float* lag_frac, const float* weight, double sample_rate_hz,
                    int n_windows, int max_iter, double* pos, double* cost, int32_t* iters, unsigned flags) {
    if (!c) return RMX_E_INVAL;
    if (!buoy_xyz || !lag_int || !lag_frac || !pos || !cost || !iters) return fail(c, RMX_E_INVAL, "NULL buffer");
    if (n_buoys < 2 || n_buoys > rmx::kSolveMaxBuoys)
        return fail(c, RMX_E_INVAL, "n_buoys %d not in 2..%d", n_buoys, rmx::kSolveMaxBuoys);
    if (!(sample_rate_hz > 0.0)) return fail(c, RMX_E_INVAL, "sample_rate_hz must be positive");
    if (n_windows < 0 || max_iter < 1) return fail(c, RMX_E_INVAL, "n_windows %d, max_iter %d", n_windows, max_iter);
    const int all_pairs = n_buoys * (n_buoys - 1) / 2;
    std::vector<int32_t> pl;
    if (!pairs) {
        if (n_pairs != 0 && n_pairs != all_pairs)
            return fail(c, RMX_E_INVAL, "pairs == NULL needs n_pairs == 0 or %d, got %d", all_pairs, n_pairs);
        n_pairs = all_pairs;
        for (int i = 0; i < n_buoys; ++i)
            for (int j = i + 1; j < n_buoys; ++j) { pl.push_back(i); pl.push_back(j); }
    } else {
        if (n_pairs < 1 || n_pairs > rmx::kSolveMaxPairs) return fail(c, RMX_E_INVAL, "n_pairs %d out of range", n_pairs);
        for (int q = 0; q < 2 * n_pairs; ++q) {
            if (pairs[q] < 0 || pairs[q] >= n_buoys) return fail(c, RMX_E_INVAL, "pair %d out of range", q / 2);
            pl.push_back(pairs[q]);
        }
    }
    if (n_windows == 0) return RMX_OK;
    RMX_HIP(c, hipSetDevice(c->device));
    if (!c->sv_buoys) RMX_HIP(c, hipMalloc((void**)&c->sv_buoys, rmx::kSolveMaxBuoys * 3 * sizeof(double)));
    if (c->sv_pairs_cap < pl.size()) {
        if (c->sv_pairs) (void)hipFree(c->sv_pairs);
        c->sv_pairs = nullptr; c->sv_pairs_cap = 0;
        RMX_HIP(c, hipMalloc((void**)&c->sv_pairs, pl.size() * sizeof(int)));
        c->sv_pairs_cap = pl.size();
    }
    // (pageable host sources: these copies are synchronous with respect to the host buffers)
    RMX_HIP(c, hipMemcpyAsync(c->sv_buoys, buoy_xyz, (size_t)n_buoys * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RMX_HIP(c, hipMemcpyAsync(c->sv_pairs, pl.data(), pl.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    RMX_HIP(c, hipStreamSynchronize(c->stream));
    const bool in_dev = flags & RMX_IN_DEVICE, out_dev = flags & RMX_OUT_DEVICE;
    const size_t ne = (size_t)n_windows * n_pairs;
    const int* d_li = lag_int;
    const float* d_lf = lag_frac;
    const float* d_wg = weight;
    if (!in_dev) {
        const size_t need = ne * 4 * 3;
        if (c->sv_in_bytes < need) {
            if (c->sv_in) (void)hipFree(c->sv_in);
            c->sv_in = nullptr; c->sv_in_bytes = 0;
            RMX_HIP(c, hipMalloc(&c->sv_in, need));
            c->sv_in_bytes = need;
        }
        char* base = (char*)c->sv_in;
        RMX_HIP(c, hipMemcpyAsync(base, lag_int, ne * 4, hipMemcpyHostToDevice, c->stream));
        RMX_HIP(c, hipMemcpyAsync(base + ne * 4, lag_frac, ne * 4, hipMemcpyHostToDevice, c->stream));
        if (weight) RMX_HIP(c, hipMemcpyAsync(base + ne * 8, weight, ne * 4, hipMemcpyHostToDevice, c->stream));
        d_li = (const int*)base;
        d_lf = (const float*)(base + ne * 4);
        d_wg = weight ? (const float*)(base + ne * 8) : nullptr;
    }
    double *d_pos = pos, *d_cost = cost;
    int* d_it = iters;
    if (!out_dev) {
        const size_t need = (size_t)n_windows * (3 * 8 + 8 + 4);
        if (c->sv_out_bytes < need) {
            if (c->sv_out) (void)hipFree(c->sv_out);
            c->sv_out = nullptr; c->sv_out_bytes = 0;
            RMX_HIP(c, hipMalloc(&c->sv_out, need));
            c->sv_out_bytes = need;
        }
        d_pos = (double*)c->sv_out;
        d_cost = d_pos + (size_t)n_windows * 3;
        d_it = (int*)(d_cost + n_windows);
    }
    hipLaunchKernelGGL(rmx::k_solve, dim3((n_windows + 63) / 64), dim3(64), 0, c->stream, c->sv_buoys, n_buoys, c->sv_pairs,
                       n_pairs, d_li, d_lf, d_wg, 299792458.0 / sample_rate_hz, n_windows, max_iter, d_pos, d_cost, d_it);
    RMX_HIP(c, hipGetLastError());
    if (!out_dev) {
        RMX_HIP(c, hipMemcpyAsync(pos, d_pos, (size_t)n_windows * 3 * 8, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipMemcpyAsync(cost, d_cost, (size_t)n_windows * 8, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipMemcpyAsync(iters, d_it, (size_t)n_windows * 4, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RMX_OK;
}

int rmx_detect_batch(rmx_ctx* c, const void* iq, int n_windows, int n_samples, float threshold_db, int distance,
                     double dc_exclude_bins, float min_confidence, int max_peaks, int32_t* count, int32_t* bin,
                     float* power_db, float* snr_db, float* confidence, float* noise_floor_db, unsigned flags) {
    if (!c) return RMX_E_INVAL;
    if (!iq || !count || !bin || !power_db || !snr_db || !confidence || !noise_floor_db)
        return fail(c, RMX_E_INVAL, "NULL buffer");
    int logn = 0;
    while ((1 << logn) < n_samples) ++logn;
    if (n_samples < 16 || n_samples > 16384 || (1 << logn) != n_samples)
        return fail(c, RMX_E_INVAL, "n_samples %d is not a power of two in 16..16384", n_samples);
    if (n_windows < 0 || distance < 1 || max_peaks < 1)
        return fail(c, RMX_E_INVAL, "n_windows %d, distance %d, max_peaks %d", n_windows, distance, max_peaks);
    if (n_windows == 0) return RMX_OK;
    RMX_HIP(c, hipSetDevice(c->device));
    const int N = n_samples;
    if (c->dt_logn != logn) {
        std::vector<float2> t;
        rmx::gen::make_row_table(t, N);
        if (c->dt_tw) { RMX_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->dt_tw); c->dt_tw = nullptr; }
        RMX_HIP(c, hipMalloc((void**)&c->dt_tw, t.size() * sizeof(float2)));
        RMX_HIP(c, hipMemcpy(c->dt_tw, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
        c->dt_logn = logn;
        RMX_HIP(c, hipFuncSetAttribute((const void*)rmx::det::d_fft_db<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rmx::gen::lp(16384) * 8));
        RMX_HIP(c, hipFuncSetAttribute((const void*)rmx::det::d_fft_db<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rmx::gen::lp(16384) * 8));
        RMX_HIP(c, hipFuncSetAttribute((const void*)rmx::det::d_peaks, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       16384 * 6 + 8192));
    }
    const bool in_dev = flags & RMX_IN_DEVICE, out_dev = flags & RMX_OUT_DEVICE, u8 = flags & RMX_IN_U8;
    auto grow = [&](void** p, size_t* have, size_t need) -> int {
        if (*have >= need) return RMX_OK;
        RMX_HIP(c, hipStreamSynchronize(c->stream));
        if (*p) (void)hipFree(*p);
        *p = nullptr; *have = 0;
        RMX_HIP(c, hipMalloc(p, need));
        *have = need;
        return RMX_OK;
    };
    int rc = grow((void**)&c->dt_pdb, &c->dt_pdb_bytes, (size_t)n_windows * N * sizeof(float));
    if (rc != RMX_OK) return rc;
    const void* d_iq = iq;
    if (!in_dev) {
        const size_t nb = (size_t)n_windows * N * (u8 ? 2 : 8);
        rc = grow(&c->dt_in, &c->dt_in_bytes, nb);
        if (rc != RMX_OK) return rc;
        RMX_HIP(c, hipMemcpyAsync(c->dt_in, iq, nb, hipMemcpyHostToDevice, c->stream));
        d_iq = c->dt_in;
    }
    int *d_count = count, *d_bin = bin;
    float *d_pw = power_db, *d_snr = snr_db, *d_conf = confidence, *d_floor = noise_floor_db;
    const size_t per = (size_t)n_windows * max_peaks;
    if (!out_dev) {
        rc = grow(&c->dt_out, &c->dt_out_bytes, per * 16 + (size_t)n_windows * 8);
        if (rc != RMX_OK) return rc;
        char* b = (char*)c->dt_out;
        d_bin = (int*)b; d_pw = (float*)(b + per * 4); d_snr = (float*)(b + per * 8); d_conf = (float*)(b + per * 12);
        d_count = (int*)(b + per * 16); d_floor = (float*)(b + per * 16 + (size_t)n_windows * 4);
    }
    const int fthr = N >= 4096 ? 1024 : (N >= 1024 ? 256 : 64);
    if (u8)
        hipLaunchKernelGGL(rmx::det::d_fft_db<true>, dim3(n_windows), dim3(fthr), (size_t)rmx::gen::lp(N) * 8, c->stream, d_iq, c->dt_pdb,
                           c->dt_tw, logn);
    else
        hipLaunchKernelGGL(rmx::det::d_fft_db<false>, dim3(n_windows), dim3(fthr), (size_t)rmx::gen::lp(N) * 8, c->stream, d_iq, c->dt_pdb,
                           c->dt_tw, logn);
    RMX_HIP(c, hipGetLastError());
    const int pthr = N >= 4096 ? 1024 : 256;
    const size_t plds = (size_t)N * 4 + N + (size_t)N + 256 * 4 + 8 * 4 + (size_t)pthr * 4;
    hipLaunchKernelGGL(rmx::det::d_peaks, dim3(n_windows), dim3(pthr), plds, c->stream, c->dt_pdb, logn, threshold_db, distance,
                       dc_exclude_bins, min_confidence, max_peaks, d_count, d_bin, d_pw, d_snr, d_conf, d_floor);
    RMX_HIP(c, hipGetLastError());
    if (!out_dev) {
        RMX_HIP(c, hipMemcpyAsync(bin, d_bin, per * 4, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipMemcpyAsync(power_db, d_pw, per * 4, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipMemcpyAsync(snr_db, d_snr, per * 4, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipMemcpyAsync(confidence, d_conf, per * 4, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipMemcpyAsync(count, d_count, (size_t)n_windows * 4, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipMemcpyAsync(noise_floor_db, d_floor, (size_t)n_windows * 4, hipMemcpyDeviceToHost, c->stream));
        RMX_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RMX_OK;
}

int rmx_synchronize(rmx_ctx* c) {
    if (!c) return RMX_E_INVAL;
    RMX_HIP(c, hipSetDevice(c->device));
    RMX_HIP(c, hipStreamSynchronize(c->stream));
    return RMX_OK;
}

int rmx_last_timing(rmx_ctx* c, float* fwd_ms, int* fwd_launches, float* pair_ms, int* pair_launches) {
    if (!c) return RMX_E_INVAL;
    RMX_HIP(c, hipSetDevice(c->device));
    RMX_HIP(c, hipStreamSynchronize(c->stream));
    float tf = 0, tp = 0;
    int nf = 0, np = 0;
    for (size_t k = 0; k < c->ev_kind.size(); ++k) {
        float ms = 0;
        RMX_HIP(c, hipEventElapsedTime(&ms, c->ev[2 * k], c->ev[2 * k + 1]));
        if (c->ev_kind[k] == rmx::kTkFwd4096) { tf += ms; ++nf; }
        else if (c->ev_kind[k] == rmx::kTkPair4096) { tp += ms; ++np; }
    }
    if (fwd_ms) *fwd_ms = tf;
    if (fwd_launches) *fwd_launches = nf;
    if (pair_ms) *pair_ms = tp;
    if (pair_launches) *pair_launches = np;
    return RMX_OK;
}

int rmx_last_timing_kind(rmx_ctx* c, int kind, const char** name, float* ms, int* launches) {
    if (!c) return RMX_E_INVAL;
    if (kind < 0 || kind >= rmx::kTkCount) return RMX_E_INVAL;   // (no error text: callers enumerate kinds until this)
    RMX_HIP(c, hipSetDevice(c->device));
    RMX_HIP(c, hipStreamSynchronize(c->stream));
    float t = 0;
    int n = 0;
    for (size_t k = 0; k < c->ev_kind.size(); ++k) {
        if (c->ev_kind[k] != kind) continue;
        float one = 0;
        RMX_HIP(c, hipEventElapsedTime(&one, c->ev[2 * k], c->ev[2 * k + 1]));
        t += one;
        ++n;
    }
    if (name) *name = rmx::kTimeKindName[kind];
    if (ms) *ms = t;
    if (launches) *launches = n;
    return RMX_OK;
}

#ifndef RMX_SOURCE_DIGEST
#define RMX_SOURCE_DIGEST "unknown"
#endif
// (the marker in front of the digest lets __graft_entry__._stale() read it out of the file without loading the library)
const char* rmx_build_info(void) { return "RMX_BUILD_INFO source_digest=" RMX_SOURCE_DIGEST " arch=gfx950"; }

size_t rmx_scratch_bytes(const rmx_ctx* c) { return c ? c->scratch_bytes : 0; }

}  // extern "C"
