// kwin16k.hpp -- k16_fwd / k16_pairs: the reference's capture length N = 16384 (buoy_node.py:364) on the register / LDS
// network of the fused N = 4096 kernel (kwin.hpp / fft_r16.hpp), as TWO kernels per chunk of windows.
//
// A window zero-padded to L = 32768 splits by bin residue h = k mod 4 into four 8192-point QUARTER transforms
//     X[4 kappa + h] = FFT_8192( W_L^(h m) (x[m] + (-i)^h x[m + 8192]) )[kappa],            m = 0 .. 8191,
// and an 8192-point transform is what k_win's network computes (two 4096-point sub-transforms p on 512 threads x 16
// points, thread t = 2u + p, three radix-16 passes, the radix-2 between the sub-transforms across lane pairs by DPP; see
// kwin8k.hpp for the same construction at N = 8192).  Slot q of thread (u, p) of quarter h starts from
//     [ (x0 + (-i)^h x2) + (-1)^p W_8^h (x1 + (-i)^h x3) ] W_L^((h + 4p) n),     x_r = x[n + 4096 r],  n = u + 256 q:
// the per-thread part W_L^((h + 4p) u) is one complex factor per thread and quarter on top of TW1's sixteen slot factors (the same
// for every quarter: they stay in registers), the per-slot part W_128^((h + 4p) q) is a row of a 1 KiB LDS table.
// The inverse of pair (i, j): e_h = IFFT_8192(X_j,h conj X_i,h) out of the inverse network, t_h[m] = W_L^(-h m) e_h[m]
// (lanes p = 1 hold m = n + 4096 and take the extra factor W_8^(-h)), and the four quarters meet in ONE radix-4 butterfly
//     r[m + 8192 s] = sum_h i^(h s) t_h[m]:     s = 0: a + c,  1: b + i d,  2: a - c,  3: b - i d,
//     a = t_0 + t_2,  b = t_0 - t_2,  c = t_1 + t_3,  d = t_1 - t_3
// -- quarter order 0, 2, 1, 3; b and t_1 wait in registers, a in 64 KiB of LDS (thread-private columns) --, then |.|^2 of
// the thread's 64 values and the peak search.
//
// Why two kernels and not one persistent workgroup per window (g_win_eo15, k_win8kl): with three partial results parked no
// anchor quarter can stay resident, so BOTH operands of every quarter transform stream from the spectra in HBM / cache;
// a private 2 MiB of spectra per workgroup (8 buoys) times 256 workgroups is twice the memory-side cache and never hits
// the XCD's L2 (k_win8k, the same situation at N = 8192: 4.9 us per transform at 7 TB/s of cache traffic).  Here the 32
// workgroups of an XCD work on the pairs of the SAME one or two windows at a time (item order below), whose spectra then
// come out of that XCD's 4 MiB L2: the k_win8k probe with XCD-shared spectra runs at 3.4 us per transform
// (tools/probe/k8_bench.hip -DK8_SHARE=8; LABNOTES R5.6).
//   k16_fwd    one (window, buoy) per workgroup turn: samples once (64 per thread), four quarter transforms, four 64 KiB
//              quarter spectra in thread-register order to spec[(w B + b) 4 + h]
//   k16_pairs  persistent, grid = 8 S: workgroup b belongs to XCD x = b mod 8 (round-robin dispatch) and walks the items
//              s, s + S, ... (s = b / 8) of that XCD's list [(window x, pair 0..P-1), (window x + 8, ...), ...]
// -DK16_NO_SPEC / K16_NO_PEAK / K16_NO_SAMPLE / K16_NO_STORE: timing-only builds of the harness (results wrong);
// -DK16_STORE_AUX=n: cache policy bits of the spectrum stores (measured: +-1 %).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>
#include <vector>

#include "fft_r16.hpp"
#include "kwin.hpp"
#include "kwin8k.hpp"

namespace rmx {
namespace k16 {

constexpr int kN16 = 16384;                               // window length
constexpr int kQuarterBytes = 8 * kThreads * 16;          // one quarter spectrum, thread-register order (65536)
constexpr int kMaxPairs16 = 640;
constexpr int kRing = 2;                                  // record ring (a pair is resolved behind the next barrier)

// pair kernel LDS
constexpr int kLdsPark = kLdsWinImg;                                  // a = t_0 + t_2: [8][512] float4
constexpr int kLdsHalo = kLdsPark + 8 * kThreads * 16;                // [ring][8 waves][4 rows][64] float
constexpr int kLdsRed = kLdsHalo + kRing * 8 * 4 * 64 * 4;            // [ring][8 waves][2 phases] float4
constexpr int kLdsOidx = kLdsRed + kRing * 8 * 2 * 16;                // [ring] long long: output index of the pair
constexpr int kLdsTw2p = kLdsOidx + kRing * 8;
constexpr int kLdsTws = kLdsTw2p + kLdsTw2;                           // [8][16] float2: W_128^(R q)
constexpr int kLdsPairs = kLdsTws + 8 * 16 * 8;
constexpr int kLdsPairBytes = kLdsPairs + kMaxPairs16 * 8;
static_assert(kLdsPairBytes <= 160 * 1024, "k16_pairs LDS");
// forward kernel LDS: the exchange image, TW2, the twist rows
constexpr int kLdsFTw2 = kLdsWinImg;
constexpr int kLdsFTws = kLdsFTw2 + kLdsTw2;
constexpr int kLdsFwdBytes = kLdsFTws + 8 * 16 * 8;

using k8::Pair2;

// 'full' index kk (0 .. 2N-2) -> owner thread, value slot sg = 16 rho + q (rho = (s + 2) & 3: the negative lags first)
__device__ __forceinline__ void k_to_owner16(int kk, int& tt, int& sg) {
    const int rho = (kk + 1) >> 13, m = (kk + 1) & 8191;       // kk + 1 = 8192 rho + m,  m = n + 4096 p
    const int pp = m >> 12, n = m & 4095;
    tt = 2 * (n & 255) + pp;
    sg = 16 * rho + (n >> 8);
}

// one pending pair: lane r < 16 looks at record r = 2 wave + phase
__device__ __forceinline__ void resolve16(int lane, const float4* red, const float* halo, const long long* oidx, int slot,
                                          float out_scale, int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                          float* __restrict__ peak) {
    const int r = lane & 15;
    const bool act = lane < 16;
    const float* rf = reinterpret_cast<const float*>(red) + 4 * (slot * 16 + r);
    const int* ri = reinterpret_cast<const int*>(rf);
    const float ex = act ? rf[0] : -3.0f;
    const int k = act ? ri[1] : 0x7fffffff;
    const float tm = rf[2], tp = rf[3];
    const long long out = oidx[slot];
    float gmax = ex;
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0xB1>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x4E>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x141>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x140>(__builtin_bit_cast(int, gmax))));
    int kstar = (ex == gmax) ? k : 0x7fffffff;
    kstar = min(kstar, dpp_i<0xB1>(kstar));
    kstar = min(kstar, dpp_i<0x4E>(kstar));
    kstar = min(kstar, dpp_i<0x141>(kstar));
    kstar = min(kstar, dpp_i<0x140>(kstar));
    const bool win = act && ex == gmax && k == kstar;
    auto halo_tap = [&](int kk) -> float {
        kk = kk < 0 ? 0 : (kk > 2 * kN16 - 2 ? 2 * kN16 - 2 : kk);
        int tt, sg;
        k_to_owner16(kk, tt, sg);
        const int ln = tt & 63;
        const int row = ln < 2 ? ln : (ln >= 62 ? ln - 60 : 0);
        return halo[(((slot * 8 + (tt >> 6)) * 4) + row) * 64 + sg];
    };
    const int kc = win ? k : (kN16 - 1);
    const float hm = halo_tap(kc - 1), hp = halo_tap(kc + 1);
    const float b = sqrtf(fmaxf(ex, 0.0f)) * out_scale;
    const float a = sqrtf(tm >= 0.0f ? tm : hm) * out_scale;
    const float c = sqrtf(tp >= 0.0f ? tp : hp) * out_scale;
    const double den = (double)a - 2.0 * (double)b + (double)c;
    float frac = 0.0f;
    if (kc > 0 && kc < 2 * kN16 - 2 && den != 0.0) frac = (float)(0.5 * ((double)a - (double)c) / den);
    if (win) {
        lag_int[out] = kc - (kN16 - 1);
        lag_frac[out] = frac;
        peak[out] = b;
    }
}

// v[q] *= row[q], q = 0..15 (row: eight float4 of the LDS twist table)
__device__ __forceinline__ void mul_row(float2 (&v)[16], const float4* row) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 w = row[j];
        v[2 * j] = cmul(v[2 * j], make_float2(w.x, w.y));
        v[2 * j + 1] = cmul(v[2 * j + 1], make_float2(w.z, w.w));
    }
}

// ---- forward: samples of (window, buoy) -> four quarter spectra ---------------------------------------------------------
template <bool U8>
__global__ __launch_bounds__(kThreads, 2) void k16_fwd(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                      const float4* __restrict__ tw1_g,     // [8][512]: W_4096^(u k0) 2^-6
                                                      const float2* __restrict__ gq_g,      // [4 quarters][512]: W_L^((h + 4p) u)
                                                      const float2* __restrict__ tw2_g, const float2* __restrict__ tws_g,
                                                      long first_item,     // (window, buoy) index of item 0 in iq_v
                                                      int n_items) {       // windows x buoys of this launch; spec index 0 ..
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* img0 = reinterpret_cast<float2*>(smem);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLdsFTw2);
    float2* tws_lds = reinterpret_cast<float2*>(smem + kLdsFTws);

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;
    load_tw2_to_lds_grouped(tw2_lds, tw2_g, t);
    if (t < 128) tws_lds[t] = tws_g[t];
    const float4* tw2row = reinterpret_cast<const float4*>(tw2_lds + (t & 15) * kTw2RowF2);
    const int loc_m0 = __builtin_amdgcn_readfirstlane(wave * kLocWave);
    const int loc_rd = wave * kLocWave + loc_read_off(lane);
    const float sgn = p ? -1.0f : 1.0f;
    constexpr float kRh = 0.70710678118654752440f;
    const float sgr = sgn * kRh;
    __syncthreads();

    const int samp_bytes = U8 ? 2 : 8;
    const int soff = t * 16;
    // TW1 = (W_4096^(u k0) 2^-6) g_h,  g_h = W_L^((h + 4p) u): the 16 slot factors are the same for every quarter and stay in
    // registers; g_h is one complex number per thread and quarter (a per-quarter TW1 table would be 64 KiB of L2 reads per
    // transform, a third of the pair kernel's traffic)
    float2 tw1[16];
    load_tw1(tw1, tw1_g, t);
    float2 g4[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) g4[h] = gq_g[h * kThreads + t];
    // samples x[n + 4096 r], n = u + 256 q, raw (uint8: the two bytes in the low half of the word); after fold_all:
    // xr[h] = quarter h's folded input
    C16 x0, x1, x2, x3;
    const int xoff = u * samp_bytes;
    auto item_rsrc = [&](long item) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(iq_v)) + (first_item + item) * (long)kN16 * samp_bytes, 0,
            kN16 * samp_bytes, 0x00020000);
    };
    auto load_group = [&](C16& d, const __amdgpu_buffer_rsrc_t& xs, auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                d.re[q] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(xs, xoff, (r * 4096 + q * 256) * 2, 0));
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(xs, xoff, (r * 4096 + q * 256) * 8, 0);
                d.set(q, __uint_as_float(w.x), __uint_as_float(w.y));
            }
        }
    };
    auto cvt = [&](C16& d) __attribute__((always_inline)) {
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const unsigned r = __float_as_uint(d.re[q]);
                d.set(q, (float)(r & 0xffu) - 127.5f, (float)(r >> 8) - 127.5f);
            }
        }
    };

    long item = blockIdx.x;
    if (item >= n_items) return;
    {
        const __amdgpu_buffer_rsrc_t xs = item_rsrc(item);
        load_group(x0, xs, std::integral_constant<int, 0>{}); load_group(x1, xs, std::integral_constant<int, 1>{});
        load_group(x2, xs, std::integral_constant<int, 2>{}); load_group(x3, xs, std::integral_constant<int, 3>{});
    }
    for (; item < n_items; item += gridDim.x) {
        const long nxt = item + gridDim.x < n_items ? item + gridDim.x : item;
        const __amdgpu_buffer_rsrc_t xn = item_rsrc(nxt);
        const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(spec) + item * (long)(4 * kQuarterBytes), 0, 4 * kQuarterBytes, 0x00020000);
        cvt(x0); cvt(x1); cvt(x2); cvt(x3);
        // (opaque from here on: the uint8 and the complex64 build must run the SAME arithmetic on these values)
#pragma unroll
        for (int q = 0; q < 16; ++q)
            asm volatile("" : "+v"(x0.re[q]), "+v"(x0.im[q]), "+v"(x1.re[q]), "+v"(x1.im[q]), "+v"(x2.re[q]), "+v"(x2.im[q]),
                         "+v"(x3.re[q]), "+v"(x3.im[q]));
        // All four folds at once, in place (x_h <- quarter h's input): the sixteen registers pairs of quarter h are then free
        // as soon as ITS transform has started, and the next item's sample group h is requested into them right there -- 16
        // requests per quarter transform.  (All 64 behind the last fold instead -- one burst per item, more requests in flight
        // than vmcnt can count -- made the forward kernel 1.6 x slower: 0.177 against 0.113 ms without any sample request.)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float ar = x0.re[q] + x2.re[q], ai = x0.im[q] + x2.im[q];        // h = 0: A = x0 + x2, B = x1 + x3
            const float br = x1.re[q] + x3.re[q], bi = x1.im[q] + x3.im[q];
            const float cr = x0.re[q] - x2.re[q], ci = x0.im[q] - x2.im[q];        // h = 2: A = x0 - x2, B = x1 - x3, W_8^2 B = (B.im, -B.re)
            const float dr = x1.re[q] - x3.re[q], di = x1.im[q] - x3.im[q];
            const float er = x0.re[q] + x2.im[q], ei = x0.im[q] - x2.re[q];        // h = 1: A = x0 - i x2, B = x1 - i x3, W_8 B = rh (B.re + B.im, B.im - B.re)
            const float fr = x1.re[q] + x3.im[q], fi = x1.im[q] - x3.re[q];
            const float gr = x0.re[q] - x2.im[q], gi = x0.im[q] + x2.re[q];        // h = 3: A = x0 + i x2, B = x1 + i x3, W_8^3 B = rh (B.im - B.re, -(B.re + B.im))
            const float kr = x1.re[q] - x3.im[q], ki = x1.im[q] + x3.re[q];
            x0.set(q, fmaf(sgn, br, ar), fmaf(sgn, bi, ai));
            x1.set(q, fmaf(sgr, fr + fi, er), fmaf(sgr, fi - fr, ei));
            x2.set(q, fmaf(sgn, di, cr), fmaf(-sgn, dr, ci));
            x3.set(q, fmaf(sgr, ki - kr, gr), fmaf(-sgr, kr + ki, gi));
        }
        auto quarter = [&](auto hc, C16& xin) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
            float2 x[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = xin.get(q);
#pragma unroll
            for (int q = 0; q < 16; q += 4)
                asm volatile("" : "+v"(x[q].x), "+v"(x[q].y), "+v"(x[q + 1].x), "+v"(x[q + 1].y), "+v"(x[q + 2].x),
                             "+v"(x[q + 2].y), "+v"(x[q + 3].x), "+v"(x[q + 3].y));
            // (behind the last item: its own samples once more, into dead registers -- unconditional requests, see kwin8k.hpp)
#ifndef K16_NO_SAMPLE
            load_group(xin, xn, hc);
#endif
            __builtin_amdgcn_sched_barrier(0);
            mul_row(x, reinterpret_cast<const float4*>(tws_lds + (h + 4 * p) * 16));
            dft16(x);
            mul_tw1(x, tw1);
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = cmul(x[q], g4[h]);
            __syncthreads();                   // (some wave may still be at the wave-local reads of the transform before)
            xchg_a2_write(img0, x, t);
            __syncthreads();
            xchg_b2_read(img0, x, t);
            dft16(x);
            const float4 r0 = tw2row[0], r1 = tw2row[1];
            loc_write16(loc_m0, x);
            wave_lds_order();
            loc_read16(smem + loc_rd, x);
            dft16_tw_row(x, tw2row, r0, r1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float e0 = x[2 * j].x, e1 = x[2 * j].y, e2 = x[2 * j + 1].x, e3 = x[2 * j + 1].y;
                asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
                const u32x4 w = {__float_as_uint(e0), __float_as_uint(e1), __float_as_uint(e2), __float_as_uint(e3)};
#ifndef K16_NO_STORE
#ifndef K16_STORE_AUX
#define K16_STORE_AUX 0
#endif
                __builtin_amdgcn_raw_buffer_store_b128(w, ss, soff + (h * 8 + j) * (kThreads * 16), 0, K16_STORE_AUX);
#else
                if (e0 == 12345.678f) __builtin_amdgcn_raw_buffer_store_b128(w, ss, soff + (h * 8 + j) * (kThreads * 16), 0, 0);
#endif
            }
        };
        quarter(std::integral_constant<int, 0>{}, x0);
        quarter(std::integral_constant<int, 1>{}, x1);
        quarter(std::integral_constant<int, 2>{}, x2);
        quarter(std::integral_constant<int, 3>{}, x3);
    }
}

// ---- pairs --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 2) void k16_pairs(const float4* __restrict__ spec,     // [window][buoy][4][8][512]
                                                        const float4* __restrict__ tw1_g, const float2* __restrict__ gq_g,
                                                        const float2* __restrict__ tw2_g, const float2* __restrict__ tws_g,
                                                        int n_buoys,
                                                        const Pair2* __restrict__ pairs, int n_pairs,
                                                        long out_first,       // output index of (window 0, pair 0)
                                                        int n_win, int flat,  // flat: items b, b + grid, ... of [(window 0, pair 0 .. P-1), (window 1, ...)]
                                                        float out_scale, int* __restrict__ lag_int,
                                                        float* __restrict__ lag_frac, float* __restrict__ peak) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* img0 = reinterpret_cast<float2*>(smem);
    float4* park = reinterpret_cast<float4*>(smem + kLdsPark);
    float* halo = reinterpret_cast<float*>(smem + kLdsHalo);
    float4* red = reinterpret_cast<float4*>(smem + kLdsRed);
    long long* oidx = reinterpret_cast<long long*>(smem + kLdsOidx);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLdsTw2p);
    float2* tws_lds = reinterpret_cast<float2*>(smem + kLdsTws);
    Pair2* plist = reinterpret_cast<Pair2*>(smem + kLdsPairs);

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;
    const int B = n_buoys, P = n_pairs;

    // this workgroup's items: XCD x = blockIdx mod 8 owns the windows x, x + 8, ...; its workgroups s = 0 .. S-1 take the
    // items s, s + S, ... of [(window, pair 0 .. P-1), (window + 8, ...), ...]
    // (flat: fewer windows than would keep eight XCDs evenly busy -- one window of 16 buoys is 120 items for ONE XCD otherwise)
    const int xcd = flat ? 0 : blockIdx.x & 7, S = flat ? gridDim.x : gridDim.x >> 3, wstep = flat ? 1 : 8;
    const int nwx = flat ? n_win : (n_win > xcd ? (n_win - xcd + 7) >> 3 : 0);
    const long n_it = (long)nwx * P;
    long it = flat ? blockIdx.x : blockIdx.x >> 3;
    if (it >= n_it) return;

    load_tw2_to_lds_grouped(tw2_lds, tw2_g, t);
    if (t < 128) tws_lds[t] = tws_g[t];
    for (int q = t; q < P; q += kThreads) plist[q] = pairs[q];
    const float4* tw2row = reinterpret_cast<const float4*>(tw2_lds + (t & 15) * kTw2RowF2);
    const int loc_m0 = __builtin_amdgcn_readfirstlane(wave * kLocWave);
    const int loc_rd = wave * kLocWave + loc_read_off(lane);
    const float sgn = p ? -1.0f : 1.0f;
    const int kbase = u + 4096 * p - 1;                    // 'full' index of value sg = 16 rho + q: kbase + 256 q + 8192 rho
    const int hl = lane < 2 ? lane : lane - 60;
    const bool is_halo = lane < 2 || lane >= 62;
    constexpr float kRh = 0.70710678118654752440f;
    // lanes p = 1: the factor W_8^(-h) as (a, b) of x' = a x + b y, y' = a y - b x on the held (Im, Re); lanes p = 0: (1, 0)
    const float f1a = p ? kRh : 1.0f, f1b = p ? kRh : 0.0f;
    const float f3a = p ? -kRh : 1.0f, f3b = p ? kRh : 0.0f;
    __syncthreads();

    const int soff = t * 16;
    float2 tw1[16];      // W_4096^(u k0) 2^-6, the same for every quarter; g4[h] = W_L^((h + 4p) u) completes quarter h's TW1
    load_tw1(tw1, tw1_g, t);
    float2 g4[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) g4[h] = gq_g[h * kThreads + t];
    // one eighth (part 0..7) of the quarter spectrum at index sidx = b * 4 + h of the window behind `rs`
    auto load_part = [&](C16& d, const __amdgpu_buffer_rsrc_t& rs, int sidx, auto part) __attribute__((always_inline)) {
        constexpr int J = decltype(part)::value;
        int bo = __builtin_amdgcn_readfirstlane(sidx) * kQuarterBytes;
        asm volatile("" : "+s"(bo));
#ifndef K16_NO_SPEC
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, soff, bo + J * (kThreads * 16), 0);
        d.set(2 * J, __uint_as_float(w.x), __uint_as_float(w.y));
        d.set(2 * J + 1, __uint_as_float(w.z), __uint_as_float(w.w));
#endif
    };
    auto window_rsrc = [&](int w) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(spec)) + (long)w * B * (long)(4 * kQuarterBytes), 0,
            B * 4 * kQuarterBytes, 0x00020000);
    };

    C16 sa, sb;          // X_i,h and X_j,h of the NEXT quarter transform
    C16 p0, p1;          // t_0, then b = t_0 - t_2;  t_1
    int seq = 0, npair = 0, npend = 0;

    // first half of a quarter transform: (X_j conj X_i) through the first two passes into the exchange image; the
    // operands of the transform behind it are requested part by part from inside the butterfly layers
    auto pair_h1 = [&](auto prefetch) __attribute__((always_inline)) {
        float2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_float2(sb.im[q], sb.re[q]);
        dft16_tw_l1<false>(v, sa);
#pragma unroll
        for (int q = 0; q < 16; q += 4)
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q + 1].x), "+v"(v[q + 1].y), "+v"(v[q + 2].x),
                         "+v"(v[q + 2].y), "+v"(v[q + 3].x), "+v"(v[q + 3].y));
        __builtin_amdgcn_sched_barrier(0);
        dft16_layer2_emit(v, [&](auto kac, const float2& a0, const float2& a1, const float2& a2, const float2& a3)
                                 __attribute__((always_inline)) {
            constexpr int ka = decltype(kac)::value;
            loc_write4<ka, ka + 4, ka + 8, ka + 12>(loc_m0, a0, a1, a2, a3);
            prefetch(kac);
        });
        const float4 r0 = tw2row[0], r1 = tw2row[1];
        wave_lds_order();
        loc_read16(smem + loc_rd, v);
        dft16_tw_row_l1(v, tw2row, r0, r1);
        float2* xb = img0 + xb2_base(t);
        dft16_layer2_emit(v, [&](auto kac, const float2& a0, const float2& a1, const float2& a2, const float2& a3)
                                 __attribute__((always_inline)) {
            constexpr int ka = decltype(kac)::value;
            xb[ka * 32] = make_float2(a0.x, a0.y);
            xb[(ka + 4) * 32] = make_float2(a1.x, a1.y);
            xb[(ka + 8) * 32] = make_float2(a2.x, a2.y);
            xb[(ka + 12) * 32] = make_float2(a3.x, a3.y);
            prefetch(std::integral_constant<int, ka + 4>{});
        });
    };
    // second half: last pass, twists, the radix-2 between the sub-transforms; v = t_h (lanes p = 1: -W-corrected, see top)
    auto pair_h2 = [&](auto hc, float2 (&v)[16]) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
        dft16_tw<false>(v, tw1);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = cmul(v[q], g4[h]);
        mul_row(v, reinterpret_cast<const float4*>(tws_lds + (h + 4 * p) * 16));
        pair_fmac8(v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y, sgn);
        pair_fmac8(v[4].x, v[4].y, v[5].x, v[5].y, v[6].x, v[6].y, v[7].x, v[7].y, sgn);
        pair_fmac8(v[8].x, v[8].y, v[9].x, v[9].y, v[10].x, v[10].y, v[11].x, v[11].y, sgn);
        pair_fmac8(v[12].x, v[12].y, v[13].x, v[13].y, v[14].x, v[14].y, v[15].x, v[15].y, sgn);
        if constexpr (h == 1 || h == 3) {
            const float fa = h == 1 ? f1a : f3a, fb = h == 1 ? f1b : f3b;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float x = v[q].x, y = v[q].y;
                v[q].x = fmaf(fb, y, fa * x);
                v[q].y = fmaf(-fb, x, fa * y);
            }
        } else if constexpr (h == 2) {         // W_8^(-2) = i on lanes p = 1: (x, y) -> (y, -x)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float x = v[q].x, y = v[q].y;
                v[q].x = p ? y : x;
                v[q].y = p ? -x : y;
            }
        }
    };

    auto barrier_hook = [&]() __attribute__((always_inline)) {
        __syncthreads();
        if (npend) {
            if (wave == (seq & 7))
                resolve16(lane, red, halo, oidx, (npair - 1) & (kRing - 1), out_scale, lag_int, lag_frac, peak);
            npend = 0;
        }
    };

    // ---- the item loop
    int wl = xcd + wstep * (int)(it / P);      // window of the current item, pair index
    int pq = (int)(it % P);
    auto all_parts = [&](C16& d, const __amdgpu_buffer_rsrc_t& rs, int sidx) __attribute__((always_inline)) {
        load_part(d, rs, sidx, std::integral_constant<int, 0>{}); load_part(d, rs, sidx, std::integral_constant<int, 1>{});
        load_part(d, rs, sidx, std::integral_constant<int, 2>{}); load_part(d, rs, sidx, std::integral_constant<int, 3>{});
        load_part(d, rs, sidx, std::integral_constant<int, 4>{}); load_part(d, rs, sidx, std::integral_constant<int, 5>{});
        load_part(d, rs, sidx, std::integral_constant<int, 6>{}); load_part(d, rs, sidx, std::integral_constant<int, 7>{});
    };
    {
        const __amdgpu_buffer_rsrc_t rs = window_rsrc(wl);
        const int2 pr = reinterpret_cast<const int2*>(plist)[pq];
        all_parts(sa, rs, __builtin_amdgcn_readfirstlane(pr.x) * 4);
        all_parts(sb, rs, __builtin_amdgcn_readfirstlane(pr.y) * 4);
    }
    for (; it < n_it; it += S) {
        // the item behind this one (behind the last: this one again -- indices that exist, results unused)
        int wn = wl, qn = pq + S;
        while (qn >= P) { qn -= P; wn += wstep; }
        const bool has_next = it + S < n_it;
        if (!has_next) { wn = wl; qn = pq; }
        const int2 prc = reinterpret_cast<const int2*>(plist)[pq];
        const int2 prn = reinterpret_cast<const int2*>(plist)[qn];
        const int ci = __builtin_amdgcn_readfirstlane(prc.x) * 4, cj = __builtin_amdgcn_readfirstlane(prc.y) * 4;
        const int ni = __builtin_amdgcn_readfirstlane(prn.x) * 4, nj = __builtin_amdgcn_readfirstlane(prn.y) * 4;
        const __amdgpu_buffer_rsrc_t rc = window_rsrc(wl);
        const __amdgpu_buffer_rsrc_t rn = window_rsrc(wn);
        const long long out = out_first + (long long)wl * P + pq;
        float2 v[16];

        // ---- quarter 0
        pair_h1([&](auto part) __attribute__((always_inline)) { load_part(sa, rc, ci + 2, part); load_part(sb, rc, cj + 2, part); });
        barrier_hook();
        xchg_a2_read(img0, v, t);
        __syncthreads();
        pair_h2(std::integral_constant<int, 0>{}, v);
#pragma unroll
        for (int q = 0; q < 16; ++q) p0.set(q, v[q].x, v[q].y);
        ++seq;
        // ---- quarter 2: a = t_0 + t_2 -> LDS, b = t_0 - t_2 stays
        pair_h1([&](auto part) __attribute__((always_inline)) { load_part(sa, rc, ci + 1, part); load_part(sb, rc, cj + 1, part); });
        barrier_hook();
        xchg_a2_read(img0, v, t);
        __syncthreads();
        pair_h2(std::integral_constant<int, 2>{}, v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            park[j * kThreads + t] = make_float4(p0.re[2 * j] + v[2 * j].x, p0.im[2 * j] + v[2 * j].y,
                                                 p0.re[2 * j + 1] + v[2 * j + 1].x, p0.im[2 * j + 1] + v[2 * j + 1].y);
#pragma unroll
        for (int q = 0; q < 16; ++q) p0.set(q, p0.re[q] - v[q].x, p0.im[q] - v[q].y);
        ++seq;
        // ---- quarter 1
        pair_h1([&](auto part) __attribute__((always_inline)) { load_part(sa, rc, ci + 3, part); load_part(sb, rc, cj + 3, part); });
        barrier_hook();
        xchg_a2_read(img0, v, t);
        __syncthreads();
        pair_h2(std::integral_constant<int, 1>{}, v);
#pragma unroll
        for (int q = 0; q < 16; ++q) p1.set(q, v[q].x, v[q].y);
        ++seq;
        // ---- quarter 3, then the radix-4 butterfly and the peak search
        pair_h1([&](auto part) __attribute__((always_inline)) { load_part(sa, rn, ni, part); load_part(sb, rn, nj, part); });
        barrier_hook();
        xchg_a2_read(img0, v, t);
        __syncthreads();
        pair_h2(std::integral_constant<int, 3>{}, v);
        ++seq;
        {
            // held (x, y) = (Im, Re); i z = (z.y, -z.x) there.  Value arrays by rho (ascending 'full' index): 0: a - c (s = 2),
            // 1: b - i d (s = 3), 2: a + c (s = 0), 3: b + i d (s = 1).  The search runs in TWO phases of 32 values (64 live
            // magnitudes beside the next item's operands spill): rho = 1, 3 from b, d; then rho = 0, 2 from a (LDS), c.  Each
            // phase leaves one record per wave; resolve16 takes the larger (ties: the lower 'full' index) of the sixteen.
            const int rb = npair & (kRing - 1);
            auto search32 = [&](const float (&lo)[16], const float (&hi)[16], auto rlo_c, auto rhi_c, auto phase_c)
                                __attribute__((always_inline)) {
                constexpr int rlo = decltype(rlo_c)::value, rhi = decltype(rhi_c)::value, phase = decltype(phase_c)::value;
#ifdef K16_NO_PEAK
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc += lo[q] + hi[q];
                if (acc == 12345.678f) lag_int[0] = 1;
#else
                if (is_halo) {
                    float4* hp = reinterpret_cast<float4*>(halo + ((rb * 8 + wave) * 4 + hl) * 64);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        hp[4 * rlo + q4] = make_float4(lo[4 * q4], lo[4 * q4 + 1], lo[4 * q4 + 2], lo[4 * q4 + 3]);
                        hp[4 * rhi + q4] = make_float4(hi[4 * q4], hi[4 * q4 + 1], hi[4 * q4 + 2], hi[4 * q4 + 3]);
                    }
                }
                float tmax = fmaxf(lo[0], hi[0]);
#pragma unroll
                for (int q = 1; q < 16; ++q) tmax = fmaxf(tmax, fmaxf(lo[q], hi[q]));
                const int ql = k8::first_slot_eq(lo, tmax), qh = k8::first_slot_eq(hi, tmax);
                const int sgsel = ql < 16 ? ql : 16 + qh;                       // phase-local value slot, lower 'full' index first
                const int kq = kbase + 256 * (sgsel & 15) + (sgsel >= 16 ? 8192 * rhi : 8192 * rlo);
                const float wmax = wave_max_f32(tmax);
                const unsigned long long hit = __ballot(tmax == wmax);
                int kw, ls, sgs;
                if (__popcll(hit) == 1) {
                    ls = __ffsll((long long)hit) - 1;
                    kw = __builtin_amdgcn_readlane(kq, ls);
                    sgs = __builtin_amdgcn_readlane(sgsel, ls);
                } else {
                    kw = wave_min_i32(tmax == wmax ? kq : 0x7fffffff);
                    int ts, sg64;
                    k_to_owner16(kw, ts, sg64);
                    sgs = ((sg64 >> 4) == rhi ? 16 : 0) + (sg64 & 15);
                    ls = ts & 63;
                }
                typedef float f32v __attribute__((ext_vector_type(32)));
                const f32v mv = {lo[0], lo[1], lo[2],  lo[3],  lo[4],  lo[5],  lo[6],  lo[7],  lo[8],  lo[9],  lo[10],
                                 lo[11], lo[12], lo[13], lo[14], lo[15], hi[0],  hi[1],  hi[2],  hi[3],  hi[4],  hi[5],
                                 hi[6],  hi[7],  hi[8],  hi[9],  hi[10], hi[11], hi[12], hi[13], hi[14], hi[15]};
                const float sel = mv[__builtin_amdgcn_readfirstlane(sgs)];
                const int seli = __builtin_bit_cast(int, sel);
                const float tapm = ls >= 2 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(seli, ls >= 2 ? ls - 2 : 0)) : -2.0f;
                const float tapp = ls <= 61 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(seli, ls <= 61 ? ls + 2 : 63)) : -2.0f;
                if (lane == 0) {
                    const u32x4 rec = {__float_as_uint(wmax), (unsigned)kw, __float_as_uint(tapm), __float_as_uint(tapp)};
                    *reinterpret_cast<u32x4*>(red + (rb * 8 + wave) * 2 + phase) = rec;
                    if (wave == 0 && phase == 0) oidx[rb] = out;
                }
#endif
            };
            {
                float m1[16], m3[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float dx = p1.re[q] - v[q].x, dy = p1.im[q] - v[q].y;        // d = t_1 - t_3
                    const float ax = p0.re[q] - dy, ay = p0.im[q] + dx;                 // b - i d
                    const float bx = p0.re[q] + dy, by = p0.im[q] - dx;                 // b + i d
                    p1.set(q, p1.re[q] + v[q].x, p1.im[q] + v[q].y);                    // c = t_1 + t_3
                    m1[q] = fmaf(ax, ax, ay * ay);
                    m3[q] = fmaf(bx, bx, by * by);
                }
                search32(m1, m3, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
            }
            {
                float m0[16], m2[16];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 a = park[j * kThreads + t];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int q = 2 * j + e;
                        const float ar = e ? a.z : a.x, ai = e ? a.w : a.y;
                        const float sx = ar - p1.re[q], sy = ai - p1.im[q], tx = ar + p1.re[q], ty = ai + p1.im[q];
                        m0[q] = fmaf(sx, sx, sy * sy);
                        m2[q] = fmaf(tx, tx, ty * ty);
                    }
                }
                if (p == 0 && u == 0) m0[0] = -1.0f;     // m = 0 of s = 2: lag -N is not part of the 'full' output
                search32(m0, m2, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
            }
            ++npend;
            ++npair;
        }
        wl = wn;
        pq = qn;
    }
    seq = 0;
    barrier_hook();
}

// host: the slot factors of TW1 ([8][512] float4, register order as build_tables), the quarters' per-thread factors g
// ([4][512]), the per-slot twists W_128^(R q) ([8][16]); TW2 is k_win's table
inline void build_tables16k(std::vector<float4>& tw1, std::vector<float2>& gq, std::vector<float2>& tws) {
    const double two_pi = 6.283185307179586476925286766559;
    tw1.resize(8 * kThreads);
    gq.resize(4 * kThreads);
    for (int j = 0; j < 8; ++j)
        for (int t = 0; t < kThreads; ++t) {
            const int u = t >> 1;
            float2 w[2];
            for (int e = 0; e < 2; ++e) {
                const double ang = -two_pi * (double)((u * (2 * j + e)) % kM) / (double)kM;      // W_4096^(u k0), scaled by 2^-6
                w[e] = make_float2((float)(std::cos(ang) * kTw1Scale), (float)(std::sin(ang) * kTw1Scale));
            }
            tw1[j * kThreads + t] = make_float4(w[0].x, w[0].y, w[1].x, w[1].y);
        }
    for (int h = 0; h < 4; ++h)
        for (int t = 0; t < kThreads; ++t) {
            const int p = t & 1, u = t >> 1;
            const double ang = -two_pi * (double)((h + 4 * p) * u) / 32768.0;
            gq[h * kThreads + t] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
    tws.resize(8 * 16);
    for (int r = 0; r < 8; ++r)
        for (int q = 0; q < 16; ++q) {
            const double ang = -two_pi * (double)((r * q) % 128) / 128.0;
            tws[r * 16 + q] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
}

}  // namespace k16
}  // namespace rmx
