// kwin8k.hpp -- k_win8kl: the reference's streaming capture length N = 8192 (iq_stream_client.py:459) on the machinery of
// the fused N = 4096 kernel (kwin.hpp / fft_r16.hpp).  Round 4 built the first two versions (tools/experiments/kwin8k.hpp:
// k_win8k without a resident anchor, k_win8ka with both anchor halves in registers: neither beat g_win_scr14); round 5's
// third version -- ONE bin-parity half of the anchor resident in LDS, the first anchor's pairs interleaved with the forward
// transforms -- does (8 buoys x 512 windows 0.51 against 0.73 ms, 16 x 256 1.00 against 1.42, 3 x 1024 0.19 against 0.285) and is
// the product kernel from 5/16 of a workgroup per CU on (rmx_hip.hip: generic_batch; option kwin8k = 0: g_win_scr14).
//
// A window zero-padded from N = 8192 to L = 16384 splits by bin parity h into two 8192-point transforms,
//     X[2 kappa + h] = FFT_8192( x[n] W_L^(h n) )[kappa],
// and an 8192-point transform is what k_win's register / LDS network computes: two 4096-point sub-transforms p (bins
// kappa = 2 kappa' + p) on 512 threads x 16 points, thread t = 2u + p, three radix-16 passes, the radix-2 between the
// sub-transforms across lane pairs by DPP.  The only new pieces are the ends:
//   forward  the first radix-2 is not free here (the half is not zero-padded): slot q of thread (u, p) starts from
//            a + (-1)^p (-i)^h b,  a = x[n], b = x[n + 4096], n = u + 256 q,  twisted by W_L^((h + 2p) n): the per-slot part
//            W_64^((h + 2p) q) as constants, the per-thread part W_L^((h + 2p) u) as the half's one factor on top of TW1;
//   inverse  e_h = IFFT_8192(X_j,h conj X_i,h) comes out of k_win's inverse network (same tables: the network runs on
//            (im, re)-swapped data, so every forward factor acts as its conjugate); e_0 waits in 32 registers while e_1
//            is computed, then r[m] = e_0 + T, r[m + 8192] = e_0 - T with T = W_L^(-m) e_1 (the twist is half 1's own
//            sub-transform twist; lanes p = 1 hold m = n + 4096 and take the extra factor +i), |.|^2 of both, and the
//            peak search over the thread's 32 values.
// Per window: 2 B forward and 2 P pair transforms; TW1 = sixteen slot factors in registers (the same for both halves) times one
// per-thread factor per half, the TW2 row comes from LDS (the 30 registers go to e_0).  Default pair list: k_win's schedule -- X_0 straight into the anchor
// (never stored), every further X_j transformed, stored and used at once for (0, j), then the anchors 1 ... B-2 with X_j
// streaming.  Any other pair list: all forward transforms first (an anchor run = consecutive pairs with the same first buoy).
// Scratch per persistent workgroup: [b][half] x 64 KiB in thread-register order.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>
#include <vector>

#include "fft_r16.hpp"
#include "kwin.hpp"

namespace rmx {
namespace k8 {

constexpr int kN8 = 8192;                 // window length
constexpr int kSlots8 = 4, kBatch8 = 3;   // record ring / pairs per resolve (halo rows are 32 wide here)
constexpr int kLds8Tw2 = 2 * kLdsWinImg;
constexpr int kLds8Halo = kLds8Tw2 + kLdsTw2;                          // [slots][8 waves][4 rows][32] float
constexpr int kLds8Red = kLds8Halo + kSlots8 * 8 * 4 * 32 * 4;        // [slots][8] float4
constexpr int kLds8Oidx = kLds8Red + kSlots8 * 8 * 16;                // [slots] int
constexpr int kLds8Pairs = kLds8Oidx + kSlots8 * 4;                    // custom pair list, at most kMaxPairs8 entries
constexpr int kMaxPairs8 = 640;
constexpr int kLds8Bytes = kLds8Pairs + kMaxPairs8 * 8;
static_assert(kLds8Bytes <= 160 * 1024, "k_win8k LDS");

struct Pair2 { int i, j; };               // same layout as gen::GPair

__device__ __forceinline__ float2 w64(int e) {   // exp(-2 pi i e / 64)
    constexpr float c[64] = {1.0f, 0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f, 0.9238795042037964f, 0.8819212913513184f, 0.8314695954322815f, 0.7730104327201843f, 0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f, 0.4713967442512512f, 0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f, 0.0f, -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f, -0.3826834261417389f, -0.4713967442512512f, -0.5555702447891235f, -0.6343932747840881f, -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f, -0.8819212913513184f, -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f, -1.0f, -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f, -0.9238795042037964f, -0.8819212913513184f, -0.8314695954322815f, -0.7730104327201843f, -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f, -0.4713967442512512f, -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f, 0.0f, 0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f, 0.3826834261417389f, 0.4713967442512512f, 0.5555702447891235f, 0.6343932747840881f, 0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f, 0.8819212913513184f, 0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f};
    constexpr float s[64] = {0.0f, -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f, -0.3826834261417389f, -0.4713967442512512f, -0.5555702447891235f, -0.6343932747840881f, -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f, -0.8819212913513184f, -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f, -1.0f, -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f, -0.9238795042037964f, -0.8819212913513184f, -0.8314695954322815f, -0.7730104327201843f, -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f, -0.4713967442512512f, -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f, 0.0f, 0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f, 0.3826834261417389f, 0.4713967442512512f, 0.5555702447891235f, 0.6343932747840881f, 0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f, 0.8819212913513184f, 0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f, 1.0f, 0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f, 0.9238795042037964f, 0.8819212913513184f, 0.8314695954322815f, 0.7730104327201843f, 0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f, 0.4713967442512512f, 0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f};
    return make_float2(c[e & 63], s[e & 63]);
}

// v[q] *= W_64^(R q), q = 1..15, in place (constants as scalar operands: the forms of mul_w32_odd in kwin.hpp)
template <int R>
__device__ __forceinline__ void mul_twist(float2 (&v)[16]) {
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const float2 w = w64(R * q);
        float x = v[q].x, y = v[q].y;
        cmul_inplace(x, y, w.x, w.y);
        v[q].x = x;
        v[q].y = y;
    }
#pragma unroll
    for (int q = 4; q < 16; q += 4) {
        float x0 = v[q].x, y0 = v[q].y, x1 = v[q + 1].x, y1 = v[q + 1].y;
        float x2 = v[q + 2].x, y2 = v[q + 2].y, x3 = v[q + 3].x, y3 = v[q + 3].y;
        cmul4_inplace(x0, y0, x1, y1, x2, y2, x3, y3, w64(R * q), w64(R * (q + 1)), w64(R * (q + 2)), w64(R * (q + 3)));
        v[q].x = x0; v[q].y = y0; v[q + 1].x = x1; v[q + 1].y = y1;
        v[q + 2].x = x2; v[q + 2].y = y2; v[q + 3].x = x3; v[q + 3].y = y3;
    }
}

// 'full' index kk (0 .. 2N-2) -> owner thread, value slot sg = 16 (1 - j) + q  (j = 1: r[m + 8192], the negative lags)
__device__ __forceinline__ void k_to_owner8(int kk, int& tt, int& sg) {
    const int j = kk < kN8 - 1 ? 1 : 0;
    const int m = j ? kk + 1 : kk - (kN8 - 1);        // = n + 4096 p, 0 .. 8191
    const int pp = m >> 12, n = m & 4095;
    tt = 2 * (n & 255) + pp;
    sg = (j ? 0 : 16) + (n >> 8);
}

// resolve_batch of kwin.hpp for this kernel's records: lane = 8 g + r looks at wave r's record of the g-th pending pair
__device__ __forceinline__ void resolve_batch8(int lane, const float4* red, const float* halo, const int* oidx, int first,
                                               int cnt, long obase, float out_scale, int* __restrict__ lag_int,
                                               float* __restrict__ lag_frac, float* __restrict__ peak) {
    const int g = lane >> 3, r = lane & 7;
    const bool act = g < cnt;
    const int slot = (first + (act ? g : 0)) & (kSlots8 - 1);
    const float* rf = reinterpret_cast<const float*>(red) + 4 * (slot * 8 + r);
    const int* ri = reinterpret_cast<const int*>(rf);
    const float ex = act ? rf[0] : -3.0f;
    const int k = act ? ri[1] : 0x7fffffff;
    const float tm = rf[2], tp = rf[3];
    const int out = oidx[slot];
    float gmax = ex;
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0xB1>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x4E>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x141>(__builtin_bit_cast(int, gmax))));
    int kstar = (ex == gmax) ? k : 0x7fffffff;
    kstar = min(kstar, dpp_i<0xB1>(kstar));
    kstar = min(kstar, dpp_i<0x4E>(kstar));
    kstar = min(kstar, dpp_i<0x141>(kstar));
    const bool win = act && ex == gmax && k == kstar;
    auto halo_tap = [&](int kk) -> float {
        kk = kk < 0 ? 0 : (kk > 2 * kN8 - 2 ? 2 * kN8 - 2 : kk);
        int tt, sg;
        k_to_owner8(kk, tt, sg);
        const int ln = tt & 63;
        const int row = ln < 2 ? ln : (ln >= 62 ? ln - 60 : 0);
        return halo[(((slot * 8 + (tt >> 6)) * 4) + row) * 32 + sg];
    };
    const int kc = win ? k : (kN8 - 1);
    const float hm = halo_tap(kc - 1), hp = halo_tap(kc + 1);
    const float b = sqrtf(fmaxf(ex, 0.0f)) * out_scale;
    const float a = sqrtf(tm >= 0.0f ? tm : hm) * out_scale;
    const float c = sqrtf(tp >= 0.0f ? tp : hp) * out_scale;
    const double den = (double)a - 2.0 * (double)b + (double)c;
    float frac = 0.0f;
    if (kc > 0 && kc < 2 * kN8 - 2 && den != 0.0) frac = (float)(0.5 * ((double)a - (double)c) / den);
    if (win) {
        lag_int[obase + out] = kc - (kN8 - 1);
        lag_frac[obase + out] = frac;
        peak[obase + out] = b;
    }
}

// lowest slot of m[0..15] that equals t (16 if none): four select chains, descending so that lower slots win
__device__ __forceinline__ int first_slot_eq(const float (&m)[16], float t) {
    int qa = 16, qb = 16, qc = 16, qd = 16;
    argsel4<12>(qa, qb, qc, qd, m[12], m[13], m[14], m[15], t);
    argsel4<8>(qa, qb, qc, qd, m[8], m[9], m[10], m[11], t);
    argsel4<4>(qa, qb, qc, qd, m[4], m[5], m[6], m[7], t);
    argsel4<0>(qa, qb, qc, qd, m[0], m[1], m[2], m[3], t);
    return min(min(qa, qb), min(qc, qd));
}

// ---- k_win8kl: ONE bin-parity half of the anchor spectrum resident in LDS -----------------------------------------------
// The limit of a kernel without a resident anchor (k_win8k, tools/experiments/) is the scratch traffic of an anchor that
// alternates between its two halves (9 MiB per window at 8 buoys); two register-resident halves (k_win8ka) spill.  Here
// X_i's half 0 stays in the 32 anchor registers and its half 1 in 64 KiB of LDS (thread-private columns: every thread reads back the float4s it stored, conflict free) for the whole run of
// pairs that share the anchor; per transform only the streamed X_j,h travels (4.4 MiB of scratch reads per window at 8 buoys
// instead of 8).  The 64 KiB come out of the second exchange image, so a transform pays a second workgroup barrier (behind
// its role-A reads) and the half-order staggering is gone.
constexpr int kLdsLAnc = kLdsWinImg;                                   // [8][512] float4
constexpr int kLdsLHalo = kLdsLAnc + 8 * kThreads * 16;
constexpr int kLdsLRed = kLdsLHalo + kSlots8 * 8 * 4 * 32 * 4;
constexpr int kLdsLOidx = kLdsLRed + kSlots8 * 8 * 16;
constexpr int kLdsLTw2 = kLdsLOidx + kSlots8 * 4;
constexpr int kLdsLPairs = kLdsLTw2 + kLdsTw2;
constexpr int kLdsLBytes = kLdsLPairs + kMaxPairs8 * 8;
static_assert(kLdsLBytes <= 160 * 1024, "k_win8kl LDS");

template <bool U8>
__global__ __launch_bounds__(kThreads, 2) void k_win8kl(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                       const float4* __restrict__ tw1_g,     // [2 halves][8][512]
                                                       const float2* __restrict__ tw2_g, int n_buoys,
                                                       const Pair2* __restrict__ pairs, int n_pairs, long first_window,
                                                       float out_scale, int* __restrict__ lag_int,
                                                       float* __restrict__ lag_frac, float* __restrict__ peak, int n_win,
                                                       int stag) {   // (stag: unused here -- one exchange image, no half-order staggering; kept so the launch sites of the three N = 8192 kernels match)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* img0 = reinterpret_cast<float2*>(smem);                      // the ONE exchange image
    float4* anc = reinterpret_cast<float4*>(smem + kLdsLAnc);            // anchor half 1: [8][512] float4, thread-private columns
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLdsLTw2);
    float* halo = reinterpret_cast<float*>(smem + kLdsLHalo);
    float4* red = reinterpret_cast<float4*>(smem + kLdsLRed);
    int* oidx = reinterpret_cast<int*>(smem + kLdsLOidx);
    Pair2* plist = reinterpret_cast<Pair2*>(smem + kLdsLPairs);

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;
    const int B = n_buoys;

    load_tw2_to_lds_grouped(tw2_lds, tw2_g, t);
    if (pairs)
        for (int q = t; q < n_pairs; q += kThreads) plist[q] = pairs[q];
    const float4* tw2row = reinterpret_cast<const float4*>(tw2_lds + (t & 15) * kTw2RowF2);
    const int loc_m0 = __builtin_amdgcn_readfirstlane(wave * kLocWave);
    const int loc_rd = wave * kLocWave + loc_read_off(lane);
    const float sgn = p ? -1.0f : 1.0f;
    const int kbase = u + 4096 * p - 1;                    // 'full' index of value sg: kbase + 256 sg + (sg >= 16 ? 4096 : 0)
    const int hl = lane < 2 ? lane : lane - 60;
    const bool is_halo = lane < 2 || lane >= 62;
    __syncthreads();

    const int samp_bytes = U8 ? 2 : 8;
    const int soff = t * 16;
    // TW1 of half h = (W_4096^(u k0) 2^-6) * g_h, g_h = W_L^((h + 2p) u): the sixteen slot factors are the same for both halves and
    // stay in the registers for the whole kernel; g_h is one complex number per thread and half.  (Round 5 first kept a full
    // table per half and re-requested the other half's behind every use -- 64 KiB of L2 reads per transform, as much as the
    // streamed spectrum: 8 x 512 0.513 -> 0.504 ms, 3 x 1024 0.197 -> 0.189, 16 x 256 1.000 -> 0.980 with the split.)
    float2 tw1[16];
    load_tw1(tw1, tw1_g, t);
    float2 g2[2];
    {
        const float2* gq = reinterpret_cast<const float2*>(tw1_g + 8 * kThreads);
        g2[0] = gq[t];
        g2[1] = gq[kThreads + t];
    }

    for (int wl = blockIdx.x; wl < n_win; wl += gridDim.x) {
    C16 sa, sb;      // the two spectra of the next / current pair transform (X_i,h and X_j,h); sample buffers in phase 1
    C16 ev;          // e_0 of the current pair
    const long wbase = (long)blockIdx.x * B * 2;                     // this workgroup's scratch: [b][h] x 64 KiB
    const long obase = (first_window + wl) * (long)n_pairs;
    int seq = 0, npair = 0, npend = 0;

    auto barrier_hook = [&](bool flush) __attribute__((always_inline)) {
        __syncthreads();
        if (npend == kBatch8 || (flush && npend > 0)) {
            if (wave == (seq & 7))
                resolve_batch8(lane, red, halo, oidx, (npair - npend) & (kSlots8 - 1), npend, obase, out_scale, lag_int,
                               lag_frac, peak);
            npend = 0;
        }
    };
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(iq_v)) + (first_window + wl) * (long)B * kN8 * samp_bytes, 0,
        B * kN8 * samp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(spec) + wbase * (long)(8 * kThreads * 16), 0, B * 2 * (8 * kThreads * 16), 0x00020000);
    const int xoff = u * samp_bytes;
    // samples x[n0 + u + 256 q] of buoy b, q = 0..15 (n0 = 0 or 4096), raw
    auto load_x = [&](C16& d, int b, int n0) __attribute__((always_inline)) {
        int bo = (b * kN8 + n0) * samp_bytes;
        asm volatile("" : "+s"(bo));
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                d.re[q] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(xs, xoff, bo + q * 256 * 2, 0));
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(xs, xoff, bo + q * 256 * 8, 0);
                d.set(q, __uint_as_float(r.x), __uint_as_float(r.y));
            }
        }
    };
    auto cvt_x = [&](C16& d) __attribute__((always_inline)) {
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const unsigned r = __float_as_uint(d.re[q]);
                d.set(q, (float)(r & 0xffu) - 127.5f, (float)(r >> 8) - 127.5f);
            }
        }
    };
    // one eighth (part 0..7) of the spectrum at scratch index sidx = 2 b + h into d
    auto load_spec_part = [&](C16& d, int sidx, auto part) __attribute__((always_inline)) {
        constexpr int J = decltype(part)::value;
        int bo = __builtin_amdgcn_readfirstlane(sidx) * (8 * kThreads * 16);
        asm volatile("" : "+s"(bo));
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(ss, soff, bo + J * (kThreads * 16), 0);
        d.set(2 * J, __uint_as_float(w.x), __uint_as_float(w.y));
        d.set(2 * J + 1, __uint_as_float(w.z), __uint_as_float(w.w));
    };
    auto store_spec = [&](const float2 (&x)[16], int sidx) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float e0 = x[2 * j].x, e1 = x[2 * j].y, e2 = x[2 * j + 1].x, e3 = x[2 * j + 1].y;
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
            const u32x4 w = {__float_as_uint(e0), __float_as_uint(e1), __float_as_uint(e2), __float_as_uint(e3)};
            // (whole offset in the VGPR, soffset immediate: the store-data hazard of kwin.hpp's store_spec)
            __builtin_amdgcn_raw_buffer_store_b128(w, ss, soff + (sidx * 8 + j) * (kThreads * 16), 0, 0);
        }
    };
    // per-slot twist W_64^((h + 2p) q) of sub-transform p of half h
    auto twist = [&](float2 (&v)[16], int h) __attribute__((always_inline)) {
        if (h == 0) {
            if (p) mul_twist<2>(v);
        } else {
            if (p) mul_twist<3>(v);
            else mul_twist<1>(v);
        }
    };

    // ---- phase 2: per pair, half 0 then half 1.  X_i's half 0 stays in `sa` and its half 1 in LDS (`anc`, thread-private
    // columns) for the anchor's run of pairs; only the streamed X_j,h is requested per transform.  One exchange image:
    // a second barrier per transform, right behind the role-A reads, frees the image for the next transform's stores.
    // H = the half as a compile-time constant.
    auto pair_h1 = [&](auto hc, const auto& s, auto prefetch) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        float2* img = img0;
        float2 v[16];
        if constexpr (std::is_same_v<std::remove_cv_t<std::remove_reference_t<decltype(s)>>, C16>) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = make_float2(s.im[q], s.re[q]);
        } else {                           // (a spectrum that has just been computed: float2[16])
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = make_float2(s[q].y, s[q].x);
        }
        if constexpr (H == 0) {
            dft16_tw_l1<false>(v, sa);
        } else {
            // the anchor's half 1 from LDS, group by group: float4 j of a thread holds its slots 2j, 2j + 1, so the first-layer
            // groups q0 = 0, 1 (slots q0 + 4m) read j = 0, 2, 4, 6 and the groups 2, 3 read j = 1, 3, 5, 7
            const float4* at = anc + t;
            {
                const float4 a0 = at[0 * kThreads], a2 = at[2 * kThreads], a4 = at[4 * kThreads], a6 = at[6 * kThreads];
                dft4_tw<false>(v[0], v[4], v[8], v[12], make_float2(a0.x, a0.y), make_float2(a2.x, a2.y), make_float2(a4.x, a4.y), make_float2(a6.x, a6.y));
                dft4_tw<false>(v[1], v[5], v[9], v[13], make_float2(a0.z, a0.w), make_float2(a2.z, a2.w), make_float2(a4.z, a4.w), make_float2(a6.z, a6.w));
            }
            {
                const float4 a1 = at[1 * kThreads], a3 = at[3 * kThreads], a5 = at[5 * kThreads], a7 = at[7 * kThreads];
                dft4_tw<false>(v[2], v[6], v[10], v[14], make_float2(a1.x, a1.y), make_float2(a3.x, a3.y), make_float2(a5.x, a5.y), make_float2(a7.x, a7.y));
                dft4_tw<false>(v[3], v[7], v[11], v[15], make_float2(a1.z, a1.w), make_float2(a3.z, a3.w), make_float2(a5.z, a5.w), make_float2(a7.z, a7.w));
            }
        }
#pragma unroll
        for (int q = 0; q < 16; q += 4)
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q + 1].x), "+v"(v[q + 1].y), "+v"(v[q + 2].x),
                         "+v"(v[q + 2].y), "+v"(v[q + 3].x), "+v"(v[q + 3].y));
        __builtin_amdgcn_sched_barrier(0);
        dft16_layer2_emit(v, [&](auto kac, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                 __attribute__((always_inline)) {
            constexpr int ka = decltype(kac)::value;
            loc_write4<ka, ka + 4, ka + 8, ka + 12>(loc_m0, x0, x1, x2, x3);
            prefetch(kac);
        });
        const float4 r0 = tw2row[0], r1 = tw2row[1];
        wave_lds_order();
        loc_read16(smem + loc_rd, v);
        dft16_tw_row_l1(v, tw2row, r0, r1);
        float2* xb = img + xb2_base(t);
        dft16_layer2_emit(v, [&](auto kac, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                 __attribute__((always_inline)) {
            constexpr int ka = decltype(kac)::value;
            xb[ka * 32] = make_float2(x0.x, x0.y);
            xb[(ka + 4) * 32] = make_float2(x1.x, x1.y);
            xb[(ka + 8) * 32] = make_float2(x2.x, x2.y);
            xb[(ka + 12) * 32] = make_float2(x3.x, x3.y);
            prefetch(std::integral_constant<int, ka + 4>{});
        });
    };
    // everything of h2 behind the role-A reads (v = the 16 values read from the image)
    auto pair_h2 = [&](auto hc, float2 (&v)[16], int out_idx) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
        dft16_tw<false>(v, tw1);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = cmul(v[q], g2[h]);
#ifndef K8_NO_TWIST
        twist(v, h);
#endif
        pair_fmac8(v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y, sgn);
        pair_fmac8(v[4].x, v[4].y, v[5].x, v[5].y, v[6].x, v[6].y, v[7].x, v[7].y, sgn);
        pair_fmac8(v[8].x, v[8].y, v[9].x, v[9].y, v[10].x, v[10].y, v[11].x, v[11].y, sgn);
        pair_fmac8(v[12].x, v[12].y, v[13].x, v[13].y, v[14].x, v[14].y, v[15].x, v[15].y, sgn);
        if constexpr (h == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) ev.set(q, v[q].x, v[q].y);
            return;
        } else {
#ifdef K8_NO_PEAK
        {
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += v[q].x * ev.re[q] + v[q].y * ev.im[q];
            if (acc == 12345.678f) lag_int[0] = 1;
            ++npend; ++npair;
            return;
        }
#endif
        // held (x, y) = (Im, Re); lanes p = 1 take T' = i V = (V.y, -V.x): r0 = E + T' (m), r1 = E - T' (m + 8192)
        float m0[16], m1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float tx = p ? v[q].y : v[q].x, ty = p ? -v[q].x : v[q].y;
            const float ax = ev.re[q] + tx, ay = ev.im[q] + ty;
            const float bx = ev.re[q] - tx, by = ev.im[q] - ty;
            m0[q] = fmaf(ax, ax, ay * ay);
            m1[q] = fmaf(bx, bx, by * by);
        }
        if (p == 0 && u == 0) m1[0] = -1.0f;     // m = 8192: lag -N is not part of the 'full' output
        const int rb = npair & (kSlots8 - 1);
        if (is_halo) {
            float4* hp = reinterpret_cast<float4*>(halo + ((rb * 8 + wave) * 4 + hl) * 32);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                hp[q4] = make_float4(m1[4 * q4], m1[4 * q4 + 1], m1[4 * q4 + 2], m1[4 * q4 + 3]);
                hp[4 + q4] = make_float4(m0[4 * q4], m0[4 * q4 + 1], m0[4 * q4 + 2], m0[4 * q4 + 3]);
            }
        }
        float tmax = fmaxf(m0[0], m1[0]);
#pragma unroll
        for (int q = 1; q < 16; ++q) tmax = fmaxf(tmax, fmaxf(m0[q], m1[q]));
        // lowest 'full' index holding the lane's max: the negative lags (m1, value slots 0..15) come first
        const int q1 = first_slot_eq(m1, tmax);
        const int q0 = first_slot_eq(m0, tmax);
        const int sgsel = q1 < 16 ? q1 : 16 + q0;
        const int kq = kbase + 256 * sgsel + (sgsel >= 16 ? 4096 : 0);
        const float wmax = wave_max_f32(tmax);
        const unsigned long long hit = __ballot(tmax == wmax);
        int kw, ls, sgs;
        if (__popcll(hit) == 1) {
            ls = __ffsll((long long)hit) - 1;
            kw = __builtin_amdgcn_readlane(kq, ls);
            sgs = __builtin_amdgcn_readlane(sgsel, ls);
        } else {
            kw = wave_min_i32(tmax == wmax ? kq : 0x7fffffff);
            int ts;
            k_to_owner8(kw, ts, sgs);
            ls = ts & 63;
        }
        typedef float f32v __attribute__((ext_vector_type(32)));
        const f32v mv = {m1[0], m1[1], m1[2],  m1[3],  m1[4],  m1[5],  m1[6],  m1[7],  m1[8],  m1[9],  m1[10],
                         m1[11], m1[12], m1[13], m1[14], m1[15], m0[0],  m0[1],  m0[2],  m0[3],  m0[4],  m0[5],
                         m0[6],  m0[7],  m0[8],  m0[9],  m0[10], m0[11], m0[12], m0[13], m0[14], m0[15]};
        const float sel = mv[__builtin_amdgcn_readfirstlane(sgs)];
        const int seli = __builtin_bit_cast(int, sel);
        const float tapm = ls >= 2 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(seli, ls >= 2 ? ls - 2 : 0)) : -2.0f;
        const float tapp = ls <= 61 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(seli, ls <= 61 ? ls + 2 : 63)) : -2.0f;
        if (lane == 0) {
            const u32x4 rec = {__float_as_uint(wmax), (unsigned)kw, __float_as_uint(tapm), __float_as_uint(tapp)};
            *reinterpret_cast<u32x4*>(red + rb * 8 + wave) = rec;
            if (wave == 0) oidx[rb] = out_idx;
        }
        ++npend;
        ++npair;
        }
    };
    auto all_parts = [&](C16& d, int sidx) __attribute__((always_inline)) {
        load_spec_part(d, sidx, std::integral_constant<int, 0>{}); load_spec_part(d, sidx, std::integral_constant<int, 1>{});
        load_spec_part(d, sidx, std::integral_constant<int, 2>{}); load_spec_part(d, sidx, std::integral_constant<int, 3>{});
        load_spec_part(d, sidx, std::integral_constant<int, 4>{}); load_spec_part(d, sidx, std::integral_constant<int, 5>{});
        load_spec_part(d, sidx, std::integral_constant<int, 6>{}); load_spec_part(d, sidx, std::integral_constant<int, 7>{});
    };
    // ev (eight float4 of an anchor's half 1, as requested from the scratch) -> this thread's column of the LDS anchor
    auto park_anchor = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) anc[j * kThreads + t] = make_float4(ev.re[2 * j], ev.im[2 * j], ev.re[2 * j + 1], ev.im[2 * j + 1]);
    };

    // ---- forward transform of half h of buoy b.  A buoy's samples arrive in pa / pb (x[n], x[n + 4096]: lane pairs ask for
    // the same sample); fold_both() turns them into the two halves' folded inputs in place, fwd_half(h) transforms one into x,
    // in registers.  `lead`: a barrier in front of the role-A stores (the transform before was a forward one too: some
    // wave may still be at its wave-local reads; behind a pair transform its second barrier already says so).  `nb`:
    // the buoy whose samples of this half are requested behind the copy into x.
    C16 pa, pb;
    // both folds of a buoy at once, in place: pa <- a + (-1)^p b (half 0), pb <- a + (-1)^p (-i) b (half 1)
    auto fold_both = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float ar = pa.re[q], ai = pa.im[q], br = pb.re[q], bi = pb.im[q];
            pa.set(q, fmaf(sgn, br, ar), fmaf(sgn, bi, ai));
            pb.set(q, fmaf(sgn, bi, ar), fmaf(-sgn, br, ai));
        }
        // (opaque from here on: the uint8 and the complex64 build must run the SAME arithmetic on these values)
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(pa.re[q]), "+v"(pa.im[q]), "+v"(pb.re[q]), "+v"(pb.im[q]));
    };
    auto fwd_half = [&](auto hc, float2 (&x)[16], bool lead, int nb) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = h == 0 ? pa.get(q) : pb.get(q);
#pragma unroll
        for (int q = 0; q < 16; q += 4)
            asm volatile("" : "+v"(x[q].x), "+v"(x[q].y), "+v"(x[q + 1].x), "+v"(x[q + 1].y), "+v"(x[q + 2].x),
                         "+v"(x[q + 2].y), "+v"(x[q + 3].x), "+v"(x[q + 3].y));
        // The folded half is in x: its sample registers are free, and the next buoy's samples of THAT half travel from here
        // on -- 16 requests per half transform (all 32 behind the second fold: 3 % slower at 3 and 8 buoys, the burst hits
        // every CU of the chip at the same time).  UNCONDITIONAL requests (behind the last buoy: its own samples once more,
        // into dead registers): hipcc merges a condition around them with the loop's exit test and sinks them to the loop
        // latch, where the next fold waits.
#ifndef K8_NO_SAMPLE
        {
            const int nbu = __builtin_amdgcn_readfirstlane(nb);
            if constexpr (h == 0) load_x(pa, nbu, 0);
            else load_x(pb, nbu, kN8 / 2);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        twist(x, h);
        dft16(x);
        mul_tw1(x, tw1);
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = cmul(x[q], g2[h]);
        if (lead) __syncthreads();
        xchg_a2_write(img0, x, t);
        __syncthreads();
        xchg_b2_read(img0, x, t);
        dft16(x);
        const float4 r0 = tw2row[0], r1 = tw2row[1];
        loc_write16(loc_m0, x);
        wave_lds_order();
        loc_read16(smem + loc_rd, x);
        dft16_tw_row(x, tw2row, r0, r1);
        ++seq;
    };

    load_x(pa, 0, 0);
    load_x(pb, 0, kN8 / 2);
    int q0 = 0;                            // first pair of the generic loop below
    int ni = 0, nj = 1;                    // default list: the pair the generic loop is at
#ifdef K8_NO_FWD
    if (n_win < 0)
#endif
    if (!pairs && B >= 2) {
        // ---- default pair list: the first anchor's run INTERLEAVED with the forward transforms, k_win's schedule.  With all
        // forward transforms of a window in one block every CU of the chip streams spectra out (and samples in) at the same
        // time and a forward half transform takes 6.9 us against 4.2-4.5 on a half-empty chip (a pair half transform: 3.3
        // either way); spread over the first run's pairs the chip-wide rate halves.  X_0 is never stored: its half 0 goes
        // straight into the anchor registers, its half 1 into the LDS anchor; every further X_j is transformed, stored and
        // used at once, from registers, for the pair (0, j).
        float2 x[16];
        cvt_x(pa);
        cvt_x(pb);
        fold_both();
        fwd_half(std::integral_constant<int, 0>{}, x, true, 1);
#pragma unroll
        for (int q = 0; q < 16; ++q) sa.set(q, x[q].x, x[q].y);
        fwd_half(std::integral_constant<int, 1>{}, x, true, 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) anc[j * kThreads + t] = make_float4(x[2 * j].x, x[2 * j].y, x[2 * j + 1].x, x[2 * j + 1].y);
        for (int j = 1; j < B; ++j) {
            float2 v[16];
            cvt_x(pa);
            cvt_x(pb);
            fold_both();
            // ---- half 0: X_j,0, stored for the later anchors, and e_0 of (0, j)
            fwd_half(std::integral_constant<int, 0>{}, x, j == 1, j + 1 < B ? j + 1 : j);
#ifndef K8_NO_STORE
            store_spec(x, 2 * j);
#endif
            pair_h1(std::integral_constant<int, 0>{}, x, [&](auto) __attribute__((always_inline)) {});
            barrier_hook(false);
            xchg_a2_read(img0, v, t);
            __syncthreads();
            pair_h2(std::integral_constant<int, 0>{}, v, j - 1);
            // ---- half 1
            fwd_half(std::integral_constant<int, 1>{}, x, false, j + 1 < B ? j + 1 : j);
#ifndef K8_NO_STORE
            store_spec(x, 2 * j + 1);
#endif
            pair_h1(std::integral_constant<int, 1>{}, x, [&](auto) __attribute__((always_inline)) {});
            barrier_hook(false);
            xchg_a2_read(img0, v, t);
            __syncthreads();
            pair_h2(std::integral_constant<int, 1>{}, v, j - 1);
        }
        q0 = B - 1;
        nj = B - 1;                        // (the generic loop's pair counters stand at (0, B - 1))
    } else {
        // ---- a custom pair list: all forward transforms first, every spectrum stored
        for (int b = 0; b < B; ++b) {
            float2 x[16];
            cvt_x(pa);
            cvt_x(pb);
            fold_both();
            fwd_half(std::integral_constant<int, 0>{}, x, true, b + 1 < B ? b + 1 : b);
#ifndef K8_NO_STORE
            store_spec(x, 2 * b);
#endif
            fwd_half(std::integral_constant<int, 1>{}, x, true, b + 1 < B ? b + 1 : b);
#ifndef K8_NO_STORE
            store_spec(x, 2 * b + 1);
#endif
        }
        __syncthreads();
    }

    if (q0 < n_pairs) {
        // (i, j) of pair q: the default list (pairs == nullptr) is the nested loop i < j -- two counters, no memory access;
        // a custom list is read from its copy in LDS (a scalar load here would share lgkmcnt with the exchanges)
        auto next_pair = [&](int q) -> Pair2 {            // pair q, called with q = 1, 2, 3, ... in order
            if (!pairs) {
                if (++nj >= B) { ++ni; nj = ni + 1; }
                return Pair2{ni, nj};
            }
            const int2 v = reinterpret_cast<const int2*>(plist)[q];
            return Pair2{__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y)};
        };
        Pair2 cur = pairs ? Pair2{__builtin_amdgcn_readfirstlane(plist[0].i), __builtin_amdgcn_readfirstlane(plist[0].j)}
                          : (q0 ? next_pair(q0) : Pair2{0, 1});
        all_parts(sa, 2 * cur.i);
        all_parts(ev, 2 * cur.i + 1);
        all_parts(sb, 2 * cur.j);
        park_anchor();                         // (phase 1 left half 0's TW1 table in the registers)
        pair_h1(std::integral_constant<int, 0>{}, sb, [&](auto part) __attribute__((always_inline)) {
#ifndef K8_NO_SPEC
            load_spec_part(sb, 2 * cur.j + 1, part);           // the same pair's half 1
#endif
        });
        bool pend_anchor = false;
        for (int q = q0; q < n_pairs; ++q) {
            const bool has_next = q + 1 < n_pairs;
            const Pair2 nxt = has_next ? next_pair(q + 1) : cur;
            const bool new_anchor = has_next && nxt.i != cur.i;
            float2 v[16];
            // ---- half 0
            barrier_hook(false);
            xchg_a2_read(img0, v, t);
#ifndef K8_NO_B2
            __syncthreads();                    // every wave holds its inputs: the image is free for the next transform
#endif
            if (pend_anchor) park_anchor();     // (requested one transform ago into ev, which half 0 overwrites just below)
            pend_anchor = false;
            pair_h2(std::integral_constant<int, 0>{}, v, q);
            pair_h1(std::integral_constant<int, 1>{}, sb, [&](auto part) __attribute__((always_inline)) {
#ifndef K8_NO_SPEC
                if constexpr (decltype(part)::value == 0) {
                    if (new_anchor) all_parts(sa, 2 * nxt.i);                  // sa is idle during this half (the anchor comes from LDS)
                }
                load_spec_part(sb, 2 * nxt.j, part);           // next pair's half 0 (behind the last pair: an index that exists)
#endif
            });
            ++seq;
            // ---- half 1
            barrier_hook(false);
            xchg_a2_read(img0, v, t);
#ifndef K8_NO_B2
            __syncthreads();
#endif
            pair_h2(std::integral_constant<int, 1>{}, v, q);
            if (has_next) {
                pair_h1(std::integral_constant<int, 0>{}, sb, [&](auto part) __attribute__((always_inline)) {
#ifndef K8_NO_SPEC
                    if constexpr (decltype(part)::value == 0) {
                        if (new_anchor) all_parts(ev, 2 * nxt.i + 1);          // e_0 is dead until the next half 0
                    }
                    load_spec_part(sb, 2 * nxt.j + 1, part);
#endif
                });
                pend_anchor = new_anchor;
            }
            ++seq;
            cur = nxt;
        }
    }
    seq = 0;
    barrier_hook(true);
    }   // next window of this workgroup
}

// host: TW1 of k_win8kl -- [8][512] float4: the slot factors W_4096^(u k0) 2^-6 in register order (as build_tables), then
// [2][512] float2: the halves' per-thread factors W_16384^((h + 2p) u); TW2 is k_win's table
inline void build_tables8kl(std::vector<float4>& tw1) {
    const double two_pi = 6.283185307179586476925286766559;
    tw1.assign(8 * kThreads + kThreads, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int j = 0; j < 8; ++j)
        for (int t = 0; t < kThreads; ++t) {
            const int u = t >> 1;
            float w[4];
            for (int e = 0; e < 2; ++e) {
                const double ang = -two_pi * (double)((u * (2 * j + e)) % kM) / (double)kM;
                w[2 * e] = (float)(std::cos(ang) * kTw1Scale);
                w[2 * e + 1] = (float)(std::sin(ang) * kTw1Scale);
            }
            tw1[j * kThreads + t] = make_float4(w[0], w[1], w[2], w[3]);
        }
    float2* gq = reinterpret_cast<float2*>(tw1.data() + 8 * kThreads);
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < kThreads; ++t) {
            const int p = t & 1, u = t >> 1;
            const double ang = -two_pi * (double)((h + 2 * p) * u) / 16384.0;
            gq[h * kThreads + t] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
}

}  // namespace k8
}  // namespace rmx
