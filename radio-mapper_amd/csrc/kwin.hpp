// kwin.hpp -- the fused window kernel of the tuned N = 4096 path (k_win), its peak-record resolve routine, the
// table loaders it shares with k_fwd / the pair kernels, and the host-side builder of its twiddle tables.
// Split out of rmx_hip.hip so that tools/probe/kwin_bench.hip can compile and time this kernel (and experimental
// variants of it) on its own; rmx_hip.hip includes it unchanged.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>
#include <vector>

#include "fft_r16.hpp"

// cache policy of the spectrum scratch accesses (buffer instruction aux bits on gfx950: 1 = sc0, 2 = sc1, 4 = nt); A/B'd in
// tools/probe/kwin_bench.hip (LABNOTES R4.7): the default policy stays
#ifndef RMX_KWIN_LOAD_AUX
#define RMX_KWIN_LOAD_AUX 0
#endif
#ifndef RMX_KWIN_STORE_AUX
#define RMX_KWIN_STORE_AUX 0
#endif

namespace rmx {

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
using u32x2 = unsigned int __attribute__((ext_vector_type(2)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));
// NOTE: __builtin_bit_cast(float, vec.y) on a vector ELEMENT is mis-lowered by this hipcc (every
// element reads lane 0 of the vector); always bit_cast the whole vector, then take elements.

// LDS carve (bytes) of both kernels: exchange image, TW2 table, reduction words
constexpr int kLdsXchg = kXchgF2 * 8;                  // 69632
constexpr int kLdsTw2 = 16 * kTw2RowF2 * 8;            // 2304
constexpr int kLdsRed = 64;
constexpr int kLdsBytes = kLdsXchg + kLdsTw2 + kLdsRed;

__device__ __forceinline__ void load_tw2_to_lds(float2* tw2_lds, const float2* __restrict__ tw2_g, int t) {
    // tw2_g: [16][16] complex; LDS rows padded to kTw2RowF2
    if (t < 256) tw2_lds[(t >> 4) * kTw2RowF2 + (t & 15)] = tw2_g[t];
}

// same table in layer-1 group order for dft16_tw_row: stored[4*q0 + m - 1] = tw2[a][q0 + 4*m]; the
// unused tw2[a][0] = 1 goes to the pad slot 15
__device__ __forceinline__ void load_tw2_to_lds_grouped(float2* tw2_lds, const float2* __restrict__ tw2_g, int t) {
    if (t < 256) {
        const int a = t >> 4, q = t & 15;
        tw2_lds[a * kTw2RowF2 + (q == 0 ? 15 : 4 * (q & 3) + (q >> 2) - 1)] = tw2_g[t];
    }
}

__device__ __forceinline__ void load_tw1(float2 (&tw1)[16], const float4* __restrict__ tw1_g, int t) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 w = tw1_g[j * kThreads + t];
        tw1[2 * j] = make_float2(w.x, w.y);
        tw1[2 * j + 1] = make_float2(w.z, w.w);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused window kernel: one workgroup (512 threads, <= 256 VGPRs, one per CU) per capture window.
//   phase 1  forward spectra of all B buoys -> this window's scratch (thread-private layout: every
//            thread later re-reads exactly the float4s it stored, so no visibility protocol is needed)
//   phase 2  anchor runs over the pair list: X_i stays in registers for its run, X_j streams one pair
//            ahead; conj-multiply merged into the first radix-16 pass; ONE workgroup barrier per pair.
// LDS: two exchange images (alternating by transform/pair, which is what makes one barrier enough:
// a wave only writes its own half-wave regions of the image the others are not reading), the TW2
// table, and two small double-buffered records per wave for the argmax: the wave's winner with its
// in-wave neighbour taps, and a "halo" of the |r|^2 of its lanes 0,1,62,63 for neighbours that sit
// in another wave.  The pair's winner is resolved by one lane after the NEXT pair's barrier.
constexpr int kLdsWinImg = kLdsXchg;                                 // 69632 each, two of them
constexpr int kLdsWinTw2 = 2 * kLdsWinImg;
#ifdef RMX_KWIN_PEAK2
// Peak search, second generation (round 5, VERDICT r04 #1b/c; an A/B build, NOT the default): a wave only finds its maximum
// and the LANE that holds it; that lane and its two neighbours l*-2, l*+2 park their sixteen |r|^2 beside the halo rows, and
// the slot, the 'full' index and the taps are worked out by the resolver (once per batch, on one of the early waves)
// instead of by sixteen compares + sixteen selects in every lane of every pair.  Seven rows per wave and slot instead of
// four: the ring shrinks to 4 slots, batches of 3.  Measured (tools/probe/kwin_bench, same box, bit-identical outputs):
// SQ_INSTS_VALU 9.03e8 -> 8.62e8 per launch (-4.5 %), SQ_WAIT_ANY 3.95e8 -> 4.52e8 (+14 %), 1.675 -> 1.673 ms: the
// instructions removed come back as waiting -- the interval between two barriers is set by dependent latencies (LDS round
// trips, the DPP chain, the barrier itself), not by VALU issue slots.  LABNOTES.md R5.1.
constexpr int kResSlots = 4;
constexpr int kResBatch = 3;
constexpr int kHaloRows = 7;   // lanes 0, 1, 62, 63, then l*-2, l*, l*+2
#ifndef RMX_KWIN_RES_MASK
#define RMX_KWIN_RES_MASK 3    /* the resolving wave rotates over waves 0-3 (A/B: 7 = over all eight) */
#endif
#else
#ifndef RMX_KWIN_RES_MASK
#define RMX_KWIN_RES_MASK 7
#endif
constexpr int kResSlots = 8;   // record ring; winners are resolved in batches of kResBatch pairs
constexpr int kResBatch = 7;   // < kResSlots: the pair after a batch writes a slot the resolver is not reading
constexpr int kHaloRows = 4;
#endif
constexpr int kLdsWinHalo = kLdsWinTw2 + kLdsTw2;                           // [slots][8][rows][16] float
constexpr int kLdsWinRed = kLdsWinHalo + kResSlots * 8 * kHaloRows * 16 * 4;   // [slots][8] float4
constexpr int kLdsWinOidx = kLdsWinRed + kResSlots * 8 * 16;                // [slots] int: output slot of the pair
constexpr int kLdsWinBytes = kLdsWinOidx + kResSlots * 4;
static_assert(kLdsWinBytes <= 160 * 1024, "k_win LDS");

__device__ __forceinline__ void k_to_owner(int kk, int& tt, int& q) {
    const int par = (kk >= kM - 1) ? 0 : 1;
    const int n = par ? (kk + 1) : (kk - (kM - 1));
    tt = 2 * (n & 255) + par;
    q = n >> 8;
}

// Executed by ONE whole wave after a barrier that published the records of `cnt` <= 7 pairs (ring
// slots first, first+1, ...): lane = 8*g + r looks at wave r's record of the g-th pair, two DPP
// reductions over each group of 8 lanes pick (max |r|^2, lowest 'full' index), the neighbour taps
// come from the winner's own record or from the halo rows, and the winning lane of every group
// stores the pair's 12 bytes.  One resolve per 7 pairs instead of one per pair: the resolving wave
// is late to its next barrier by the length of this routine, and the other seven wait for it.
__device__ __forceinline__ void resolve_batch(int lane, const float4* red, const float* halo, const int* oidx,
                                              int first, int cnt, long obase, float out_scale,
                                              int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                              float* __restrict__ peak) {
    const int g = lane >> 3, r = lane & 7;
    const bool act = g < cnt;
    const int slot = (first + g) & (kResSlots - 1);
    // record = {max |r|^2, its lowest 'full' index (int bits), tap k*-1, tap k*+1}; scalar LDS reads
    const float* rf = reinterpret_cast<const float*>(red) + 4 * (slot * 8 + r);
    const int* ri = reinterpret_cast<const int*>(rf);
    const float ex = act ? rf[0] : -3.0f;
    const int k = act ? ri[1] : 0x7fffffff;
    const float tm = rf[2], tp = rf[3];
    const int out = oidx[slot];
    float gmax = ex;                                     // max over the 8 lanes of the group
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0xB1>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x4E>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x141>(__builtin_bit_cast(int, gmax))));
    int kstar = (ex == gmax) ? k : 0x7fffffff;
    kstar = min(kstar, dpp_i<0xB1>(kstar));
    kstar = min(kstar, dpp_i<0x4E>(kstar));
    kstar = min(kstar, dpp_i<0x141>(kstar));
    const bool win = act && ex == gmax && k == kstar;     // exactly one lane per active group
    // halo rows (always read, clamped): only lanes 0,1,62,63 of a wave can own a cross-wave neighbour
    auto halo_tap = [&](int kk) -> float {
        kk = kk < 0 ? 0 : (kk > 2 * kM - 2 ? 2 * kM - 2 : kk);
        int tt, q;
        k_to_owner(kk, tt, q);
        const int ln = tt & 63;
        const int row = ln < 2 ? ln : (ln >= 62 ? ln - 60 : 0);
        return halo[(((slot * 8 + (tt >> 6)) * kHaloRows) + row) * 16 + q];
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

#ifdef RMX_KWIN_PEAK2
// The same for the second-generation records: red[slot][wave] = {max |r|^2 of the wave, the lane l* that holds it}; the
// halo block holds per (slot, wave) seven rows of sixteen |r|^2: lanes 0, 1, 62, 63 (rows 0-3) and l*-2, l*, l*+2 (rows 4-6,
// where those are not halo lanes themselves).  lane = 8*g + r works on wave r's record of the g-th pair of the batch: the
// slot is the lowest one of l*'s row that equals the maximum, the 'full' index follows from (wave, lane, slot).
__device__ __forceinline__ void resolve_batch2(int lane, const float4* red, const float* halo, const int* oidx,
                                               int first, int cnt, long obase, float out_scale,
                                               int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                               float* __restrict__ peak) {
    const int g = lane >> 3, r = lane & 7;
    const bool act = g < cnt;
    const int slot = (first + g) & (kResSlots - 1);
    const float* rf = reinterpret_cast<const float*>(red) + 4 * (slot * 8 + r);
    const float ex_raw = rf[0];
    const int ls = reinterpret_cast<const int*>(rf)[1] & 63;
    auto row_of = [&](int ln) -> int { return ln < 2 ? ln : (ln >= 62 ? ln - 60 : 5); };
    const float4* wrow = reinterpret_cast<const float4*>(halo + ((slot * 8 + r) * kHaloRows + row_of(ls)) * 16);
    int qs = 15;
#pragma unroll
    for (int j = 3; j >= 0; --j) {          // descending: the lowest slot wins
        const float4 v = wrow[j];
        if (v.w == ex_raw) qs = 4 * j + 3;
        if (v.z == ex_raw) qs = 4 * j + 2;
        if (v.y == ex_raw) qs = 4 * j + 1;
        if (v.x == ex_raw) qs = 4 * j;
    }
    const int tw = r * 64 + ls, pw = tw & 1, uw = tw >> 1;
    const float ex = act ? ex_raw : -3.0f;
    const int k = act ? (pw ? uw - 1 : uw + kM - 1) + 256 * qs : 0x7fffffff;
    const int out = oidx[slot];
    float gmax = ex;                                     // max over the 8 lanes of the group
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0xB1>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x4E>(__builtin_bit_cast(int, gmax))));
    gmax = fmaxf(gmax, __builtin_bit_cast(float, dpp_i<0x141>(__builtin_bit_cast(int, gmax))));
    int kstar = (ex == gmax) ? k : 0x7fffffff;
    kstar = min(kstar, dpp_i<0xB1>(kstar));
    kstar = min(kstar, dpp_i<0x4E>(kstar));
    kstar = min(kstar, dpp_i<0x141>(kstar));
    const bool win = act && ex == gmax && k == kstar;     // exactly one lane per active group
    // neighbour taps k*-1, k*+1: lanes l*-2 / l*+2 of the winner's wave (rows 4 / 6), or a halo lane of some wave
    auto tap = [&](int kk) -> float {
        kk = kk < 0 ? 0 : (kk > 2 * kM - 2 ? 2 * kM - 2 : kk);
        int tt, q;
        k_to_owner(kk, tt, q);
        const int ln = tt & 63;
        int row = ln < 2 ? ln : (ln >= 62 ? ln - 60 : 5 + ((ln - ls) >> 1));
        row = row < 0 ? 0 : (row > kHaloRows - 1 ? kHaloRows - 1 : row);   // (lanes that lost their group compute garbage: keep it inside)
        return halo[((slot * 8 + (tt >> 6)) * kHaloRows + row) * 16 + q];
    };
    const int kc = win ? k : (kM - 1);
    const float tm = tap(kc - 1), tp = tap(kc + 1);
    const float b = sqrtf(fmaxf(ex, 0.0f)) * out_scale;
    const float a = sqrtf(fmaxf(tm, 0.0f)) * out_scale;
    const float c = sqrtf(fmaxf(tp, 0.0f)) * out_scale;
    const double den = (double)a - 2.0 * (double)b + (double)c;
    float frac = 0.0f;
    if (kc > 0 && kc < 2 * kM - 2 && den != 0.0) frac = (float)(0.5 * ((double)a - (double)c) / den);
    if (win) {
        lag_int[obase + out] = kc - (kM - 1);
        lag_frac[obase + out] = frac;
        peak[obase + out] = b;
    }
}
#endif

#ifdef RMX_KWIN_STAMPS
// diagnostic build (tools/probe/kwin_bench.hip -DRMX_KWIN_STAMPS): shader-clock stamps of workgroup's wave 0 at the phase
// boundaries of every window it processes; no stamp exists in the product build
__device__ long long rmx_stamps[256 * 64 * 4];
__device__ int rmx_stamps_vm[256 * 64 * 8];
__device__ int rmx_stamps_bar[256 * 64 * 8 * 2];
__device__ int rmx_stamps_p1[256 * 64 * 8 * 8];   // per wave: ticks of phase 1 by piece (RMX_LAP buckets)
__device__ int rmx_stamps_pc[256 * 64 * 8 * 2];    // per wave: ticks in the first / second piece between two barriers of the anchor loop   // per wave: ticks draining its LDS stores in front of the barriers / waiting at them   // per wave: ticks spent in the vmcnt wait at the head of h1, summed over the window's pairs
#define RMX_STAMP(slot)                                                                              \
    do {                                                                                             \
        if (t == 0 && (wl / (int)gridDim.x) < 64)                                                    \
            rmx_stamps[((int)blockIdx.x * 64 + wl / (int)gridDim.x) * 4 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define RMX_STAMP(slot) do { } while (0)
#endif

// Schedule of one window (all pairs i<j of B buoys; the anchor spectrum X_i is resident in registers,
// X_j streams one pair ahead):
//   anchor 0      X_0 is transformed straight into the anchor registers (never stored); every further
//                 X_e is transformed once, stored once, and used at once, from registers, for (0,e);
//   anchor i>=1   one anchor load, then the X_j stream, walking j down for odd i and up for even i so
//                 that each anchor starts on the spectra the previous one touched last.
// HBM/L2 traffic per window at B = 8: 8 inputs (256 KiB) + 7 spectrum stores (448 KiB) + 27
// spectrum loads of 64 KiB, of which ~8 are L2-hot, instead of 8 stores + 35 loads.
template <bool U8>
__global__ __launch_bounds__(kThreads, 2) void k_win(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                     const float4* __restrict__ tw1_g,
                                                     const float2* __restrict__ tw2_g, int n_buoys,
                                                     long first_window, float out_scale,
                                                     int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                                     float* __restrict__ peak, int n_win, int dbg_rt, int stag) {
#ifdef RMX_ABLATE
    const int dbg = dbg_rt;   // timing-only ablation build (wrong results): tools/ablate.sh, tools/ablate_run.py
#else
    constexpr int dbg = 0;
    (void)dbg_rt;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* img0 = reinterpret_cast<float2*>(smem);
    float2* img1 = reinterpret_cast<float2*>(smem + kLdsWinImg);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLdsWinTw2);
    float* halo = reinterpret_cast<float*>(smem + kLdsWinHalo);
    float4* red = reinterpret_cast<float4*>(smem + kLdsWinRed);
    int* oidx = reinterpret_cast<int*>(smem + kLdsWinOidx);

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;
    const int B = n_buoys;
    const int n_pairs = B * (B - 1) / 2;

    load_tw2_to_lds_grouped(tw2_lds, tw2_g, t);
    float2 tw1[16];
    load_tw1(tw1, tw1_g, t);
#ifndef RMX_KWIN_LDS1
    // second-generation exchanges (fft_r16.hpp): roles B and C keep their own digit (n0 / k1) in lane bits 0-3
    const float4* tw2row = reinterpret_cast<const float4*>(tw2_lds + (t & 15) * kTw2RowF2);
    const int loc_m0[2] = {__builtin_amdgcn_readfirstlane(wave * kLocWave),                 // this wave's region of image 0 / 1
                           __builtin_amdgcn_readfirstlane(kLdsWinImg + wave * kLocWave)};
    const int loc_rd = wave * kLocWave + loc_read_off(lane);
#else
    const float4* tw2row = reinterpret_cast<const float4*>(tw2_lds + (u & 15) * kTw2RowF2);
#endif
    const float sgn = p ? -1.0f : 1.0f;
    const int kbase = p ? (u - 1) : (u + kM - 1);
    const int hl = lane < 2 ? lane : lane - 60;            // halo row of lanes 0,1,62,63
    const bool is_halo = lane < 2 || lane >= 62;
#ifdef RMX_KWIN_PEAK2
    const int peak_ca = is_halo ? hl * 64 : lane * 32 + 320;   // row byte offset of this lane's |r|^2 row = peak_ca - peak_cb * l*
    const int peak_cb = is_halo ? 0 : 32;
    const int wave_halo = __builtin_amdgcn_readfirstlane(wave * (kHaloRows * 64));
#endif
    __syncthreads();
#ifndef RMX_TW2_LDS
    // this thread's TW2 row W_256^(n0*k1), k1 = 0..15, kept in registers for the whole launch (30 of the 60 VGPRs
    // this kernel left unused at 2 waves per SIMD) instead of eight ds_read_b128 per transform: LDS array time is
    // not hidden behind the butterflies in this kernel, so the 11 % of it that these reads
    // were came off the launch time one for one (1.778 -> 1.728 ms); -DRMX_TW2_LDS restores the LDS reads
    C16 tw2r;
    {
        const float2* rowf2 = reinterpret_cast<const float2*>(tw2row);
        tw2r.set(0, 1.0f, 0.0f);
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float2 w = rowf2[4 * (q & 3) + (q >> 2) - 1];
            tw2r.set(q, w.x, w.y);
        }
    }
#endif

    // persistent workgroup: the tables above are loaded once, then windows blockIdx.x, +gridDim.x, ...
    for (int wl = blockIdx.x; wl < n_win; wl += gridDim.x) {
    C16 sa, sb;   // anchor spectrum X_i and the streamed X_j (scalar arrays: see C16)
    // spectrum scratch is per WORKGROUP, not per window: the persistent workgroup reuses the same
    // B x 64 KiB for every window it processes (256 x 448 KiB = 115 MB live for the whole launch,
    // resident in the 256 MB Infinity Cache, rewritten before most of it is ever evicted to HBM)
    const long wbase = (long)blockIdx.x * B;
    const long obase = (first_window + wl) * (long)n_pairs;
    int seq = 0;         // transform counter: selects the exchange image
    int npair = 0;       // pair counter: selects the record slot (ring of kResSlots)
    int npend = 0;       // pairs whose records await a resolve
#ifdef RMX_KWIN_STAMPS
    int stamp_vm = 0, stamp_drain = 0, stamp_bar = 0, stamp_pc1 = 0, stamp_pc2 = 0;
    int lap_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long lap_last = __builtin_readcyclecounter();
    bool lap_on = true;
    // lap timer of phase 1: the time since the previous lap goes to bucket k (0 forward role A, 1 its barrier, 2 roles B + C,
    // 3 spectrum store, 4 h1, 5 the pair's barrier, 6 h2)
#define RMX_LAP(k) do { if (lap_on) { const long long c_ = __builtin_readcyclecounter(); lap_acc[k] += (int)(c_ - lap_last); lap_last = c_; } } while (0)
#else
#define RMX_LAP(k) do { } while (0)
#endif


    auto barrier_hook = [&](bool flush) __attribute__((always_inline)) {
#ifdef RMX_KWIN_STAMPS
        {   // how long does this wave wait for its own LDS stores to drain, and then at the barrier for the others?
            const long long c0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long c1 = __builtin_readcyclecounter();
            __syncthreads();
            const long long c2 = __builtin_readcyclecounter();
            stamp_drain += (int)(c1 - c0);
            stamp_bar += (int)(c2 - c1);
        }
#else
        if (!(dbg & 1)) __syncthreads();
#endif
        if (npend == kResBatch || (flush && npend > 0)) {
#ifdef RMX_KWIN_PEAK2
            // one of waves 0-3: they reach every barrier ~950 ticks ahead of waves 4-7 (the older wave of a SIMD wins every
            // arbiter), so the resolve is paid out of their waiting time
            if (!(dbg & 2) && !(dbg & 256) && wave == (seq & RMX_KWIN_RES_MASK))
                resolve_batch2(lane, red, halo, oidx, (npair - npend) & (kResSlots - 1), npend, obase, out_scale, lag_int,
                               lag_frac, peak);
#else
            if (!(dbg & 2) && !(dbg & 256) && wave == (seq & RMX_KWIN_RES_MASK))
                resolve_batch(lane, red, halo, oidx, (npair - npend) & (kResSlots - 1), npend, obase, out_scale, lag_int,
                              lag_frac, peak);
#endif
            npend = 0;
        }
    };
    // odd lanes: v[q] *= W32^q, the per-slot part of the odd sub-transform's W_L^n (in place)
    auto mul_w32_odd = [&](float2 (&v)[16]) __attribute__((always_inline)) {
        if (p) {
            {   // q = 1..3 one by one, then three groups of four in one asm statement each
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float2 w = w32(q);
                    float x = v[q].x, y = v[q].y;   // scalars by value: keeps the array out of scratch
                    cmul_inplace(x, y, w.x, w.y);
                    v[q].x = x;
                    v[q].y = y;
                }
            }
#pragma unroll
            for (int q = 4; q < 16; q += 4) {
                float x0 = v[q].x, y0 = v[q].y, x1 = v[q + 1].x, y1 = v[q + 1].y;
                float x2 = v[q + 2].x, y2 = v[q + 2].y, x3 = v[q + 3].x, y3 = v[q + 3].y;
                cmul4_inplace(x0, y0, x1, y1, x2, y2, x3, y3, w32(q), w32(q + 1), w32(q + 2), w32(q + 3));
                v[q].x = x0; v[q].y = y0; v[q + 1].x = x1; v[q + 1].y = y1;
                v[q + 2].x = x2; v[q + 2].y = y2; v[q + 3].x = x3; v[q + 3].y = y3;
            }
        }
    };
    // All global traffic of the loop goes through buffer descriptors held in SGPRs: address = SRD
    // base + one shared 32-bit VGPR offset + an SGPR/immediate offset.  (With flat 64-bit addressing
    // hipcc keeps ~100 VGPRs of loop-invariant addresses alive and spills the twiddles instead.)
    const int samp_bytes = U8 ? 2 : 8;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(iq_v)) + (first_window + wl) * (long)B * kM * samp_bytes, 0,
        B * kM * samp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(spec) + wbase * (long)(8 * kThreads * 16), 0, B * (8 * kThreads * 16), 0x00020000);
    const int xoff = u * samp_bytes, soff = t * 16;
    // raw window samples of buoy b into d (uint8 pairs stay packed in d[q].x until cvt_x)
    auto load_x = [&](C16& d, int b) __attribute__((always_inline)) {
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                d.re[q] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(xs, xoff, (b * kM + q * 256) * 2, 0));
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(xs, xoff, (b * kM + q * 256) * 8, 0);
                d.set(q, __uint_as_float(r.x), __uint_as_float(r.y));   // (by value: see NOTE)
            }
        }
    };
    // quarter G of the same loads (slots 4G..4G+3): issued between the groups of a butterfly layer
    auto load_x_part_from = [&](const __amdgpu_buffer_rsrc_t& rs, C16& d, int b, auto part) __attribute__((always_inline)) {
        constexpr int G = decltype(part)::value;
        if (dbg & 64) return;   // ablation: no window-sample requests
#ifndef RMX_KWIN_NO_SPREAD
        constexpr int Q0 = 2 * G, Q1 = 2 * G + 2;   // eighths: issued from the groups of BOTH butterfly layers of h1
#else
        constexpr int Q0 = 4 * G, Q1 = 4 * G + 4;
#endif
        // (the buoy's byte offset is made opaque HERE so that each request's SGPR offset is computed in front of it (s_mov +
        // s_addk): left to itself hipcc precomputes all of them ahead of the loop, runs out of SGPRs, and every request
        // then pays v_readlane + s_nop 4 to get its offset back out of a spill lane.  One `s_add_i32` per request in a
        // volatile asm is one instruction fewer and measured +1.5 %: volatile statements keep their order among
        // themselves, which pins every request between the exchange stores around it.)
        int bo = b * (kM * samp_bytes);
        asm volatile("" : "+s"(bo));
        if constexpr (U8) {
#pragma unroll
            for (int q = Q0; q < Q1; ++q)
                d.re[q] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, xoff, bo + q * 256 * 2, 0));
        } else {
#pragma unroll
            for (int q = Q0; q < Q1; ++q) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, xoff, bo + q * 256 * 8, 0);
                d.set(q, __uint_as_float(r.x), __uint_as_float(r.y));
            }
        }
    };
    auto load_x_part = [&](C16& d, int b, auto part) __attribute__((always_inline)) { load_x_part_from(xs, d, b, part); };
    auto cvt_x = [&](C16& d) __attribute__((always_inline)) {
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const unsigned r = __float_as_uint(d.re[q]);
                d.set(q, (float)(r & 0xffu) - 127.5f, (float)(r >> 8) - 127.5f);
            }
        }
    };
    auto load_spec_part = [&](C16& d, int b, auto part) __attribute__((always_inline)) {
        constexpr int G = decltype(part)::value;
        if (dbg & 32) b = 1;   // ablation: every request hits the same (cache-resident) spectrum
        if (dbg & 128) return;  // ablation: no spectrum requests
#ifndef RMX_KWIN_NO_SPREAD
        constexpr int J0 = G, J1 = G + 1;
#else
        constexpr int J0 = 2 * G, J1 = 2 * G + 2;
#endif
        int bo = b * (8 * kThreads * 16);   // (opaque: see load_x_part_from)
        asm volatile("" : "+s"(bo));
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(ss, soff, bo + j * (kThreads * 16), RMX_KWIN_LOAD_AUX);
            d.set(2 * j, __uint_as_float(w.x), __uint_as_float(w.y));
            d.set(2 * j + 1, __uint_as_float(w.z), __uint_as_float(w.w));
        }
    };
    auto store_spec = [&](const C16& d, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (opaque copies: hipcc otherwise widens these four scalar reads into overlapping 16-byte
            // loads of the register array, which pins half of it in scratch memory)
            float e0 = d.re[2 * j], e1 = d.im[2 * j], e2 = d.re[2 * j + 1], e3 = d.im[2 * j + 1];
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
            const u32x4 w = {__float_as_uint(e0), __float_as_uint(e1), __float_as_uint(e2), __float_as_uint(e3)};
            // The whole byte offset goes into the VGPR offset, soffset = 0.  Root cause of the corruption seen with
            // an SGPR soffset (round 2, tools/exp_soffset.py + tools/probe/soffset_probe.hip): a store of more
            // than 64 bits reads its data VGPRs late, so a VALU write to them needs a wait state behind the store
            // (ISA "required software-inserted wait states").  hipcc 7.2's hazard recognizer waives that wait
            // state when the store has an SGPR soffset, but on gfx950 the hazard is still there: with soffset in
            // an SGPR the next group's `v_mov_b32 v3, v86` followed the store of v[2:5] directly and half of all
            // pair-windows came out wrong on every call; the same stores with two wait states forced behind each
            // (an asm that keeps e0..e3 live) were right 200 calls out of 200, as is this immediate-soffset form, for
            // which the compiler inserts the s_nop itself.  (The SGPR form alone is fine: the probe, whose stores
            // do not reuse their data registers, has no wrong float.)
            __builtin_amdgcn_raw_buffer_store_b128(w, ss, soff + (b * 8 + j) * (kThreads * 16), 0, RMX_KWIN_STORE_AUX);
        }
    };
    // forward spectrum of the samples in x, in place (carries the 2^-6 of the TW1 table)
    auto fwd = [&](C16& xc) __attribute__((always_inline)) {
        float2* img = (seq & 1) ? img1 : img0;
        float2 x[16];
#ifdef RMX_KWIN_STAMPS
        RMX_LAP(6);                                       // (cvt_x and whatever else sits between h2 and here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // bucket 7: waiting for the window samples (and the spectrum stores)
        RMX_LAP(7);
#endif
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = xc.get(q);
        mul_w32_odd(x);            // odd sub-transform input x*W32^q (W_L^u is folded into tw1)
        dft16(x);
        mul_tw1(x, tw1);
#ifndef RMX_KWIN_LDS1
        if (!(dbg & 8)) xchg_a2_write(img, x, t);
        RMX_LAP(0);
        barrier_hook(false);
        RMX_LAP(1);
        if (!(dbg & 8)) xchg_b2_read(img, x, t);
#else
        if (!(dbg & 8)) xchg_a_write(img, x, t);
        barrier_hook(false);
        if (!(dbg & 8)) xchg_b_read(img, x, t);
#endif
        dft16(x);
#ifdef RMX_TW2_LDS
        const float4 r0 = tw2row[0], r1 = tw2row[1];   // ahead of the exchange reads (see dft16_tw_row_l1)
#endif
        if (!(dbg & 4)) {
#ifndef RMX_KWIN_LDS1
        loc_write16(loc_m0[seq & 1], x);
        wave_lds_order();
        loc_read16(smem + (seq & 1) * kLdsWinImg + loc_rd, x);
#else
        xchg_bc_write_b(img, x, t);
        wave_lds_order();
        xchg_bc_read_c(img, x, t);
#endif
        }
#ifndef RMX_TW2_LDS
        dft16_tw<true>(x, tw2r);
#else
        dft16_tw_row(x, tw2row, r0, r1);   // W_256^(n0*k1) as pre-twiddle of the last pass
#endif
#pragma unroll
        for (int q = 0; q < 16; ++q) xc.set(q, x[q].x, x[q].y);   // (scaled by 2^-6 through the TW1 table)
        ++seq;
        RMX_LAP(2);
    };
    // One pair = two halves around its only workgroup barrier.
    //   h1  conj-multiply merged into the role-C pass, wave-local exchange, role-B pass, stores into
    //       exchange image `tr & 1` (this wave's own regions); `prefetch(part)` is called eight times,
    //       from the groups of both butterfly layers (parts 0-3 behind the role-C pass, 4-7 behind the role-B pass: one
    //       16-byte request per part instead of bursts of two, tools/probe/kwin_bench.hip: -0.6 %), always
    //       after the last read of a and s (their registers may be reloaded there)
    //   h2  reads image `tr & 1` (all waves' regions), role-A pass, last radix-2, |.|^2, peak records
    // h2 of pair n and h1 of pair n+1 sit between the same two barriers and do not depend on each
    // other (different images, disjoint registers), so the two waves that share a SIMD can run them in
    // opposite order: see the phase-2 loop.
    auto pair_h1 = [&](const C16& a, const C16& s, int tr, auto prefetch) __attribute__((always_inline)) {
        float2* img = (tr & 1) ? img1 : img0;
        float2 v[16];
#ifdef RMX_KWIN_STAMPS
        {   // how long does this wave wait for the spectra it requested a pair ahead?
            const long long c0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long c1 = __builtin_readcyclecounter();
            stamp_vm += (int)(c1 - c0);
        }
#endif
        // R = X_j conj(X_i), (im,re)-swapped == swap(X_j) * X_i: merged into the first radix-16 pass
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_float2(s.im[q], s.re[q]);
        dft16_tw_l1<false>(v, a);                // k2 -> n0   (role C), layer 1: the last reads of a and s
#pragma unroll
        for (int q = 0; q < 16; q += 4)          // pin: the requests below must follow the reads above
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q + 1].x), "+v"(v[q + 1].y), "+v"(v[q + 2].x),
                         "+v"(v[q + 2].y), "+v"(v[q + 3].x), "+v"(v[q + 3].y));
        __builtin_amdgcn_sched_barrier(0);
        {   // layer 2 group by group: each group's outputs go to the wave-local image at once, and a
            // quarter of the next spectra is requested behind it
            float2* wb = img + (u >> 4) * kBcHalf + (u & 15) * kBcRow + p;
            (void)wb;
            dft16_layer2_emit(v, [&](auto kac, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                     __attribute__((always_inline)) {
                constexpr int ka = decltype(kac)::value;
                if (!(dbg & 4)) {
#ifndef RMX_KWIN_LDS1
                    loc_write4<ka, ka + 4, ka + 8, ka + 12>(loc_m0[tr & 1], x0, x1, x2, x3);
#else
                    wb[2 * ka] = make_float2(x0.x, x0.y);
                    wb[2 * (ka + 4)] = make_float2(x1.x, x1.y);
                    wb[2 * (ka + 8)] = make_float2(x2.x, x2.y);
                    wb[2 * (ka + 12)] = make_float2(x3.x, x3.y);
#endif
                }
                if (!(dbg & 16)) prefetch(kac);
            });
        }
#ifdef RMX_TW2_LDS
        const float4 r0 = tw2row[0], r1 = tw2row[1];   // ahead of the exchange reads (see dft16_tw_row_l1)
#endif
        if (!(dbg & 4)) {
            wave_lds_order();
#ifndef RMX_KWIN_LDS1
            loc_read16(smem + (tr & 1) * kLdsWinImg + loc_rd, v);
#else
            xchg_bc_read_b(img, v, t);
#endif
        }
#ifndef RMX_TW2_LDS
        dft16_tw_l1<true>(v, tw2r);
#else
        dft16_tw_row_l1(v, tw2row, r0, r1);      // W_256^(n0*k1), k1 -> n1   (role B), layer 1
#endif
        {
#ifndef RMX_KWIN_LDS1
            float2* xb = img + xb2_base(t);                             // own k0 row of the [k0][n1][p][n0] image
#else
            float2* xb = img + (u >> 4) * kBcHalf + (u & 15) * 2 + p;   // own half-wave regions
#endif
            dft16_layer2_emit(v, [&](auto kac, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                     __attribute__((always_inline)) {
                constexpr int ka = decltype(kac)::value;
                if (!(dbg & 8)) {
                    xb[ka * 32] = make_float2(x0.x, x0.y);
                    xb[(ka + 4) * 32] = make_float2(x1.x, x1.y);
                    xb[(ka + 8) * 32] = make_float2(x2.x, x2.y);
                    xb[(ka + 12) * 32] = make_float2(x3.x, x3.y);
                }
#ifndef RMX_KWIN_NO_SPREAD
                if (!(dbg & 16)) prefetch(std::integral_constant<int, ka + 4>{});
#endif
            });
        }
    };
    auto pair_h2 = [&](int tr, int out_idx) __attribute__((always_inline)) {
        const float2* img = (tr & 1) ? img1 : img0;
        const int rb = npair & (kResSlots - 1);
        float2 v[16];
#ifndef RMX_KWIN_LDS1
        if (!(dbg & 8)) xchg_a2_read(img, v, t);
#else
        if (!(dbg & 8)) xchg_a_read(img, v, t);
#endif
        dft16_tw<false>(v, tw1);                 // W_M^(u*k0) [* W_L^u odd], k0 -> n2   (role A)
        mul_w32_odd(v);                          // odd lanes: * W32^q
        // last radix-2 stage across the lane pair, up to a sign that |.| does not see:
        // even lane e + o' = r[n], odd lane o' - e = -r[n+M]
        pair_fmac8(v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y, sgn);
        pair_fmac8(v[4].x, v[4].y, v[5].x, v[5].y, v[6].x, v[6].y, v[7].x, v[7].y, sgn);
        pair_fmac8(v[8].x, v[8].y, v[9].x, v[9].y, v[10].x, v[10].y, v[11].x, v[11].y, sgn);
        pair_fmac8(v[12].x, v[12].y, v[13].x, v[13].y, v[14].x, v[14].y, v[15].x, v[15].y, sgn);
        float mag[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) mag[q] = fmaf(v[q].x, v[q].x, v[q].y * v[q].y);
        if (p && u == 0) mag[0] = -1.0f;         // lag -M is not part of the 'full' output
        if (dbg & 2) {
            float s = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += mag[q];
            if (s == 12345.678f) lag_int[0] = 1;
            ++npend; ++npair;
            return;
        }
#ifdef RMX_KWIN_PEAK2
        {
            // lane maximum (8 x v_max3_f32), row maxima by four DPP steps, wave maximum on the scalar unit: |r|^2 >= +0 and
            // the one sentinel is -1, so the float order is the signed-integer order of the bit patterns (s_max_i32)
            float tmax, wrow;
            asm volatile("v_max3_f32 %[t], %[m0], %[m1], %[m2]\n\tv_max3_f32 %[w], %[m3], %[m4], %[m5]\n\t"
                         "v_max3_f32 %[t], %[t], %[m6], %[m7]\n\tv_max3_f32 %[w], %[w], %[m8], %[m9]\n\t"
                         "v_max3_f32 %[t], %[t], %[ma], %[mb]\n\tv_max3_f32 %[w], %[w], %[mc], %[md]\n\t"
                         "v_max3_f32 %[t], %[t], %[me], %[mf]\n\tv_max_f32 %[t], %[t], %[w]\n\t"
                         "s_nop 1\n\t"
                         "v_max_f32_dpp %[w], %[t], %[t] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_max_f32_dpp %[w], %[w], %[w] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_max_f32_dpp %[w], %[w], %[w] row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_max_f32_dpp %[w], %[w], %[w] row_mirror row_mask:0xf bank_mask:0xf"
                         : [t] "=&v"(tmax), [w] "=&v"(wrow)
                         : [m0] "v"(mag[0]), [m1] "v"(mag[1]), [m2] "v"(mag[2]), [m3] "v"(mag[3]), [m4] "v"(mag[4]),
                           [m5] "v"(mag[5]), [m6] "v"(mag[6]), [m7] "v"(mag[7]), [m8] "v"(mag[8]), [m9] "v"(mag[9]),
                           [ma] "v"(mag[10]), [mb] "v"(mag[11]), [mc] "v"(mag[12]), [md] "v"(mag[13]), [me] "v"(mag[14]),
                           [mf] "v"(mag[15]));
            // wave maximum on the scalar unit (four v_readlane, three s_max_i32), winner lane by ballot + s_ff1: everything
            // behind the DPP steps is one dependent chain, and a dependent instruction costs a wave ~10 cycles where an
            // independent one costs 4 -- the chain is kept short (first PEAK2 build: ~40 links, no faster than the 75
            // mostly independent instructions it replaced)
            const int wi = __builtin_bit_cast(int, wrow);   // every lane: the max of its row of 16
            const int r0 = __builtin_amdgcn_readlane(wi, 0), r1 = __builtin_amdgcn_readlane(wi, 16);
            const int r2 = __builtin_amdgcn_readlane(wi, 32), r3 = __builtin_amdgcn_readlane(wi, 48);
            int wmaxi, wtmp;
            asm("s_max_i32 %0, %2, %3\n\ts_max_i32 %1, %4, %5\n\ts_max_i32 %0, %0, %1"
                : "=&s"(wmaxi), "=&s"(wtmp) : "s"(r0), "s"(r1), "s"(r2), "s"(r3));
            const unsigned long long hit = __ballot(__builtin_bit_cast(int, tmax) == wmaxi);
            int ls;
            if (__builtin_expect(__popcll(hit) == 1, 1)) {
                ls = __ffsll((long long)hit) - 1;
            } else {
                // several lanes hold the maximum exactly (an exact tie, or an all-zero window): the lowest 'full' index decides,
                // found the old way -- lowest slot per lane, wave minimum of the indices
                const float wmaxf = __builtin_bit_cast(float, wmaxi);
                int qa = 16, qb = 16, qc = 16, qd = 16;
                argsel4<12>(qa, qb, qc, qd, mag[12], mag[13], mag[14], mag[15], wmaxf);
                argsel4<8>(qa, qb, qc, qd, mag[8], mag[9], mag[10], mag[11], wmaxf);
                argsel4<4>(qa, qb, qc, qd, mag[4], mag[5], mag[6], mag[7], wmaxf);
                argsel4<0>(qa, qb, qc, qd, mag[0], mag[1], mag[2], mag[3], wmaxf);
                const int qsel = min(min(qa, qb), min(qc, qd));
                const int kw = wave_min_i32(qsel < 16 ? kbase + qsel * 256 : 0x7fffffff);
                int ts, qs;
                k_to_owner(kw, ts, qs);
                ls = __builtin_amdgcn_readfirstlane(ts & 63);
            }
            // rows: the halo lanes always, l*-2 / l* / l*+2 where they are not halo lanes (whose row the resolver reads
            // instead).  Writers' mask on the scalar unit: bits l*-2, l*, l*+2 (what falls off either end of the shifts is a
            // halo lane or no lane) | lanes 0, 1, 62, 63.  Row byte offset: halo lanes hl * 64; the others
            // (4 + (lane - l* + 2) / 2) * 64 = lane * 32 + 320 - l* * 32, i.e. peak_ca - peak_cb * l* with two per-lane constants.
            const unsigned long long wmask = ((0x15ull << ls) >> 2) | 0xC000000000000003ull;
            const int sbase = kLdsWinHalo + rb * (8 * kHaloRows * 64) + wave_halo;   // (scalar: wave_halo is held in an SGPR)
#ifndef RMX_ABLATE
            {
                int addr;
                unsigned long long sv;
                const f32x4 d0 = {mag[0], mag[1], mag[2], mag[3]}, d1 = {mag[4], mag[5], mag[6], mag[7]};
                const f32x4 d2 = {mag[8], mag[9], mag[10], mag[11]}, d3 = {mag[12], mag[13], mag[14], mag[15]};
                asm volatile("v_mad_i32_i24 %[a], %[cb], %[nls], %[ca]\n\t"
                             "v_add_u32 %[a], %[sb], %[a]\n\t"
                             "s_mov_b64 %[sv], exec\n\t"
                             "s_mov_b64 exec, %[m]\n\t"
                             "ds_write_b128 %[a], %[d0]\n\t"
                             "ds_write_b128 %[a], %[d1] offset:16\n\t"
                             "ds_write_b128 %[a], %[d2] offset:32\n\t"
                             "ds_write_b128 %[a], %[d3] offset:48\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [a] "=&v"(addr), [sv] "=&s"(sv)
                             : [cb] "v"(peak_cb), [nls] "s"(-ls), [ca] "v"(peak_ca), [sb] "s"(sbase), [m] "s"(wmask), [d0] "v"(d0),
                               [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3)
                             : "memory");
            }
#else
            if (((wmask >> lane) & 1) && !(dbg & 512)) {
                float4* hp = reinterpret_cast<float4*>(smem + sbase + peak_ca - peak_cb * ls);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    hp[q4] = make_float4(mag[4 * q4], mag[4 * q4 + 1], mag[4 * q4 + 2], mag[4 * q4 + 3]);
            }
#endif
            if (lane == 0) {
                const u32x2 rec = {(unsigned)wmaxi, (unsigned)ls};
                *reinterpret_cast<u32x2*>(red + rb * 8 + wave) = rec;
                if (wave == 0) oidx[rb] = out_idx;
            }
        }
#else
        if (is_halo && !(dbg & 512)) {
            float4* hp = reinterpret_cast<float4*>(halo + ((rb * 8 + wave) * kHaloRows + hl) * 16);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                hp[q4] = make_float4(mag[4 * q4], mag[4 * q4 + 1], mag[4 * q4 + 2], mag[4 * q4 + 3]);
        }
        float tmax = mag[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) tmax = fmaxf(tmax, mag[q]);
#ifndef RMX_KWIN_OLD_PEAK
        // Lowest slot holding the lane's max (four select chains, descending so that lower slots win), with the four
        // row steps of the wave maximum issued BETWEEN the chains' groups: the DPP steps depend on each other, the
        // groups do not depend on them, so neither the 2 wait states in front of a DPP read nor the steps' latency
        // are ever waited for.  One asm statement (hipcc separates consecutive statements by s_nop).
        int qa = 16, qb = 16, qc = 16, qd = 16;
        float wrow;
        {
            unsigned long long k0, k1, k2, k3;
#define RMX_AS4(M0, M1, M2, M3, Q)                                                        \
    "v_cmp_eq_f32_e64 %[k0], %[" #M0 "], %[t]\n\tv_cmp_eq_f32_e64 %[k1], %[" #M1 "], %[t]\n\t" \
    "v_cmp_eq_f32_e64 %[k2], %[" #M2 "], %[t]\n\tv_cmp_eq_f32_e64 %[k3], %[" #M3 "], %[t]\n\t" \
    "v_cndmask_b32_e64 %[qa], %[qa], " #Q ", %[k0]\n\tv_cndmask_b32_e64 %[qb], %[qb], " #Q "+1, %[k1]\n\t" \
    "v_cndmask_b32_e64 %[qc], %[qc], " #Q "+2, %[k2]\n\tv_cndmask_b32_e64 %[qd], %[qd], " #Q "+3, %[k3]\n\t"
            asm volatile(RMX_AS4(mc, md, me, mf, 12)
                         "v_max_f32_dpp %[w], %[t], %[t] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         RMX_AS4(m8, m9, ma, mb, 8)
                         "v_max_f32_dpp %[w], %[w], %[w] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                         RMX_AS4(m4, m5, m6, m7, 4)
                         "v_max_f32_dpp %[w], %[w], %[w] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                         RMX_AS4(m0, m1, m2, m3, 0)
                         "v_max_f32_dpp %[w], %[w], %[w] row_mirror row_mask:0xf bank_mask:0xf"
                         : [qa] "+v"(qa), [qb] "+v"(qb), [qc] "+v"(qc), [qd] "+v"(qd), [w] "=&v"(wrow), [k0] "=&s"(k0),
                           [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
                         : [t] "v"(tmax), [m0] "v"(mag[0]), [m1] "v"(mag[1]), [m2] "v"(mag[2]), [m3] "v"(mag[3]),
                           [m4] "v"(mag[4]), [m5] "v"(mag[5]), [m6] "v"(mag[6]), [m7] "v"(mag[7]), [m8] "v"(mag[8]),
                           [m9] "v"(mag[9]), [ma] "v"(mag[10]), [mb] "v"(mag[11]), [mc] "v"(mag[12]), [md] "v"(mag[13]),
                           [me] "v"(mag[14]), [mf] "v"(mag[15]));
#undef RMX_AS4
        }
        const int qsel = min(min(qa, qb), min(qc, qd));
        const int kq = kbase + qsel * 256;
        const int wi = __builtin_bit_cast(int, wrow);   // every lane: the max of its row of 16
        const float wmax = fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(wi, 0)),
                                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(wi, 16))),
                                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(wi, 32)),
                                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(wi, 48))));
        // which lane holds it?  One lane almost always: its index and slot come over by readlane.  Several lanes (an
        // exact tie between lanes): the lowest 'full' index decides, found by the wave minimum as before.
        const unsigned long long hit = __ballot(tmax == wmax);
        int kw, ls, qs;
        if (__popcll(hit) == 1) {
            ls = __ffsll((long long)hit) - 1;
            kw = __builtin_amdgcn_readlane(kq, ls);
            qs = __builtin_amdgcn_readlane(qsel, ls);
        } else {
            kw = wave_min_i32(tmax == wmax ? kq : 0x7fffffff);
            int ts;
            k_to_owner(kw, ts, qs);
            ls = ts & 63;
        }
#else
        // lowest slot holding the max: four independent select chains
        int qa = 16, qb = 16, qc = 16, qd = 16;
        argsel4<12>(qa, qb, qc, qd, mag[12], mag[13], mag[14], mag[15], tmax);   // descending: lower slots win
        argsel4<8>(qa, qb, qc, qd, mag[8], mag[9], mag[10], mag[11], tmax);
        argsel4<4>(qa, qb, qc, qd, mag[4], mag[5], mag[6], mag[7], tmax);
        argsel4<0>(qa, qb, qc, qd, mag[0], mag[1], mag[2], mag[3], tmax);
        const int qsel = min(min(qa, qb), min(qc, qd));
        const int kq = kbase + qsel * 256;
        const float wmax = wave_max_f32(tmax);
        const int kw = wave_min_i32(tmax == wmax ? kq : 0x7fffffff);
        // the winner's neighbours k*-1, k*+1 live in lanes l*-2, l*+2 (same slot) when those exist
        int ts, qs;
        k_to_owner(kw, ts, qs);
        const int ls = ts & 63;
#endif
        // qs is wave-uniform (it comes out of the wave reductions): one indexed register read
        // (s_set_gpr_idx) instead of a 16-way select chain
        typedef float f16v __attribute__((ext_vector_type(16)));
        const f16v mv = {mag[0], mag[1], mag[2],  mag[3],  mag[4],  mag[5],  mag[6],  mag[7],
                         mag[8], mag[9], mag[10], mag[11], mag[12], mag[13], mag[14], mag[15]};
        const float sel = mv[__builtin_amdgcn_readfirstlane(qs)];
        const int seli = __builtin_bit_cast(int, sel);
        const float tapm = ls >= 2 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(seli, ls >= 2 ? ls - 2 : 0)) : -2.0f;
        const float tapp = ls <= 61 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(seli, ls <= 61 ? ls + 2 : 63)) : -2.0f;
        if (lane == 0) {
            const u32x4 rec = {__float_as_uint(wmax), (unsigned)kw, __float_as_uint(tapm), __float_as_uint(tapp)};
            *reinterpret_cast<u32x4*>(red + rb * 8 + wave) = rec;
            if (wave == 0) oidx[rb] = out_idx;
        }
#endif   // RMX_KWIN_PEAK2
        ++npend;
        ++npair;
    };
    auto pair = [&](const C16& a, const C16& s, int out_idx, auto prefetch) __attribute__((always_inline)) {
        pair_h1(a, s, seq, prefetch);
        RMX_LAP(4);
        barrier_hook(false);                     // the pair's only barrier
        RMX_LAP(5);
        pair_h2(seq, out_idx);
        RMX_LAP(6);
        ++seq;
    };
    auto out_of = [&](int i, int j) -> int { return i * B - (i * (i + 1)) / 2 + (j - i - 1); };

    // ---- anchor 0: X_0 goes straight into the anchor registers (never stored); every other X_e is
    // transformed once, stored once for the later anchors, and used at once from registers for (0,e)
    RMX_STAMP(0);
    load_x(sa, 0);
    if (B > 1) load_x(sb, 1);          // sb is free: X_1's samples travel while X_0 is transformed
    cvt_x(sa);
    fwd(sa);
    // (the last buoy is peeled off the loop: its pair requests spectra instead of samples; with both
    // request kinds in one loop body the compiler's waitcnt bookkeeping merges their destination
    // registers across the back edge)
    for (int e = 1; e + 1 < B; ++e) {
        cvt_x(sb);
        fwd(sb);
        store_spec(sb, e);
        RMX_LAP(3);
        pair(sa, sb, out_of(0, e), [&](auto part) __attribute__((always_inline)) { load_x_part(sb, e + 1, part); });
    }
    if (B > 1) {
        cvt_x(sb);
        fwd(sb);
        store_spec(sb, B - 1);
        RMX_LAP(3);
        pair(sa, sb, out_of(0, B - 1), [&](auto part) __attribute__((always_inline)) {
            if (B > 2) {               // next anchor 1 streams downwards from B-1: X_{B-1} is L2-hot
                load_spec_part(sa, 1, part);
                load_spec_part(sb, B - 1, part);
            }
        });
    }
    RMX_STAMP(1);
#ifdef RMX_KWIN_STAMPS
    lap_on = false;
#endif
    // ---- anchors 1..B-2: the stream direction alternates (odd anchors walk j down, even ones up), so
    // the first spectra an anchor streams are the ones the previous anchor touched last (L2 hits).
    // Between two barriers sit h2 of pair m and h1 of pair m+1, which are independent: waves 0-3 run
    // them in that order and waves 4-7 (the second wave of each SIMD) in the opposite order, so that
    // one wave's LDS / barrier / DPP-chain stalls fall on the other's butterfly arithmetic instead of
    // on the same stalls (all eight waves are otherwise barrier-aligned in lockstep).
    {
        const int M2 = (B - 1) * (B - 2) / 2;            // pairs of this phase
        // SIMD pairs {a, a+2} vs {a+1, a+3} (stag 1): measured best of the splits; 0 = nobody, 5 = everybody late
        // (2: odd waves = SIMDs 1, 3; 3: the second wave of every SIMD; 4: one wave of every SIMD, alternating between SIMDs)
        const bool late_h2 = stag == 1 ? ((wave >> 1) & 1) : stag == 2 ? (wave & 1) : stag == 3 ? (wave >> 2) :
                             stag == 4 ? ((wave ^ (wave >> 2)) & 1) : (stag == 5);
        auto j_of = [&](int i, int s) -> int { return (i & 1) ? (B - 1 - s) : (i + 1 + s); };
        int ci = 1, cs = 0;                              // pair m     (anchor, position in its run)
        int ni = 1, ns = 1;                              // pair m + 1
        if (ns >= B - 1 - ni) { ++ni; ns = 0; }
        auto h1_of = [&](int hi, int hs, int tr) __attribute__((always_inline)) {
            // spectra for the pair after (hi, hs) are requested here
            int pi = hi, ps = hs + 1;
            if (ps >= B - 1 - pi) { ++pi; ps = 0; }
            const bool valid = pi + 1 < B;
#ifdef RMX_KWIN_FORCE_ANCHOR   // timing experiment (results unchanged): the anchor spectrum is requested again for EVERY pair -- the
            const bool new_anchor = valid;   // scratch traffic of a design that cannot keep its anchor resident (LABNOTES R4.6)
#else
            const bool new_anchor = valid && pi != hi;
#endif
#ifndef RMX_KWIN_REQ_BRANCHY
            // Every instruction costs the issuing wave ~2 ns whatever its kind (tools/probe/valu_forms.hip), and the two
            // uniform branches around each of the eight requests were 45 scalar instructions per pair plus the vector
            // instructions hipcc used to carry their conditions: the streamed spectrum is now requested unconditionally
            // (behind the window's last pair an index that exists: the registers are dead there), and a new anchor's
            // eight requests go out together behind ONE branch, in the first callback (6 of a window's 21 pairs).
            const int pj = valid ? j_of(pi, ps) : B - 1;
            pair_h1(sa, sb, tr, [&](auto part) __attribute__((always_inline)) {
                if constexpr (decltype(part)::value == 0) {
                    if (new_anchor) {
                        load_spec_part(sa, pi, std::integral_constant<int, 0>{});
                        load_spec_part(sa, pi, std::integral_constant<int, 1>{});
                        load_spec_part(sa, pi, std::integral_constant<int, 2>{});
                        load_spec_part(sa, pi, std::integral_constant<int, 3>{});
                        load_spec_part(sa, pi, std::integral_constant<int, 4>{});
                        load_spec_part(sa, pi, std::integral_constant<int, 5>{});
                        load_spec_part(sa, pi, std::integral_constant<int, 6>{});
                        load_spec_part(sa, pi, std::integral_constant<int, 7>{});
                    }
                }
                load_spec_part(sb, pj, part);
            });
#else
            const int pj = j_of(pi, ps);
            pair_h1(sa, sb, tr, [&](auto part) __attribute__((always_inline)) {
                if (valid) {
                    if (pi != hi) load_spec_part(sa, pi, part);
                    load_spec_part(sb, pj, part);
                }
            });
#endif
        };
        if (M2 > 0) h1_of(ci, cs, seq);
        for (int m = 0; m < M2; ++m) {
            barrier_hook(false);
            const bool has_next = m + 1 < M2;
            const int out_idx = out_of(ci, j_of(ci, cs));
#ifndef RMX_KWIN_PRIO_MODE
#define RMX_KWIN_PRIO_MODE 1
#endif
#if !defined(RMX_KWIN_PRIO) && !defined(RMX_KWIN_NO_PRIO)
#define RMX_KWIN_PRIO 1      /* default: on (-0.8 ... -1.0 % in three A/B runs, tools/probe/kwin_bench.hip) */
#endif
#ifdef RMX_KWIN_PRIO
            // The two waves of a SIMD share its issue slots (and the CU's LDS / vector-memory request paths), and every
            // arbiter prefers the OLDER one: in-kernel stamps (tools/probe/kwin_bench.hip -DRMX_KWIN_STAMPS) show waves 0-3
            // spending 51-60 k ticks per window in the two pieces between barriers where waves 4-7 need 57-64 k, and then
            // waiting ~950 ticks at every barrier for them (waves 4-7: ~270).  Priority outranks age, so waves 4-7 run the
            // FIRST of their two pieces at priority 1 and the second at 0.  It only helps the piece that is VALU-bound (h2
            // as first piece: 63.9 -> 58.1 k ticks; h1 does not react to priority), so the launch gains 0.8-1.0 %, not the 8 %
            // an even split would give; priority during h2 only (mode 2) and priority 3 measured the same or less.
#if RMX_KWIN_PRIO_MODE == 2   /* priority during h2 only (where it was seen to help), whichever piece that is */
            if (wave >= 4) { if (late_h2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(RMX_KWIN_PRIO); }
#define RMX_PRIO_MID() do { if (wave >= 4) { if (late_h2) __builtin_amdgcn_s_setprio(RMX_KWIN_PRIO); else __builtin_amdgcn_s_setprio(0); } } while (0)
#else
            if (wave >= 4) __builtin_amdgcn_s_setprio(RMX_KWIN_PRIO);
#define RMX_PRIO_MID() do { if (wave >= 4) __builtin_amdgcn_s_setprio(0); } while (0)
#endif
#else
#define RMX_PRIO_MID() do { } while (0)
#endif
#ifdef RMX_KWIN_STAMPS
            const long long p0 = __builtin_readcyclecounter();
            long long p1;
#define RMX_PIECE_MID() p1 = __builtin_readcyclecounter()
#else
#define RMX_PIECE_MID() do { } while (0)
#endif
            if (late_h2) {
                if (has_next) h1_of(ni, ns, seq + 1);
                RMX_PRIO_MID();
                RMX_PIECE_MID();
                pair_h2(seq, out_idx);
            } else {
                pair_h2(seq, out_idx);
                RMX_PRIO_MID();
                RMX_PIECE_MID();
                if (has_next) h1_of(ni, ns, seq + 1);
            }
#ifdef RMX_KWIN_STAMPS
            {
                const long long p2 = __builtin_readcyclecounter();
                stamp_pc1 += (int)(p1 - p0);
                stamp_pc2 += (int)(p2 - p1);
            }
#endif
#undef RMX_PIECE_MID
#undef RMX_PRIO_MID
            ++seq;
            ci = ni; cs = ns;
            ++ns;
            if (ns >= B - 1 - ni) { ++ni; ns = 0; }
        }
    }
    RMX_STAMP(2);
#ifdef RMX_KWIN_STAMPS
    if (lane == 0 && (wl / (int)gridDim.x) < 64) {
        rmx_stamps_vm[((int)blockIdx.x * 64 + wl / (int)gridDim.x) * 8 + wave] = stamp_vm;
        rmx_stamps_bar[(((int)blockIdx.x * 64 + wl / (int)gridDim.x) * 8 + wave) * 2] = stamp_drain;
        rmx_stamps_bar[(((int)blockIdx.x * 64 + wl / (int)gridDim.x) * 8 + wave) * 2 + 1] = stamp_bar;
        for (int k = 0; k < 8; ++k) rmx_stamps_p1[(((int)blockIdx.x * 64 + wl / (int)gridDim.x) * 8 + wave) * 8 + k] = lap_acc[k];
        rmx_stamps_pc[(((int)blockIdx.x * 64 + wl / (int)gridDim.x) * 8 + wave) * 2] = stamp_pc1;
        rmx_stamps_pc[(((int)blockIdx.x * 64 + wl / (int)gridDim.x) * 8 + wave) * 2 + 1] = stamp_pc2;
    }
#endif
    seq = 0;   // any wave may resolve the last pairs; take wave 0
    barrier_hook(true);
    }   // next window of this workgroup
}

// ---- host: twiddle tables of k_fwd / k_win / the pair kernels ---------------------------------------
constexpr int kTw1ScaleLog2 = 6;                       // spectra carry 2^-6 (L = 2^13: sqrt(L) rounded down)
constexpr double kTw1Scale = 1.0 / (1 << kTw1ScaleLog2);
inline void build_tables(std::vector<float4>& tw1, std::vector<float2>& tw2) {
    const double two_pi = 6.283185307179586476925286766559;
    tw1.resize(8 * kThreads);
    std::vector<float2> t1(16 * kThreads);
    for (int t = 0; t < kThreads; ++t) {
        const int p = t & 1, u = t >> 1;
        for (int k0 = 0; k0 < 16; ++k0) {
            // W_M^(u*k0) * (p ? W_L^u : 1), W_n = exp(-2*pi*i/n)
            double ang = -two_pi * (double)((u * k0) % kM) / (double)kM;
            if (p) ang += -two_pi * (double)u / (double)kL;
            // the power-of-two scale of the forward spectra rides on this table (exact): the forward
            // transform multiplies by it once, the inverse once more (undone in out_scale)
            t1[k0 * kThreads + t] = make_float2((float)(std::cos(ang) * kTw1Scale), (float)(std::sin(ang) * kTw1Scale));
        }
    }
    for (int j = 0; j < 8; ++j)
        for (int t = 0; t < kThreads; ++t) {
            const float2 a = t1[(2 * j) * kThreads + t], b = t1[(2 * j + 1) * kThreads + t];
            tw1[j * kThreads + t] = make_float4(a.x, a.y, b.x, b.y);
        }
    tw2.resize(256);
    for (int a = 0; a < 16; ++a)
        for (int b = 0; b < 16; ++b) {
            const double ang = -two_pi * (double)((a * b) % 256) / 256.0;
            tw2[a * 16 + b] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
}

}  // namespace rmx
