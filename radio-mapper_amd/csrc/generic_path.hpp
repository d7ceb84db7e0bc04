// generic_path.hpp -- the same path for every power-of-two window length N (16 .. 4 Mi samples).
//
// N = 4096 (BASELINE cfg3/cfg4) has the register/LDS-resident radix-16 kernels of rmx_hip.hip.  All
// other lengths run here: simpler kernels, same definition, same output contract.
//   512 <= L = 2N <= 16384 (N = 8192: the reference's iq_stream_client captures), batches that fill the chip:
//                    a workgroup owns a WINDOW and runs all its B + P transforms, spectra never travel as spectra:
//                      g_win_fused   <= 4 buoys, L <= 4096: the spectra stay in registers
//                      g_win_scr     any buoy count: spectra in a per-workgroup, cache-resident scratch, persistent grid
//                      g_win_scr14   L = 16384: the same with 512 threads x two butterflies per pass
//   L = 2N <= 8192   small batches, the Doppler search, L < 512: one workgroup per transform, the whole zero-padded
//                    window in LDS:
//                      g_fwd_small   (window, buoy)  : radix-2 DIF, spectrum left in bit-reversed order
//                      g_pair_small  (window, pair)  : X_j conj(X_i) -> radix-2 DIT (takes bit-reversed
//                                                      input, natural output) -> |.|, argmax, parabola
//   L = 2N >= 16384  four-step transform through HBM in two passes per transform, no transposes: the
//                    sequence is the row-major matrix [L1][L2] (L1 <= 1024 columns-length, L2 = L/L1
//                    <= 8192), n = n1*L2 + n2:
//                      forward : g_cols_fwd  tiles of 16 columns x all L1 rows in LDS (up to 128 KiB of
//                                            the CU's 160 KiB): zero-padded load, DIF over n1,
//                                            * W_L^(n2 k1), store [k1'][n2]
//                                g_rows      DIF over n2 in place -> spectrum [k1'][k2'] (digit-reversed;
//                                            the product is pointwise and the inverse undoes the order)
//                      inverse : g_rows      X_j conj(X_i) formed on load, DIT over k2', * conj W_L^(n2 k1)
//                                g_cols_inv  DIT over k1' -> r tile by tile in LDS only: the tile's partial
//                                            argmax in 'full' order, its peak's taps, its two edge columns
//                                            (halo); g_final: reduce, taps, parabola.  r never reaches HBM.
//                    HBM bytes per window: B*(8N + 3*16N) + P*(2*16N + 2*16N) = (56 B + 64 P) N (+ 2-4 N P of
//                    halo), against the 16 N P of the algorithmic model (cfg2: 360 N vs 48 N): the two-pass
//                    minimum SURVEY.md section 8d quotes.
// Twiddles: W_R^k tables per row length computed in double on the host; the large W_L^(a*b) factor of
// the four-step is the product of two table entries (a*b mod L split into high and low digits).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>

namespace rmx {
namespace gen {

#ifndef RMX_PROD_BATCH
#define RMX_PROD_BATCH 4
#endif
constexpr int kGThreads = 256;
// threads per row of the row kernels: two threads per radix-16 group (the passes leave half of them idle, the
// streaming loops use all: measured better than one thread per group on the 2048-point rows of cfg2)
__host__ __device__ constexpr int rows_tpr(int R) { return (R >> 3) < kGThreads ? ((R >> 3) > 0 ? (R >> 3) : 1) : kGThreads; }

__device__ __forceinline__ float2 g_cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 g_cmulc(float2 a, float2 b) {   // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

// Exchange between two passes.  LOCAL: every butterfly of the reading pass takes its inputs from lanes of its own
// wave (blocks of at most 64 x 16 elements: the threads of a block are consecutive) -- a wave's LDS instructions
// execute in issue order, so the compiler-only fence is enough and no wave waits for another; else a barrier.
template <bool LOCAL>
__device__ __forceinline__ void xsync() {
#ifdef RMX_EXP_NOBAR      // timing experiment only (wrong results): what would the remaining barriers of the register-block kernels cost?
    wave_lds_order();
#else
    if constexpr (LOCAL) wave_lds_order();
    else __syncthreads();
#endif
}
// In-place transforms of R = 2^logR points held in LDS; tw[k] = W_R^k, k < R/2.  Up to FOUR radix-2 stages
// are fused per pass: a radix-16 (8, 4, 2) butterfly in registers -- the blocks of fft_r16.hpp / win8.hpp -- so a
// pass costs one LDS read and write of the data and one barrier per four stages (2048 points: 16 x 16 x 8, three
// passes).  The data flow, and with it the output order, is that of the radix-2 network: a fused pass over
// blocks of 2^b points with stride q = 2^(b-M) computes y[k] = DFT_{2^M}(x[j + m q])[k], multiplies by
// W_{2^b}^(l k) (l = element index inside the sub-block) and stores it at j + bitrev_M(k) q.  LOGT > 0: 2^LOGT
// interleaved transforms at once (element e of transform c at x[(e << LOGT) | c]: a tile of columns of a
// row-major matrix; consecutive threads take consecutive columns, so LDS accesses stay conflict free).
// De-rotation of a window sample by a Doppler phasor (rmx_caf_batch): separately rounded products and sums, as
// numpy multiplies complex64 arrays (no contraction), so that the rotated window equals the oracle's bit for bit.
__device__ __forceinline__ float2 rot_mul(float2 v, float2 r) {
#pragma clang fp contract(off)
    const float re = v.x * r.x - v.y * r.y;
    const float im = v.x * r.y + v.y * r.x;
    return make_float2(re, im);
}

// Strided copy loop with U loads in flight per thread: all U global loads of a batch are issued before the
// first value is consumed (a plain `for (n = tid; n < end; n += stride) dst[n] = src[n]` with run-time bounds
// is compiled to one load, one wait and one store per trip: the kernels here are streaming kernels and
// need the memory-level parallelism).
template <int U, class LoadFn, class StoreFn>
__device__ __forceinline__ void batched(int start, int end, int stride, LoadFn&& ld, StoreFn&& st) {
    for (int n0 = start; n0 < end; n0 += U * stride) {
        decltype(ld(0)) tmp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u * stride;
            if (n < end) tmp[u] = ld(n);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u * stride;
            if (n < end) st(n, tmp[u]);
        }
    }
}

// LDS element index -> padded position: one complex of padding per 16 keeps the stride-16 accesses of the last
// fused passes (each thread walks 16 neighbouring elements) on distinct banks: lane i's element 16 i + m sits at
// 17 i + m, and 17 is odd.  Every access to a transform buffer goes through lp().
__host__ __device__ constexpr long lp(long e) { return e + (e >> 4); }
template <int M>
__device__ __forceinline__ constexpr int brev_m(int k) {
    int r = 0;
    for (int i = 0; i < M; ++i) r |= ((k >> i) & 1) << (M - 1 - i);
    return r;
}
template <int RAD>
__device__ __forceinline__ void dft_reg(float2 (&v)[RAD]) {   // natural order in and out
    if constexpr (RAD == 16) dft16(v);
    else if constexpr (RAD == 8) w8::dft8(v);
    else if constexpr (RAD == 4) dft4(v[0], v[1], v[2], v[3]);
    else {
        const float2 a = v[0], b = v[1];
        v[0] = make_float2(a.x + b.x, a.y + b.y);
        v[1] = make_float2(a.x - b.x, a.y - b.y);
    }
}
// W_R^e for any e < R from the half table
__device__ __forceinline__ float2 tw_full(const float2* __restrict__ tw, int e, int half) {
    const float2 w = tw[e & (half - 1)];
    return (e & half) ? make_float2(-w.x, -w.y) : w;
}
// Where a pass reads its group from and writes it to.  A group's first element has the (unpadded) index
// E0 = (j << LOGT) | c, its m-th element sits o = m * (q << LOGT) further on.
struct LdsIO {      // the transform buffer in LDS, padded by lp()
    float2* x;
    static __host__ __device__ constexpr int pos(int e) { return e + (e >> 4); }
    struct H {
        float2* p;
        // lp(E0 + o) = lp(E0) + o + (o >> 4): E0's 16-block has fewer than the stride's worth of elements in front
        // of E0 whenever the stride is below 16, and whole blocks are crossed otherwise
        __device__ __forceinline__ float2 ld(int o) const { return p[o + (o >> 4)]; }
        __device__ __forceinline__ void st(int o, float2 v) const { p[o + (o >> 4)] = v; }
    };
    __device__ __forceinline__ H open(int E0) const { return H{x + (E0 + (E0 >> 4))}; }
};
// Tiles of 2^LOGT interleaved columns (LOGT > 0) are NOT padded: a half-wave (16 columns x 2 rows, or 8 x 4) walks
// 32 consecutive complex numbers in every pass whose lanes take neighbouring rows -- conflict free -- and lp()'s
// extra element per 16 made each of those accesses, and every tile load / store loop, a 2-way conflict (62 % of the
// LDS cycles of g_cols_fwd were conflict cycles).  Only the last pass (lanes 2^M rows apart) stays 2-way.
struct LdsFlat {
    float2* x;
    static __host__ __device__ constexpr int pos(int e) { return e; }
    struct H {
        float2* p;
        __device__ __forceinline__ float2 ld(int o) const { return p[o]; }
        __device__ __forceinline__ void st(int o, float2 v) const { p[o] = v; }
    };
    __device__ __forceinline__ H open(int E0) const { return H{x + E0}; }
};
#ifndef RMX_TILE_FLAT_FWD
#define RMX_TILE_FLAT_FWD 1
#endif
#ifndef RMX_TILE_FLAT_INV
#define RMX_TILE_FLAT_INV 0
#endif
template <int LOGT, bool FLAT>
using LdsIOFor = std::conditional_t<LOGT == 0 || !FLAT, LdsIO, LdsFlat>;
using TileFwd = LdsIOFor<4, RMX_TILE_FLAT_FWD != 0>;      // tile layout of g_cols_fwd / g_cols_inv
using TileInv = LdsIOFor<4, RMX_TILE_FLAT_INV != 0>;
// Second layout of the same buffer: 16 complex of padding per 2^BM elements, e -> e + 16 (e >> BM).  lp() keeps a
// thread's 16 neighbours (and lanes 16 elements apart) on distinct banks but puts 32 CONSECUTIVE elements on 34
// slots (lanes 0 and 31 collide: every unit-stride access a 2-way conflict); this one keeps unit-stride accesses
// and the 16-lanes-here, 16-lanes-2^BM-further pattern of a q = 16 pass conflict free.  g_rows_fused alternates:
// passes whose lanes walk consecutive elements use this layout, the 16-neighbour butterflies lp().  A pass that
// reads one layout and writes the other is only safe when a butterfly's elements occupy the same REGION in both
// (its lanes sit in one wave, whose reads precede its writes): blocks of 2^BM elements must start at the same
// position in both layouts, which holds for BM = 8 (272 h) and not for BM = 6 (80 h against 68 h).
template <int BM>
struct LdsIOA {
    float2* x;
    static __host__ __device__ constexpr int pos(int e) { return e + ((e >> BM) << 4); }
    struct H {
        float2* p;
        __device__ __forceinline__ float2 ld(int o) const { return p[o + ((o >> BM) << 4)]; }     // o: below 2^BM or a multiple of it
        __device__ __forceinline__ void st(int o, float2 v) const { p[o + ((o >> BM) << 4)] = v; }
    };
    __device__ __forceinline__ H open(int E0) const { return H{x + (E0 + ((E0 >> BM) << 4))}; }
};
// adaptors: a pass end given as a callable on the element index (source) or on (E0, o, value) (sink)
template <class F>
struct SrcFn {
    F f;
    struct H {
        const F& f;
        int E0;
        __device__ __forceinline__ float2 ld(int o) const { return f(E0 + o); }
    };
    __device__ __forceinline__ H open(int E0) const { return H{f, E0}; }
};
template <class F>
struct DstFn {
    F f;
    struct H {
        const F& f;
        int E0;
        __device__ __forceinline__ void st(int o, float2 v) const { f(E0, o, v); }
    };
    __device__ __forceinline__ H open(int E0) const { return H{f, E0}; }
};
// a first-pass source whose elements from offset zoff on (the zero-padded half of a column) are zero: no load, and
// with compile-time strides the butterfly's additions of those inputs fold away
template <class IO>
struct SrcZeroTail {
    IO io;
    int zoff;
    struct H {
        decltype(std::declval<IO>().open(0)) h;
        int zoff;
        __device__ __forceinline__ float2 ld(int o) const { return o >= zoff ? make_float2(0.f, 0.f) : h.ld(o); }
    };
    __device__ __forceinline__ H open(int E0) const { return H{io.open(E0), zoff}; }
};
template <class F>
__device__ __forceinline__ SrcFn<F> make_src(F f) { return SrcFn<F>{f}; }
template <class F>
__device__ __forceinline__ DstFn<F> make_dst(F f) { return DstFn<F>{f}; }

// DIF pass: M stages on blocks of 2^b (forward, W = exp(-2 pi i / R)); no barrier inside
// Where a pass takes W_R^(l k 2^(logR-b)) from (TWM): 0 = the half table tw[e], e < R/2, with the sign fix-up;
// 2 = this pass's own table tw[(k-1) q + l] (consecutive lanes read consecutive entries: the strided reads of
// mode 0 are 2..16-way LDS bank conflicts -- SQ_LDS_BANK_CONFLICT was half of the LDS cycles of g_rows_fused)
// 3 = registers: tw.w[k-1], the thread's own 2^M - 1 twiddles of this pass (a pass with one butterfly per thread:
// they are the same for every transform the thread ever does)
struct TwRegs {
    float2 w[15];
};
template <int TWM, class TwT>
__device__ __forceinline__ float2 pass_tw(const TwT& tw, int es, int l, int k, int q, int half) {
    if constexpr (TWM == 3) return tw.w[k - 1];
    else if constexpr (TWM == 2) return tw[(k - 1) * q + l];
    else return tw_full(tw, es * k, half);
}
template <int M, int LOGT, int TWM = 0, class Src, class Dst, class TwT = const float2*>
__device__ __forceinline__ void dif_pass(int logR, int b, const TwT& tw, int tid, int nthr, const Src& src,
                                         const Dst& dst) {
    constexpr int RAD = 1 << M, T = 1 << LOGT;
    const int qb = b - M, q = 1 << qb, half = 1 << (logR - 1);
    const int work = (1 << (logR - M)) << LOGT;
    for (int idx = tid; idx < work; idx += nthr) {
        const int c = idx & (T - 1), i = idx >> LOGT;
        const int l = i & (q - 1);
        const int j = ((i >> qb) << b) | l;
        const int E0 = (j << LOGT) + c;
        const int qs = q << LOGT;
        const auto hs = src.open(E0);
        const auto hd = dst.open(E0);
        float2 v[RAD];
#pragma unroll
        for (int m = 0; m < RAD; ++m) v[m] = hs.ld(m * qs);
        dft_reg<RAD>(v);
        hd.st(0, v[0]);
        if (qb > 0) {
            const int es = l << (logR - b);
#pragma unroll
            for (int k = 1; k < RAD; ++k) hd.st(brev_m<M>(k) * qs, g_cmul(v[k], pass_tw<TWM>(tw, es, l, k, q, half)));
        } else {
#pragma unroll
            for (int k = 1; k < RAD; ++k) hd.st(brev_m<M>(k) * qs, v[k]);
        }
    }
}
// DIT pass with conjugated twiddles: the exact inverse data flow (unnormalised); no barrier inside
template <int M, int LOGT, int TWM = 0, class Src, class Dst, class TwT = const float2*>
__device__ __forceinline__ void dit_pass(int logR, int b, const TwT& tw, int tid, int nthr, const Src& src,
                                         const Dst& dst) {
    constexpr int RAD = 1 << M, T = 1 << LOGT;
    const int qb = b - M, q = 1 << qb, half = 1 << (logR - 1);
    const int work = (1 << (logR - M)) << LOGT;
    for (int idx = tid; idx < work; idx += nthr) {
        const int c = idx & (T - 1), i = idx >> LOGT;
        const int l = i & (q - 1);
        const int j = ((i >> qb) << b) | l;
        const int E0 = (j << LOGT) + c;
        const int qs = q << LOGT;
        const auto hs = src.open(E0);
        const auto hd = dst.open(E0);
        float2 v[RAD];      // (im, re)-swapped: swap o DFT o swap = conj(DFT)
        {
            const float2 e = hs.ld(0);
            v[0] = make_float2(e.y, e.x);
        }
        if (qb > 0) {
            const int es = l << (logR - b);
#pragma unroll
            for (int k = 1; k < RAD; ++k) {
                const float2 e = g_cmulc(hs.ld(brev_m<M>(k) * qs), pass_tw<TWM>(tw, es, l, k, q, half));
                v[k] = make_float2(e.y, e.x);
            }
        } else {
#pragma unroll
            for (int k = 1; k < RAD; ++k) {
                const float2 e = hs.ld(brev_m<M>(k) * qs);
                v[k] = make_float2(e.y, e.x);
            }
        }
        dft_reg<RAD>(v);
#pragma unroll
        for (int m = 0; m < RAD; ++m) hd.st(m * qs, make_float2(v[m].y, v[m].x));
    }
}
// run-time stage count -> the instantiated pass
template <int LOGT, class Src, class Dst>
__device__ __forceinline__ void dif_pass_m(int M, int logR, int b, const float2* __restrict__ tw, int tid, int nthr,
                                           const Src& src, const Dst& dst) {
    if (M == 4) dif_pass<4, LOGT>(logR, b, tw, tid, nthr, src, dst);
    else if (M == 3) dif_pass<3, LOGT>(logR, b, tw, tid, nthr, src, dst);
    else if (M == 2) dif_pass<2, LOGT>(logR, b, tw, tid, nthr, src, dst);
    else dif_pass<1, LOGT>(logR, b, tw, tid, nthr, src, dst);
}
template <int LOGT, class Src, class Dst>
__device__ __forceinline__ void dit_pass_m(int M, int logR, int b, const float2* __restrict__ tw, int tid, int nthr,
                                           const Src& src, const Dst& dst) {
    if (M == 4) dit_pass<4, LOGT>(logR, b, tw, tid, nthr, src, dst);
    else if (M == 3) dit_pass<3, LOGT>(logR, b, tw, tid, nthr, src, dst);
    else if (M == 2) dit_pass<2, LOGT>(logR, b, tw, tid, nthr, src, dst);
    else dit_pass<1, LOGT>(logR, b, tw, tid, nthr, src, dst);
}
// Whole transforms with caller-supplied ends: the FIRST pass reads through `first`, the LAST writes through
// `last`, everything between lives in the LDS buffer x.  With first / last = LdsIO{x} the transform is in place in
// LDS; a kernel that passes its global source / sink here saves one LDS round trip and one barrier per end.
// Passes: radix 16 while more than four stages remain, the rest (1..4 stages) in the last pass.  A barrier
// follows every pass but the last.
// DIF: natural in -> bit-reversed out (forward).
template <int LOGT, class First, class Last>
__device__ __forceinline__ void fft_dif(float2* x, int logR, const float2* __restrict__ tw, int tid, int nthr,
                                        const First& first, const Last& last) {
    const LdsIOFor<LOGT, RMX_TILE_FLAT_FWD != 0> mid{x};
    int b = logR;
    if (b <= 4) {
        dif_pass_m<LOGT>(b, logR, b, tw, tid, nthr, first, last);
        return;
    }
    // (five left-over stages as ONE radix-32 pass instead of 16 + 2, tried on the column tiles: cfg2 4.06 -> 4.60
    // ms -- 64 data registers, half the threads idle in that pass)
    dif_pass<4, LOGT>(logR, b, tw, tid, nthr, first, mid);
    __syncthreads();
    for (b -= 4; b > 4; b -= 4) {
        dif_pass<4, LOGT>(logR, b, tw, tid, nthr, mid, mid);
        __syncthreads();
    }
    dif_pass_m<LOGT>(b, logR, b, tw, tid, nthr, mid, last);
}
// DIT with conjugated twiddles: bit-reversed in -> natural out (inverse, unnormalised)
// tw16 (optional, logR > 8): the second pass's own table [k = 1..15][l < 16] = W_R^(l k 2^(logR-8)) (build_tw16): its
// reads of the shared table are 16 lanes 2^(logR-8) k entries apart -- all on one or two banks, an 8..16-way conflict
template <int LOGT, class First, class Last>
__device__ __forceinline__ void fft_dit_inv(float2* x, int logR, const float2* __restrict__ tw, int tid, int nthr,
                                            const First& first, const Last& last, const float2* __restrict__ tw16 = nullptr) {
    const LdsIOFor<LOGT, RMX_TILE_FLAT_INV != 0> mid{x};
    if (logR <= 4) {
        dit_pass_m<LOGT>(logR, logR, logR, tw, tid, nthr, first, last);
        return;
    }
    // radix 16 first (a first pass that reads from HBM has 16 loads in flight per thread), what is left over
    // (1..4 stages) in the last pass
    int b = 0;
    for (; b + 4 < logR; b += 4) {
        if (b == 0) dit_pass<4, LOGT>(logR, 4, tw, tid, nthr, first, mid);
        else if (b == 4 && tw16) dit_pass<4, LOGT, 2>(logR, 8, tw16, tid, nthr, mid, mid);
        else dit_pass<4, LOGT>(logR, b + 4, tw, tid, nthr, mid, mid);
        __syncthreads();
    }
    dit_pass_m<LOGT>(logR - b, logR, logR, tw, tid, nthr, mid, last);
}
__device__ __forceinline__ void build_tw16(float2* t, const float2* __restrict__ tw, int logR, int tid, int nthr) {
    for (int e = tid; e < 240; e += nthr) t[e] = tw_full(tw, ((e & 15) << (logR - 8)) * ((e >> 4) + 1), 1 << (logR - 1));
}
// in place in LDS, barrier behind the last pass too
template <int LOGT = 0>
__device__ __forceinline__ void lds_dif(float2* x, int logR, const float2* __restrict__ tw, int tid, int nthr) {
    const LdsIOFor<LOGT, RMX_TILE_FLAT_FWD != 0> io{x};
    fft_dif<LOGT>(x, logR, tw, tid, nthr, io, io);
    __syncthreads();
}
template <int LOGT = 0>
__device__ __forceinline__ void lds_dit_inv(float2* x, int logR, const float2* __restrict__ tw, int tid, int nthr) {
    const LdsIOFor<LOGT, RMX_TILE_FLAT_INV != 0> io{x};
    fft_dit_inv<LOGT>(x, logR, tw, tid, nthr, io, io);
    __syncthreads();
}

// 'full' order index (lag ascending from -(N-1)) of circular index m of an L = 2N point correlation;
// m == N (lag -N) is not part of the 'full' output: returns -1.
__device__ __forceinline__ int full_index(int m, int N) { return m < N ? m + N - 1 : (m == N ? -1 : m - N - 1); }
__device__ __forceinline__ int circ_index(int k, int N) { return k >= N - 1 ? k - (N - 1) : k + N + 1; }

// One output of an inverse's last pass into a thread's running (max |r|^2, lowest 'full' index), branch-free: the
// index is n + N - 1 below N and n - N - 1 above (n == N, lag -N, is not part of the 'full' output: its value is
// replaced by -2, which never beats the -1 the search starts from).  With n = tid + a multiple of R / 16 the
// comparisons against N fold at compile time for all but one element.
__device__ __forceinline__ void scan_update(float& best, int& bk, float m2, int n, int N) {
    const int k = n < N ? n + (N - 1) : n - (N + 1);
    const float v = n == N ? -2.0f : m2;
    const bool better = (v > best) | ((v == best) & (k < bk));      // (bitwise: short-circuit forms become branches)
    best = better ? v : best;
    bk = better ? k : bk;
}
// workgroup argmax of (value, lowest index): every thread passes its best candidate
__device__ __forceinline__ void block_argmax(float& v, int& k, float* sv, int* sk, int tid, int nthr) {
    sv[tid] = v;
    sk[tid] = k;
    __syncthreads();
    for (int s = nthr >> 1; s > 0; s >>= 1) {
        if (tid < s) {
            const float ov = sv[tid + s];
            const int ok = sk[tid + s];
            if (ov > sv[tid] || (ov == sv[tid] && ok < sk[tid])) { sv[tid] = ov; sk[tid] = ok; }
        }
        __syncthreads();
    }
    v = sv[0];
    k = sk[0];
    __syncthreads();
}

// the same with a wave-level shuffle reduction first: one (value, index) entry per wave in LDS (<= 16)
__device__ __forceinline__ void block_argmax_w(float& v, int& k, float* sv, int* sk, int tid, int nthr) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int ok = __shfl_xor(k, off, 64);
        if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
    }
    const int nw = (nthr + 63) >> 6;
    if ((tid & 63) == 0) { sv[tid >> 6] = v; sk[tid >> 6] = k; }
    __syncthreads();
    v = sv[0];
    k = sk[0];
    for (int w = 1; w < nw; ++w) {
        const float ov = sv[w];
        const int ok = sk[w];
        if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
    }
    __syncthreads();
}

__device__ __forceinline__ float parabola(float a, float b, float c) {
    const double den = (double)a - 2.0 * (double)b + (double)c;
    return den == 0.0 ? 0.0f : (float)(0.5 * ((double)a - (double)c) / den);
}

// ---- small path -------------------------------------------------------------------------------
// spectrum layout: [item][L] complex, bit-reversed order, scaled by `scale`
template <bool U8>
__global__ __launch_bounds__(1024) void g_fwd_small(const void* __restrict__ iq, float2* __restrict__ spec,
                                                         const float2* __restrict__ tw, int N, int logL,
                                                         long first_item, float scale,
                                                         const float2* __restrict__ rot = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float2* x = reinterpret_cast<float2*>(gsm);
    const int L = 1 << logL, tid = threadIdx.x, nthr = blockDim.x;
    const long item = first_item + blockIdx.x;
    for (int n = N + tid; n < L; n += nthr) x[lp(n)] = make_float2(0.f, 0.f);   // the zero-padded half
    batched<8>(tid, N, nthr,
               [&](int n) -> float2 {
                   if constexpr (U8) {
                       const uchar2 b = reinterpret_cast<const uchar2*>(iq)[item * N + n];
                       return make_float2((float)b.x - 127.5f, (float)b.y - 127.5f);
                   } else {
                       return reinterpret_cast<const float2*>(iq)[item * N + n];
                   }
               },
               [&](int n, float2 v) { x[lp(n)] = rot ? rot_mul(v, rot[n]) : v; });
    __syncthreads();
    lds_dif(x, logL, tw, tid, nthr);
    float2* out = spec + (long)blockIdx.x * L;
    for (int n = tid; n < L; n += nthr) {
        const float2 e = x[lp(n)];
        out[n] = make_float2(e.x * scale, e.y * scale);
    }
}

struct GPair {
    int i, j;
};

__global__ __launch_bounds__(1024) void g_pair_small(const float2* __restrict__ spec, const float2* __restrict__ spec_j,
                                                          const float2* __restrict__ tw,
                                                          const GPair* __restrict__ pairs, int n_pairs,
                                                          int n_buoys, int N, int logL, long first_window,
                                                          float out_scale, int* __restrict__ lag_int,
                                                          float* __restrict__ lag_frac, float* __restrict__ peak) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float2* x = reinterpret_cast<float2*>(gsm);
    const int L = 1 << logL, tid = threadIdx.x, nthr = blockDim.x;
    float* sv = reinterpret_cast<float*>(gsm + (size_t)lp(L) * 8);
    int* sk = reinterpret_cast<int*>(sv + nthr);
    const int wl = blockIdx.x / n_pairs, q = blockIdx.x % n_pairs;
    const GPair pr = pairs[q];
    const float2* xi = spec + ((long)wl * n_buoys + pr.i) * L;
    const float2* xj = spec_j + ((long)wl * n_buoys + pr.j) * L;
    batched<4>(tid, L, nthr, [&](int n) -> float4 { const float2 a = xj[n], b = xi[n]; return make_float4(a.x, a.y, b.x, b.y); },
               [&](int n, float4 v) { x[lp(n)] = g_cmulc(make_float2(v.x, v.y), make_float2(v.z, v.w)); });   // X_j conj(X_i)
    __syncthreads();
    lds_dit_inv(x, logL, tw, tid, nthr);
    float best = -1.0f;
    int bk = 0x7fffffff;
    for (int m = tid; m < L; m += nthr) {
        const int k = full_index(m, N);
        if (k < 0) continue;
        const float2 e = x[lp(m)];
        const float v = e.x * e.x + e.y * e.y;
        if (v > best || (v == best && k < bk)) { best = v; bk = k; }
    }
    block_argmax(best, bk, sv, sk, tid, nthr);
    if (tid == 0) {
        const float b = sqrtf(best) * out_scale;
        float frac = 0.0f;
        if (bk > 0 && bk < 2 * N - 2) {
            const float2 ra = x[lp(circ_index(bk - 1, N))], rc = x[lp(circ_index(bk + 1, N))];
            frac = parabola(sqrtf(ra.x * ra.x + ra.y * ra.y) * out_scale, b,
                            sqrtf(rc.x * rc.x + rc.y * rc.y) * out_scale);
        }
        const long o = (first_window + wl) * (long)n_pairs + q;
        lag_int[o] = bk - (N - 1);
        lag_frac[o] = frac;
        peak[o] = b;
    }
}

// ---- large path -------------------------------------------------------------------------------
__device__ __forceinline__ int brev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// W_L^(m), m in [0, L): product of the two host tables (hi digit, lo digit)
__device__ __forceinline__ float2 big_tw(long m, int lo_bits, const float2* __restrict__ thi,
                                         const float2* __restrict__ tlo) {
    return g_cmul(thi[m >> lo_bits], tlo[m & ((1L << lo_bits) - 1)]);
}

// Row transforms of length R over a [batch * n_rows][R] array, in place.  A workgroup takes
// 256 / min(256, R/4) rows at once (short rows would leave most of its threads idle).
//   FWD = true : DIF; then (TW) multiply position q by W_L^(row_in_batch * brev(q))
//   FWD = false: DIT inverse; then (TW) multiply position n by conj W_L^(n * brev(row_in_batch))
//   PROD (inverse only): the row is not read from `data` but formed on the fly as
//        X_j[row] conj(X_i[row]) from the spectra (slot = window-in-chunk * n_pairs + pair), which
//        saves the product's own pass through HBM; the result is written to `data`.
// LOGR > 0: the row length as a compile-time constant (pass loops unrolled, strides and LDS offsets immediates)
template <bool FWD, bool TW, bool PROD = false, int LOGR = 0>
__global__ __launch_bounds__(kGThreads) void g_rows(float2* __restrict__ data, const float2* __restrict__ tw, int logR_arg,
                                                    int n_rows, int row_bits, long Ltot, int lo_bits,
                                                    const float2* __restrict__ thi,
                                                    const float2* __restrict__ tlo, float scale, long total_rows,
                                                    const float2* __restrict__ spec = nullptr,
                                                    const float2* __restrict__ spec_j = nullptr,
                                                    const GPair* __restrict__ pairs = nullptr, int n_pairs = 0,
                                                    int n_buoys = 0, int tpr_arg = 0) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int logR = LOGR > 0 ? LOGR : logR_arg;
    const int R = 1 << logR;
    const int tpr = LOGR > 0 ? rows_tpr(1 << LOGR) : (tpr_arg > 0 ? tpr_arg : rows_tpr(R));   // threads per row
    const int rpw = kGThreads / tpr;                                                   // rows per workgroup
    const int g = threadIdx.x / tpr, tid = threadIdx.x % tpr;
    float2* x = reinterpret_cast<float2*>(gsm) + (long)g * lp(R);
    // the W_R table goes to LDS (butterflies would otherwise fetch their twiddles
    // through the vector memory path, a dependent cache-latency access in the inner loop)
    float2* twl = reinterpret_cast<float2*>(gsm) + (long)rpw * lp(R) + (TW ? (long)rpw * ((1 << (logR >> 1)) + (R >> (logR >> 1))) : 0);
    for (int k = threadIdx.x; k < (R >> 1); k += kGThreads) twl[k] = tw[k];
    float2* tw16 = twl + (R >> 1);                // [15][16]: the inverse's q = 16 pass (fft_dit_inv)
    // (rows of 4096 go without: three workgroups of 52 KiB share a CU, and 2 KiB more make it two -- 5.46 -> 5.88 ms)
    const bool own16 = !FWD && logR > 8 && logR < 12;
    if (own16) build_tw16(tw16, tw, logR, threadIdx.x, kGThreads);
    static_assert((FWD && !TW && !PROD) || (!FWD && TW && PROD), "the two row passes of the four-step");
    const LdsIO lds{x};
    // (one block of rpw rows per workgroup.  A persistent variant -- the workgroup walking blocks blockIdx.x, +
    // gridDim.x, ... with the W_R table loaded once -- measured 15 % slower on 2048-point rows and equal on 4096-point
    // ones: hipcc schedules the loop body worse than the straight-line kernel.)
    // (slot-major.  Walking all pairs of a window over a block of 8 rows per XCD, so that the spectra rows stay in
    // L2, measured 7 % SLOWER on cfg5: the kernel is not waiting for HBM reads.)
    const long ridx = (long)blockIdx.x * rpw + g;
    const bool live = ridx < total_rows;
    float2* row = data + ridx * R;
    const int rib = (int)(ridx & (n_rows - 1));   // row index inside its batch element (n_rows = 2^row_bits)
    if constexpr (FWD) {
        __syncthreads();                          // W_R table
        // forward rows: the first pass reads the row straight from HBM (its 2^M inputs are R / 2^M apart, so
        // consecutive threads read consecutive elements), the last leaves the spectrum in LDS for the store loop
        fft_dif<0>(x, logR, twl, tid, tpr, make_src([&](int E) -> float2 { return live ? row[E] : make_float2(0.f, 0.f); }), lds);
        __syncthreads();
        if (live)
            batched<8>(tid, R, tpr, [&](int n) -> float2 { return x[lp(n)]; },
                       [&](int n, float2 e) { row[n] = make_float2(e.x * scale, e.y * scale); });
    } else {
        // inverse rows: X_j conj(X_i) formed on load (the product never exists in HBM)
        const int a = logR >> 1, n1 = 1 << a, n2 = R >> a;
        float2* t1 = reinterpret_cast<float2*>(gsm) + (long)(kGThreads / tpr) * lp(R) + (long)g * (n1 + n2);
        float2* t2 = t1 + n1;
        {   // W_L^(c*e) for this row's multiplier c and every exponent e < R = T1[e & (2^a - 1)] * T2[e >> a]
            const long c = (long)brev(rib, row_bits);
            for (int e = tid; e < n1 + n2; e += tpr) {
                const long ee = e < n1 ? (long)e : ((long)(e - n1) << a);
                t1[e] = big_tw((c * ee) & (Ltot - 1), lo_bits, thi, tlo);
            }
        }
        if (live) {
            const long slot = ridx >> row_bits;
            const int wl = (int)(slot / n_pairs), q = (int)(slot % n_pairs);
            const GPair pr = pairs[q];
            const float2* xi = spec + (((long)wl * n_buoys + pr.i) * n_rows + rib) * R;
            const float2* xj = spec_j + (((long)wl * n_buoys + pr.j) * n_rows + rib) * R;
            batched<RMX_PROD_BATCH>(tid, R, tpr, [&](int n) -> float4 { const float2 u = xj[n], v = xi[n]; return make_float4(u.x, u.y, v.x, v.y); },
                       [&](int n, float4 v) { x[lp(n)] = g_cmulc(make_float2(v.x, v.y), make_float2(v.z, v.w)); });
        }
        __syncthreads();
        fft_dit_inv<0>(x, logR, twl, tid, tpr, lds, lds, own16 ? tw16 : nullptr);
        __syncthreads();
        // (measured on cfg2: the last pass straight to HBM through the twiddle 1.73 ms, this loop batched eight
        // deep 2.05 ms, plain 1.68 ms)
        if (live)
            for (int n = tid; n < R; n += tpr) {
                const float2 w = g_cmul(t1[n & (n1 - 1)], t2[n >> a]);
                const float2 r = g_cmulc(x[lp(n)], w);
                row[n] = make_float2(r.x * scale, r.y * scale);
            }
    }
}

// Inverse rows with a resident anchor (default pair list: all i < j, i-major; more than four buoys, the Doppler search).
// g_rows<inverse> above loads X_i's and X_j's row for every pair: 2 P row loads per (window, row index).  Here a
// workgroup takes the rows `rib` of ONE (window, anchor i) and walks the anchor's pairs (i, j), j = i + 1 .. B - 1: X_i's
// row stays in registers (R / tpr = 8 complex per thread), so P + B - 1 rows are loaded instead of 2 P, and the row's
// W_L^(c e) factors (T1, T2: 64 products of two table entries per row) are built once per workgroup instead of once per
// pair.  Workgroup order: all anchors of a row block one after the other on ONE XCD (consecutive workgroup ids go round
// robin over the 8 XCDs), so the X_j rows that the anchors i < j share are L2 hits.
// grid: x = (n_rows / rpw) * (n_buoys - 1), y = windows of the chunk.
template <int LOGR>
__global__ __launch_bounds__(kGThreads) void g_rows_anchor(float2* __restrict__ data, const float2* __restrict__ tw,
                                                           int n_rows, int row_bits, long Ltot, int lo_bits,
                                                           const float2* __restrict__ thi, const float2* __restrict__ tlo,
                                                           float scale, const float2* __restrict__ spec,
                                                           const float2* __restrict__ spec_j, int n_pairs, int n_buoys) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    constexpr int logR = LOGR, R = 1 << LOGR, tpr = rows_tpr(R), rpw = kGThreads / tpr, EPT = R / tpr;   // elements per thread
    const int g = threadIdx.x / tpr, tid = threadIdx.x % tpr;
    float2* x = reinterpret_cast<float2*>(gsm) + (long)g * lp(R);
    constexpr int a = logR >> 1, n1 = 1 << a, n2 = R >> a;
    float2* t1 = reinterpret_cast<float2*>(gsm) + (long)rpw * lp(R) + (long)g * (n1 + n2);
    float2* t2 = t1 + n1;
    float2* twl = reinterpret_cast<float2*>(gsm) + (long)rpw * lp(R) + (long)rpw * (n1 + n2);
    for (int k = threadIdx.x; k < (R >> 1); k += kGThreads) twl[k] = tw[k];
    float2* tw16 = twl + (R >> 1);
    constexpr bool own16 = logR > 8 && logR < 12;
    if constexpr (own16) build_tw16(tw16, tw, logR, threadIdx.x, kGThreads);
    // (the 256-block pass's twiddles in registers instead of this table: 8 buoys x 16384 1.36 -> 1.43 ms, cfg5 +-0: not used)
    const int nrb = n_rows / rpw, na = n_buoys - 1, bid = blockIdx.x;
    int rb, i;
    if ((nrb & 7) == 0) {
        const int s = bid >> 3;
        rb = (s / na) * 8 + (bid & 7);
        i = s % na;
    } else {
        rb = bid / na;
        i = bid % na;
    }
    const int rib = rb * rpw + g;
    const long wl = blockIdx.y;
    {   // W_L^(c*e) for this row's multiplier c and every exponent e < R = T1[e & (2^a - 1)] * T2[e >> a]
        const long c = (long)brev(rib, row_bits);
        for (int e = tid; e < n1 + n2; e += tpr) {
            const long ee = e < n1 ? (long)e : ((long)(e - n1) << a);
            t1[e] = big_tw((c * ee) & (Ltot - 1), lo_bits, thi, tlo);
        }
    }
    // rows travel as float4 (two complex per lane and load): thread tid owns elements 2 tid + {0, 1} + 2 tpr k
    static_assert(EPT % 2 == 0, "two complex per load");
    float4 anc[EPT / 2];
    {
        const float4* xi = reinterpret_cast<const float4*>(spec + (((long)wl * n_buoys + i) * n_rows + rib) * R);
#pragma unroll
        for (int k = 0; k < EPT / 2; ++k) anc[k] = xi[tid + k * tpr];
    }
    const LdsIO lds{x};
    const int q0 = i * n_buoys - (i * (i + 1)) / 2 - (i + 1);        // pair (i, j) is number q0 + j of the i-major list
    auto load_row = [&](float4 (&d)[EPT / 2], int j) __attribute__((always_inline)) {
        const float4* xj = reinterpret_cast<const float4*>(spec_j + (((long)wl * n_buoys + j) * n_rows + rib) * R);
#pragma unroll
        for (int k = 0; k < EPT / 2; ++k) d[k] = xj[tid + k * tpr];
    };
    float4 nx[EPT / 2];
    load_row(nx, i + 1);
    // the thread's store twiddles conj-applied below, W_L^(c n) = T1[n & (2^a - 1)] T2[n >> a] for its EPT elements: the same for
    // every pair of the run, so they live in registers (16 LDS reads and 8 complex products per pair and thread less)
    __syncthreads();                                      // t1 / t2 (and the pass tables) are complete
    float2 wst[EPT];
#pragma unroll
    for (int k = 0; k < EPT / 2; ++k) {
        const int n = 2 * (tid + k * tpr);
        wst[2 * k] = g_cmul(t1[n & (n1 - 1)], t2[n >> a]);
        wst[2 * k + 1] = g_cmul(t1[(n + 1) & (n1 - 1)], t2[(n + 1) >> a]);
    }
    for (int j = i + 1; j < n_buoys; ++j) {
#pragma unroll
        for (int k = 0; k < EPT / 2; ++k) {               // X_j conj(X_i)
            const int n = 2 * (tid + k * tpr);
            x[lp(n)] = g_cmulc(make_float2(nx[k].x, nx[k].y), make_float2(anc[k].x, anc[k].y));
            x[lp(n + 1)] = g_cmulc(make_float2(nx[k].z, nx[k].w), make_float2(anc[k].z, anc[k].w));
        }
        if (j + 1 < n_buoys) load_row(nx, j + 1);         // the next pair's row travels during this inverse
        __syncthreads();                                  // (the first trip: the tables too)
        fft_dit_inv<0>(x, logR, twl, tid, tpr, lds, lds, own16 ? tw16 : nullptr);
        __syncthreads();
        float4* row = reinterpret_cast<float4*>(data + (((long)wl * n_pairs + (q0 + j)) * n_rows + rib) * R);
#pragma unroll
        for (int k = 0; k < EPT / 2; ++k) {               // two neighbouring elements per 16-byte store
            const int n = 2 * (tid + k * tpr);
            const float2 r0 = g_cmulc(x[lp(n)], wst[2 * k]), r1 = g_cmulc(x[lp(n + 1)], wst[2 * k + 1]);
            row[tid + k * tpr] = make_float4(r0.x * scale, r0.y * scale, r1.x * scale, r1.y * scale);
        }
        __syncthreads();                                  // x is rewritten
    }
}

// ---- both row passes in one kernel (few buoys) ---------------------------------------------------------------
// The inverse row pass of pair (i, j) needs row `rib` of X_i and X_j only, and that row of a spectrum is the forward
// row transform of row `rib` of the column pass's output.  With few buoys a workgroup therefore does everything for
// one (window, rib): the NB forward row transforms, whose LAST pass -- a radix-16 butterfly on 16 neighbouring
// elements per thread -- ends in registers (NB x 16 complex per thread: the register file is the larger memory, 512
// KiB per CU against 160 KiB of LDS), then for every pair the product out of those registers straight into the
// FIRST pass of the inverse (the same 16 neighbours), the remaining passes in LDS and the twiddled store.  The
// spectra are never written: per window (3 NB + 3 P) L x 8 bytes of HBM traffic instead of (5 NB + 5 P) L x 8 for
// the whole four-step (cfg2: 14.4 GB instead of 24 GB).  R / 16 threads per row; the scale of the forward
// transform (a power of two) is applied once, squared, at the store.
// Passes on the way down to the 16-point blocks (DIF) / up from them (DIT): radix 16, with the 1..3 left-over
// stages in a first radix-8 / 4 / 2 pass (8 then 4 for five stages).
// The kernel is compiled per row length (LOGR = 9..12): block sizes, strides and trip counts are constants, LDS
// offsets immediates.
__host__ __device__ constexpr int fused_pass_m(int left) {
    return (left & 3) == 0 ? 4 : ((left & 3) == 1 && left > 1 ? 3 : (left & 3));
}
// Twiddles: every pass (block size 2^B, M stages) has its own table [k = 1 .. 2^M - 1][l < 2^(B-M)] =
// W_R^(l k 2^(LOGR-B)) in LDS, one behind the other from the first pass (B = LOGR) down; the inverse's passes are
// the forward's in reverse order and use the same tables conjugated.  Lanes read consecutive entries (the
// strided reads of a shared W_R table were 2..16-way bank conflicts: half of this kernel's LDS cycles).
__host__ __device__ constexpr int fused_tab_off(int logR, int B) {      // offset of pass B's table, in entries
    int cur = logR, acc = 0;
    while (cur > B) {
        const int M = fused_pass_m(cur - 4);
        acc += ((1 << M) - 1) << (cur - M);
        cur -= M;
    }
    return acc;
}
__host__ __device__ constexpr int fused_tab_total(int logR) { return fused_tab_off(logR, 4); }
template <int LOGR, int B>
__device__ __forceinline__ void fused_tab_build(float2* tab, const float2* __restrict__ tw) {
    if constexpr (B > 4) {
        constexpr int M = fused_pass_m(B - 4), q = 1 << (B - M), n = ((1 << M) - 1) * q;
        float2* t = tab + fused_tab_off(LOGR, B);
        for (int e = threadIdx.x; e < n; e += kGThreads) {
            const int k = e / q + 1, l = e % q;
            t[e] = tw_full(tw, (l << (LOGR - B)) * k, 1 << (LOGR - 1));
        }
        fused_tab_build<LOGR, B - M>(tab, tw);
    }
}
// The thread's 16 raw inputs of a row's FIRST pass (radix 2^M on the whole row: butterfly idx = tid + it * nthr takes
// the elements idx + m * R / 2^M) sit in registers, raw[it * 2^M + m], loaded by the caller -- all rows of a unit at
// once, one HBM latency per unit instead of one per row, in the registers that later hold the spectra.
// LDS layouts (LdsIOA above): the first forward pass writes, the one middle pass (q = 16, blocks of 2^BM) reads
// layout A; the middle pass writes and the 16-neighbour butterfly reads lp(); the inverse the other way round.
template <int LOGR>
struct FusedPlan {
    static constexpr int M0 = fused_pass_m(LOGR - 4), BM = LOGR - M0, MM = BM - 4;      // first pass, middle pass
    static_assert(MM >= 1 && MM <= 4 && fused_pass_m(BM - 4) == MM, "rows of 2^9 .. 2^12: first pass, one middle pass, 16-point blocks");
    // the unit-stride layout where its blocks line up with lp()'s (LdsIOA); rows of 512 keep lp() throughout
    using IOA = std::conditional_t<BM == 8, LdsIOA<8>, LdsIO>;
    static constexpr int buf = LdsIO::pos(1 << LOGR) > IOA::pos(1 << LOGR) ? LdsIO::pos(1 << LOGR) : IOA::pos(1 << LOGR);
};
// Twiddle providers of the helpers below: a pointer to the passes' LDS tables (fused_tab_build), or FusedTw -- the
// thread's own twiddles of the first / last pass (a) and of the middle pass (m) in registers (rows of 4096: one
// butterfly per thread and pass, so the 2 x 15 values never change; 60 LDS reads per forward + inverse pair and the
// 32 KiB of tables are gone for 60 VGPRs, which the default-plan build has to spare)
struct FusedTw {
    TwRegs a, m;
};
template <int LOGR, class TW>
__device__ __forceinline__ void dif_first_from_regs(float2* x, const TW& tab, int tid, const float2 (&raw)[16]) {
    using P = FusedPlan<LOGR>;
    constexpr int M = P::M0, RAD = 1 << M, NIT = 16 / RAD, q = 1 << (LOGR - M), nthr = 1 << (LOGR - 4);
    const typename P::IOA io{x};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * nthr;
        const auto hd = io.open(idx);
        float2 v[RAD];
#pragma unroll
        for (int m = 0; m < RAD; ++m) v[m] = raw[it * RAD + m];
        dft_reg<RAD>(v);
        hd.st(0, v[0]);
#pragma unroll
        for (int k = 1; k < RAD; ++k) {
            float2 w;
            if constexpr (std::is_same_v<TW, FusedTw>) w = tab.a.w[k - 1];
            else w = tab[(k - 1) * q + idx];
            hd.st(brev_m<M>(k) * q, g_cmul(v[k], w));
        }
    }
}
// the middle pass and the 16-point blocks, whose butterfly ends in registers
template <int LOGR, class TW>
__device__ __forceinline__ void fft_dif_rest_to_regs(float2* x, const TW& tab, int tid, float2 (&out)[16]) {
    using P = FusedPlan<LOGR>;
    const LdsIO nb{x};
    if constexpr (std::is_same_v<TW, FusedTw>)
        dif_pass<P::MM, 0, 3>(LOGR, P::BM, tab.m, tid, 1 << (LOGR - 4), typename P::IOA{x}, nb);
    else
        dif_pass<P::MM, 0, 2>(LOGR, P::BM, tab + fused_tab_off(LOGR, P::BM), tid, 1 << (LOGR - 4), typename P::IOA{x}, nb);
    xsync<true>();      // blocks of 2^BM <= 256 elements = 16 consecutive threads: inside a wave
    const auto h = nb.open(tid << 4);
#pragma unroll
    for (int m = 0; m < 16; ++m) out[m] = h.ld(m);
    dft16(out);         // out[k] belongs at position 16 tid + bitrev4(k): the inverse's first pass reads it as its input k
}
// v[k]: the (unswapped) input k of the thread's first DIT butterfly; result in LDS (natural order, layout A), no
// barrier behind the last pass
// `last`: where the last pass writes (default: the buffer, layout A)
template <int LOGR, class TW, class Last>
__device__ __forceinline__ void fft_dit_inv_from_regs(float2* x, const TW& tab, int tid, float2 (&v)[16], const Last& last) {
    using P = FusedPlan<LOGR>;
    const LdsIO nb{x};
    const typename P::IOA io{x};
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = make_float2(v[k].y, v[k].x);
    dft16(v);
    const auto h = nb.open(tid << 4);
#pragma unroll
    for (int m = 0; m < 16; ++m) h.st(m, make_float2(v[m].y, v[m].x));
    xsync<true>();                                  // neighbours -> middle pass: inside a wave
    constexpr bool one_wave = (1 << (LOGR - 4)) <= 64;      // the whole row belongs to one wave
    if constexpr (std::is_same_v<TW, FusedTw>) {
        dit_pass<P::MM, 0, 3>(LOGR, P::BM, tab.m, tid, 1 << (LOGR - 4), nb, io);
        xsync<one_wave>();
        dit_pass<P::M0, 0, 3>(LOGR, LOGR, tab.a, tid, 1 << (LOGR - 4), io, last);
    } else {
        dit_pass<P::MM, 0, 2>(LOGR, P::BM, tab + fused_tab_off(LOGR, P::BM), tid, 1 << (LOGR - 4), nb, io);
        xsync<one_wave>();
        dit_pass<P::M0, 0, 2>(LOGR, LOGR, tab, tid, 1 << (LOGR - 4), io, last);
    }
}
template <int LOGR, class TW>
__device__ __forceinline__ void fft_dit_inv_from_regs(float2* x, const TW& tab, int tid, float2 (&v)[16]) {
    fft_dit_inv_from_regs<LOGR>(x, tab, tid, v, typename FusedPlan<LOGR>::IOA{x});
}
template <int NB>
__device__ __forceinline__ void constexpr_pair(int ij, const float2 (&S)[NB][16], float2 (&v)[16]) {
    const int i = ij / NB, j = ij % NB;       // constant after unrolling
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = g_cmulc(S[j][k], S[i][k]);
}
// rows of 4096, default plan, at most 3 buoys: the passes' twiddles live in registers (FusedTw)
__host__ __device__ constexpr bool fused_tw_regs(int nb, int logR, bool def) {
#ifdef RMX_FUSED_NO_TWREG
    return false;
#else
    return def && logR == 12 && nb <= 3;
#endif
}
template <int NB, int LOGR, bool DEF = false>      // DEF: the default plan (all pairs i < j in order), pair loop unrolled
__global__ __launch_bounds__(kGThreads, 2) void g_rows_fused(const float2* __restrict__ cols, float2* __restrict__ prod,
                                                          const float2* __restrict__ tw, int n_rows, int row_bits,
                                                          long Ltot, int lo_bits, const float2* __restrict__ thi,
                                                          const float2* __restrict__ tlo, float scale, long n_units,
                                                          const GPair* __restrict__ pairs, int n_pairs) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    constexpr int R = 1 << LOGR, tpr = R >> 4, upw = kGThreads / tpr;   // threads per row, (window, rib) units per workgroup
    constexpr bool one_wave = tpr <= 64;                                // a row's threads sit in one wave: no workgroup barriers (xsync)
    using P = FusedPlan<LOGR>;
    constexpr int M0 = P::M0;                                           // stages of the forward rows' first pass
    constexpr int a = LOGR >> 1, n1 = 1 << a, n2 = R >> a;
    const int g = threadIdx.x / tpr, tid = threadIdx.x % tpr;
    float2* x = reinterpret_cast<float2*>(gsm) + g * P::buf;
    float2* t1 = reinterpret_cast<float2*>(gsm) + upw * P::buf + g * (n1 + n2);
    float2* t2 = t1 + n1;
    constexpr bool TWREG = fused_tw_regs(NB, LOGR, DEF);
    using TwSrc = std::conditional_t<TWREG, FusedTw, const float2*>;
    TwSrc twl;
    if constexpr (TWREG) {
        static_assert(P::M0 == 4 && P::MM == 4 && tpr == kGThreads, "one radix-16 butterfly per thread and pass");
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            twl.a.w[k - 1] = tw_full(tw, tid * k, R >> 1);                                   // first / last pass: l = tid
            twl.m.w[k - 1] = tw_full(tw, ((tid & 15) << (LOGR - P::BM)) * k, R >> 1);        // middle pass: l = tid & 15
        }
    } else {
        float2* tl_ = reinterpret_cast<float2*>(gsm) + upw * (P::buf + n1 + n2);
        fused_tab_build<LOGR, LOGR>(tl_, tw);
        twl = tl_;
    }
    __syncthreads();
    // One block of upw units per workgroup.  (While the kernel still had a run-time pair loop a persistent workgroup --
    // tables built once -- was 3 % faster; with the pair loop unrolled and the 4096-point twiddles in registers the
    // straight-line kernel is as fast or up to 4 % faster.)
    {   const long blk = blockIdx.x;
        // the block's units are rows rib0 .. rib0 + upw - 1 of one window (upw divides n_rows); addresses below are a
        // workgroup-uniform base (SGPRs) plus a 32-bit offset per thread
        const long unit0 = blk * upw;
        const bool live = unit0 + g < n_units;
        const int rib0 = (int)(unit0 & (n_rows - 1)), rib = rib0 + g;
        const long wl = unit0 >> row_bits;
        const unsigned goff = (unsigned)g << LOGR;
        float2 S[NB][16];
        {   // every row's first-pass inputs, all loads in flight together
            constexpr int RAD = 1 << M0, qq = R >> M0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float2* row = cols + (((long)wl * NB + b) * n_rows + rib0) * R;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    S[b][e] = live ? row[goff + (unsigned)(tid + (e / RAD) * tpr + (e % RAD) * qq)] : make_float2(0.f, 0.f);
            }
        }
        {   // W_L^(c*e), c = this row's multiplier: T1[e & (2^a - 1)] * T2[e >> a] (as g_rows)
            const long c = (long)brev(rib, row_bits);
            for (int e = tid; e < n1 + n2; e += tpr) {
                const long ee = e < n1 ? (long)e : ((long)(e - n1) << a);
                t1[e] = big_tw((c * ee) & (Ltot - 1), lo_bits, thi, tlo);
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int tl = tid;                         // (opaque copy: keeps the passes' address arithmetic inside the loop)
            asm volatile("" : "+v"(tl));
            dif_first_from_regs<LOGR>(x, twl, tl, S[b]);
            xsync<one_wave>();
            fft_dif_rest_to_regs<LOGR>(x, twl, tl, S[b]);
            __builtin_amdgcn_sched_barrier(0);    // (the last butterfly is not to be interleaved with the next first pass)
            xsync<one_wave>();                    // the next transform's first pass overwrites x
        }
        // inverse of the product in v (input k of the first butterfly) and the twiddled store to pair slot q
        auto finish = [&](float2 (&v)[16], int q) {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            fft_dit_inv_from_regs<LOGR>(x, twl, tl, v);
            xsync<true>();                        // (the store loop reads back the thread's own outputs of the last pass: n = tid mod R/16)
            if (live) {
                float2* row = prod + (((long)wl * n_pairs + q) * n_rows + rib0) * R;
#pragma unroll 4
                for (int n = tl; n < R; n += tpr) {
                    const float2 w = g_cmul(t1[n & (n1 - 1)], t2[n >> a]);
                    const float2 r = g_cmulc(x[P::IOA::pos(n)], w);
                    row[goff + (unsigned)n] = make_float2(r.x * scale, r.y * scale);
                }
            }
            xsync<one_wave>();                    // x (and, behind the last pair, t1) are rewritten
        };
        if constexpr (DEF) {
            // the default plan, (i, j) with i < j, i-major: the pair loop unrolled, every product's registers known at
            // compile time (no branch chain; a buoy's spectrum registers are dead after its last pair)
            int q = 0;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = i + 1; j < NB; ++j) {
                    float2 v[16];
                    constexpr_pair<NB>(i * NB + j, S, v);
                    finish(v, q++);
                }
        } else {
            for (int q = 0; q < n_pairs; ++q) {
                const GPair pr = pairs[q];
                float2 v[16];
                // the pair's two spectra: register arrays cannot be indexed at run time, so one (workgroup-uniform)
                // branch per ordered pair
                const int code = pr.i * NB + pr.j;
                constexpr_pair<NB>(0, S, v);      // (defined on every path: no value carried around the pair loop)
#pragma unroll
                for (int ij = 1; ij < NB * NB; ++ij)
                    if (code == ij) {
                        asm volatile("" ::: "memory");    // keeps the branch: hipcc otherwise computes all NB^2 products and selects
                        constexpr_pair<NB>(ij, S, v);
                    }
                finish(v, q);
            }
        }
    }
}

// ---- whole windows in one kernel (few buoys, 512 <= L <= 4096) -------------------------------------------------
// The two-kernel LDS path (g_fwd_small, g_pair_small) writes every spectrum to HBM and reads two of them back per
// pair: (24 B + 32 P) N bytes per window against 8 B N of input -- at 4096 windows of 2048 samples the spectra
// (400 MB for 3 buoys) do not even stay in the memory-side cache.  With at most four buoys a window's spectra fit the
// registers of the R / 16 threads that transform it (g_rows_fused's blocks, the "row" being the whole zero-padded
// window): the kernel reads the samples once (the zero half never: literal zeros in the first butterfly), forms every
// pair's product in registers, runs the inverse in LDS and folds the peak scan into the inverse's last pass, which
// leaves only |r|^2 behind for the two neighbour taps.  HBM traffic: the input and 12 bytes per pair.
// 256 / (R / 16) windows per workgroup.  sv / sk: two words per wave for the cross-wave step of the argmax.
template <int TPR>
__device__ __forceinline__ void group_argmax(float& v, int& k, float* sv, int* sk, int gtid, int g) {
    constexpr int W = TPR < 64 ? TPR : 64;
#pragma unroll
    for (int off = W >> 1; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int ok = __shfl_xor(k, off, 64);
        if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
    }
    if constexpr (TPR > 64) {
        constexpr int NW = TPR / 64;
        if ((gtid & 63) == 0) { sv[g * NW + (gtid >> 6)] = v; sk[g * NW + (gtid >> 6)] = k; }
        __syncthreads();
        v = sv[g * NW];
        k = sk[g * NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float ov = sv[g * NW + w];
            const int ok = sk[g * NW + w];
            if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
        }
    } else {
        wave_lds_order();       // the caller's tap reads follow the other lanes' |r|^2 stores
    }
}
__host__ __device__ constexpr int default_pair_code(int nb, int q) {     // i * nb + j of the q-th pair i < j, i-major
    int i = 0;
    while (q >= nb - 1 - i) { q -= nb - 1 - i; ++i; }
    return i * nb + (i + 1 + q);
}
template <class F, int... Q>
__device__ __forceinline__ void for_each_q(F&& f, std::integer_sequence<int, Q...>) { (f(std::integral_constant<int, Q>{}), ...); }
template <int NB, class F>
__device__ __forceinline__ void for_each_default_pair(F&& f) { for_each_q(f, std::make_integer_sequence<int, NB * (NB - 1) / 2>{}); }
template <int NB, int LOGR, bool DEF, bool U8>
__global__ __launch_bounds__(kGThreads, 2) void g_win_fused(const void* __restrict__ iq, const float2* __restrict__ tw,
                                                         long n_windows, long first_window, float fwd_scale,
                                                         float out_scale, const GPair* __restrict__ pairs, int n_pairs,
                                                         int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                                         float* __restrict__ peak) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    constexpr int R = 1 << LOGR, N = R >> 1, tpr = R >> 4, upw = kGThreads / tpr;   // threads per window, windows per workgroup
    constexpr bool one_wave = tpr <= 64;                                            // a window's threads sit in one wave
    using P = FusedPlan<LOGR>;
    constexpr int M0 = P::M0;
    const int g = threadIdx.x / tpr, tid = threadIdx.x % tpr;
    float2* x = reinterpret_cast<float2*>(gsm) + g * P::buf;
    constexpr bool TWREG = fused_tw_regs(NB, LOGR, DEF);
    using TwSrc = std::conditional_t<TWREG, FusedTw, const float2*>;
    TwSrc twl;
    float* sv;
    if constexpr (TWREG) {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            twl.a.w[k - 1] = tw_full(tw, tid * k, R >> 1);
            twl.m.w[k - 1] = tw_full(tw, ((tid & 15) << (LOGR - P::BM)) * k, R >> 1);
        }
        sv = reinterpret_cast<float*>(reinterpret_cast<float2*>(gsm) + upw * P::buf);
    } else {
        float2* tl_ = reinterpret_cast<float2*>(gsm) + upw * P::buf;
        fused_tab_build<LOGR, LOGR>(tl_, tw);
        twl = tl_;
        sv = reinterpret_cast<float*>(tl_ + fused_tab_total(LOGR));
    }
    int* sk = reinterpret_cast<int*>(sv + 8);
    __syncthreads();
    const long w = (long)blockIdx.x * upw + g;
    const bool live = w < n_windows;
    float2 S[NB][16];
    {   // every buoy's first-pass inputs, all loads in flight together; the zero-padded half is never read
        constexpr int RAD = 1 << M0, qq = R >> M0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const long base = (w * NB + b) * (long)N;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if ((e % RAD) < RAD / 2 && live) {
                    const long n = base + (tid + (e / RAD) * tpr + (e % RAD) * qq);
                    if constexpr (U8) {
                        const uchar2 b8 = reinterpret_cast<const uchar2*>(iq)[n];
                        S[b][e] = make_float2((float)b8.x - 127.5f, (float)b8.y - 127.5f);
                    } else {
                        S[b][e] = reinterpret_cast<const float2*>(iq)[n];
                    }
                } else {
                    S[b][e] = make_float2(0.f, 0.f);
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int tl = tid;                         // (opaque copy: keeps the passes' address arithmetic inside the loop)
        asm volatile("" : "+v"(tl));
        dif_first_from_regs<LOGR>(x, twl, tl, S[b]);
        xsync<one_wave>();
        fft_dif_rest_to_regs<LOGR>(x, twl, tl, S[b]);
#pragma unroll
        for (int k = 0; k < 16; ++k) S[b][k] = make_float2(S[b][k].x * fwd_scale, S[b][k].y * fwd_scale);
        __builtin_amdgcn_sched_barrier(0);
        xsync<one_wave>();                    // the next transform's first pass overwrites x
    }
    const long obase = (first_window + w) * (long)n_pairs;
    // inverse of the product in v; the last pass keeps (max |r|^2, lowest 'full' index) per thread and leaves |r|^2
    auto finish = [&](float2 (&v)[16], int q) __attribute__((always_inline)) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
        float best = -1.0f;
        int bk = 0x7fffffff;
        fft_dit_inv_from_regs<LOGR>(x, twl, tl, v, make_dst([&](int E0, int off, float2 e) __attribute__((always_inline)) {
            const int n = E0 + off;
            const float m2 = e.x * e.x + e.y * e.y;
            scan_update(best, bk, m2, n, N);
            x[P::IOA::pos(n)].x = m2;
        }));
        group_argmax<tpr>(best, bk, sv, sk, tl, g);       // (its barrier, or wave order, also publishes the |r|^2 for the taps)
        if (tl == 0 && live) {
            const float b = sqrtf(best) * out_scale;
            float frac = 0.0f;
            if (bk > 0 && bk < 2 * N - 2) {
                const float ra = x[P::IOA::pos(circ_index(bk - 1, N))].x, rc = x[P::IOA::pos(circ_index(bk + 1, N))].x;
                frac = parabola(sqrtf(ra) * out_scale, b, sqrtf(rc) * out_scale);
            }
            lag_int[obase + q] = bk - (N - 1);
            lag_frac[obase + q] = frac;
            peak[obase + q] = b;
        }
        xsync<one_wave>();                    // x is rewritten by the next pair (sv: behind the next inverse's barrier)
    };
    if constexpr (DEF) {
        // the default plan, pair q = (i, j) with i < j, i-major: unrolled by pack expansion (a "#pragma unroll" loop nest
        // of this size is left rolled by hipcc at 4 buoys x 4096 points, and the spectra then live in scratch memory)
        for_each_default_pair<NB>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            float2 v[16];
            constexpr int code = default_pair_code(NB, q);
            constexpr_pair<NB>(code, S, v);
            finish(v, q);
        });
    } else {
        for (int q = 0; q < n_pairs; ++q) {
            const GPair pr = pairs[q];
            float2 v[16];
            const int code = pr.i * NB + pr.j;
            constexpr_pair<NB>(0, S, v);
#pragma unroll
            for (int ij = 1; ij < NB * NB; ++ij)
                if (code == ij) {
                    asm volatile("" ::: "memory");
                    constexpr_pair<NB>(ij, S, v);
                }
            finish(v, q);
        }
    }
}

// ---- whole windows in one kernel, any number of buoys, 512 <= L <= 16384 (N = 8192: the reference's capture length) --
// The same idea with the spectra in a per-workgroup scratch instead of registers: a persistent workgroup (R / 16
// threads per window, one radix-16 butterfly per thread and pass) transforms the window's buoys one after the other
// and parks each spectrum -- in the register order of its last butterfly, 16 bytes per lane, coalesced -- in its own
// n_buoys x 8 R bytes of global memory, which it overwrites window after window: the scratch of the whole grid stays
// in the L2 / memory-side cache, HBM sees the input.  Every pair then loads X_j back (each thread exactly the values it
// stored: no visibility protocol; the anchor X_i stays in registers while consecutive pairs share it), multiplies in
// registers, and runs the inverse with the peak search behind its last pass.  The first pass's twiddles live in registers
// (one butterfly per thread: they never change), the others' in small LDS tables.  L = 16384 (one 136 KiB transform per
// CU) runs g_win_scr14 below -- 512 threads x two butterflies -- and this kernel's 1024-thread build only on request.
// Build knobs (A/B experiments): RMX_WS_TWREG_FROM / RMX_WS_TWG_FROM (first-pass twiddles in
// registers / from the global table from that log2 L on), RMX_WS_TW2REG (second pass's in registers too).
// Pass order: radix 16 from the whole window down (DIF) while more than four stages remain, the left-over 1..4
// stages last, on the thread's 16 neighbouring elements: they end in registers, and the inverse starts there.
template <int LOGR>
struct WinPlan {
    static constexpr int R = 1 << LOGR, tpr = R >> 4;
    static constexpr int ML = ((LOGR - 1) & 3) + 1, RADL = 1 << ML, NITL = 16 / RADL;   // the neighbour pass
    static constexpr int NP = (LOGR - ML) / 4;                                            // radix-16 passes, blocks LOGR, LOGR - 4, ..
#ifndef RMX_WS_TWG_FROM
#define RMX_WS_TWG_FROM 99
#endif
#ifndef RMX_WS_TWREG_FROM
#define RMX_WS_TWREG_FROM 9
#endif
    static constexpr bool TW1REG = LOGR >= RMX_WS_TWREG_FROM && LOGR < RMX_WS_TWG_FROM;                  // first pass's twiddles in registers
    static constexpr bool TW1GLOBAL = LOGR >= RMX_WS_TWG_FROM;                            // ... or read from the global table
#ifndef RMX_WS_TW2REG
#define RMX_WS_TW2REG 0
#endif
    static constexpr bool TW2REG = RMX_WS_TW2REG && NP >= 2 && tpr < 1024;               // the second pass's too
    static constexpr int thr = tpr < kGThreads ? kGThreads : tpr, upw = thr / tpr;         // threads, windows per workgroup
    static constexpr int tab_off(int b) {                                                 // entries in front of pass b's LDS table
        int acc = 0;
        for (int c = LOGR; c > b; c -= 4)
            if (!(c == LOGR && (TW1REG || TW1GLOBAL)) && !(c == LOGR - 4 && TW2REG)) acc += 15 << (c - 4);
        return acc;
    }
    static constexpr int tab_total = tab_off(ML);
    static constexpr size_t lds_bytes = ((size_t)upw * lp(R) + tab_total + 16) * 8;
};
// The transforms of those kernels: one window's R / 16 threads, buffer x (lp() layout), the passes' twiddles
template <int LOGR>
struct WinXf {
    using P = WinPlan<LOGR>;
    static constexpr int R = P::R, tpr = P::tpr, ML = P::ML, RADL = P::RADL, NITL = P::NITL;
    // the exchange on the small side of the radix-16 pass over blocks of 2^b is wave-local when those blocks' 2^(b-4)
    // threads fit a wave (b <= 10): at L = 16384 only the pass over the whole window needs workgroup barriers
    static constexpr bool one_wave = tpr <= 64;
    float2* x;
    float2* tab;
    const float2* __restrict__ tw;
    TwRegs tw1, tw2;
    int tid;
    // tables [k = 1..15][l < q]: W_R^(l k 2^(LOGR - b)), consecutive lanes read consecutive entries; all threads of the
    // workgroup; the caller's barrier follows
    __device__ __forceinline__ void setup() {
        auto build = [&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value, q = 1 << (b - 4), off = P::tab_off(b);
            for (int e = threadIdx.x; e < 15 * q; e += P::thr)
                tab[off + e] = tw_full(tw, ((e & (q - 1)) << (LOGR - b)) * ((e >> (b - 4)) + 1), R >> 1);
        };
        if constexpr (!P::TW1REG && !P::TW1GLOBAL) build(std::integral_constant<int, LOGR>{});
        if constexpr (P::NP >= 2 && !P::TW2REG) build(std::integral_constant<int, LOGR - 4>{});
        if constexpr (P::NP >= 3) build(std::integral_constant<int, LOGR - 8>{});
        if constexpr (P::TW1REG) {
#pragma unroll
            for (int k = 1; k < 16; ++k) tw1.w[k - 1] = tw_full(tw, tid * k, R >> 1);
        }
        if constexpr (P::TW2REG) {
            constexpr int b = LOGR - 4, q = 1 << (b - 4);
#pragma unroll
            for (int k = 1; k < 16; ++k) tw2.w[k - 1] = tw_full(tw, ((tid & (q - 1)) << (LOGR - b)) * k, R >> 1);
        }
    }
    // forward: v[m] = element tid + m R/16 (m >= 8: the zero-padded half, literal zeros) -> the spectrum in the register
    // order of the last butterflies: v[it 2^ML + k] is the bin at position 16 tid + it 2^ML + bitrev_ML(k) of the
    // bit-reversed spectrum.  Ends with the exchange that lets the next transform overwrite x.
    __device__ __forceinline__ void fwd(float2 (&v)[16]) const {
        const LdsIO lds{x};
        {   // first pass: the whole window
            dft16(v);
            const auto hd = lds.open(tid);
            hd.st(0, v[0]);
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                float2 wk;
                if constexpr (P::TW1REG) wk = tw1.w[k - 1];
                else if constexpr (P::TW1GLOBAL) wk = tw_full(tw, tid * k, R >> 1);
                else wk = tab[(k - 1) * tpr + tid];
                hd.st(brev_m<4>(k) * tpr, g_cmul(v[k], wk));
            }
        }
        xsync<one_wave>();
        if constexpr (P::NP >= 2) {
            constexpr int bb = LOGR - 4, off = P::tab_off(bb);
            if constexpr (P::TW2REG) dif_pass<4, 0, 3>(LOGR, bb, tw2, tid, tpr, lds, lds);
            else dif_pass<4, 0, 2>(LOGR, bb, tab + off, tid, tpr, lds, lds);
            xsync<(bb <= 10)>();
        }
        if constexpr (P::NP >= 3) {
            constexpr int bb = LOGR - 8, off = P::tab_off(bb);
            dif_pass<4, 0, 2>(LOGR, bb, tab + off, tid, tpr, lds, lds);
            xsync<(bb <= 10)>();
        }
        {   // the thread's 16 neighbours: 16 / 2^ML butterflies, outputs stay in registers (butterfly order)
            const auto h = lds.open(tid << 4);
#pragma unroll
            for (int it = 0; it < NITL; ++it) {
                float2 t[RADL];
#pragma unroll
                for (int m = 0; m < RADL; ++m) t[m] = h.ld(it * RADL + m);
                dft_reg<RADL>(t);
#pragma unroll
                for (int m = 0; m < RADL; ++m) v[it * RADL + m] = t[m];
            }
        }
        xsync<one_wave>();                    // the next transform's first pass overwrites x
    }
    // inverse (unnormalised) of the product in v (same register order); the last pass hands every output (natural index
    // n) to `scan`
    template <class Scan>
    __device__ __forceinline__ void inv(float2 (&v)[16], const Scan& scan) const {
        const LdsIO lds{x};
        {   // first pass: the same neighbour butterflies ((im, re)-swapped data: swap o DFT o swap = conj DFT)
            const auto h = lds.open(tid << 4);
#pragma unroll
            for (int it = 0; it < NITL; ++it) {
                float2 t[RADL];
#pragma unroll
                for (int k = 0; k < RADL; ++k) t[k] = make_float2(v[it * RADL + k].y, v[it * RADL + k].x);
                dft_reg<RADL>(t);
#pragma unroll
                for (int m = 0; m < RADL; ++m) h.st(it * RADL + m, make_float2(t[m].y, t[m].x));
            }
        }
        if constexpr (P::NP >= 3) {
            constexpr int bb = LOGR - 8, off = P::tab_off(bb);
            xsync<(bb <= 10)>();
            dit_pass<4, 0, 2>(LOGR, bb, tab + off, tid, tpr, lds, lds);
        }
        if constexpr (P::NP >= 2) {
            constexpr int bb = LOGR - 4, off = P::tab_off(bb);
            xsync<(bb <= 10)>();
            if constexpr (P::TW2REG) dit_pass<4, 0, 3>(LOGR, bb, tw2, tid, tpr, lds, lds);
            else dit_pass<4, 0, 2>(LOGR, bb, tab + off, tid, tpr, lds, lds);
        }
        xsync<one_wave>();                    // the pass over the whole window reads every wave's blocks
        if constexpr (P::TW1REG) dit_pass<4, 0, 3>(LOGR, LOGR, tw1, tid, tpr, lds, scan);
        else if constexpr (P::TW1GLOBAL) dit_pass<4, 0, 0>(LOGR, LOGR, tw, tid, tpr, lds, scan);
        else dit_pass<4, 0, 2>(LOGR, LOGR, tab, tid, tpr, lds, scan);
    }
};
template <int LOGR, bool U8>
__global__ __launch_bounds__(WinPlan<LOGR>::thr) void g_win_scr(const void* __restrict__ iq, float4* __restrict__ scratch,
                                                             const float2* __restrict__ tw, int n_buoys, long n_windows,
                                                             long first_window, float fwd_scale, float out_scale,
                                                             const GPair* __restrict__ pairs, int n_pairs,
                                                             int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                                             float* __restrict__ peak) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    using P = WinPlan<LOGR>;
    constexpr int R = P::R, N = R >> 1, tpr = P::tpr, upw = P::upw;
    constexpr bool one_wave = tpr <= 64;
    const int g = threadIdx.x / tpr, tid = threadIdx.x % tpr;
    float2* x = reinterpret_cast<float2*>(gsm) + g * lp(R);
    float2* tab = reinterpret_cast<float2*>(gsm) + upw * lp(R);
    float* sv = reinterpret_cast<float*>(tab + P::tab_total);
    int* sk = reinterpret_cast<int*>(sv + 16);
    WinXf<LOGR> xf{x, tab, tw, {}, {}, tid};
    xf.setup();
    __syncthreads();
    float4* scr = scratch + ((long)blockIdx.x * upw + g) * n_buoys * (8L * tpr) + tid;
    for (long blk = blockIdx.x; blk * upw < n_windows; blk += gridDim.x) {
        const long w = blk * upw + g;
        const bool live = w < n_windows;
        auto load_in = [&](float2 (&d)[8], int b) __attribute__((always_inline)) {
            const long base = (w * n_buoys + b) * (long)N + tid;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (!live) { d[m] = make_float2(0.f, 0.f); continue; }
                if constexpr (U8) {
                    const uchar2 b8 = reinterpret_cast<const uchar2*>(iq)[base + m * tpr];
                    d[m] = make_float2((float)b8.x - 127.5f, (float)b8.y - 127.5f);
                } else {
                    d[m] = reinterpret_cast<const float2*>(iq)[base + m * tpr];
                }
            }
        };
        float2 nx[8];
        load_in(nx, 0);
        for (int b = 0; b < n_buoys; ++b) {
            float2 v[16];
#pragma unroll
            for (int m = 0; m < 8; ++m) { v[m] = nx[m]; v[m + 8] = make_float2(0.f, 0.f); }   // the zero-padded half: literal zeros
            if (b + 1 < n_buoys) load_in(nx, b + 1);                                           // travels during this transform
            xf.fwd(v);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                scr[(long)(b * 8 + kk) * tpr] = make_float4(v[2 * kk].x * fwd_scale, v[2 * kk].y * fwd_scale,
                                                             v[2 * kk + 1].x * fwd_scale, v[2 * kk + 1].y * fwd_scale);
        }
        const long obase = (first_window + w) * (long)n_pairs;
        // the anchor X_i stays in registers while consecutive pairs share it (the default list is i-major: B - 1 anchor
        // loads and P loads of X_j per window instead of 2 P loads); not at L = 16384, whose 1024 threads have 128 VGPRs
        // (measured there: 39 spilled registers, 8 buoys x 512 windows of 8192 0.81 -> 1.07 ms)
        constexpr bool kAnchor = P::thr < 1024;
        float4 anc[8];
        int anc_i = -1;
        for (int q = 0; q < n_pairs; ++q) {
            const GPair pr = pairs[q];
            float2 v[16];
            if constexpr (kAnchor) {
                if (pr.i != anc_i) {              // (workgroup-uniform)
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) anc[kk] = scr[(long)(pr.i * 8 + kk) * tpr];
                    anc_i = pr.i;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {      // X_j conj(X_i), element by element of the thread's own 16
                const float4 a = scr[(long)(pr.j * 8 + kk) * tpr];
                float4 c;
                if constexpr (kAnchor) c = anc[kk];
                else c = scr[(long)(pr.i * 8 + kk) * tpr];
                v[2 * kk] = g_cmulc(make_float2(a.x, a.y), make_float2(c.x, c.y));
                v[2 * kk + 1] = g_cmulc(make_float2(a.z, a.w), make_float2(c.z, c.w));
            }
            // the last pass hands thread tid its outputs n = tid + s R/16, s = 0 .. 15: |r|^2 into mg[s] and into the buffer (taps)
            float mg[16];
            xf.inv(v, make_dst([&](int E0, int off, float2 e) __attribute__((always_inline)) {
                const float m2 = e.x * e.x + e.y * e.y;
                mg[off / tpr] = m2;                           // (off = s R/16 is a constant once the pass is unrolled)
                x[lp(E0 + off)].x = m2;
            }));
            // the thread's (max, lowest 'full' index): 'full' index n - N - 1 for s >= 8 (lag -N itself, n = N: thread 0, s = 8,
            // is not part of the output), n + N - 1 for s < 8 -- ascending in the order s = 8 .. 15, 0 .. 7
            if (tid == 0) mg[8] = -2.0f;
            float best = fmaxf(fmaxf(fmaxf(fmaxf(mg[0], mg[1]), fmaxf(mg[2], mg[3])), fmaxf(fmaxf(mg[4], mg[5]), fmaxf(mg[6], mg[7]))),
                               fmaxf(fmaxf(fmaxf(mg[8], mg[9]), fmaxf(mg[10], mg[11])), fmaxf(fmaxf(mg[12], mg[13]), fmaxf(mg[14], mg[15]))));
            int ssel = 7;                                     // lowest-index slot holding the max: later assignments win
#pragma unroll
            for (int s = 6; s >= 0; --s) ssel = mg[s] == best ? s : ssel;
#pragma unroll
            for (int s = 15; s >= 8; --s) ssel = mg[s] == best ? s : ssel;
            int bk = tid + (ssel & 7) * tpr + (ssel >= 8 ? -1 : N - 1);
            group_argmax<tpr>(best, bk, sv, sk, tid, g);      // (its barrier, or wave order, also publishes the |r|^2 for the taps)
            if (tid == 0 && live) {
                const float bpk = sqrtf(best) * out_scale;
                float frac = 0.0f;
                if (bk > 0 && bk < 2 * N - 2) {
                    const float ra = x[lp(circ_index(bk - 1, N))].x, rc = x[lp(circ_index(bk + 1, N))].x;
                    frac = parabola(sqrtf(ra) * out_scale, bpk, sqrtf(rc) * out_scale);
                }
                lag_int[obase + q] = bk - (N - 1);
                lag_frac[obase + q] = frac;
                peak[obase + q] = bpk;
            }
            xsync<one_wave>();                    // x is rewritten (sv: behind the next inverse's barrier)
        }
    }
}

// L = 16384 (N = 8192) with 512 threads and TWO butterflies per thread and pass (indices tid and tid + 512): two waves per
// SIMD with 256 VGPRs each instead of four with 128, which is what the anchor spectrum (64 registers here) and the
// register twiddles of the pass over the whole window (2 x 30) need -- g_win_scr<14> at 1024 threads has neither and spills.
// Same passes (16 x 16 x 16 x 4), same LDS image, same scratch idea: [b][u][8][512] float4 per window slot.
template <bool U8>
__global__ __launch_bounds__(512, 2) void g_win_scr14(const void* __restrict__ iq, float4* __restrict__ scratch,
                                                   const float2* __restrict__ tw, int n_buoys, long n_windows,
                                                   long first_window, float fwd_scale, float out_scale,
                                                   const GPair* __restrict__ pairs, int n_pairs,
                                                   int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                                   float* __restrict__ peak) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    constexpr int LOGR = 14, R = 1 << LOGR, N = R >> 1, NT = 512, Q = R >> 4;      // Q = 1024 butterflies per radix-16 pass
    constexpr int off10 = 0, off6 = 15 * 64, tab_total = 15 * 64 + 15 * 4;          // LDS tables of the passes over blocks of 2^10, 2^6
    const int tid = threadIdx.x;
    float2* x = reinterpret_cast<float2*>(gsm);
    float2* tab = x + lp(R);
    float* sv = reinterpret_cast<float*>(tab + tab_total);
    int* sk = reinterpret_cast<int*>(sv + 16);
    for (int e = tid; e < 15 * 64; e += NT) tab[off10 + e] = tw_full(tw, ((e & 63) << 4) * ((e >> 6) + 1), R >> 1);
    for (int e = tid; e < 15 * 4; e += NT) tab[off6 + e] = tw_full(tw, ((e & 3) << 8) * ((e >> 2) + 1), R >> 1);
    TwRegs tw1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 1; k < 16; ++k) tw1[u].w[k - 1] = tw_full(tw, (tid + u * NT) * k, R >> 1);
    __syncthreads();
    const LdsIO lds{x};
    float4* scr = scratch + (long)blockIdx.x * n_buoys * (16L * NT) + tid;          // [b][u][kk][tid]
    for (long w = blockIdx.x; w < n_windows; w += gridDim.x) {
        auto load_in = [&](float2 (&d)[2][8], int b) __attribute__((always_inline)) {
            const long base = (w * n_buoys + b) * (long)N + tid;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if constexpr (U8) {
                        const uchar2 b8 = reinterpret_cast<const uchar2*>(iq)[base + u * NT + m * Q];
                        d[u][m] = make_float2((float)b8.x - 127.5f, (float)b8.y - 127.5f);
                    } else {
                        d[u][m] = reinterpret_cast<const float2*>(iq)[base + u * NT + m * Q];
                    }
                }
        };
        float2 nx[2][8];
        load_in(nx, 0);
        for (int b = 0; b < n_buoys; ++b) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {         // first pass: the whole window, elements i + m R/16, i = tid + 512 u
                float2 v[16];
#pragma unroll
                for (int m = 0; m < 8; ++m) { v[m] = nx[u][m]; v[m + 8] = make_float2(0.f, 0.f); }
                dft16(v);
                const auto hd = lds.open(tid + u * NT);
                hd.st(0, v[0]);
#pragma unroll
                for (int k = 1; k < 16; ++k) hd.st(brev_m<4>(k) * Q, g_cmul(v[k], tw1[u].w[k - 1]));
            }
            if (b + 1 < n_buoys) load_in(nx, b + 1);                                // travels during the rest of this transform
            __syncthreads();
            dif_pass<4, 0, 2>(LOGR, 10, tab + off10, tid, NT, lds, lds);
            xsync<true>();
            dif_pass<4, 0, 2>(LOGR, 6, tab + off6, tid, NT, lds, lds);
            xsync<true>();
#pragma unroll
            for (int u = 0; u < 2; ++u) {         // the 16 neighbours of butterfly i: four radix-4 butterflies, outputs to the scratch
                float2 v[16];
                const auto h = lds.open((tid + u * NT) << 4);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    float2 t[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) t[m] = h.ld(it * 4 + m);
                    dft_reg<4>(t);
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[it * 4 + m] = t[m];
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    scr[(long)((b * 2 + u) * 8 + kk) * NT] = make_float4(v[2 * kk].x * fwd_scale, v[2 * kk].y * fwd_scale,
                                                                          v[2 * kk + 1].x * fwd_scale, v[2 * kk + 1].y * fwd_scale);
            }
            __syncthreads();                      // the next transform's first pass overwrites x
        }
        const long obase = (first_window + w) * (long)n_pairs;
        float4 anc[2][8];                         // the anchor X_i, while consecutive pairs share it
        int anc_i = -1;
        for (int q = 0; q < n_pairs; ++q) {
            const GPair pr = pairs[q];
            if (pr.i != anc_i) {                  // (workgroup-uniform)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) anc[u][kk] = scr[(long)((pr.i * 2 + u) * 8 + kk) * NT];
                anc_i = pr.i;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {         // X_j conj(X_i) into the neighbour butterflies of the inverse
                float2 v[16];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float4 a = scr[(long)((pr.j * 2 + u) * 8 + kk) * NT], c = anc[u][kk];
                    v[2 * kk] = g_cmulc(make_float2(a.x, a.y), make_float2(c.x, c.y));
                    v[2 * kk + 1] = g_cmulc(make_float2(a.z, a.w), make_float2(c.z, c.w));
                }
                const auto h = lds.open((tid + u * NT) << 4);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    float2 t[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = make_float2(v[it * 4 + k].y, v[it * 4 + k].x);
                    dft_reg<4>(t);
#pragma unroll
                    for (int m = 0; m < 4; ++m) h.st(it * 4 + m, make_float2(t[m].y, t[m].x));
                }
            }
            xsync<true>();
            dit_pass<4, 0, 2>(LOGR, 6, tab + off6, tid, NT, lds, lds);
            xsync<true>();
            dit_pass<4, 0, 2>(LOGR, 10, tab + off10, tid, NT, lds, lds);
            __syncthreads();                      // the pass over the whole window reads every wave's blocks
            // last pass: outputs n = tid + s 512, s = u + 2 m; |r|^2 into mg[s] and into the buffer (taps)
            float mg[32];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float2 v[16];
                const auto h = lds.open(tid + u * NT);
                {
                    const float2 e = h.ld(0);
                    v[0] = make_float2(e.y, e.x);
                }
#pragma unroll
                for (int k = 1; k < 16; ++k) {
                    const float2 e = g_cmulc(h.ld(brev_m<4>(k) * Q), tw1[u].w[k - 1]);
                    v[k] = make_float2(e.y, e.x);
                }
                dft16(v);
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const float m2 = v[m].x * v[m].x + v[m].y * v[m].y;
                    mg[u + 2 * m] = m2;
                    x[lp(tid + u * NT + m * Q)].x = m2;
                }
            }
            // 'full' index n - N - 1 for s >= 16 (lag -N itself: thread 0, s = 16, is not part of the output), n + N - 1 below:
            // ascending in the order s = 16 .. 31, 0 .. 15
            if (tid == 0) mg[16] = -2.0f;
            float best = mg[0];
#pragma unroll
            for (int s2 = 1; s2 < 32; ++s2) best = fmaxf(best, mg[s2]);
            int ssel = 15;
#pragma unroll
            for (int s2 = 14; s2 >= 0; --s2) ssel = mg[s2] == best ? s2 : ssel;
#pragma unroll
            for (int s2 = 31; s2 >= 16; --s2) ssel = mg[s2] == best ? s2 : ssel;
            int bk = tid + (ssel & 15) * NT + (ssel >= 16 ? -1 : N - 1);
            group_argmax<NT>(best, bk, sv, sk, tid, 0);      // (its barrier also publishes the |r|^2 for the taps)
            if (tid == 0) {
                const float bpk = sqrtf(best) * out_scale;
                float frac = 0.0f;
                if (bk > 0 && bk < 2 * N - 2) {
                    const float ra = x[lp(circ_index(bk - 1, N))].x, rc = x[lp(circ_index(bk + 1, N))].x;
                    frac = parabola(sqrtf(ra) * out_scale, bpk, sqrtf(rc) * out_scale);
                }
                lag_int[obase + q] = bk - (N - 1);
                lag_frac[obase + q] = frac;
                peak[obase + q] = bpk;
            }
            __syncthreads();                      // x is rewritten (sv: behind the next inverse's barrier)
        }
    }
}

// ---- column passes of the four-step (no transposes through HBM) ----------------------------------
// The L-point sequence is the row-major matrix [L1][L2], n = n1*L2 + n2.  A workgroup takes a tile of
// T = 16 adjacent columns (128-byte row segments: full cache lines; 8 when L1 = 1024) with all L1 rows into LDS
// ([L1][16], up to 128 KiB of the CU's 160 KiB), transforms the 16 columns there and writes the tile
// back in place of a transpose + row pass + transpose.
// columns per tile: 16 (128-byte row segments), or 8 when L1 = 1024 so that two 64 KiB tiles share a CU and
// one workgroup's loads and stores overlap the other's butterflies (measured: cfg2 8.7 -> 8.1 ms)
__host__ __device__ constexpr int col_log_t(int l1) { return l1 >= 10 ? 3 : 4; }
// forward: zero-padded window -> out[k1'][n2] = W_L^(n2*k1) * sum_n1 x[n1][n2] W_L1^(n1*k1)
//   (k1' = bit-reversed k1).  grid (L2/T, items); only the rows n1 < L1/2 are non-zero and read.
// Workgroup -> tile: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, so with tile =
// blockIdx.x the 128-byte row segments of neighbouring tiles -- one DRAM page -- are fetched through 8 different L2s
// at unrelated times.  Remapped, XCD k walks the tiles [k n/8, (k+1) n/8): its 64 resident workgroups read runs of
// adjacent segments of the same rows together.
__device__ __forceinline__ int xcd_tile(int bid, int n) {
#ifdef RMX_NO_XCD_REMAP
    return bid;
#else
    return (n & 7) == 0 ? (bid & 7) * (n >> 3) + (bid >> 3) : bid;
#endif
}
// threads of a column kernel: one radix-16 work item per thread and pass, 64 .. 1024
__host__ __device__ constexpr int cols_threads(int l1, int log_t) {
    return ((1 << l1) << log_t) / 16 >= 1024 ? 1024 : (((1 << l1) << log_t) / 16 < 64 ? 64 : ((1 << l1) << log_t) / 16);
}
// L1C > 0: the column length (and with it the thread count) as compile-time constants
template <bool U8, int kColLogT, int L1C = 0>
__global__ __launch_bounds__(1024) void g_cols_fwd(const void* __restrict__ iq, float2* __restrict__ out,
                                                   const float2* __restrict__ tw, int l1_arg, int l2, long first_item,
                                                   int lo_bits, const float2* __restrict__ thi,
                                                   const float2* __restrict__ tlo,
                                                   const float2* __restrict__ rot = nullptr) {
    constexpr int kColT = 1 << kColLogT;
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float2* x = reinterpret_cast<float2*>(gsm);
    const int l1 = L1C > 0 ? L1C : l1_arg;
    const int L1 = 1 << l1, L2 = 1 << l2, tid = threadIdx.x, nthr = L1C > 0 ? cols_threads(L1C, kColLogT) : (int)blockDim.x;
    const long L = (long)L1 << l2, N = L >> 1;
    const int tile = xcd_tile(blockIdx.x, gridDim.x), c0 = tile * kColT;
    const long item = first_item + blockIdx.y;
    // per-column twiddle tables W_L^(n2*e) = T1[e & (2^a-1)] * T2[e >> a] (n2*e < L: no reduction) and the W_L1 table
    const int a = l1 >> 1, na = 1 << a, nb = L1 >> a;
    float2* tab = x + lp((long)L1 << kColLogT);                 // [T][na + nb + 1]: an odd stride, or the 16 columns' reads of one
    constexpr int kTabPad = 1;                                //   entry sit on two banks (8-way conflict)
    float2* twl = tab + (long)kColT * (na + nb + kTabPad);
    for (int k = tid; k < (L1 >> 1); k += nthr) twl[k] = tw[k];
    for (int idx = tid; idx < kColT * (na + nb); idx += nthr) {
        const int c = idx / (na + nb), e = idx % (na + nb);
        const long ee = e < na ? (long)e : ((long)(e - na) << a);
        tab[c * (na + nb + kTabPad) + e] = big_tw((long)(c0 + c) * ee, lo_bits, thi, tlo);
    }
    // (no barrier here on the staged path: the tile's loads below go out behind the tables' and both latencies overlap;
    // the barrier behind the staging covers the tables too)
    if constexpr (L1C > 0 && L1C <= 6) __syncthreads();
    // (first pass from HBM / last pass to HBM measured slower here: 2.0 vs 1.58 ms on cfg2 -- the sink's table
    // lookups on top of a radix-16 pass spill)
    const int nz = (L1 >> 1) << kColLogT;                     // only the rows n1 < L1/2 are non-zero
    auto sample = [&](int idx) -> float2 {                    // tile element idx = (row << LOGT) | column, from HBM
        const int n = ((idx >> kColLogT) << l2) + c0 + (idx & (kColT - 1));   // < N: 32-bit
        float2 v;
        if constexpr (U8) {
            const uchar2 b8 = reinterpret_cast<const uchar2*>(iq)[item * N + n];
            v = make_float2((float)b8.x - 127.5f, (float)b8.y - 127.5f);
        } else {
            v = reinterpret_cast<const float2*>(iq)[item * N + n];
        }
        return rot ? rot_mul(v, rot[n]) : v;
    };
    // short columns (L1 <= 64: N = 8192 0.49 -> 0.46 ms) hand the first pass its inputs straight from HBM; longer ones
    // stage the tile in LDS first (cfg2 3.60 vs 3.66 ms, 8 buoys x 2^18 0.91 vs 0.94 ms)
    if constexpr (!(L1C > 0 && L1C <= 6)) {
    batched<8>(tid, nz, nthr, sample, [&](int idx, float2 v) { x[TileFwd::pos(idx)] = v; });
    __syncthreads();
    {
        const TileFwd io{x};
        fft_dif<kColLogT>(x, l1, twl, tid, nthr, SrcZeroTail<TileFwd>{io, nz}, io);
        __syncthreads();
    }
    } else {   // the first pass reads its (eight non-zero of sixteen) inputs straight from HBM: rows L1/16 apart of one
        // column per thread, consecutive threads on consecutive columns -- the same 128-byte row segments a tile load
        // fetches, without the tile's trip through LDS
        const TileFwd io{x};
        const auto hbm = make_src(sample);
        fft_dif<kColLogT>(x, l1, twl, tid, nthr, SrcZeroTail<decltype(hbm)>{hbm, nz}, io);
        __syncthreads();
    }
    float2* o = out + (long)blockIdx.y * L;
    if constexpr (L1C >= 8 && kColLogT == 4) {
        // compile-time column length: thread tid stores elements tid + k nthr, k < 16 -- always column c = tid & 15, rows
        // plo + k 2^pb.  The bit-reversed row index k1 then splits into bitrev(k) (its low four bits: the T1 entry, an
        // immediate offset) and bitrev(plo) (the T2 entry: one read per thread instead of one per element)
        constexpr int pb = L1C - 4;
        static_assert((cols_threads(L1C, 4) >> 4) == (1 << pb) && (L1C >> 1) == 4, "16 elements per thread, a = 4");
        const int c = tid & 15, plo = tid >> 4;
        const float2* tc = tab + c * (na + nb + kTabPad);
        const float2 t2v = tc[na + brev(plo, pb)];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float2 v = x[TileFwd::pos(tid + k * nthr)];
            const float2 w = g_cmul(tc[brev_m<4>(k)], t2v);
            o[((plo + (k << pb)) << l2) + c0 + c] = g_cmul(v, w);
        }
        return;
    }
    batched<8>(tid, L1 << kColLogT, nthr, [&](int idx) -> float2 { return x[TileFwd::pos(idx)]; },
               [&](int idx, float2 v) {
                   const int c = idx & (kColT - 1), pos = idx >> kColLogT;
                   const int k1 = brev(pos, l1);
                   const float2* tc = tab + c * (na + nb + kTabPad);
                   const float2 w = g_cmul(tc[k1 & (na - 1)], tc[na + (k1 >> a)]);
                   o[(pos << l2) + c0 + c] = g_cmul(v, w);             // index < L <= 2^23
               });
}
// inverse: in[k1'][n2] (after the inverse row pass and its twiddle) -> this tile of r[n1][n2] in LDS only: the
// tile's partial argmax of |r|^2 in 'full' order with the peak's two neighbour taps when they sit inside the
// tile, and the |r|^2 of the tile's first and last column (the "halo" a neighbouring tile's peak may need).
// r itself is never written to HBM (16 N bytes per pair-window saved).  grid (L2/T, slots); parts = L2/T.
struct GTile {
    float v;        // max |r|^2 of the tile
    int k;          // its lowest 'full' index
    float tm, tp;   // |r|^2 at k-1, k+1, or -1 where that lag lives in another tile (or does not exist)
};
template <int kColLogT, int L1C = 0>
__global__ __launch_bounds__(1024) void g_cols_inv(const float2* __restrict__ in, const float2* __restrict__ tw, int l1_arg,
                                                   int l2, GTile* __restrict__ rec, float* __restrict__ halo) {
    constexpr int kColT = 1 << kColLogT;
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float2* x = reinterpret_cast<float2*>(gsm);
    const int l1 = L1C > 0 ? L1C : l1_arg;
    const int L1 = 1 << l1, L2 = 1 << l2, tid = threadIdx.x, nthr = L1C > 0 ? cols_threads(L1C, kColLogT) : (int)blockDim.x;
    const long L = (long)L1 << l2;
    const int N = (int)(L >> 1);
    const int tile = xcd_tile(blockIdx.x, gridDim.x), c0 = tile * kColT;
    float2* twl = x + lp((long)L1 << kColLogT) + (long)kColT * ((1 << (l1 >> 1)) + (L1 >> (l1 >> 1)) + 1);
    float* sv = reinterpret_cast<float*>(x + lp((long)L1 << kColLogT));   // 16 + 16 words in the (unused here) table region
    int* sk = reinterpret_cast<int*>(sv + 16);
    for (int k = tid; k < (L1 >> 1); k += nthr) twl[k] = tw[k];
    const float2* src = in + (long)blockIdx.y * L;
    float best = -1.0f;
    int bk = 0x7fffffff;
    // first pass straight from HBM (a thread's 2^M inputs are neighbouring rows of one column; consecutive
    // threads take consecutive columns: the same row segments a tile load would fetch); the last pass never
    // stores r: every output goes into the thread's running (max |r|^2, lowest 'full' index) and leaves only its
    // |r|^2 in the buffer, for the taps and the halo
#ifdef RMX_EXP_COLS_NOCOMP   // timing experiment only (wrong results): the tile's loads alone, same row segments, 16 in flight per thread
    {
        float acc = 0.0f;
        batched<16>(tid, L1 << kColLogT, nthr,
                    [&](int E) -> float2 { return src[((E >> kColLogT) << l2) + c0 + (E & (kColT - 1))]; },
                    [&](int, float2 e) { acc += e.x * e.x + e.y * e.y; });
        best = acc;
        bk = tid;
        x[tid].x = acc;
    }
    if (false)
#endif
    fft_dit_inv<kColLogT>(
        x, l1, twl, tid, nthr,
#ifdef RMX_EXP_COLS_NOLOAD   // timing experiment only (wrong results): the tile's arithmetic alone, no global loads
        make_src([&](int E) -> float2 { return make_float2(1e-6f * (float)(E + c0), 1.0f); }),
#else
        make_src([&](int E) -> float2 { return src[((E >> kColLogT) << l2) + c0 + (E & (kColT - 1))]; }),
#endif
        make_dst([&](int E0, int off, float2 e) {
            const int E = E0 + off;
            const float v = e.x * e.x + e.y * e.y;
            if (v >= best) {                      // (rare after the first few elements: the index arithmetic stays off the common path)
                const int m = ((E >> kColLogT) << l2) + c0 + (E & (kColT - 1));
                const int k = full_index(m, N);
                if (k >= 0 && (v > best || k < bk)) { best = v; bk = k; }
            }
            x[TileInv::pos(E0) + TileInv::pos(off)].x = v;        // (4-byte store: only .x is read back, by the halo and the taps)
        }));
    __syncthreads();
    // halo: |r|^2 of columns c0 and c0 + T - 1, all rows: [slot][tile][2][L1]
    float* hb = halo + ((long)blockIdx.y * gridDim.x + tile) * 2L * L1;
    for (int n1 = tid; n1 < 2 * L1; n1 += nthr) {
        const int row = n1 & (L1 - 1), col = n1 < L1 ? 0 : kColT - 1;
        hb[n1] = x[TileInv::pos((row << kColLogT) + col)].x;
    }
    block_argmax_w(best, bk, sv, sk, tid, nthr);
    if (tid == 0) {
        GTile t;
        t.v = best;
        t.k = bk;
        t.tm = t.tp = -1.0f;
        if (bk > 0 && bk < 2 * N - 2) {
            const long m = circ_index(bk, N);
            const int c = (int)(m & (L2 - 1)) - c0, row = (int)(m >> l2);
            // 'full' neighbours are the circular neighbours m -+ 1 (the excluded lag -N sits between the two
            // ends of the 'full' range): inside this tile when the column is
            if (c > 0) t.tm = x[TileInv::pos((row << kColLogT) + c - 1)].x;
            if (c < kColT - 1) t.tp = x[TileInv::pos((row << kColLogT) + c + 1)].x;
        }
        rec[(long)blockIdx.y * gridDim.x + tile] = t;
    }
}

// final reduction over a slot's tile records (one wave per slot), neighbour taps from the winning record or
// from the halo columns of the adjacent tile, parabola
__global__ __launch_bounds__(64) void g_final(int N, int l1, int l2, int col_log_t, const GTile* __restrict__ rec,
                                              const float* __restrict__ halo, int parts, int n_slots, long out_base,
                                              float out_scale, int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                              float* __restrict__ peak) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= n_slots) return;
    float best = -1.0f;
    int bk = 0x7fffffff, bp = 0;
    for (int p = lane; p < parts; p += 64) {
        const GTile t = rec[(long)slot * parts + p];
        if (t.v > best || (t.v == best && t.k < bk)) { best = t.v; bk = t.k; bp = p; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int ok = __shfl_xor(bk, off, 64), op = __shfl_xor(bp, off, 64);
        if (ov > best || (ov == best && ok < bk)) { best = ov; bk = ok; bp = op; }
    }
    if (lane != 0) return;
    const GTile t = rec[(long)slot * parts + bp];
    const int L1 = 1 << l1, L2 = 1 << l2, T = 1 << col_log_t;
    const long L = (long)L1 << l2;
    const float b = sqrtf(best) * out_scale;
    float frac = 0.0f;
    if (bk > 0 && bk < 2 * N - 2) {
        const long m = circ_index(bk, N);
        auto tap = [&](float inside, long mm) -> float {
            if (inside >= 0.0f) return inside;
            mm &= (L - 1);                                   // circular neighbour, in another tile: an edge column
            const int row = (int)(mm >> l2), col = (int)(mm & (L2 - 1));
            const int tile = col >> col_log_t, cc = col & (T - 1);
            return halo[(((long)slot * parts + tile) * 2 + (cc == 0 ? 0 : 1)) * L1 + row];
        };
        const float ra = tap(t.tm, m - 1 + L), rc = tap(t.tp, m + 1);
        frac = parabola(sqrtf(ra) * out_scale, b, sqrtf(rc) * out_scale);
    }
    lag_int[out_base + slot] = bk - (N - 1);
    lag_frac[out_base + slot] = frac;
    peak[out_base + slot] = b;
}

// ---- host-side tables ---------------------------------------------------------------------------
inline void make_row_table(std::vector<float2>& t, int R) {   // W_R^k, k < R/2
    t.resize(R / 2 > 0 ? R / 2 : 1);
    for (int k = 0; k < R / 2; ++k) {
        const double a = -6.283185307179586476925286766559 * (double)k / (double)R;
        t[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    if (R / 2 == 0) t[0] = make_float2(1.f, 0.f);
}
inline void make_big_tables(std::vector<float2>& thi, std::vector<float2>& tlo, long L, int lo_bits) {
    const long nlo = 1L << lo_bits, nhi = (L + nlo - 1) / nlo;
    tlo.resize(nlo);
    thi.resize(nhi);
    for (long j = 0; j < nlo; ++j) {
        const double a = -6.283185307179586476925286766559 * (double)j / (double)L;
        tlo[j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (long i = 0; i < nhi; ++i) {
        const double a = -6.283185307179586476925286766559 * (double)(i * nlo) / (double)L;
        thi[i] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
}

}  // namespace gen
}  // namespace rmx
