// generic_path.hpp -- the same path for every power-of-two window length N (16 .. 4 Mi samples).
//
// N = 4096 (BASELINE cfg3/cfg4) has the register/LDS-resident radix-16 kernels of rmx_hip.hip.  All
// other lengths run here: simpler kernels, same definition, same output contract.
//   L = 2N <= 16384  one workgroup per transform, the whole zero-padded window in LDS (128 KiB at L = 16384,
//                    i.e. the reference's 8192-sample captures):
//                      g_fwd_small   (window, buoy)  : radix-2 DIF, spectrum left in bit-reversed order
//                      g_pair_small  (window, pair)  : X_j conj(X_i) -> radix-2 DIT (takes bit-reversed
//                                                      input, natural output) -> |.|, argmax, parabola
//   L = 2N  > 16384  four-step transform through HBM in two passes per transform, no transposes: the
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
#include <vector>

namespace rmx {
namespace gen {

constexpr int kGThreads = 256;

__device__ __forceinline__ float2 g_cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 g_cmulc(float2 a, float2 b) {   // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

// In-place transforms of R = 2^logR points held in LDS; tw[k] = W_R^k, k < R/2.  Two radix-2 stages
// are fused per pass (a radix-4 butterfly in registers: one LDS read and write and one barrier per
// TWO stages; an odd logR leaves one plain radix-2 stage); the data flow, and with it the output
// order, is that of the radix-2 network.  LOGT > 0: 2^LOGT interleaved transforms at once (element e
// of transform c at x[(e << LOGT) | c]: a tile of columns of a row-major matrix; consecutive threads
// take consecutive columns, so LDS accesses stay conflict free).
// DIF: natural in -> bit-reversed out (forward, W = exp(-2 pi i / R)).
template <int LOGT = 0>
__device__ __forceinline__ void lds_dif(float2* x, int logR, const float2* __restrict__ tw, int tid, int nthr) {
    constexpr int T = 1 << LOGT;
    int s = logR - 1;
    for (; s >= 1; s -= 2) {          // stages s (half = 2q) and s-1 (half = q)
        const int q = 1 << (s - 1);
        const int work = (1 << (logR - 2)) << LOGT;
        for (int idx = tid; idx < work; idx += nthr) {
            const int c = idx & (T - 1), i = idx >> LOGT;
            const int l = i & (q - 1);
            const int j = ((i >> (s - 1)) << (s + 1)) | l;
            float2* p = x + ((long)j << LOGT) + c;
            const int qs = q << LOGT;
            const float2 x0 = p[0], x1 = p[qs], x2 = p[2 * qs], x3 = p[3 * qs];
            const float2 wa = tw[l << (logR - 1 - s)];          // stage s, element j
            const float2 wb = make_float2(wa.y, -wa.x);         // stage s, element j+q: wa * W_R^(R/4) = -i wa
            const float2 w2 = tw[l << (logR - s)];              // stage s-1
            const float2 a0 = make_float2(x0.x + x2.x, x0.y + x2.y);
            const float2 a2 = g_cmul(make_float2(x0.x - x2.x, x0.y - x2.y), wa);
            const float2 a1 = make_float2(x1.x + x3.x, x1.y + x3.y);
            const float2 a3 = g_cmul(make_float2(x1.x - x3.x, x1.y - x3.y), wb);
            p[0] = make_float2(a0.x + a1.x, a0.y + a1.y);
            p[qs] = g_cmul(make_float2(a0.x - a1.x, a0.y - a1.y), w2);
            p[2 * qs] = make_float2(a2.x + a3.x, a2.y + a3.y);
            p[3 * qs] = g_cmul(make_float2(a2.x - a3.x, a2.y - a3.y), w2);
        }
        __syncthreads();
    }
    if (s == 0) {                     // odd logR: the last stage (half = 1, twiddle 1)
        const int work = (1 << (logR - 1)) << LOGT;
        for (int idx = tid; idx < work; idx += nthr) {
            const int c = idx & (T - 1), i = idx >> LOGT;
            float2* p = x + ((long)(2 * i) << LOGT) + c;
            const float2 a = p[0], b = p[T];
            p[0] = make_float2(a.x + b.x, a.y + b.y);
            p[T] = make_float2(a.x - b.x, a.y - b.y);
        }
        __syncthreads();
    }
}
// DIT with conjugated twiddles: bit-reversed in -> natural out (inverse, unnormalised).
template <int LOGT = 0>
__device__ __forceinline__ void lds_dit_inv(float2* x, int logR, const float2* __restrict__ tw, int tid, int nthr) {
    constexpr int T = 1 << LOGT;
    int s = 0;
    if (logR & 1) {                   // odd logR: the first stage (half = 1, twiddle 1) on its own
        const int work = (1 << (logR - 1)) << LOGT;
        for (int idx = tid; idx < work; idx += nthr) {
            const int c = idx & (T - 1), i = idx >> LOGT;
            float2* p = x + ((long)(2 * i) << LOGT) + c;
            const float2 a = p[0], b = p[T];
            p[0] = make_float2(a.x + b.x, a.y + b.y);
            p[T] = make_float2(a.x - b.x, a.y - b.y);
        }
        __syncthreads();
        s = 1;
    }
    for (; s + 1 < logR; s += 2) {    // stages s (half = q) and s+1 (half = 2q)
        const int q = 1 << s;
        const int work = (1 << (logR - 2)) << LOGT;
        for (int idx = tid; idx < work; idx += nthr) {
            const int c = idx & (T - 1), i = idx >> LOGT;
            const int l = i & (q - 1);
            const int j = ((i >> s) << (s + 2)) | l;
            float2* p = x + ((long)j << LOGT) + c;
            const int qs = q << LOGT;
            const float2 w1 = tw[l << (logR - 1 - s)];          // stage s
            const float2 wa = tw[l << (logR - 2 - s)];          // stage s+1, element j
            const float2 wb = make_float2(wa.y, -wa.x);         // stage s+1, element j+q: -i wa (conjugated below)
            const float2 x0 = p[0], x1 = g_cmulc(p[qs], w1), x2 = p[2 * qs], x3 = g_cmulc(p[3 * qs], w1);
            const float2 a0 = make_float2(x0.x + x1.x, x0.y + x1.y), a1 = make_float2(x0.x - x1.x, x0.y - x1.y);
            const float2 b0 = g_cmulc(make_float2(x2.x + x3.x, x2.y + x3.y), wa);
            const float2 b1 = g_cmulc(make_float2(x2.x - x3.x, x2.y - x3.y), wb);
            p[0] = make_float2(a0.x + b0.x, a0.y + b0.y);
            p[2 * qs] = make_float2(a0.x - b0.x, a0.y - b0.y);
            p[qs] = make_float2(a1.x + b1.x, a1.y + b1.y);
            p[3 * qs] = make_float2(a1.x - b1.x, a1.y - b1.y);
        }
        __syncthreads();
    }
}

// 'full' order index (lag ascending from -(N-1)) of circular index m of an L = 2N point correlation;
// m == N (lag -N) is not part of the 'full' output: returns -1.
__device__ __forceinline__ int full_index(int m, int N) { return m < N ? m + N - 1 : (m == N ? -1 : m - N - 1); }
__device__ __forceinline__ int circ_index(int k, int N) { return k >= N - 1 ? k - (N - 1) : k + N + 1; }

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
                                                         long first_item, float scale) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float2* x = reinterpret_cast<float2*>(gsm);
    const int L = 1 << logL, tid = threadIdx.x, nthr = blockDim.x;
    const long item = first_item + blockIdx.x;
    for (int n = tid; n < L; n += nthr) {
        float2 v = make_float2(0.f, 0.f);
        if (n < N) {
            if constexpr (U8) {
                const uchar2 b = reinterpret_cast<const uchar2*>(iq)[item * N + n];
                v = make_float2((float)b.x - 127.5f, (float)b.y - 127.5f);
            } else {
                v = reinterpret_cast<const float2*>(iq)[item * N + n];
            }
        }
        x[n] = v;
    }
    __syncthreads();
    lds_dif(x, logL, tw, tid, nthr);
    float2* out = spec + (long)blockIdx.x * L;
    for (int n = tid; n < L; n += nthr) out[n] = make_float2(x[n].x * scale, x[n].y * scale);
}

struct GPair {
    int i, j;
};

__global__ __launch_bounds__(1024) void g_pair_small(const float2* __restrict__ spec,
                                                          const float2* __restrict__ tw,
                                                          const GPair* __restrict__ pairs, int n_pairs,
                                                          int n_buoys, int N, int logL, long first_window,
                                                          float out_scale, int* __restrict__ lag_int,
                                                          float* __restrict__ lag_frac, float* __restrict__ peak) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float2* x = reinterpret_cast<float2*>(gsm);
    const int L = 1 << logL, tid = threadIdx.x, nthr = blockDim.x;
    float* sv = reinterpret_cast<float*>(gsm + (size_t)L * 8);
    int* sk = reinterpret_cast<int*>(sv + nthr);
    const int wl = blockIdx.x / n_pairs, q = blockIdx.x % n_pairs;
    const GPair pr = pairs[q];
    const float2* xi = spec + ((long)wl * n_buoys + pr.i) * L;
    const float2* xj = spec + ((long)wl * n_buoys + pr.j) * L;
    for (int n = tid; n < L; n += nthr) x[n] = g_cmulc(xj[n], xi[n]);   // X_j conj(X_i)
    __syncthreads();
    lds_dit_inv(x, logL, tw, tid, nthr);
    float best = -1.0f;
    int bk = 0x7fffffff;
    for (int m = tid; m < L; m += nthr) {
        const int k = full_index(m, N);
        if (k < 0) continue;
        const float v = x[m].x * x[m].x + x[m].y * x[m].y;
        if (v > best || (v == best && k < bk)) { best = v; bk = k; }
    }
    block_argmax(best, bk, sv, sk, tid, nthr);
    if (tid == 0) {
        const float b = sqrtf(best) * out_scale;
        float frac = 0.0f;
        if (bk > 0 && bk < 2 * N - 2) {
            const float2 ra = x[circ_index(bk - 1, N)], rc = x[circ_index(bk + 1, N)];
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
template <bool FWD, bool TW, bool PROD = false>
__global__ __launch_bounds__(kGThreads) void g_rows(float2* __restrict__ data, const float2* __restrict__ tw, int logR,
                                                    int n_rows, int row_bits, long Ltot, int lo_bits,
                                                    const float2* __restrict__ thi,
                                                    const float2* __restrict__ tlo, float scale, long total_rows,
                                                    const float2* __restrict__ spec = nullptr,
                                                    const GPair* __restrict__ pairs = nullptr, int n_pairs = 0,
                                                    int n_buoys = 0) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int R = 1 << logR;
    const int tpr = (R >> 2) < kGThreads ? ((R >> 2) > 0 ? (R >> 2) : 1) : kGThreads;   // threads per row
    const int rpw = kGThreads / tpr;                                                   // rows per workgroup
    const int g = threadIdx.x / tpr, tid = threadIdx.x % tpr;
    const long ridx = (long)blockIdx.x * rpw + g;
    const bool live = ridx < total_rows;
    float2* x = reinterpret_cast<float2*>(gsm) + (long)g * R;
    // the row's twiddle table goes to LDS once (butterflies would otherwise fetch two entries per
    // radix-4 group through the vector memory path, a dependent cache-latency access in the inner loop)
    float2* twl = reinterpret_cast<float2*>(gsm) + (long)rpw * R + (TW ? (long)rpw * ((1 << (logR >> 1)) + (R >> (logR >> 1))) : 0);
    for (int k = threadIdx.x; k < (R >> 1); k += kGThreads) twl[k] = tw[k];
    float2* row = data + ridx * R;
    const int rib = (int)(ridx & (n_rows - 1));   // row index inside its batch element (n_rows = 2^row_bits)
    if (live) {
        if constexpr (PROD) {
            const long slot = ridx >> row_bits;
            const int wl = (int)(slot / n_pairs), q = (int)(slot % n_pairs);
            const GPair pr = pairs[q];
            const float2* xi = spec + (((long)wl * n_buoys + pr.i) * n_rows + rib) * R;
            const float2* xj = spec + (((long)wl * n_buoys + pr.j) * n_rows + rib) * R;
            for (int n = tid; n < R; n += tpr) x[n] = g_cmulc(xj[n], xi[n]);
        } else {
            for (int n = tid; n < R; n += tpr) x[n] = row[n];
        }
    }
    __syncthreads();
    if (FWD) lds_dif(x, logR, twl, tid, tpr); else lds_dit_inv(x, logR, twl, tid, tpr);
    if constexpr (TW) {
        // W_L^(c*e) for this row's fixed multiplier c and every exponent e < R, as the product of two
        // per-row LDS tables T1[e & (2^a - 1)] * T2[e >> a] (2^a + R/2^a big-table lookups per row
        // instead of two uncoalesced gathers per element)
        const int a = logR >> 1, n1 = 1 << a, n2 = R >> a;
        float2* t1 = reinterpret_cast<float2*>(gsm) + (long)(kGThreads / tpr) * R + (long)g * (n1 + n2);
        float2* t2 = t1 + n1;
        const long c = FWD ? (long)rib : (long)brev(rib, row_bits);
        for (int e = tid; e < n1 + n2; e += tpr) {
            const long ee = e < n1 ? (long)e : ((long)(e - n1) << a);
            t1[e] = big_tw((c * ee) & (Ltot - 1), lo_bits, thi, tlo);
        }
        __syncthreads();
        if (!live) return;
        for (int n = tid; n < R; n += tpr) {
            const int e = FWD ? brev(n, logR) : n;
            const float2 w = g_cmul(t1[e & (n1 - 1)], t2[e >> a]);
            const float2 v = FWD ? g_cmul(x[n], w) : g_cmulc(x[n], w);
            row[n] = make_float2(v.x * scale, v.y * scale);
        }
    } else {
        if (!live) return;
        for (int n = tid; n < R; n += tpr) row[n] = make_float2(x[n].x * scale, x[n].y * scale);
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
template <bool U8, int kColLogT>
__global__ __launch_bounds__(1024) void g_cols_fwd(const void* __restrict__ iq, float2* __restrict__ out,
                                                   const float2* __restrict__ tw, int l1, int l2, long first_item,
                                                   int lo_bits, const float2* __restrict__ thi,
                                                   const float2* __restrict__ tlo) {
    constexpr int kColT = 1 << kColLogT;
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float2* x = reinterpret_cast<float2*>(gsm);
    const int L1 = 1 << l1, L2 = 1 << l2, tid = threadIdx.x, nthr = blockDim.x;
    const long L = (long)L1 << l2, N = L >> 1;
    const int c0 = blockIdx.x * kColT;
    const long item = first_item + blockIdx.y;
    float2* twl = x + ((long)L1 << kColLogT) + (long)kColT * ((1 << (l1 >> 1)) + (L1 >> (l1 >> 1)));   // behind x and the tables
    for (int k = tid; k < (L1 >> 1); k += nthr) twl[k] = tw[k];
    for (int idx = tid; idx < (L1 << kColLogT); idx += nthr) {
        const int c = idx & (kColT - 1), n1 = idx >> kColLogT;
        float2 v = make_float2(0.f, 0.f);
        if (n1 < (L1 >> 1)) {
            const long n = (long)n1 * L2 + c0 + c;
            if constexpr (U8) {
                const uchar2 b = reinterpret_cast<const uchar2*>(iq)[item * N + n];
                v = make_float2((float)b.x - 127.5f, (float)b.y - 127.5f);
            } else {
                v = reinterpret_cast<const float2*>(iq)[item * N + n];
            }
        }
        x[idx] = v;
    }
    __syncthreads();
    lds_dif<kColLogT>(x, l1, twl, tid, nthr);
    // per-column twiddle tables: W_L^(n2*e) = T1[e & (2^a-1)] * T2[e >> a]   (n2*e < L: no reduction)
    const int a = l1 >> 1, na = 1 << a, nb = L1 >> a;
    float2* tab = x + ((long)L1 << kColLogT);               // [16][na + nb]
    for (int idx = tid; idx < kColT * (na + nb); idx += nthr) {
        const int c = idx / (na + nb), e = idx % (na + nb);
        const long ee = e < na ? (long)e : ((long)(e - na) << a);
        tab[idx] = big_tw((long)(c0 + c) * ee, lo_bits, thi, tlo);
    }
    __syncthreads();
    float2* o = out + (long)blockIdx.y * L;
    for (int idx = tid; idx < (L1 << kColLogT); idx += nthr) {
        const int c = idx & (kColT - 1), pos = idx >> kColLogT;
        const int k1 = brev(pos, l1);
        const float2* tc = tab + c * (na + nb);
        const float2 w = g_cmul(tc[k1 & (na - 1)], tc[na + (k1 >> a)]);
        o[(long)pos * L2 + c0 + c] = g_cmul(x[idx], w);
    }
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
template <int kColLogT>
__global__ __launch_bounds__(1024) void g_cols_inv(const float2* __restrict__ in, const float2* __restrict__ tw, int l1,
                                                   int l2, GTile* __restrict__ rec, float* __restrict__ halo) {
    constexpr int kColT = 1 << kColLogT;
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float2* x = reinterpret_cast<float2*>(gsm);
    const int L1 = 1 << l1, L2 = 1 << l2, tid = threadIdx.x, nthr = blockDim.x;
    const long L = (long)L1 << l2;
    const int N = (int)(L >> 1);
    const int c0 = blockIdx.x * kColT;
    float2* twl = x + ((long)L1 << kColLogT) + (long)kColT * ((1 << (l1 >> 1)) + (L1 >> (l1 >> 1)));
    float* sv = reinterpret_cast<float*>(x + ((long)L1 << kColLogT));   // 16 + 16 words in the (unused here) table region
    int* sk = reinterpret_cast<int*>(sv + 16);
    for (int k = tid; k < (L1 >> 1); k += nthr) twl[k] = tw[k];
    const float2* src = in + (long)blockIdx.y * L;
    for (int idx = tid; idx < (L1 << kColLogT); idx += nthr)
        x[idx] = src[(long)(idx >> kColLogT) * L2 + c0 + (idx & (kColT - 1))];
    __syncthreads();
    lds_dit_inv<kColLogT>(x, l1, twl, tid, nthr);
    float best = -1.0f;
    int bk = 0x7fffffff;
    for (int idx = tid; idx < (L1 << kColLogT); idx += nthr) {
        const long m = (long)(idx >> kColLogT) * L2 + c0 + (idx & (kColT - 1));
        const float2 e = x[idx];
        const int k = full_index((int)m, N);
        const float v = e.x * e.x + e.y * e.y;
        if (k >= 0 && (v > best || (v == best && k < bk))) { best = v; bk = k; }
    }
    // halo: |r|^2 of columns c0 and c0 + T - 1, all rows: [slot][tile][2][L1]
    float* hb = halo + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2L * L1;
    for (int n1 = tid; n1 < 2 * L1; n1 += nthr) {
        const int row = n1 & (L1 - 1), col = n1 < L1 ? 0 : kColT - 1;
        const float2 e = x[(row << kColLogT) + col];
        hb[n1] = e.x * e.x + e.y * e.y;
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
            if (c > 0) { const float2 e = x[(row << kColLogT) + c - 1]; t.tm = e.x * e.x + e.y * e.y; }
            if (c < kColT - 1) { const float2 e = x[(row << kColLogT) + c + 1]; t.tp = e.x * e.x + e.y * e.y; }
        }
        rec[(long)blockIdx.y * gridDim.x + blockIdx.x] = t;
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
