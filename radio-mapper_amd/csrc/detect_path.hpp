// detect_path.hpp -- spectral detection: the reference's per-capture detector
// (buoy_node.py:401-433, iq_stream_client.py:186-217) batched over capture windows.
//
//   P[k]  = 20 log10(|FFT_N(iq)[k]| + 1e-12)            N-point, unpadded, unwindowed, float32
//   peaks = scipy.signal.find_peaks(P, height=thr, distance=d): local maxima (plateau -> midpoint) with
//           P >= thr, then highest-first removal of every peak closer than d bins to a kept one
//   floor = median(P); snr = P[peak] - floor; confidence = min(max(snr/20, 0), 1)
//   a peak is reported unless |signed bin| < dc_exclude_bins (buoy_node.py:419) or confidence <
//   min_confidence (:430)
//
// Two kernels, one workgroup per window: d_fft_db (the whole window in LDS, radix-4 DIF, dB spectrum in
// natural order to HBM) and d_peaks (spectrum back into LDS: candidates, distance filter, median by
// radix select, ordered compaction; 1024 threads, because its chains of dependent LDS reads need
// every wave the workgroup can have: cycle stamps showed 51 % in the filter rounds and 38 % in the
// median at 256 threads).  The greedy highest-first filter of scipy is evaluated in
// parallel rounds: an undecided candidate that outranks every undecided candidate within d-1 bins is
// kept, its neighbours are removed, repeat -- the same set as the sequential sweep.  Ties in height
// rank the higher bin first (what scipy's argsort-from-the-end does on the short peak lists it
// insertion-sorts; exact ties between neighbouring float32 dB values do not occur in practice).
#pragma once
#include <hip/hip_runtime.h>

#include "generic_path.hpp"

namespace rmx {
namespace det {

using gen::brev;
using gen::lds_dif;

template <bool U8>
__global__ __launch_bounds__(1024) void d_fft_db(const void* __restrict__ iq, float* __restrict__ pdb,
                                                 const float2* __restrict__ tw, int logN) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    float2* x = reinterpret_cast<float2*>(dsm);
    const int N = 1 << logN, tid = threadIdx.x, nthr = blockDim.x;
    const long w = blockIdx.x;
    for (int n = tid; n < N; n += nthr) {
        if constexpr (U8) {
            const uchar2 b = reinterpret_cast<const uchar2*>(iq)[w * N + n];
            x[gen::lp(n)] = make_float2((float)b.x - 127.5f, (float)b.y - 127.5f);
        } else {
            x[gen::lp(n)] = reinterpret_cast<const float2*>(iq)[w * N + n];
        }
    }
    __syncthreads();
    lds_dif<0>(x, logN, tw, tid, nthr);
    float* o = pdb + w * N;
    for (int pos = tid; pos < N; pos += nthr) {
        const float2 v = x[gen::lp(pos)];
        const float a = sqrtf(v.x * v.x + v.y * v.y);
        o[brev(pos, logN)] = 20.0f * log10f(a + 1e-12f);
    }
}

__device__ __forceinline__ unsigned f32_key(float f) {   // monotonic float -> uint
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// rank-th smallest (0-based) of the N floats in p (LDS): 4 passes of 8-bit histograms (LDS atomics; the
// hardware copes with the many same-address adds that clustered dB values produce -- measured 4x
// faster than a counting select with per-wave adds)
__device__ __forceinline__ float lds_select(const float* p, int N, int rank, unsigned* hist, unsigned* bc, int tid, int nthr) {
    unsigned prefix = 0, mask = 0;
    int r = rank;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += nthr) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < N; i += nthr) {
            const unsigned k = f32_key(p[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {     // first bucket whose cumulative count exceeds r: four buckets per lane, one wave scan
            const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned tot = h0 + h1 + h2 + h3;
            unsigned incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(incl, off, 64);
                if (tid >= off) incl += o;
            }
            const unsigned excl = incl - tot;
            if (excl <= (unsigned)r && (unsigned)r < incl) {
                unsigned acc = excl;
                int b = 4 * tid;
                if (acc + h0 <= (unsigned)r) { acc += h0; ++b;
                    if (acc + h1 <= (unsigned)r) { acc += h1; ++b;
                        if (acc + h2 <= (unsigned)r) { acc += h2; ++b; } } }
                bc[0] = (unsigned)b;
                bc[1] = acc;
            }
        }
        __syncthreads();
        prefix |= bc[0] << shift;
        mask |= 255u << shift;
        r -= (int)bc[1];
        __syncthreads();
    }
    return key_f32(prefix);
}

// (dependent LDS reads dominate: the kernel wants as many waves per window as the workgroup allows)
__global__ __launch_bounds__(1024) void d_peaks(const float* __restrict__ pdb, int logN, float thr, int dist,
                                               double dc_exclude_bins, float min_conf, int max_peaks,
                                               int* __restrict__ count, int* __restrict__ bin,
                                               float* __restrict__ power_db, float* __restrict__ snr_db,
                                               float* __restrict__ confidence, float* __restrict__ floor_db) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int N = 1 << logN, tid = threadIdx.x, nthr = blockDim.x;
    float* p = reinterpret_cast<float*>(dsm);                                   // [N]
    unsigned char* st = reinterpret_cast<unsigned char*>(p + N);                // [N] 0 none 1 undecided 2 kept 3 removed 4 new
    unsigned short* cand = reinterpret_cast<unsigned short*>(st + N);           // [N/2] candidate positions
    unsigned* hist = reinterpret_cast<unsigned*>(cand + N / 2);                 // [256]
    unsigned* bc = hist + 256;                                                  // [8] scalars
    unsigned* scan = bc + 8;                                                    // [nthr]
    const long w = blockIdx.x;
    const float* src = pdb + w * N;
    for (int i = tid; i < N; i += nthr) { p[i] = src[i]; st[i] = 0; }
    if (tid == 0) bc[2] = 0;
    __syncthreads();
    // local maxima; a plateau counts once, at its midpoint (scipy _local_maxima_1d)
    for (int i0 = 1; i0 < N - 1; i0 += nthr) {
        const int i = i0 + tid;
        int m = -1;
        if (i < N - 1) {
            const float v = p[i];
            if (p[i - 1] < v) {
                int r = i + 1;
                while (r < N - 1 && p[r] == v) ++r;
                if (p[r] < v && p[(i + r - 1) >> 1] >= thr) m = (i + r - 1) >> 1;
            }
        }
        // one counter add per wave (a third of all bins are candidates: thousands of adds to one word)
        const unsigned long long has = __ballot(m >= 0);
        unsigned base = 0;
        if ((tid & 63) == 0 && has) base = atomicAdd(&bc[2], (unsigned)__popcll(has));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (m >= 0) {
            st[m] = 1;
            cand[base + (unsigned)__popcll(has & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)m;
        }
    }
    __syncthreads();
    const int ncand = (int)bc[2];
    // distance filter in rounds
    if (dist > 1) {
        const int span = 2 * (dist - 1) + 1;
        for (;;) {
            if (tid == 0) bc[3] = 0;
            __syncthreads();
            for (int c = tid; c < ncand; c += nthr) {
                const int i = cand[c];
                if (st[i] != 1) continue;
                const float v = p[i];
                // the 2 dist - 1 neighbours in batches of eight independent reads (a loop with an early exit is a chain
                // of dependent LDS latencies: 32 us per round at N = 16 384)
                bool top = true;
                for (int k0 = 0; k0 < span; k0 += 8) {
                    unsigned char s8[8];
                    float p8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = i - (dist - 1) + k0 + u;
                        const int jc = (k0 + u < span && j >= 0 && j < N) ? j : i;     // out of range: the candidate itself (ignored below)
                        s8[u] = st[jc];
                        p8[u] = p[jc];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = i - (dist - 1) + k0 + u;
                        const bool in = k0 + u < span && j >= 0 && j < N && j != i;
                        if (in && (s8[u] == 1 || s8[u] == 4) && (p8[u] > v || (p8[u] == v && j > i))) top = false;
                    }
                }
                if (top) st[i] = 4;
            }
            __syncthreads();
            // neighbours of this round's keeps are removed; a keep stays marked 4 until the rounds are over (no undecided
            // candidate is left within reach of one, so later rounds never meet it) -- one loop and two barriers fewer
            // per round than promoting 4 -> 2 here
            bool undecided = false;
            for (int c = tid; c < ncand; c += nthr) {
                const int i = cand[c];
                if (st[i] != 1) continue;
                bool hit = false;
                for (int k0 = 0; k0 < span; k0 += 8) {
                    unsigned char s8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = i - (dist - 1) + k0 + u;
                        s8[u] = st[(k0 + u < span && j >= 0 && j < N) ? j : i];        // (the candidate itself is 1, never 4)
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) hit = hit || s8[u] == 4;
                }
                if (hit) st[i] = 3;
                else undecided = true;
            }
            if (undecided) bc[3] = 1;                  // benign race: any writer writes 1
            __syncthreads();
            const bool more = bc[3] != 0;
            __syncthreads();
            if (!more) break;
        }
        for (int c = tid; c < ncand; c += nthr)
            if (st[cand[c]] == 4) st[cand[c]] = 2;
        __syncthreads();
    } else {
        for (int c = tid; c < ncand; c += nthr) st[cand[c]] = 2;
        __syncthreads();
    }
    // noise floor = median (mean of the two middle values for even N, as np.median)
    // (even N: the upper middle value is the lower one again, or the smallest value above it -- one counting pass
    // instead of a second four-pass select)
    float m_lo, m_hi;
    if (N & 1) {
        m_lo = m_hi = lds_select(p, N, N / 2, hist, bc, tid, nthr);
    } else {
        m_lo = lds_select(p, N, N / 2 - 1, hist, bc, tid, nthr);
        const unsigned klo = f32_key(m_lo);
        if (tid == 0) { bc[5] = 0; bc[6] = 0xffffffffu; }
        __syncthreads();
        unsigned le = 0, nxt = 0xffffffffu;
        for (int i = tid; i < N; i += nthr) {
            const unsigned k = f32_key(p[i]);
            if (k <= klo) ++le;
            else if (k < nxt) nxt = k;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            le += __shfl_xor(le, off, 64);
            const unsigned o = __shfl_xor(nxt, off, 64);
            nxt = o < nxt ? o : nxt;
        }
        if ((tid & 63) == 0) { atomicAdd(&bc[5], le); atomicMin(&bc[6], nxt); }
        __syncthreads();
        m_hi = bc[5] > (unsigned)(N / 2) ? m_lo : key_f32(bc[6]);
    }
    const float floor_v = (m_lo + m_hi) * 0.5f;
    // ordered compaction: every thread owns a contiguous run of bins
    const int per = (N + nthr - 1) / nthr;
    const int b0 = tid * per, b1 = (b0 + per) < N ? (b0 + per) : N;
    auto passes = [&](int k, float& snr, float& conf) -> bool {
        if (st[k] != 2) return false;
        const int sb = k < N / 2 ? k : k - N;
        if ((double)(sb < 0 ? -sb : sb) < dc_exclude_bins) return false;
        snr = p[k] - floor_v;
        conf = fminf(fmaxf(__fdiv_rn(snr, 20.0f), 0.0f), 1.0f);
        return !(conf < min_conf);
    };
    int mine = 0;
    float s_, c_;
    for (int k = b0; k < b1; ++k) mine += passes(k, s_, c_) ? 1 : 0;
    // exclusive prefix sum over the threads: inside each wave by shuffles, across the (at most 16) waves by one wave
    // (a serial loop of thread 0 over 1024 LDS words was 60 us of this kernel's 280 per window)
    unsigned incl = (unsigned)mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += o;
    }
    if ((tid & 63) == 63) scan[tid >> 6] = incl;   // wave totals
    __syncthreads();
    if (tid < 64) {
        const int nw = (nthr + 63) >> 6;
        unsigned t = tid < nw ? scan[tid] : 0u, ti = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(ti, off, 64);
            if (tid >= off) ti += o;
        }
        if (tid < nw) scan[64 + tid] = ti - t;      // exclusive offset of wave tid
        if (tid == nw - 1) bc[4] = ti;
    }
    __syncthreads();
    int at = (int)(scan[64 + (tid >> 6)] + incl - (unsigned)mine);
    const long ob = w * max_peaks;
    for (int k = b0; k < b1; ++k) {
        float snr, conf;
        if (passes(k, snr, conf)) {
            if (at < max_peaks) {
                bin[ob + at] = k;
                power_db[ob + at] = p[k];
                snr_db[ob + at] = snr;
                confidence[ob + at] = conf;
            }
            ++at;
        }
    }
    if (tid == 0) {
        count[w] = (int)bc[4];      // may exceed max_peaks: the arrays then hold the first max_peaks
        floor_db[w] = floor_v;
    }
}

}  // namespace det
}  // namespace rmx
