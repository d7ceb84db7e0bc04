// winpk.hpp -- k_winp: the fused window kernel k_win (rmx_hip.hip) with its butterfly arithmetic on packed
// fp32 (fft_pk.hpp).  Same decomposition, schedule, LDS images, spectrum scratch and peak search: one
// persistent 512-thread workgroup per CU, thread t = 2u + p holds 16 complex points of sub-transform p,
// three radix-16 passes, one workgroup barrier per transform.  Only the instruction selection differs:
// every complex value is one VGPR pair and the radix-4 layers, the merged twiddles and the conj-multiply
// issue as v_pk_* (half the VALU instructions of k_win for the same arithmetic).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_pk.hpp"

namespace rmx {
namespace pk {

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
using u32x2 = unsigned int __attribute__((ext_vector_type(2)));

// LDS carve: identical to k_win's (rmx_hip.hip: kLdsWin*)
constexpr int kPImg = kXchgF2 * 8;                                   // 69632 each, two of them
constexpr int kPTw2 = 2 * kPImg;
constexpr int kPSlots = 8;
constexpr int kPBatch = 7;
constexpr int kPHalo = kPTw2 + 16 * kTw2RowF2 * 8;                  // [slots][8][4][16] float
constexpr int kPRed = kPHalo + kPSlots * 8 * 4 * 16 * 4;            // [slots][8] float4
constexpr int kPOidx = kPRed + kPSlots * 8 * 16;                    // [slots] int
constexpr int kLdsWinpBytes = kPOidx + kPSlots * 4;
static_assert(kLdsWinpBytes <= 160 * 1024, "k_winp LDS");

// resolve_batch / k_to_owner / load_tw2_to_lds_grouped are k_win's (declared in rmx_hip.hip before this
// header is included)
template <bool U8, class ResolveFn>
__device__ __forceinline__ void winp_body(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                          const float4* __restrict__ tw1_g, const float2* __restrict__ tw2_g,
                                          int n_buoys, long first_window, float out_scale, int* __restrict__ lag_int,
                                          float* __restrict__ lag_frac, float* __restrict__ peak, int n_win,
                                          ResolveFn&& resolve) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f2* img0 = reinterpret_cast<f2*>(smem);
    f2* img1 = reinterpret_cast<f2*>(smem + kPImg);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kPTw2);
    float* halo = reinterpret_cast<float*>(smem + kPHalo);
    float4* red = reinterpret_cast<float4*>(smem + kPRed);
    int* oidx = reinterpret_cast<int*>(smem + kPOidx);

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;
    const int B = n_buoys;
    const int n_pairs = B * (B - 1) / 2;

    if (t < 256) {   // TW2 in layer-1 group order (see load_tw2_to_lds_grouped)
        const int a = t >> 4, q = t & 15;
        tw2_lds[a * kTw2RowF2 + (q == 0 ? 15 : 4 * (q & 3) + (q >> 2) - 1)] = tw2_g[t];
    }
    f2 tw1[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 w = tw1_g[j * kThreads + t];
        tw1[2 * j] = mk(w.x, w.y);
        tw1[2 * j + 1] = mk(w.z, w.w);
    }
    const f4* tw2row = reinterpret_cast<const f4*>(tw2_lds + (u & 15) * kTw2RowF2);
    const float sgn = p ? -1.0f : 1.0f;
    const int kbase = p ? (u - 1) : (u + kM - 1);
    const int hl = lane < 2 ? lane : lane - 60;
    const bool is_halo = lane < 2 || lane >= 62;
    __syncthreads();

    for (int wl = blockIdx.x; wl < n_win; wl += gridDim.x) {
    f2 sa[16], sb[16];   // anchor spectrum X_i and the streamed X_j
    const long wbase = (long)blockIdx.x * B;
    const long obase = (first_window + wl) * (long)n_pairs;
    int seq = 0, npair = 0, npend = 0;

    auto barrier_hook = [&](bool flush) __attribute__((always_inline)) {
        __syncthreads();
        if (npend == kPBatch || (flush && npend > 0)) {
            if (wave == (seq & 7))
                resolve(lane, red, halo, oidx, (npair - npend) & (kPSlots - 1), npend, obase, out_scale, lag_int, lag_frac,
                        peak);
            npend = 0;
        }
    };
    // odd lanes: v[q] *= W32^q (in place, exec-masked; twiddles from SGPR pairs)
    auto mul_w32_odd = [&](f2 (&v)[16]) __attribute__((always_inline)) {
        if (p) {
            auto wk = [](int q) { const float2 w = w32(q); return mk(w.x, w.y); };
#pragma unroll
            for (int q = 0; q < 16; q += 4)      // (slot 0: W32^0 = 1, kept in the group of four)
                cmul4_k(v[q], v[q + 1], v[q + 2], v[q + 3], wk(q), wk(q + 1), wk(q + 2), wk(q + 3));
        }
    };
    const int samp_bytes = U8 ? 2 : 8;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(iq_v)) + (first_window + wl) * (long)B * kM * samp_bytes, 0,
        B * kM * samp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(spec) + wbase * (long)(8 * kThreads * 16), 0, B * (8 * kThreads * 16), 0x00020000);
    const int xoff = u * samp_bytes, soff = t * 16;
    auto load_x_part = [&](f2 (&d)[16], int b, auto part) __attribute__((always_inline)) {
        constexpr int G = decltype(part)::value;
        if constexpr (U8) {
#pragma unroll
            for (int q = 4 * G; q < 4 * G + 4; ++q)
                d[q] = mk(__uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(xs, xoff, (b * kM + q * 256) * 2, 0)), 0.0f);
        } else {
#pragma unroll
            for (int q = 4 * G; q < 4 * G + 4; ++q) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(xs, xoff, (b * kM + q * 256) * 8, 0);
                d[q] = __builtin_bit_cast(f2, r);
            }
        }
    };
    auto load_x = [&](f2 (&d)[16], int b) __attribute__((always_inline)) {
        load_x_part(d, b, std::integral_constant<int, 0>{});
        load_x_part(d, b, std::integral_constant<int, 1>{});
        load_x_part(d, b, std::integral_constant<int, 2>{});
        load_x_part(d, b, std::integral_constant<int, 3>{});
    };
    auto cvt_x = [&](f2 (&d)[16]) __attribute__((always_inline)) {
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f2 e = d[q];
                const unsigned r = __float_as_uint(e.x);
                d[q] = mk((float)(r & 0xffu) - 127.5f, (float)(r >> 8) - 127.5f);
            }
        }
    };
    auto load_spec_part = [&](f2 (&d)[16], int b, auto part) __attribute__((always_inline)) {
        constexpr int G = decltype(part)::value;
#pragma unroll
        for (int j = 2 * G; j < 2 * G + 2; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(ss, soff, (b * 8 + j) * (kThreads * 16), 0);
            const f4 f = __builtin_bit_cast(f4, w);
            d[2 * j] = lo2(f);
            d[2 * j + 1] = hi2(f);
        }
    };
    auto store_spec = [&](const f2 (&d)[16], int b) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f2 e = d[2 * j], o = d[2 * j + 1];
            const f4 f = {e.x, e.y, o.x, o.y};
            // whole byte offset in the VGPR offset (see store_spec of k_win)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), ss, soff + (b * 8 + j) * (kThreads * 16), 0, 0);
        }
    };
    // forward spectrum of the samples in x, in place (carries the 2^-6 of the TW1 table)
    auto fwd = [&](f2 (&x)[16]) __attribute__((always_inline)) {
        f2* img = (seq & 1) ? img1 : img0;
        mul_w32_odd(x);
        dft16(x);
        mul_tw1(x, tw1);
        xchg_a_write(img, x, t);
        barrier_hook(false);
        xchg_b_read(img, x, t);
        dft16(x);
        const f4 r0 = tw2row[0], r1 = tw2row[1];   // ahead of the exchange reads (LDS returns in issue order)
        xchg_bc_write_b(img, x, t);
        wave_lds_order();
        xchg_bc_read_c(img, x, t);
        dft16_tw_row_l1(x, tw2row, r0, r1);        // W_256^(n0*k1) as pre-twiddle of the last pass
        dft16_layer2(x);
        ++seq;
    };
    auto pair_h1 = [&](const f2 (&a)[16], const f2 (&s)[16], int tr, auto prefetch) __attribute__((always_inline)) {
        f2* img = (tr & 1) ? img1 : img0;
        f2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = s[q];
        // R = X_j conj(X_i), (im,re)-swapped == swap(X_j) * X_i: merged into the first radix-16 pass
        dft16_tw_l1<false, true>(v, a);          // k2 -> n0   (role C), layer 1: the last reads of a and s
#pragma unroll
        for (int q = 0; q < 16; q += 4)          // pin: the requests below must follow the reads above
            asm volatile("" : "+v"(v[q]), "+v"(v[q + 1]), "+v"(v[q + 2]), "+v"(v[q + 3]));
        __builtin_amdgcn_sched_barrier(0);
        {
            f2* wb = img + (u >> 4) * kBcHalf + (u & 15) * kBcRow + p;
            dft16_layer2_emit(v, [&](auto kac, const f2& x0, const f2& x1, const f2& x2, const f2& x3)
                                     __attribute__((always_inline)) {
                constexpr int ka = decltype(kac)::value;
                wb[2 * ka] = x0;
                wb[2 * (ka + 4)] = x1;
                wb[2 * (ka + 8)] = x2;
                wb[2 * (ka + 12)] = x3;
                prefetch(kac);
            });
        }
        const f4 r0 = tw2row[0], r1 = tw2row[1];
        wave_lds_order();
        xchg_bc_read_b(img, v, t);
        dft16_tw_row_l1(v, tw2row, r0, r1);      // W_256^(n0*k1), k1 -> n1   (role B), layer 1
        {
            f2* xb = img + (u >> 4) * kBcHalf + (u & 15) * 2 + p;   // own half-wave regions
            dft16_layer2_emit(v, [&](auto kac, const f2& x0, const f2& x1, const f2& x2, const f2& x3)
                                     __attribute__((always_inline)) {
                constexpr int ka = decltype(kac)::value;
                xb[ka * 32] = x0;
                xb[(ka + 4) * 32] = x1;
                xb[(ka + 8) * 32] = x2;
                xb[(ka + 12) * 32] = x3;
            });
        }
    };
    auto pair_h2 = [&](int tr, int out_idx) __attribute__((always_inline)) {
        const f2* img = (tr & 1) ? img1 : img0;
        const int rb = npair & (kPSlots - 1);
        f2 v[16];
        xchg_a_read(img, v, t);
        dft16_tw_l1<false>(v, tw1);              // W_M^(u*k0) [* W_L^u odd], k0 -> n2   (role A)
        dft16_layer2(v);
        mul_w32_odd(v);                          // odd lanes: * W32^q
        // last radix-2 stage across the lane pair (scalar DPP on the halves), then |.|^2
        float mag[16];
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            float x0 = v[q].x, y0 = v[q].y, x1 = v[q + 1].x, y1 = v[q + 1].y;
            float x2 = v[q + 2].x, y2 = v[q + 2].y, x3 = v[q + 3].x, y3 = v[q + 3].y;
            pair_fmac8(x0, y0, x1, y1, x2, y2, x3, y3, sgn);
            mag[q] = fmaf(x0, x0, y0 * y0);
            mag[q + 1] = fmaf(x1, x1, y1 * y1);
            mag[q + 2] = fmaf(x2, x2, y2 * y2);
            mag[q + 3] = fmaf(x3, x3, y3 * y3);
        }
        if (p && u == 0) mag[0] = -1.0f;         // lag -M is not part of the 'full' output
        if (is_halo) {
            float4* hp = reinterpret_cast<float4*>(halo + ((rb * 8 + wave) * 4 + hl) * 16);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                hp[q4] = make_float4(mag[4 * q4], mag[4 * q4 + 1], mag[4 * q4 + 2], mag[4 * q4 + 3]);
        }
        float tmax = mag[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) tmax = fmaxf(tmax, mag[q]);
        int qa = 16, qb = 16, qc = 16, qd = 16;
        argsel4<12>(qa, qb, qc, qd, mag[12], mag[13], mag[14], mag[15], tmax);   // descending: lower slots win
        argsel4<8>(qa, qb, qc, qd, mag[8], mag[9], mag[10], mag[11], tmax);
        argsel4<4>(qa, qb, qc, qd, mag[4], mag[5], mag[6], mag[7], tmax);
        argsel4<0>(qa, qb, qc, qd, mag[0], mag[1], mag[2], mag[3], tmax);
        const int qsel = min(min(qa, qb), min(qc, qd));
        const int kq = kbase + qsel * 256;
        const float wmax = wave_max_f32(tmax);
        const int kw = wave_min_i32(tmax == wmax ? kq : 0x7fffffff);
        int ts, qs;
        {
            const int par = (kw >= kM - 1) ? 0 : 1;
            const int n = par ? (kw + 1) : (kw - (kM - 1));
            ts = 2 * (n & 255) + par;
            qs = n >> 8;
        }
        const int ls = ts & 63;
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
        ++npend;
        ++npair;
    };
    auto pair = [&](const f2 (&a)[16], const f2 (&s)[16], int out_idx, auto prefetch) __attribute__((always_inline)) {
        pair_h1(a, s, seq, prefetch);
        barrier_hook(false);
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
    // ---- anchors 1..B-2, halves staggered across SIMD pairs (k_win)
    {
        const int M2 = (B - 1) * (B - 2) / 2;
        const bool late_h2 = (wave >> 1) & 1;
        auto j_of = [&](int i, int s) -> int { return (i & 1) ? (B - 1 - s) : (i + 1 + s); };
        int ci = 1, cs = 0;
        int ni = 1, ns = 1;
        if (ns >= B - 1 - ni) { ++ni; ns = 0; }
        auto h1_of = [&](int hi, int hs, int tr) __attribute__((always_inline)) {
            int pi = hi, ps = hs + 1;
            if (ps >= B - 1 - pi) { ++pi; ps = 0; }
            const bool valid = pi + 1 < B;
            const bool new_anchor = pi != hi;
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
    seq = 0;
    barrier_hook(true);
    }   // next window of this workgroup
}

}  // namespace pk
}  // namespace rmx
