// kwin2.hpp -- the fused window kernel of the tuned N = 4096 path with TWO resident anchor spectra.
//
// Same transform code, tables, LDS images, peak search and scratch layout as k_win (kwin.hpp); what changes is the pair
// schedule of a window, and with it the global-memory traffic.  Healthy-data ablations of k_win on MI355X
// (tools/probe/kwin_bench.hip, DESIGN.md section 6.2: the registers keep a valid spectrum, only the request is skipped)
// put 27-29 % of its launch time on that traffic -- spectrum loads of the anchor loop 14 %, spectrum stores 10 %,
// window samples 5 % -- although all of it is served by L2 / the memory-side cache: 2.4 MB per window, 5.7 TB/s.
//
//   k_win    one resident anchor X_i, every pair streams its X_j:             27 spectrum loads + 7 stores per window (B = 8)
//   k_win2   two resident anchors (X_a, X_a+1), every streamed X_j serves BOTH: 12 spectrum loads + 6 stores
//
//   phase 1  X_0 and X_1 stay in registers (X_1 is never stored); each X_e, e >= 2, is transformed once, stored once
//            and used at once for (0, e) and (1, e);
//   phase 2  blocks a = 2, 4, ...: anchors X_a, X_a+1 are loaded once, pair (a, a+1) needs no stream, then every
//            X_j, j = B-1 ... a+2, is loaded once and used for (a, j) and (a+1, j).
// The pair code always multiplies the stream registers sb by the anchor registers sa: the second resident spectrum
// waits in sc and the two are swapped (32 v_swap) between the two pairs of a stream, so there is ONE copy of the
// pair code per half order, as in k_win.  tools/model_kwin2_schedule.py executes this schedule symbolically for
// B = 2 ... 32 (every pair once, from the right registers, nothing overwritten while still needed).
#pragma once
#include "../../radio-mapper_amd/csrc/kwin.hpp"

namespace rmx {

constexpr int kLdsWin2Bytes = kLdsWinBytes;

// STAG: the half-order staggering of k_win (1 = SIMD pairs {a+1, a+3} run h1 of pair m+1 before h2 of pair m).
// TW2LDS: the thread's TW2 row is read from LDS just in time (8 ds_read_b128 per transform) instead of living in 30 VGPRs.
// ABL:  timing-only ablation bits (wrong results, healthy data): 1 no spectrum loads in phase 2, 2 no wave reductions /
//       records, 4 no spectrum stores, 8 no sample prefetch.
template <bool U8, int STAG, int ABL = 0, bool TW2LDS = false>
__global__ __launch_bounds__(kThreads, 2) void k_win2(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                      const float4* __restrict__ tw1_g,
                                                      const float2* __restrict__ tw2_g, int n_buoys,
                                                      long first_window, float out_scale,
                                                      int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                                      float* __restrict__ peak, int n_win) {
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
    const float4* tw2row = reinterpret_cast<const float4*>(tw2_lds + (u & 15) * kTw2RowF2);
    const float sgn = p ? -1.0f : 1.0f;
    const int kbase = p ? (u - 1) : (u + kM - 1);
    const int hl = lane < 2 ? lane : lane - 60;            // halo row of lanes 0,1,62,63
    const bool is_halo = lane < 2 || lane >= 62;
    __syncthreads();
    C16 tw2r;   // this thread's TW2 row W_256^(n0*k1) in registers (see k_win)
    if constexpr (!TW2LDS) {
        const float2* rowf2 = reinterpret_cast<const float2*>(tw2row);
        tw2r.set(0, 1.0f, 0.0f);
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float2 w = rowf2[4 * (q & 3) + (q >> 2) - 1];
            tw2r.set(q, w.x, w.y);
        }
    }

    for (int wl = blockIdx.x; wl < n_win; wl += gridDim.x) {
    C16 sa, sb, sc;   // the anchor in use, the streamed X_j, the other resident anchor
    const long wbase = (long)blockIdx.x * B;
    const long obase = (first_window + wl) * (long)n_pairs;
    int seq = 0;         // transform counter: selects the exchange image
    int npair = 0;       // pair counter: selects the record slot (ring of kResSlots)
    int npend = 0;       // pairs whose records await a resolve

    auto barrier_hook = [&](bool flush) __attribute__((always_inline)) {
        __syncthreads();
        if (npend == kResBatch || (flush && npend > 0)) {
            if (wave == (seq & 7))
                resolve_batch(lane, red, halo, oidx, (npair - npend) & (kResSlots - 1), npend, obase, out_scale, lag_int,
                              lag_frac, peak);
            npend = 0;
        }
    };
    auto mul_w32_odd = [&](float2 (&v)[16]) __attribute__((always_inline)) {
        if (p) {
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float2 w = w32(q);
                float x = v[q].x, y = v[q].y;
                cmul_inplace(x, y, w.x, w.y);
                v[q].x = x;
                v[q].y = y;
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
    const int samp_bytes = U8 ? 2 : 8;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(iq_v)) + (first_window + wl) * (long)B * kM * samp_bytes, 0,
        B * kM * samp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(spec) + wbase * (long)(8 * kThreads * 16), 0, B * (8 * kThreads * 16), 0x00020000);
    const int xoff = u * samp_bytes, soff = t * 16;
    auto load_x = [&](C16& d, int b) __attribute__((always_inline)) {
        if constexpr (U8) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                d.re[q] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(xs, xoff, (b * kM + q * 256) * 2, 0));
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(xs, xoff, (b * kM + q * 256) * 8, 0);
                d.set(q, __uint_as_float(r.x), __uint_as_float(r.y));
            }
        }
    };
    auto load_x_part = [&](C16& d, int b, auto part) __attribute__((always_inline)) {
        constexpr int G = decltype(part)::value;
        if constexpr (ABL & 8) return;   // timing-only
        if constexpr (U8) {
#pragma unroll
            for (int q = 4 * G; q < 4 * G + 4; ++q)
                d.re[q] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(xs, xoff, (b * kM + q * 256) * 2, 0));
        } else {
#pragma unroll
            for (int q = 4 * G; q < 4 * G + 4; ++q) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(xs, xoff, (b * kM + q * 256) * 8, 0);
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
    auto load_spec_part = [&](C16& d, int b, auto part) __attribute__((always_inline)) {
        constexpr int G = decltype(part)::value;
#pragma unroll
        for (int j = 2 * G; j < 2 * G + 2; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(ss, soff, (b * 8 + j) * (kThreads * 16), 0);
            d.set(2 * j, __uint_as_float(w.x), __uint_as_float(w.y));
            d.set(2 * j + 1, __uint_as_float(w.z), __uint_as_float(w.w));
        }
    };
    auto store_spec = [&](const C16& d, int b) __attribute__((always_inline)) {
        if constexpr (ABL & 4) return;   // timing-only
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float e0 = d.re[2 * j], e1 = d.im[2 * j], e2 = d.re[2 * j + 1], e3 = d.im[2 * j + 1];
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
            const u32x4 w = {__float_as_uint(e0), __float_as_uint(e1), __float_as_uint(e2), __float_as_uint(e3)};
            // whole offset in the VGPR, immediate soffset 0: see the store-data hazard note in kwin.hpp
            __builtin_amdgcn_raw_buffer_store_b128(w, ss, soff + (b * 8 + j) * (kThreads * 16), 0, 0);
        }
    };
    auto fwd = [&](C16& xc) __attribute__((always_inline)) {
        float2* img = (seq & 1) ? img1 : img0;
        float2 x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = xc.get(q);
        mul_w32_odd(x);
        dft16(x);
        mul_tw1(x, tw1);
        xchg_a_write(img, x, t);
        barrier_hook(false);
        xchg_b_read(img, x, t);
        dft16(x);
        float4 r0, r1;
        if constexpr (TW2LDS) { r0 = tw2row[0]; r1 = tw2row[1]; }   // ahead of the exchange reads (see dft16_tw_row_l1)
        xchg_bc_write_b(img, x, t);
        wave_lds_order();
        xchg_bc_read_c(img, x, t);
        if constexpr (TW2LDS) dft16_tw_row(x, tw2row, r0, r1);
        else dft16_tw<true>(x, tw2r);
#pragma unroll
        for (int q = 0; q < 16; ++q) xc.set(q, x[q].x, x[q].y);
        ++seq;
    };
    // ---- the pieces of one pair ------------------------------------------------------------------
    // h1a: product X_j conj(X_i) merged into the role-C pass; its outputs go to this wave's own regions of image `tr & 1`
    auto pair_h1a = [&](float2 (&v)[16], const C16& a, const C16& s, int tr, auto prefetch) __attribute__((always_inline)) {
        float2* img = (tr & 1) ? img1 : img0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_float2(s.im[q], s.re[q]);
        dft16_tw_l1<false>(v, a);
#pragma unroll
        for (int q = 0; q < 16; q += 4)
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q + 1].x), "+v"(v[q + 1].y), "+v"(v[q + 2].x),
                         "+v"(v[q + 2].y), "+v"(v[q + 3].x), "+v"(v[q + 3].y));
        __builtin_amdgcn_sched_barrier(0);
        float2* wb = img + (u >> 4) * kBcHalf + (u & 15) * kBcRow + p;
        dft16_layer2_emit(v, [&](auto kac, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                 __attribute__((always_inline)) {
            constexpr int ka = decltype(kac)::value;
            wb[2 * ka] = make_float2(x0.x, x0.y);
            wb[2 * (ka + 4)] = make_float2(x1.x, x1.y);
            wb[2 * (ka + 8)] = make_float2(x2.x, x2.y);
            wb[2 * (ka + 12)] = make_float2(x3.x, x3.y);
            prefetch(kac);
        });
    };
    // wave-local reads of h1 (role-B side of the B<->C image)
    auto pair_h1r = [&](float2 (&v)[16], int tr, float4& r0, float4& r1) __attribute__((always_inline)) {
        const float2* img = (tr & 1) ? img1 : img0;
        if constexpr (TW2LDS) { r0 = tw2row[0]; r1 = tw2row[1]; }
        wave_lds_order();
        xchg_bc_read_b(img, v, t);
    };
    // h1b: role-B pass, stores into exchange image `tr & 1` (own half-wave regions)
    auto pair_h1b = [&](float2 (&v)[16], int tr, float4 r0, float4 r1) __attribute__((always_inline)) {
        float2* img = (tr & 1) ? img1 : img0;
        if constexpr (TW2LDS) dft16_tw_row_l1(v, tw2row, r0, r1);
        else dft16_tw_l1<true>(v, tw2r);
        float2* xb = img + (u >> 4) * kBcHalf + (u & 15) * 2 + p;
        dft16_layer2_emit(v, [&](auto kac, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                 __attribute__((always_inline)) {
            constexpr int ka = decltype(kac)::value;
            xb[ka * 32] = make_float2(x0.x, x0.y);
            xb[(ka + 4) * 32] = make_float2(x1.x, x1.y);
            xb[(ka + 8) * 32] = make_float2(x2.x, x2.y);
            xb[(ka + 12) * 32] = make_float2(x3.x, x3.y);
        });
    };
    auto pair_h1 = [&](const C16& a, const C16& s, int tr, auto prefetch) __attribute__((always_inline)) {
        float2 v[16];
        float4 r0, r1;
        pair_h1a(v, a, s, tr, prefetch);
        pair_h1r(v, tr, r0, r1);
        pair_h1b(v, tr, r0, r1);
    };
    // h2: exchange reads (all waves' regions of image `tr & 1`) | first layer of the role-A pass | the rest
    auto pair_h2r = [&](float2 (&v)[16], int tr) __attribute__((always_inline)) {
        const float2* img = (tr & 1) ? img1 : img0;
        xchg_a_read(img, v, t);
    };
    auto pair_h2a = [&](float2 (&v)[16]) __attribute__((always_inline)) {
        dft4_tw<false>(v[0], v[4], v[8], v[12], tw1[0], tw1[4], tw1[8], tw1[12]);
        dft4_tw<false>(v[1], v[5], v[9], v[13], tw1[1], tw1[5], tw1[9], tw1[13]);
        dft4_tw<false>(v[2], v[6], v[10], v[14], tw1[2], tw1[6], tw1[10], tw1[14]);
        dft4_tw<false>(v[3], v[7], v[11], v[15], tw1[3], tw1[7], tw1[11], tw1[15]);
    };
    auto pair_h2b = [&](float2 (&v)[16], int out_idx) __attribute__((always_inline)) {
        const int rb = npair & (kResSlots - 1);
        dft16_layer2(v);
        mul_w32_odd(v);
        pair_fmac8(v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y, sgn);
        pair_fmac8(v[4].x, v[4].y, v[5].x, v[5].y, v[6].x, v[6].y, v[7].x, v[7].y, sgn);
        pair_fmac8(v[8].x, v[8].y, v[9].x, v[9].y, v[10].x, v[10].y, v[11].x, v[11].y, sgn);
        pair_fmac8(v[12].x, v[12].y, v[13].x, v[13].y, v[14].x, v[14].y, v[15].x, v[15].y, sgn);
        float mag[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) mag[q] = fmaf(v[q].x, v[q].x, v[q].y * v[q].y);
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
        argsel4<12>(qa, qb, qc, qd, mag[12], mag[13], mag[14], mag[15], tmax);
        argsel4<8>(qa, qb, qc, qd, mag[8], mag[9], mag[10], mag[11], tmax);
        argsel4<4>(qa, qb, qc, qd, mag[4], mag[5], mag[6], mag[7], tmax);
        argsel4<0>(qa, qb, qc, qd, mag[0], mag[1], mag[2], mag[3], tmax);
        const int qsel = min(min(qa, qb), min(qc, qd));
        const int kq = kbase + qsel * 256;
        if constexpr (ABL & 2) {   // timing-only: no wave reductions, no records
            if (tmax == 12345.678f && kq == 77) lag_int[0] = 1;
            ++npend; ++npair;
            return;
        }
        const float wmax = wave_max_f32(tmax);
        const int kw = wave_min_i32(tmax == wmax ? kq : 0x7fffffff);
        int ts, qs;
        k_to_owner(kw, ts, qs);
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
    auto pair_h2 = [&](int tr, int out_idx) __attribute__((always_inline)) {
        float2 v[16];
        pair_h2r(v, tr);
        pair_h2a(v);
        pair_h2b(v, out_idx);
    };
    auto pair = [&](const C16& a, const C16& s, int out_idx, auto prefetch) __attribute__((always_inline)) {
        pair_h1(a, s, seq, prefetch);
        barrier_hook(false);
        pair_h2(seq, out_idx);
        ++seq;
    };
    auto out_of = [&](int i, int j) -> int { return i * B - (i * (i + 1)) / 2 + (j - i - 1); };

    // ---- requests issued inside a pair's h1a (after its last read of sa / sb).  Every conditional request is ONE-SIDED
    // (request or keep the old registers): given a choice between two kinds of loads into the same registers hipcc issues
    // both speculatively and selects between their results, i.e. waits for them on the spot.
    //   spectra: kind 2 = X_a0 -> sb;  kind 3 = next block: X_a0 -> sa, X_a0+1 -> sb;  0 = nothing
    auto request_spec = [&](int kind, int a0, auto part) __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
        if (kind == 3) load_spec_part(sa, a0, part);
        if (kind >= 2) load_spec_part(sb, kind == 3 ? a0 + 1 : a0, part);
    };
    auto swap_ac = [&]() __attribute__((always_inline)) {   // the other resident anchor becomes sa
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float r = sa.re[q], i = sa.im[q];
            sa.re[q] = sc.re[q]; sa.im[q] = sc.im[q];
            sc.re[q] = r; sc.im[q] = i;
        }
    };
    auto copy_bc = [&]() __attribute__((always_inline)) {   // sc = sb (the block's second anchor arrives in sb)
#pragma unroll
        for (int q = 0; q < 16; ++q) { sc.re[q] = sb.re[q]; sc.im[q] = sb.im[q]; }
    };

    // ---- phase F: the forward spectra, in the order 2, 3, ..., B-1, 0, 1.  X_2 ... X_B-1 go to the scratch (each is
    // streamed back once per block it is paired with); X_0 ends in sa and X_1 in sb, which is how a block starts.
    // One copy of the transform code (on sb); the next buoy's samples travel into sc meanwhile.
    auto order_of = [&](int st) -> int { return st + 2 < B ? st + 2 : st + 2 - B; };
    load_x(sb, order_of(0));
#pragma unroll 1
    for (int st = 0; st < B; ++st) {
        const int e = order_of(st);
        if (st + 1 < B) load_x(sc, order_of(st + 1));
        cvt_x(sb);
        fwd(sb);
        if (e >= 2) store_spec(sb, e);
        if (e == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { sa.re[q] = sb.re[q]; sa.im[q] = sb.im[q]; }
        }
        if (st + 1 < B) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { sb.re[q] = sc.re[q]; sb.im[q] = sc.im[q]; }
        }
    }
    int flipped = 0;   // 1: sa holds the block's SECOND anchor
    // ---- phase P: blocks of two resident anchors (a, a+1), a = 0, 2, ...; streams j = B-1 ... a+2, each used twice.
    // Pair sequence of a block: pos 0 = (a, a+1); pos 2k-1, 2k = the two pairs of stream j = B-k.
    {
        const int M2 = B * (B - 1) / 2;
        const bool late_h2 = STAG == 1 ? ((wave >> 1) & 1) : STAG == 2 ? (wave & 1) : STAG == 3 ? (wave >> 2) :
                             STAG == 4 ? ((wave ^ (wave >> 2)) & 1) : (STAG == 5);
        int ga = 0, gpos = 0;                            // generator state: block, position in the block
        int n_out = 0, n_pre = 0, n_kind = 0, n_a0 = 0;  // descriptor of the NEXT pair (pre: 1 swap, 2 copy sb -> sc)
        auto gen = [&]() __attribute__((always_inline)) {
            const int n = B - ga - 2;
            const bool nxt_block = ga + 3 <= B - 1;
            n_kind = 0; n_a0 = 0;
            if (gpos == 0) {
                n_pre = 2; flipped = 0;
                n_out = out_of(ga, ga + 1);
                if (n > 0) { n_kind = 2; n_a0 = B - 1; }
                else if (nxt_block) { n_kind = 3; n_a0 = ga + 2; }
            } else {
                const int k = (gpos + 1) >> 1, r = (gpos + 1) & 1;
                if (r == 0) n_pre = 0;
                else {
                    n_pre = 1; flipped ^= 1;
                    if (k < n) { n_kind = 2; n_a0 = B - k - 1; }
                    else if (nxt_block) { n_kind = 3; n_a0 = ga + 2; }
                }
                n_out = out_of(ga + flipped, B - k);
            }
            ++gpos;
            if (gpos > 2 * n) { gpos = 0; ga += 2; }
        };
        auto h1_next = [&](int tr) __attribute__((always_inline)) {
            if (n_pre == 1) swap_ac();
            else if (n_pre == 2) copy_bc();
            const int kind = n_kind, a0 = n_a0;
            pair_h1(sa, sb, tr, [&](auto part) __attribute__((always_inline)) { request_spec(kind, a0, part); });
        };
        int c_out = 0;
        if (M2 > 0) { gen(); h1_next(seq); }
        for (int m = 0; m < M2; ++m) {
            c_out = n_out;
            const bool has_next = m + 1 < M2;
            if (has_next) gen();
            barrier_hook(false);
            if (late_h2) {
                if (has_next) h1_next(seq + 1);
                pair_h2(seq, c_out);
            } else {
                pair_h2(seq, c_out);
                if (has_next) h1_next(seq + 1);
            }
            ++seq;
        }
    }
    seq = 0;
    barrier_hook(true);
    }   // next window of this workgroup
}

}  // namespace rmx
