// kwin8k.hpp (experiments) -- the first two versions of the N = 8192 kernel on the fused N = 4096 network, round 4: k_win8k
// (no resident anchor) and k_win8ka (both anchor halves in registers, TW1 rebuilt from four powers).  Neither beat
// g_win_scr14 (LABNOTES.md R4.6); round 5's third version, k_win8kl (one anchor half resident in LDS), does and lives in the
// product: radio-mapper_amd/csrc/kwin8k.hpp, which also holds the helpers shared with these two (constants, twists, records).
// Compiled only with -DRMX_EXPERIMENTS (option kwin8k = 2 then selects k_win8k; tools/probe/k8_bench.hip times all three).
#pragma once
#include "../../radio-mapper_amd/csrc/kwin8k.hpp"

namespace rmx {
namespace k8 {

template <bool U8>
__global__ __launch_bounds__(kThreads, 2) void k_win8k(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                       const float4* __restrict__ tw1_g,     // [2 halves][8][512]
                                                       const float2* __restrict__ tw2_g, int n_buoys,
                                                       const Pair2* __restrict__ pairs, int n_pairs, long first_window,
                                                       float out_scale, int* __restrict__ lag_int,
                                                       float* __restrict__ lag_frac, float* __restrict__ peak, int n_win,
                                                       int stag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* img0 = reinterpret_cast<float2*>(smem);
    float2* img1 = reinterpret_cast<float2*>(smem + kLdsWinImg);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLds8Tw2);
    float* halo = reinterpret_cast<float*>(smem + kLds8Halo);
    float4* red = reinterpret_cast<float4*>(smem + kLds8Red);
    int* oidx = reinterpret_cast<int*>(smem + kLds8Oidx);
    Pair2* plist = reinterpret_cast<Pair2*>(smem + kLds8Pairs);

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;
    const int B = n_buoys;

    load_tw2_to_lds_grouped(tw2_lds, tw2_g, t);
    if (pairs)
        for (int q = t; q < n_pairs; q += kThreads) plist[q] = pairs[q];
    const float4* tw2row = reinterpret_cast<const float4*>(tw2_lds + (t & 15) * kTw2RowF2);
    const int loc_m0[2] = {__builtin_amdgcn_readfirstlane(wave * kLocWave),
                           __builtin_amdgcn_readfirstlane(kLdsWinImg + wave * kLocWave)};
    const int loc_rd = wave * kLocWave + loc_read_off(lane);
    const float sgn = p ? -1.0f : 1.0f;
    const int kbase = u + 4096 * p - 1;                    // 'full' index of value sg: kbase + 256 sg + (sg >= 16 ? 4096 : 0)
    const int hl = lane < 2 ? lane : lane - 60;
    const bool is_halo = lane < 2 || lane >= 62;
    __syncthreads();

    const int samp_bytes = U8 ? 2 : 8;
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(tw1_g)), 0, 2 * 8 * kThreads * 16, 0x00020000);
    const int soff = t * 16;
    float2 tw1[16];
    // the TW1 table of half h into the twiddle registers (eight 16-byte requests; behind the last use of the other half's)
    auto load_tw1_half = [&](int h) __attribute__((always_inline)) {
        int bo = h * (8 * kThreads * 16);
        asm volatile("" : "+s"(bo));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(twr, soff, bo + j * (kThreads * 16), 0);
            tw1[2 * j] = make_float2(__uint_as_float(w.x), __uint_as_float(w.y));
            tw1[2 * j + 1] = make_float2(__uint_as_float(w.z), __uint_as_float(w.w));
        }
    };

    for (int wl = blockIdx.x; wl < n_win; wl += gridDim.x) {
    C16 sa, sb;      // the two spectra of the next / current pair transform (X_i,h and X_j,h); sample buffers in phase 1
    C16 ev;          // e_0 of the current pair
#ifdef K8_SHARE   // timing probe (results wrong): workgroups b, b + K8_SHARE, ... use ONE scratch region -- K8_SHARE = 8: all 32
                  // workgroups of an XCD stream the same 1 MiB (8 buoys) out of their L2
    const long wbase = (long)(blockIdx.x % K8_SHARE) * B * 2;
#else
    const long wbase = (long)blockIdx.x * B * 2;                     // this workgroup's scratch: [b][h] x 64 KiB
#endif
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

    // ---- phase 1: forward transforms, half-major (one TW1 table per half) ------------------------------------------
    load_x(sa, 0, 0);
    load_x(sb, 0, kN8 / 2);
#ifdef K8_NO_FWD
    if (n_win < 0)
#endif
    for (int h = 0; h < 2; ++h) {
        load_tw1_half(h);
        for (int b = 0; b < B; ++b) {
            float2* img = (seq & 1) ? img1 : img0;
            cvt_x(sa);
            cvt_x(sb);
            float2 x[16];
            if (h == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) x[q] = make_float2(fmaf(sgn, sb.re[q], sa.re[q]), fmaf(sgn, sb.im[q], sa.im[q]));
            } else {   // a + (-1)^p (-i) b
#pragma unroll
                for (int q = 0; q < 16; ++q) x[q] = make_float2(fmaf(sgn, sb.im[q], sa.re[q]), fmaf(-sgn, sb.re[q], sa.im[q]));
            }
            // (opaque from here on: the uint8 and the complex64 build must run the SAME arithmetic on these values)
#pragma unroll
            for (int q = 0; q < 16; q += 4)
                asm volatile("" : "+v"(x[q].x), "+v"(x[q].y), "+v"(x[q + 1].x), "+v"(x[q + 1].y), "+v"(x[q + 2].x),
                             "+v"(x[q + 2].y), "+v"(x[q + 3].x), "+v"(x[q + 3].y));
            {   // the next transform's samples travel during this one (the sample registers are free behind the fold)
                const int nb = b + 1 < B ? b + 1 : 0;
#ifndef K8_NO_SAMPLE
                if (b + 1 < B || h == 0) {
                    load_x(sa, nb, 0);
                    load_x(sb, nb, kN8 / 2);
                }
#endif
            }
            twist(x, h);
            dft16(x);
            mul_tw1(x, tw1);
            xchg_a2_write(img, x, t);
            __syncthreads();
            xchg_b2_read(img, x, t);
            dft16(x);
            const float4 r0 = tw2row[0], r1 = tw2row[1];
            loc_write16(loc_m0[seq & 1], x);
            wave_lds_order();
            loc_read16(smem + (seq & 1) * kLdsWinImg + loc_rd, x);
            dft16_tw_row(x, tw2row, r0, r1);
#ifndef K8_NO_STORE
            store_spec(x, 2 * b + h);
#endif
            ++seq;
        }
    }

    // ---- phase 2: pair transforms tr = 2 pair + h ------------------------------------------------------------------
    auto pair_h1 = [&](const C16& a, const C16& s, int tr, auto prefetch) __attribute__((always_inline)) {
        float2* img = (tr & 1) ? img1 : img0;
        float2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_float2(s.im[q], s.re[q]);
        dft16_tw_l1<false>(v, a);
#pragma unroll
        for (int q = 0; q < 16; q += 4)
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q + 1].x), "+v"(v[q + 1].y), "+v"(v[q + 2].x),
                         "+v"(v[q + 2].y), "+v"(v[q + 3].x), "+v"(v[q + 3].y));
        __builtin_amdgcn_sched_barrier(0);
        dft16_layer2_emit(v, [&](auto kac, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                 __attribute__((always_inline)) {
            constexpr int ka = decltype(kac)::value;
            loc_write4<ka, ka + 4, ka + 8, ka + 12>(loc_m0[tr & 1], x0, x1, x2, x3);
            prefetch(kac);
        });
        const float4 r0 = tw2row[0], r1 = tw2row[1];
        wave_lds_order();
        loc_read16(smem + (tr & 1) * kLdsWinImg + loc_rd, v);
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
    auto pair_h2 = [&](int tr, int out_idx) __attribute__((always_inline)) {
        const float2* img = (tr & 1) ? img1 : img0;
        const int h = tr & 1;
        float2 v[16];
        xchg_a2_read(img, v, t);
        dft16_tw<false>(v, tw1);
#ifndef K8_NO_TW1
        load_tw1_half(h ^ 1);                   // the other half's table travels while the rest of this piece runs
#endif
#ifndef K8_NO_TWIST
        twist(v, h);
#endif
        pair_fmac8(v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y, sgn);
        pair_fmac8(v[4].x, v[4].y, v[5].x, v[5].y, v[6].x, v[6].y, v[7].x, v[7].y, sgn);
        pair_fmac8(v[8].x, v[8].y, v[9].x, v[9].y, v[10].x, v[10].y, v[11].x, v[11].y, sgn);
        pair_fmac8(v[12].x, v[12].y, v[13].x, v[13].y, v[14].x, v[14].y, v[15].x, v[15].y, sgn);
        if (h == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) ev.set(q, v[q].x, v[q].y);
            return;
        }
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
    };

    {
        const int T2 = 2 * n_pairs;
        const bool late_h2 = stag == 1 ? ((wave >> 1) & 1) : stag == 2 ? (wave & 1) : stag == 3 ? (wave >> 2) :
                             stag == 4 ? ((wave ^ (wave >> 2)) & 1) : (stag == 5);
        // spectra of transform tr into sa / sb, one eighth per call (tr beyond the end: transform 0's, into dead registers)
        // (i, j) of pair q: the default list (pairs == nullptr) is the nested loop i < j -- two counters, no memory access;
        // a custom list is read from its copy in LDS (a scalar load here would share lgkmcnt with the exchanges)
        int ni = 0, nj = 1, nq = 0;                       // the pair the NEXT request belongs to (default list)
        auto pair_of = [&](int q) -> Pair2 {
            if (!pairs) {
                while (nq < q) { ++nq; if (++nj >= B) { ++ni; nj = ni + 1; } }
                if (q < nq) { ni = 0; nj = 1; nq = 0; }
                return Pair2{ni, nj};
            }
            const int2 v = reinterpret_cast<const int2*>(plist)[q];
            return Pair2{__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y)};
        };
        auto h1_of = [&](int tr) __attribute__((always_inline)) {
            const int nx = tr + 1 < T2 ? tr + 1 : 0;
            const Pair2 pr = pair_of(nx >> 1);
            const int si = __builtin_amdgcn_readfirstlane(2 * pr.i + (nx & 1)), sj = __builtin_amdgcn_readfirstlane(2 * pr.j + (nx & 1));
            pair_h1(sa, sb, tr, [&](auto part) __attribute__((always_inline)) {
#ifndef K8_NO_SPEC
#ifdef K8_ANCHOR_HOT      // timing experiment (wrong results): every anchor request hits one L2-resident spectrum
                load_spec_part(sa, 0, part);
#elif !defined(K8_NO_ANCHOR)
                load_spec_part(sa, si, part);
#endif
                load_spec_part(sb, sj, part);
#endif
            });
        };
        if (T2 > 0) {
            Pair2 pr = pair_of(0);
            pr.i = __builtin_amdgcn_readfirstlane(pr.i);
            pr.j = __builtin_amdgcn_readfirstlane(pr.j);
            load_spec_part(sa, 2 * pr.i, std::integral_constant<int, 0>{}); load_spec_part(sb, 2 * pr.j, std::integral_constant<int, 0>{});
            load_spec_part(sa, 2 * pr.i, std::integral_constant<int, 1>{}); load_spec_part(sb, 2 * pr.j, std::integral_constant<int, 1>{});
            load_spec_part(sa, 2 * pr.i, std::integral_constant<int, 2>{}); load_spec_part(sb, 2 * pr.j, std::integral_constant<int, 2>{});
            load_spec_part(sa, 2 * pr.i, std::integral_constant<int, 3>{}); load_spec_part(sb, 2 * pr.j, std::integral_constant<int, 3>{});
            load_spec_part(sa, 2 * pr.i, std::integral_constant<int, 4>{}); load_spec_part(sb, 2 * pr.j, std::integral_constant<int, 4>{});
            load_spec_part(sa, 2 * pr.i, std::integral_constant<int, 5>{}); load_spec_part(sb, 2 * pr.j, std::integral_constant<int, 5>{});
            load_spec_part(sa, 2 * pr.i, std::integral_constant<int, 6>{}); load_spec_part(sb, 2 * pr.j, std::integral_constant<int, 6>{});
            load_spec_part(sa, 2 * pr.i, std::integral_constant<int, 7>{}); load_spec_part(sb, 2 * pr.j, std::integral_constant<int, 7>{});
            load_tw1_half(0);                  // (phase 1 left half 1's table in the registers)
            h1_of(0);
        }
        for (int tr = 0; tr < T2; ++tr) {
            barrier_hook(false);
            const bool has_next = tr + 1 < T2;
            if (wave >= 4) __builtin_amdgcn_s_setprio(1);
            if (late_h2) {
                if (has_next) h1_of(tr + 1);
                if (wave >= 4) __builtin_amdgcn_s_setprio(0);
                pair_h2(tr, tr >> 1);
            } else {
                pair_h2(tr, tr >> 1);
                if (wave >= 4) __builtin_amdgcn_s_setprio(0);
                if (has_next) h1_of(tr + 1);
            }
            ++seq;
        }
    }
    seq = 0;
    barrier_hook(true);
    }   // next window of this workgroup
}

// ---- second version: both anchor halves resident, TW1 twiddles rebuilt from four powers per transform -------------
// k_win8k's limit is its scratch traffic (no resident anchor).  Here X_i's two halves stay in registers over the
// anchor's run of pairs (64 VGPRs) and only the streamed X_j,h travels per transform; the 32 registers come from the TW1
// table: tw1[k0] = c_h w^k0 with w = W_4096^u, built per transform from w, w^2, w^4, w^8 (exact table values) and
// c_h = 2^-6 W_16384^((h + 2p) u) -- 15 complex products, no deeper than four, three per first-layer group.
//   tws[t] = {w, w^2, w^4, w^8, c_0, c_1}  (six float2 per thread: build_tables8ka)
struct TwBase { float2 w1, w2, w4, w8, c[2]; };

// first butterfly layer group q0 of the 16 inputs pre-twiddled by c w^k0, k0 = q0 + 4 m
template <int Q0>
__device__ __forceinline__ void jit_group(float2 (&v)[16], const TwBase& tb, float2 c) {
    float2 t0 = c;
    if constexpr (Q0 == 1) t0 = cmul(c, tb.w1);
    if constexpr (Q0 == 2) t0 = cmul(c, tb.w2);
    if constexpr (Q0 == 3) t0 = cmul(cmul(c, tb.w1), tb.w2);
    const float2 t4 = cmul(t0, tb.w4), t8 = cmul(t0, tb.w8), t12 = cmul(t4, tb.w8);
    dft4_tw<false>(v[Q0], v[Q0 + 4], v[Q0 + 8], v[Q0 + 12], t0, t4, t8, t12);
}
__device__ __forceinline__ void dft16_tw_jit(float2 (&v)[16], const TwBase& tb, float2 c) {
    jit_group<0>(v, tb, c);
    jit_group<1>(v, tb, c);
    jit_group<2>(v, tb, c);
    jit_group<3>(v, tb, c);
    dft16_layer2(v);
}
__device__ __forceinline__ void mul_tw1_jit(float2 (&x)[16], const TwBase& tb, float2 c) {
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) {
        float2 t0 = c;
        if (q0 == 1) t0 = cmul(c, tb.w1);
        if (q0 == 2) t0 = cmul(c, tb.w2);
        if (q0 == 3) t0 = cmul(cmul(c, tb.w1), tb.w2);
        const float2 t4 = cmul(t0, tb.w4), t8 = cmul(t0, tb.w8), t12 = cmul(t4, tb.w8);
        x[q0] = cmul(x[q0], t0);
        x[q0 + 4] = cmul(x[q0 + 4], t4);
        x[q0 + 8] = cmul(x[q0 + 8], t8);
        x[q0 + 12] = cmul(x[q0 + 12], t12);
    }
}

template <bool U8>
__global__ __launch_bounds__(kThreads, 2) void k_win8ka(const void* __restrict__ iq_v, float4* __restrict__ spec,
                                                        const float4* __restrict__ tws_g,     // [3][512] float4 = TwBase per thread
                                                        const float2* __restrict__ tw2_g, int n_buoys,
                                                        const Pair2* __restrict__ pairs, int n_pairs, long first_window,
                                                        float out_scale, int* __restrict__ lag_int,
                                                        float* __restrict__ lag_frac, float* __restrict__ peak, int n_win,
                                                        int stag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* img0 = reinterpret_cast<float2*>(smem);
    float2* img1 = reinterpret_cast<float2*>(smem + kLdsWinImg);
    float2* tw2_lds = reinterpret_cast<float2*>(smem + kLds8Tw2);
    float* halo = reinterpret_cast<float*>(smem + kLds8Halo);
    float4* red = reinterpret_cast<float4*>(smem + kLds8Red);
    int* oidx = reinterpret_cast<int*>(smem + kLds8Oidx);
    Pair2* plist = reinterpret_cast<Pair2*>(smem + kLds8Pairs);

    const int t = threadIdx.x;
    const int p = t & 1, u = t >> 1;
    const int lane = t & 63, wave = t >> 6;
    const int B = n_buoys;

    load_tw2_to_lds_grouped(tw2_lds, tw2_g, t);
    if (pairs)
        for (int q = t; q < n_pairs; q += kThreads) plist[q] = pairs[q];
    const float4* tw2row = reinterpret_cast<const float4*>(tw2_lds + (t & 15) * kTw2RowF2);
    const int loc_m0[2] = {__builtin_amdgcn_readfirstlane(wave * kLocWave),
                           __builtin_amdgcn_readfirstlane(kLdsWinImg + wave * kLocWave)};
    const int loc_rd = wave * kLocWave + loc_read_off(lane);
    const float sgn = p ? -1.0f : 1.0f;
    const int kbase = u + 4096 * p - 1;
    const int hl = lane < 2 ? lane : lane - 60;
    const bool is_halo = lane < 2 || lane >= 62;
    TwBase tb;
    {
        const float4 a = tws_g[t], b = tws_g[kThreads + t], c = tws_g[2 * kThreads + t];
        tb.w1 = make_float2(a.x, a.y); tb.w2 = make_float2(a.z, a.w);
        tb.w4 = make_float2(b.x, b.y); tb.w8 = make_float2(b.z, b.w);
        tb.c[0] = make_float2(c.x, c.y); tb.c[1] = make_float2(c.z, c.w);
    }
    __syncthreads();

    const int samp_bytes = U8 ? 2 : 8;
    const int soff = t * 16;

    for (int wl = blockIdx.x; wl < n_win; wl += gridDim.x) {
    C16 sa0, sa1;    // the anchor's two halves
    C16 sb;          // the streamed X_j,h of the next / current transform (sample buffer b in phase 1)
    C16 ev;          // e_0 of the current pair (sample buffer a in phase 1)
    const long wbase = (long)blockIdx.x * B * 2;
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
    auto load_spec_part = [&](C16& d, int sidx, auto part) __attribute__((always_inline)) {
        constexpr int J = decltype(part)::value;
        int bo = __builtin_amdgcn_readfirstlane(sidx) * (8 * kThreads * 16);
        asm volatile("" : "+s"(bo));
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(ss, soff, bo + J * (kThreads * 16), 0);
        d.set(2 * J, __uint_as_float(w.x), __uint_as_float(w.y));
        d.set(2 * J + 1, __uint_as_float(w.z), __uint_as_float(w.w));
    };
    auto load_spec_all = [&](C16& d, int sidx) __attribute__((always_inline)) {
        load_spec_part(d, sidx, std::integral_constant<int, 0>{}); load_spec_part(d, sidx, std::integral_constant<int, 1>{});
        load_spec_part(d, sidx, std::integral_constant<int, 2>{}); load_spec_part(d, sidx, std::integral_constant<int, 3>{});
        load_spec_part(d, sidx, std::integral_constant<int, 4>{}); load_spec_part(d, sidx, std::integral_constant<int, 5>{});
        load_spec_part(d, sidx, std::integral_constant<int, 6>{}); load_spec_part(d, sidx, std::integral_constant<int, 7>{});
    };
    auto store_spec = [&](const float2 (&x)[16], int sidx) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float e0 = x[2 * j].x, e1 = x[2 * j].y, e2 = x[2 * j + 1].x, e3 = x[2 * j + 1].y;
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
            const u32x4 w = {__float_as_uint(e0), __float_as_uint(e1), __float_as_uint(e2), __float_as_uint(e3)};
            __builtin_amdgcn_raw_buffer_store_b128(w, ss, soff + (sidx * 8 + j) * (kThreads * 16), 0, 0);
        }
    };
    auto twist = [&](float2 (&v)[16], auto hc) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        if constexpr (H == 0) {
            if (p) mul_twist<2>(v);
        } else {
            if (p) mul_twist<3>(v);
            else mul_twist<1>(v);
        }
    };

    // ---- phase 1: forward transforms (sample buffers: ev = a, sb = b) ----------------------------------------------
    load_x(ev, 0, 0);
    load_x(sb, 0, kN8 / 2);
    auto fwd_half = [&](auto hc) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        for (int b = 0; b < B; ++b) {
            float2* img = (seq & 1) ? img1 : img0;
            cvt_x(ev);
            cvt_x(sb);
            float2 x[16];
            if constexpr (H == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) x[q] = make_float2(fmaf(sgn, sb.re[q], ev.re[q]), fmaf(sgn, sb.im[q], ev.im[q]));
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) x[q] = make_float2(fmaf(sgn, sb.im[q], ev.re[q]), fmaf(-sgn, sb.re[q], ev.im[q]));
            }
#pragma unroll
            for (int q = 0; q < 16; q += 4)
                asm volatile("" : "+v"(x[q].x), "+v"(x[q].y), "+v"(x[q + 1].x), "+v"(x[q + 1].y), "+v"(x[q + 2].x),
                             "+v"(x[q + 2].y), "+v"(x[q + 3].x), "+v"(x[q + 3].y));
            {
                const int nb = b + 1 < B ? b + 1 : 0;
                if (b + 1 < B || H == 0) {
                    load_x(ev, nb, 0);
                    load_x(sb, nb, kN8 / 2);
                }
            }
            twist(x, hc);
            dft16(x);
            mul_tw1_jit(x, tb, tb.c[H]);
            xchg_a2_write(img, x, t);
            __syncthreads();
            xchg_b2_read(img, x, t);
            dft16(x);
            const float4 r0 = tw2row[0], r1 = tw2row[1];
            loc_write16(loc_m0[seq & 1], x);
            wave_lds_order();
            loc_read16(smem + (seq & 1) * kLdsWinImg + loc_rd, x);
            dft16_tw_row(x, tw2row, r0, r1);
            store_spec(x, 2 * b + H);
            ++seq;
        }
    };
    fwd_half(std::integral_constant<int, 0>{});
    fwd_half(std::integral_constant<int, 1>{});

    // ---- phase 2 ------------------------------------------------------------------------------------------------------
    auto pair_h1 = [&](const C16& a, const C16& s, int tr, auto prefetch) __attribute__((always_inline)) {
        float2* img = (tr & 1) ? img1 : img0;
        float2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_float2(s.im[q], s.re[q]);
        dft16_tw_l1<false>(v, a);
#pragma unroll
        for (int q = 0; q < 16; q += 4)
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q + 1].x), "+v"(v[q + 1].y), "+v"(v[q + 2].x),
                         "+v"(v[q + 2].y), "+v"(v[q + 3].x), "+v"(v[q + 3].y));
        __builtin_amdgcn_sched_barrier(0);
        dft16_layer2_emit(v, [&](auto kac, const float2& x0, const float2& x1, const float2& x2, const float2& x3)
                                 __attribute__((always_inline)) {
            constexpr int ka = decltype(kac)::value;
            loc_write4<ka, ka + 4, ka + 8, ka + 12>(loc_m0[tr & 1], x0, x1, x2, x3);
            prefetch(kac);
        });
        const float4 r0 = tw2row[0], r1 = tw2row[1];
        wave_lds_order();
        loc_read16(smem + (tr & 1) * kLdsWinImg + loc_rd, v);
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
    auto pair_h2 = [&](int tr, int out_idx, auto hc) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        const float2* img = (tr & 1) ? img1 : img0;
        float2 v[16];
        xchg_a2_read(img, v, t);
        dft16_tw_jit(v, tb, tb.c[H]);
        twist(v, hc);
        pair_fmac8(v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y, sgn);
        pair_fmac8(v[4].x, v[4].y, v[5].x, v[5].y, v[6].x, v[6].y, v[7].x, v[7].y, sgn);
        pair_fmac8(v[8].x, v[8].y, v[9].x, v[9].y, v[10].x, v[10].y, v[11].x, v[11].y, sgn);
        pair_fmac8(v[12].x, v[12].y, v[13].x, v[13].y, v[14].x, v[14].y, v[15].x, v[15].y, sgn);
        if constexpr (H == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) ev.set(q, v[q].x, v[q].y);
        } else {
            float m0[16], m1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float tx = p ? v[q].y : v[q].x, ty = p ? -v[q].x : v[q].y;
                const float ax = ev.re[q] + tx, ay = ev.im[q] + ty;
                const float bx = ev.re[q] - tx, by = ev.im[q] - ty;
                m0[q] = fmaf(ax, ax, ay * ay);
                m1[q] = fmaf(bx, bx, by * by);
            }
            if (p == 0 && u == 0) m1[0] = -1.0f;
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

    if (n_pairs > 0) {
        const bool late_h2 = stag == 1 ? ((wave >> 1) & 1) : stag == 2 ? (wave & 1) : stag == 3 ? (wave >> 2) :
                             stag == 4 ? ((wave ^ (wave >> 2)) & 1) : (stag == 5);
        // pair q -> (i, j): default list by two counters (requests come in order), a custom list from its LDS copy
        int ci = 0, cj = 1;                              // pair cq of the default list
        int cq = 0;
        auto pair_of = [&](int q) -> Pair2 {
            if (!pairs) {
                if (q < cq) { ci = 0; cj = 1; cq = 0; }
                while (cq < q) { ++cq; if (++cj >= B) { ++ci; cj = ci + 1; } }
                return Pair2{ci, cj};
            }
            const int2 v = reinterpret_cast<const int2*>(plist)[q];
            return Pair2{__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y)};
        };
        int cur_i, cur_j;                                // the pair whose transforms are being issued
        {
            const Pair2 pr = pair_of(0);
            cur_i = __builtin_amdgcn_readfirstlane(pr.i);
            cur_j = __builtin_amdgcn_readfirstlane(pr.j);
        }
        load_spec_all(sa0, 2 * cur_i);
        load_spec_all(sa1, 2 * cur_i + 1);
        load_spec_all(sb, 2 * cur_j);
        // h1 of transform (pair q, half H): behind its last reads, the stream of the NEXT transform goes into sb and, when
        // the next pair has another anchor, that anchor's half H into the register set this transform has just finished with
        auto h1_of = [&](int q, auto hc) __attribute__((always_inline)) {
            constexpr int H = decltype(hc)::value;
            const int nq = q + 1 < n_pairs ? q + 1 : 0;
            const Pair2 nx = pair_of(nq);
            const int ni = __builtin_amdgcn_readfirstlane(nx.i), nj = __builtin_amdgcn_readfirstlane(nx.j);
            const bool new_anchor = ni != cur_i;
            const int sj = H == 0 ? 2 * cur_j + 1 : 2 * nj;      // next transform's stream
            const int tr = 2 * q + H;
            auto pf = [&](auto part) __attribute__((always_inline)) {
                if constexpr (decltype(part)::value == 0) {
                    if (new_anchor) {
                        if constexpr (H == 0) load_spec_all(sa0, 2 * ni);
                        else load_spec_all(sa1, 2 * ni + 1);
                    }
                }
                load_spec_part(sb, sj, part);
            };
            if constexpr (H == 0) pair_h1(sa0, sb, tr, pf);
            else pair_h1(sa1, sb, tr, pf);
            if constexpr (H == 1) { cur_i = ni; cur_j = nj; }
        };
        h1_of(0, std::integral_constant<int, 0>{});
        for (int q = 0; q < n_pairs; ++q) {
            // (pair q, half 0) finishes while (pair q, half 1) starts ...
            barrier_hook(false);
            if (wave >= 4) __builtin_amdgcn_s_setprio(1);
            if (late_h2) {
                h1_of(q, std::integral_constant<int, 1>{});
                if (wave >= 4) __builtin_amdgcn_s_setprio(0);
                pair_h2(2 * q, q, std::integral_constant<int, 0>{});
            } else {
                pair_h2(2 * q, q, std::integral_constant<int, 0>{});
                if (wave >= 4) __builtin_amdgcn_s_setprio(0);
                h1_of(q, std::integral_constant<int, 1>{});
            }
            ++seq;
            // ... and (pair q, half 1) finishes while (pair q + 1, half 0) starts
            barrier_hook(false);
            const bool has_next = q + 1 < n_pairs;
            if (wave >= 4) __builtin_amdgcn_s_setprio(1);
            if (late_h2) {
                if (has_next) h1_of(q + 1, std::integral_constant<int, 0>{});
                if (wave >= 4) __builtin_amdgcn_s_setprio(0);
                pair_h2(2 * q + 1, q, std::integral_constant<int, 1>{});
            } else {
                pair_h2(2 * q + 1, q, std::integral_constant<int, 1>{});
                if (wave >= 4) __builtin_amdgcn_s_setprio(0);
                if (has_next) h1_of(q + 1, std::integral_constant<int, 0>{});
            }
            ++seq;
        }
    }
    seq = 0;
    barrier_hook(true);
    }   // next window of this workgroup
}

// host: TW1 of k_win8k, one full table per half ([h][8][512] float4, register order as build_tables: W_4096^(u k0) W_16384^((h + 2p) u) 2^-6)
inline void build_tables8k(std::vector<float4>& tw1) {
    const double two_pi = 6.283185307179586476925286766559;
    tw1.resize(2 * 8 * kThreads);
    for (int h = 0; h < 2; ++h) {
        std::vector<float2> t1(16 * kThreads);
        for (int t = 0; t < kThreads; ++t) {
            const int p = t & 1, u = t >> 1;
            for (int k0 = 0; k0 < 16; ++k0) {
                // W_4096^(u k0) * W_16384^((h + 2p) u), scaled by 2^-6 like k_win's table
                const double ang = -two_pi * (double)((u * k0) % kM) / (double)kM - two_pi * (double)((h + 2 * p) * u) / 16384.0;
                t1[k0 * kThreads + t] = make_float2((float)(std::cos(ang) * kTw1Scale), (float)(std::sin(ang) * kTw1Scale));
            }
        }
        for (int j = 0; j < 8; ++j)
            for (int t = 0; t < kThreads; ++t) {
                const float2 a = t1[(2 * j) * kThreads + t], b = t1[(2 * j + 1) * kThreads + t];
                tw1[(h * 8 + j) * kThreads + t] = make_float4(a.x, a.y, b.x, b.y);
            }
    }
}


// host: per-thread twiddle bases of k_win8ka ([3][512] float4: {w, w^2}, {w^4, w^8}, {c_0, c_1})
inline void build_tables8ka(std::vector<float4>& tws) {
    const double two_pi = 6.283185307179586476925286766559;
    tws.resize(3 * kThreads);
    for (int t = 0; t < kThreads; ++t) {
        const int p = t & 1, u = t >> 1;
        auto wpow = [&](int e) { const double a = -two_pi * (double)((u * e) % kM) / (double)kM; return make_float2((float)std::cos(a), (float)std::sin(a)); };
        const float2 w1 = wpow(1), w2 = wpow(2), w4 = wpow(4), w8 = wpow(8);
        float2 c[2];
        for (int h = 0; h < 2; ++h) {
            const double a = -two_pi * (double)((h + 2 * p) * u) / 16384.0;
            c[h] = make_float2((float)(std::cos(a) * kTw1Scale), (float)(std::sin(a) * kTw1Scale));
        }
        tws[t] = make_float4(w1.x, w1.y, w2.x, w2.y);
        tws[kThreads + t] = make_float4(w4.x, w4.y, w8.x, w8.y);
        tws[2 * kThreads + t] = make_float4(c[0].x, c[0].y, c[1].x, c[1].y);
    }
}

}  // namespace k8
}  // namespace rmx
