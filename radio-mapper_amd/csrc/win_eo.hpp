// win_eo.hpp -- whole windows of N = 16384 samples (buoy_node.py:364, the reference's capture length) in one kernel.
//
// The zero-padded transform (L = 32768 complex = 256 KiB) does not fit the LDS, but its two HALVES do: with a window
// zero-padded from N to L = 2N the even bins are FFT_N(x) and the odd bins FFT_N(x W_L^n) -- the decomposition k_win
// uses at N = 4096 -- and each of those is the 16384-point LDS-resident transform of g_win_scr14 (generic_path.hpp:
// 512 threads x two butterflies per pass, 16 x 16 x 16 x 4, first-pass twiddles in registers).  So a window is
//     forward   per buoy: E = FFT_N(x), O = FFT_N(x W_L^n), both parked in the workgroup's scratch (cache resident)
//     inverse   per pair: e = IFFT_N(E_j conj E_i), kept in REGISTERS (32 complex per thread: the register file is the
//               second buffer the LDS cannot be), o = IFFT_N(O_j conj O_i), then r[n] = e[n] + W_L^-n o[n] and
//               r[n + N] = e[n] - W_L^-n o[n] in registers, |.|^2 of both into the LDS buffer (x = lag n, y = lag n - N)
//               for the neighbour taps, thread / wave / workgroup argmax with numpy's tie rule, 12 bytes out.
// W_L^n with n = i + 1024 m (i = the butterfly's index, m = its slot) is w_i W_32^m: w_i = W_L^i is a per-thread pair
// of registers (table `twl`), W_32^m are constants; in the inverse's (re, im)-swapped domain the multiplication by
// conj(W) becomes one by W (swap(z conj w) = swap(z) w).
// Against the four-step path this length ran before: no spectra, products or tile records in HBM -- 8 buoys x 256
// windows move the samples, 2 MB of scratch writes and 14 MB of scratch reads per window through L2 / the memory-side
// cache instead of 36.7 MB through HBM.
#pragma once
#include "generic_path.hpp"

namespace rmx {
namespace gen {

constexpr size_t kWinEoLds = ((size_t)lp(16384) + 15 * 64 + 15 * 4 + 16) * 8;   // as g_win_scr14

template <bool U8>
__global__ __launch_bounds__(512, 2) void g_win_eo15(const void* __restrict__ iq, float4* __restrict__ scratch,
                                                  const float2* __restrict__ tw, const float2* __restrict__ twl,
                                                  int n_buoys, long n_windows, long first_window, float fwd_scale,
                                                  float out_scale, const GPair* __restrict__ pairs, int n_pairs,
                                                  int* __restrict__ lag_int, float* __restrict__ lag_frac,
                                                  float* __restrict__ peak) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    constexpr int LOGR = 14, R = 1 << LOGR, N = R, NT = 512, Q = R >> 4;          // half transforms of R = N points
    constexpr int off10 = 0, off6 = 15 * 64, tab_total = 15 * 64 + 15 * 4;
    const int tid = threadIdx.x;
    float2* x = reinterpret_cast<float2*>(gsm);
    float2* tab = x + lp(R);
    float* sv = reinterpret_cast<float*>(tab + tab_total);
    int* sk = reinterpret_cast<int*>(sv + 16);
    for (int e = tid; e < 15 * 64; e += NT) tab[off10 + e] = tw_full(tw, ((e & 63) << 4) * ((e >> 6) + 1), R >> 1);
    for (int e = tid; e < 15 * 4; e += NT) tab[off6 + e] = tw_full(tw, ((e & 3) << 8) * ((e >> 2) + 1), R >> 1);
    // First-pass twiddles W_R^(i k), k = 1..15, of butterfly i: g_win_scr14 keeps all 2 x 15 of them in registers; here the
    // registers hold 64 of samples / of the even half as well, so only w = W_R^i stays (2 x 2 registers) and its powers
    // are rebuilt where a pass needs them -- 14 complex products, none deeper than four multiplications (a few float32
    // ulp; every other twiddle of the transform comes exact from a table).
    float2 w1[2], wl[2];                            // W_R^i and W_L^i of this thread's two butterflies, i = tid + 512 u
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        w1[u] = tw_full(tw, tid + u * NT, R >> 1);
        wl[u] = twl[tid + u * NT];
    }
    auto powers = [&](int u, float2 (&p)[16]) __attribute__((always_inline)) {   // p[k] = w^k, k = 1..15
        float2 a = w1[u];
        asm volatile("" : "+v"(a.x), "+v"(a.y));    // (opaque: keeps hipcc from hoisting the 14 products out of every loop)
        p[1] = a;
        p[2] = g_cmul(p[1], p[1]);   p[3] = g_cmul(p[2], p[1]);   p[4] = g_cmul(p[2], p[2]);   p[5] = g_cmul(p[4], p[1]);
        p[6] = g_cmul(p[3], p[3]);   p[7] = g_cmul(p[4], p[3]);   p[8] = g_cmul(p[4], p[4]);   p[9] = g_cmul(p[8], p[1]);
        p[10] = g_cmul(p[5], p[5]);  p[11] = g_cmul(p[8], p[3]);  p[12] = g_cmul(p[6], p[6]);  p[13] = g_cmul(p[8], p[5]);
        p[14] = g_cmul(p[7], p[7]);  p[15] = g_cmul(p[8], p[7]);
    };
    __syncthreads();
    const LdsIO lds{x};
    // All global traffic through buffer descriptors (base in SGPRs + ONE VGPR offset + scalar / immediate offsets): with
    // flat 64-bit addresses the 32 sample requests and the 32 scratch accesses of a transform each kept their own pair
    // of address registers alive and the kernel spilled 147 VGPRs.
    constexpr int SB = U8 ? 2 : 8;
    // scratch of this workgroup: [b][half][u][kk][tid] float4
    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(scratch) + (long)blockIdx.x * n_buoys * (32L * NT * 16), 0, n_buoys * (32 * NT * 16), 0x00020000);
    const int soff = tid * 16, xoff = tid * SB;
    auto scr_ld = [&](int b, int half, int u, int kk) __attribute__((always_inline)) -> float4 {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ss, soff, ((((b * 2 + half) * 2 + u) * 8 + kk) * NT) * 16, 0);
        return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
    };
    for (long w = blockIdx.x; w < n_windows; w += gridDim.x) {
        const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(iq)) + w * (long)n_buoys * N * SB, 0, n_buoys * N * SB, 0x00020000);
        auto load_in = [&](float2 (&d)[2][16], int b) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    if constexpr (U8) {
                        const unsigned r = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(xs, xoff, (b * N + u * NT + m * Q) * 2, 0);
                        d[u][m] = make_float2((float)(r & 0xffu) - 127.5f, (float)((r >> 8) & 0xffu) - 127.5f);
                    } else {
                        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(xs, xoff, (b * N + u * NT + m * Q) * 8, 0);
                        d[u][m] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
                    }
                }
        };
        float2 nx[2][16];
        load_in(nx, 0);
        for (int b = 0; b < n_buoys; ++b) {
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {     // first pass over the whole window: elements i + m R/16, i = tid + 512 u
                    float2 v[16];
#pragma unroll
                    for (int m = 0; m < 16; ++m) v[m] = nx[u][m];
                    if (half) {                   // odd bins: x[n] W_L^n = x[n] w_i W_32^m; w_i rides on the outputs
#pragma unroll
                        for (int m = 1; m < 16; ++m) v[m] = g_cmul(v[m], w32(m));
                    }
                    dft16(v);
                    float2 pw[16];
                    powers(u, pw);
                    const auto hd = lds.open(tid + u * NT);
                    if (half) {
                        hd.st(0, g_cmul(v[0], wl[u]));
#pragma unroll
                        for (int k = 1; k < 16; ++k) hd.st(brev_m<4>(k) * Q, g_cmul(g_cmul(v[k], pw[k]), wl[u]));
                    } else {
                        hd.st(0, v[0]);
#pragma unroll
                        for (int k = 1; k < 16; ++k) hd.st(brev_m<4>(k) * Q, g_cmul(v[k], pw[k]));
                    }
                }
                if (half && b + 1 < n_buoys) load_in(nx, b + 1);                    // travels during the rest of this transform
                __syncthreads();
                dif_pass<4, 0, 2>(LOGR, 10, tab + off10, tid, NT, lds, lds);
                xsync<true>();
                dif_pass<4, 0, 2>(LOGR, 6, tab + off6, tid, NT, lds, lds);
                xsync<true>();
#pragma unroll
                for (int u = 0; u < 2; ++u) {     // the 16 neighbours of butterfly i: four radix-4 butterflies, outputs to the scratch
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
                    for (int kk = 0; kk < 8; ++kk) {
                        // (whole offset in the VGPR, immediate soffset: the store-data hazard note in kwin.hpp)
                        float e0 = v[2 * kk].x * fwd_scale, e1 = v[2 * kk].y * fwd_scale, e2 = v[2 * kk + 1].x * fwd_scale,
                              e3 = v[2 * kk + 1].y * fwd_scale;
                        asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
                        const u32x4 pk = {__float_as_uint(e0), __float_as_uint(e1), __float_as_uint(e2), __float_as_uint(e3)};
                        __builtin_amdgcn_raw_buffer_store_b128(pk, ss, soff + ((((b * 2 + half) * 2 + u) * 8 + kk) * NT) * 16, 0, 0);
                    }
                }
                __syncthreads();                  // the next transform's first pass overwrites x
            }
        }
        const long obase = (first_window + w) * (long)n_pairs;
        for (int q = 0; q < n_pairs; ++q) {
            const GPair pr = pairs[q];
            float2 ev[2][16];                     // the even half's r-contribution e[n] ((re, im)-swapped), n = tid + 512 u + 1024 m
            float best = -1.0f;
            int bk = 0x7fffffff;
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {     // X_j conj(X_i) into the neighbour butterflies of the inverse
                    float2 v[16];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const float4 a = scr_ld(pr.j, half, u, kk), c = scr_ld(pr.i, half, u, kk);
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
                __syncthreads();                  // the pass over the whole window reads every wave's blocks
                // last pass: outputs n = tid + 512 u + 1024 m.  Even half: e[n] stays in ev.  Odd half: |e + o'|^2 (lag n) and
                // |e - o'|^2 (lag n - N) replace it there (x = lag n, y = lag n - N)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float2 v[16], pw[16];
                    powers(u, pw);
                    const auto h = lds.open(tid + u * NT);
                    {
                        float2 e = h.ld(0);
                        if (half) e = g_cmulc(e, wl[u]);
                        v[0] = make_float2(e.y, e.x);
                    }
#pragma unroll
                    for (int k = 1; k < 16; ++k) {
                        float2 e = g_cmulc(h.ld(brev_m<4>(k) * Q), pw[k]);
                        if (half) e = g_cmulc(e, wl[u]);
                        v[k] = make_float2(e.y, e.x);
                    }
                    dft16(v);
                    if (!half) {
#pragma unroll
                        for (int m = 0; m < 16; ++m) ev[u][m] = v[m];
                    } else {
#pragma unroll
                        for (int m = 0; m < 16; ++m) {
                            const float2 o = m ? g_cmul(v[m], w32(m)) : v[m];   // swapped domain: * W, not * conj(W)
                            const float lx = ev[u][m].x + o.x, ly = ev[u][m].y + o.y;
                            const float hx = ev[u][m].x - o.x, hy = ev[u][m].y - o.y;
                            ev[u][m] = make_float2(lx * lx + ly * ly, hx * hx + hy * hy);
                        }
                    }
                }
                __syncthreads();                  // every thread has read x: the next half's butterflies / the |r|^2 may overwrite it
                if (half) {
                    // 'full' index: lag n -> n + N - 1, lag n - N -> n - 1 (n = 0: lag -N is not part of the output).  All "hi"
                    // candidates come before all "lo" ones, each ascending in s = u + 2 m (n = tid + 512 s)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int m = 0; m < 16; ++m) x[lp(tid + u * NT + m * Q)] = ev[u][m];
                    if (tid == 0) ev[0][0].y = -2.0f;
                    float tb = ev[0][0].x;
#pragma unroll
                    for (int m = 0; m < 16; ++m)
#pragma unroll
                        for (int u = 0; u < 2; ++u) tb = fmaxf(tb, fmaxf(ev[u][m].x, ev[u][m].y));
                    int ssel = 63;                 // position in scan order: hi s = 0..31, then lo s = 0..31
#pragma unroll
                    for (int s2 = 31; s2 >= 0; --s2) ssel = ev[s2 & 1][s2 >> 1].x == tb ? 32 + s2 : ssel;   // lo, descending
#pragma unroll
                    for (int s2 = 31; s2 >= 0; --s2) ssel = ev[s2 & 1][s2 >> 1].y == tb ? s2 : ssel;        // hi: lower indices win
                    best = tb;
                    bk = tid + (ssel & 31) * NT + (ssel >= 32 ? N - 1 : -1);
                }
            }
            group_argmax<NT>(best, bk, sv, sk, tid, 0);      // (its barrier also publishes the |r|^2 for the taps)
            if (tid == 0) {
                const float bpk = sqrtf(best) * out_scale;
                float frac = 0.0f;
                auto tap = [&](int k) -> float { return k >= N - 1 ? x[lp(k - (N - 1))].x : x[lp(k + 1)].y; };
                if (bk > 0 && bk < 2 * N - 2) frac = parabola(sqrtf(tap(bk - 1)) * out_scale, bpk, sqrtf(tap(bk + 1)) * out_scale);
                lag_int[obase + q] = bk - (N - 1);
                lag_frac[obase + q] = frac;
                peak[obase + q] = bpk;
            }
            __syncthreads();                      // x is rewritten (sv: behind the next inverse's barrier)
        }
    }
}

}  // namespace gen
}  // namespace rmx
