// fft_r16.hpp -- register-resident radix-16 building blocks for the LDS-resident xcorr kernels.
//
// Work decomposition for one length-L = 2M transform, M = 4096 = 16^3 (gfx950, wave64):
//   * the L-point transform is split by bin parity into two M-point transforms (bins 2k and 2k+1);
//     for a window zero-padded from N = M to L = 2M samples that first radix-2 stage is free:
//         X[2k]   = FFT_M(x[n])            X[2k+1] = FFT_M(x[n] * W_L^n)
//   * 512 threads; thread t = 2u + p holds 16 complex points of sub-transform p (the parity sits on
//     lane bit 0, so that the last radix-2 stage of the inverse is one DPP quad_perm exchange);
//   * three radix-16 passes in registers, two exchanges through LDS between them:
//         role A  u = 16*n1 + n0   slots n2 / k0      (time side)
//         role B  u = 16*k0 + n0   slots n1 / k1
//         role C  u = 16*k0 + k1   slots n0 / k2      (frequency side; bin = 2(k0+16k1+256k2)+p)
//     exchange A<->B crosses waves (workgroup barrier); exchange B<->C stays inside one 32-lane
//     half wave (fixed k0), so it needs no barrier, only the wave's own LDS ordering.
//   * the inverse is the same forward blocks in reverse order applied to (im, re)-swapped data:
//     swap o F o swap = conj(F) for every linear block, and |.| does not see the final swap.
// All LDS images are bank-conflict free for ds_write_b64 / ds_read_b64 (see the index functions).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace rmx {

constexpr int kM = 4096;       // sub-transform length (= window length N of the LDS path)
constexpr int kL = 2 * kM;     // zero-padded transform length
constexpr int kThreads = 512;  // threads per workgroup
constexpr int kSlots = 16;     // complex points per thread

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// 4-point DFT (W4 = -i), natural order in and out, in place.
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2);
    const float2 t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = make_float2(t1.x + t3.y, t1.y - t3.x);  // t1 - i*t3
    a3 = make_float2(t1.x - t3.y, t1.y + t3.x);  // t1 + i*t3
}

// W16^e for the inner twiddles of the 16-point DFT (e = q0*ka, q0, ka in 0..3).
#define RMX_C1 0.92387953251128675613f  /* cos(pi/8) */
#define RMX_S1 0.38268343236508977173f  /* sin(pi/8) */
#define RMX_RH 0.70710678118654752440f   /* sqrt(1/2) */

template <int E>
__device__ __forceinline__ float2 mul_w16(float2 a) {
    if constexpr (E == 0) return a;
    else if constexpr (E == 1) return cmul(a, make_float2(RMX_C1, -RMX_S1));
    else if constexpr (E == 2) return make_float2((a.x + a.y) * RMX_RH, (a.y - a.x) * RMX_RH);
    else if constexpr (E == 3) return cmul(a, make_float2(RMX_S1, -RMX_C1));
    else if constexpr (E == 4) return make_float2(a.y, -a.x);
    else if constexpr (E == 6) return make_float2((a.y - a.x) * RMX_RH, -(a.x + a.y) * RMX_RH);
    else if constexpr (E == 9) return cmul(a, make_float2(-RMX_C1, RMX_S1));
    else return a;
}

// acc + a*w (4 FMAs)
__device__ __forceinline__ float2 cfma(float2 acc, float2 a, float2 w) {
    float re = fmaf(a.x, w.x, acc.x);
    re = fmaf(-a.y, w.y, re);
    float im = fmaf(a.x, w.y, acc.y);
    im = fmaf(a.y, w.x, im);
    return make_float2(re, im);
}
// 2*a - b (1 FMA per component): the second output of a twiddled radix-2 butterfly,
// t1 = A - w*s = 2A - (A + w*s), costs 2 instead of 4 instructions.
__device__ __forceinline__ float2 twice_minus(float2 a, float2 b) {
    return make_float2(fmaf(2.0f, a.x, -b.x), fmaf(2.0f, a.y, -b.y));
}
__device__ __forceinline__ void dft4_tail(float2& a0, float2& a1, float2& a2, float2& a3, float2 t0, float2 t1,
                                          float2 t2, float2 t3) {
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = make_float2(t1.x + t3.y, t1.y - t3.x);  // t1 - i*t3
    a3 = make_float2(t1.x - t3.y, t1.y + t3.x);  // t1 + i*t3
}
// DFT4 of (w0*a0, w1*a1, w2*a2, w3*a3): pre-twiddles merged into the first butterfly layer
// (28 instructions instead of 16 + 16; 24 when w0 == 1).
template <bool W0_IS_ONE>
__device__ __forceinline__ void dft4_tw(float2& a0, float2& a1, float2& a2, float2& a3, float2 w0, float2 w1,
                                        float2 w2, float2 w3) {
    const float2 A0 = W0_IS_ONE ? a0 : cmul(a0, w0);
    const float2 t0 = cfma(A0, a2, w2);
    const float2 t1 = twice_minus(A0, t0);
    const float2 A1 = cmul(a1, w1);
    const float2 t2 = cfma(A1, a3, w3);
    const float2 t3 = twice_minus(A1, t2);
    dft4_tail(a0, a1, a2, a3, t0, t1, t2, t3);
}

// component-wise swap (a struct copy of float2 can keep the array element in scratch memory)
__device__ __forceinline__ void swap2(float2& a, float2& b) {
    const float ax = a.x, ay = a.y;
    a.x = b.x; a.y = b.y;
    b.x = ax; b.y = ay;
}

// second radix-4 layer of the 16-point DFT with its constant inner twiddles W16^(q0*ka) merged
__device__ __forceinline__ void dft16_layer2(float2 (&v)[16]) {
    dft4(v[0], v[1], v[2], v[3]);                                                  // ka = 0
    dft4_tw<true>(v[4], v[5], v[6], v[7], make_float2(1.f, 0.f), make_float2(RMX_C1, -RMX_S1),
                  make_float2(RMX_RH, -RMX_RH), make_float2(RMX_S1, -RMX_C1));      // ka = 1: W1 W2 W3
    {                                                                              // ka = 2: W2 W4 W6
        float2 &a0 = v[8], &a1 = v[9], &a2 = v[10], &a3 = v[11];
        const float2 t0 = make_float2(a0.x + a2.y, a0.y - a2.x);                   // a0 + (-i) a2
        const float2 t1 = make_float2(a0.x - a2.y, a0.y + a2.x);
        const float2 A1 = make_float2((a1.x + a1.y) * RMX_RH, (a1.y - a1.x) * RMX_RH);
        const float2 t2 = cfma(A1, a3, make_float2(-RMX_RH, -RMX_RH));
        const float2 t3 = twice_minus(A1, t2);
        dft4_tail(a0, a1, a2, a3, t0, t1, t2, t3);
    }
    dft4_tw<true>(v[12], v[13], v[14], v[15], make_float2(1.f, 0.f), make_float2(RMX_S1, -RMX_C1),
                  make_float2(-RMX_RH, -RMX_RH), make_float2(-RMX_C1, RMX_S1));    // ka = 3: W3 W6 W9
    // un-transpose (pure register renaming): X[ka + 4kb] sits in v[4ka + kb]
    swap2(v[1], v[4]);
    swap2(v[2], v[8]);
    swap2(v[3], v[12]);
    swap2(v[6], v[9]);
    swap2(v[7], v[13]);
    swap2(v[11], v[14]);
}

// The same layer, handing each group's four outputs X[ka], X[ka+4], X[ka+8], X[ka+12] to `emit` as soon
// as they exist (ka as an integral_constant), with a scheduling fence after each group: the caller's
// LDS stores and global prefetch requests are then spread over the layer's arithmetic instead of
// hitting the LDS store path / the texture-address queue in one burst at its end (all eight waves
// of the workgroup run in barrier-aligned lockstep, so a burst is eight bursts at once).  v is left
// in layer order (not un-transposed).
template <class F>
__device__ __forceinline__ void dft16_layer2_emit(float2 (&v)[16], F&& emit) {
    dft4(v[0], v[1], v[2], v[3]);                                                  // ka = 0
    emit(std::integral_constant<int, 0>{}, v[0], v[1], v[2], v[3]);
    __builtin_amdgcn_sched_barrier(0);
    dft4_tw<true>(v[4], v[5], v[6], v[7], make_float2(1.f, 0.f), make_float2(RMX_C1, -RMX_S1),
                  make_float2(RMX_RH, -RMX_RH), make_float2(RMX_S1, -RMX_C1));      // ka = 1
    emit(std::integral_constant<int, 1>{}, v[4], v[5], v[6], v[7]);
    __builtin_amdgcn_sched_barrier(0);
    {                                                                              // ka = 2
        float2 &a0 = v[8], &a1 = v[9], &a2 = v[10], &a3 = v[11];
        const float2 t0 = make_float2(a0.x + a2.y, a0.y - a2.x);
        const float2 t1 = make_float2(a0.x - a2.y, a0.y + a2.x);
        const float2 A1 = make_float2((a1.x + a1.y) * RMX_RH, (a1.y - a1.x) * RMX_RH);
        const float2 t2 = cfma(A1, a3, make_float2(-RMX_RH, -RMX_RH));
        const float2 t3 = twice_minus(A1, t2);
        dft4_tail(a0, a1, a2, a3, t0, t1, t2, t3);
    }
    emit(std::integral_constant<int, 2>{}, v[8], v[9], v[10], v[11]);
    __builtin_amdgcn_sched_barrier(0);
    dft4_tw<true>(v[12], v[13], v[14], v[15], make_float2(1.f, 0.f), make_float2(RMX_S1, -RMX_C1),
                  make_float2(-RMX_RH, -RMX_RH), make_float2(-RMX_C1, RMX_S1));    // ka = 3
    emit(std::integral_constant<int, 3>{}, v[12], v[13], v[14], v[15]);
    __builtin_amdgcn_sched_barrier(0);
}

// 16-point DFT, X[k] = sum_q v[q] W16^(qk), natural order in, natural order out (150 instructions).
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    // q = 4*q1 + q0, k = ka + 4*kb.  Layer 1: DFT4 over q1 for each q0 -> y[q0][ka] in v[q0+4ka].
    dft4(v[0], v[4], v[8], v[12]);
    dft4(v[1], v[5], v[9], v[13]);
    dft4(v[2], v[6], v[10], v[14]);
    dft4(v[3], v[7], v[11], v[15]);
    dft16_layer2(v);
}

// 16-point DFT of (v[q] * w[q]): the per-slot pre-twiddles (TW1, TW2, W32, or a whole spectrum for
// the conj-multiply) are merged into layer 1.  W0_IS_ONE: w[0] == 1 (slot 0 of TW2 / W32).
template <bool W0_IS_ONE>
__device__ __forceinline__ void dft16_tw(float2 (&v)[16], const float2 (&w)[16]) {
    dft4_tw<W0_IS_ONE>(v[0], v[4], v[8], v[12], w[0], w[4], w[8], w[12]);
    dft4_tw<false>(v[1], v[5], v[9], v[13], w[1], w[5], w[9], w[13]);
    dft4_tw<false>(v[2], v[6], v[10], v[14], w[2], w[6], w[10], w[14]);
    dft4_tw<false>(v[3], v[7], v[11], v[15], w[3], w[7], w[11], w[15]);
    dft16_layer2(v);
}

// 16 complex values as two scalar arrays.  Long-lived register arrays (anchor / stream spectra) use
// this instead of float2[16]: hipcc's SROA leaves the tail of a float2 array in scratch memory as
// soon as one access is a struct copy or a vector-typed view, and every scratch access in the pair
// loop costs a vmcnt(0) that drains the spectra requested a pair ahead.
struct C16 {
    float re[16], im[16];
    __device__ __forceinline__ float2 get(int q) const { return make_float2(re[q], im[q]); }
    __device__ __forceinline__ void set(int q, float x, float y) { re[q] = x; im[q] = y; }
};
template <bool W0_IS_ONE>
__device__ __forceinline__ void dft16_tw_l1(float2 (&v)[16], const C16& w) {   // layer 1 only
    dft4_tw<W0_IS_ONE>(v[0], v[4], v[8], v[12], w.get(0), w.get(4), w.get(8), w.get(12));
    dft4_tw<false>(v[1], v[5], v[9], v[13], w.get(1), w.get(5), w.get(9), w.get(13));
    dft4_tw<false>(v[2], v[6], v[10], v[14], w.get(2), w.get(6), w.get(10), w.get(14));
    dft4_tw<false>(v[3], v[7], v[11], v[15], w.get(3), w.get(7), w.get(11), w.get(15));
}
template <bool W0_IS_ONE>
__device__ __forceinline__ void dft16_tw(float2 (&v)[16], const C16& w) {
    dft16_tw_l1<W0_IS_ONE>(v, w);
    dft16_layer2(v);
}

// 16-point DFT of (v[q] * tw[q]) with the twiddle row fetched just in time from LDS.  tw[0] is 1 (TW2
// rows) and is not stored: `row` holds the other 15 twiddles in layer-1 group order, s = 4*q0 + m - 1
// for tw[q0 + 4*m], packed two per float4 from a 16-byte aligned base, so the row is eight full
// ds_read_b128 (256 B/clk; the compiler pairs a row that starts on the unused tw[0] into
// ds_read2_b64, 128 B/clk) and only ~8 twiddle registers are live at a time.
// (f0, f1 = row[0], row[1] may be fetched by the caller BEFORE it issues the exchange reads that fill v:
// a wave's LDS reads return in issue order, so twiddles requested behind the sixteen exchange reads
// would hold the first butterfly group back until all sixteen have arrived)
__device__ __forceinline__ void dft16_tw_row_l1(float2 (&v)[16], const float4* row, float4 f0, float4 f1) {   // layer 1 only
    dft4_tw<true>(v[0], v[4], v[8], v[12], make_float2(1.0f, 0.0f), make_float2(f0.x, f0.y), make_float2(f0.z, f0.w),
                  make_float2(f1.x, f1.y));
    const float4 f2 = row[2], f3 = row[3];
    dft4_tw<false>(v[1], v[5], v[9], v[13], make_float2(f1.z, f1.w), make_float2(f2.x, f2.y), make_float2(f2.z, f2.w),
                   make_float2(f3.x, f3.y));
    const float4 f4 = row[4], f5 = row[5];
    dft4_tw<false>(v[2], v[6], v[10], v[14], make_float2(f3.z, f3.w), make_float2(f4.x, f4.y), make_float2(f4.z, f4.w),
                   make_float2(f5.x, f5.y));
    const float4 f6 = row[6], f7 = row[7];
    dft4_tw<false>(v[3], v[7], v[11], v[15], make_float2(f5.z, f5.w), make_float2(f6.x, f6.y), make_float2(f6.z, f6.w),
                   make_float2(f7.x, f7.y));
}
__device__ __forceinline__ void dft16_tw_row_l1(float2 (&v)[16], const float4* row) {
    dft16_tw_row_l1(v, row, row[0], row[1]);
}
__device__ __forceinline__ void dft16_tw_row(float2 (&v)[16], const float4* row, float4 f0, float4 f1) {
    dft16_tw_row_l1(v, row, f0, f1);
    dft16_layer2(v);
}
__device__ __forceinline__ void dft16_tw_row(float2 (&v)[16], const float4* row) {
    dft16_tw_row_l1(v, row);
    dft16_layer2(v);
}

// (x + iy) *= (wr + i wi) in place, 4 VALU instructions and one temporary, no result copies: used
// under the odd-lane branch for the W32^q factors (the compiler's version copies every result back).
__device__ __forceinline__ void cmul_inplace(float& x, float& y, float wr, float wi) {
    float tmp;
    asm volatile("v_mul_f32 %2, %0, %4\n\tv_mul_f32 %0, %0, %3\n\tv_fma_f32 %0, -%1, %4, %0\n\tv_fma_f32 %1, %1, %3, %2"
                 : "+v"(x), "+v"(y), "=&v"(tmp)
                 : "s"(wr), "s"(wi));
}

// Four of them in ONE asm statement (hipcc puts an s_nop between consecutive asm statements):
// (x_k + i y_k) *= (wr_k + i wi_k), k = 0..3.
__device__ __forceinline__ void cmul4_inplace(float& x0, float& y0, float& x1, float& y1, float& x2, float& y2,
                                              float& x3, float& y3, float2 w0, float2 w1, float2 w2, float2 w3) {
    float tmp;
#define RMX_CM(X, Y, WR, WI)                                                                              \
    "v_mul_f32 %8, %" #X ", %" #WI "\n\tv_mul_f32 %" #X ", %" #X ", %" #WR "\n\tv_fma_f32 %" #X ", -%" #Y \
    ", %" #WI ", %" #X "\n\tv_fma_f32 %" #Y ", %" #Y ", %" #WR ", %8\n\t"
    asm volatile(RMX_CM(0, 1, 9, 10) RMX_CM(2, 3, 11, 12) RMX_CM(4, 5, 13, 14) RMX_CM(6, 7, 15, 16)
                 : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2), "+v"(x3), "+v"(y3), "=&v"(tmp)
                 : "s"(w0.x), "s"(w0.y), "s"(w1.x), "s"(w1.y), "s"(w2.x), "s"(w2.y), "s"(w3.x), "s"(w3.y));
#undef RMX_CM
}

// W32^q, q = 0..15 (exp(-2*pi*i*q/32)): the per-slot part of the odd sub-transform's W_L^n.
__device__ __forceinline__ float2 w32(int q) {
    constexpr float c[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                             0.83146961230254523708f, 0.70710678118654752440f,
                             0.55557023301960222474f, 0.38268343236508977173f,
                             0.19509032201612826785f, 0.0f, -0.19509032201612826785f,
                             -0.38268343236508977173f, -0.55557023301960222474f,
                             -0.70710678118654752440f, -0.83146961230254523708f,
                             -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float s[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f,
                             0.55557023301960222474f, 0.70710678118654752440f,
                             0.83146961230254523708f, 0.92387953251128675613f,
                             0.98078528040323044913f, 1.0f, 0.98078528040323044913f,
                             0.92387953251128675613f, 0.83146961230254523708f,
                             0.70710678118654752440f, 0.55557023301960222474f,
                             0.38268343236508977173f, 0.19509032201612826785f};
    return make_float2(c[q], -s[q]);
}

// ---- LDS exchanges (complex index into a float2 array of kXchgF2 entries) -----------------------
// Both images give half wave k0 (32 lanes: fixed k0 in roles B and C) the SAME private region
// [k0*kBcHalf, (k0+1)*kBcHalf): a wave only ever overwrites LDS that itself (or nobody) still
// needs, so the only workgroup barrier of an exchange is the one between A-side and B-side accesses.
// A<->B image: idx = k0*kBcHalf + ((n1*16 + n0)*2 + p).  Role-A lanes are contiguous in it for a
// fixed slot k0; role-B lanes read/write one contiguous 256-B run per 32-lane half for a fixed slot
// n1: conflict free both ways.
constexpr int kBcRow = 34;
constexpr int kBcHalf = 16 * kBcRow;          // complex per half wave (>= 512)
constexpr int kXchgF2 = 16 * kBcHalf;         // complex in the whole exchange image

__device__ __forceinline__ void wave_lds_fence() {
    // orders this wave's LDS writes before its later LDS reads (same-wave exchange, no barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Same ordering without the wait: a wave's LDS instructions execute in issue order, so its reads
// queue behind its own writes (wavefront-scope fences are compiler-only on gfx9).
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void xchg_a_write(float2* lds, const float2 (&v)[16], int t) {
#pragma unroll
    for (int k0 = 0; k0 < 16; ++k0) lds[k0 * kBcHalf + t] = make_float2(v[k0].x, v[k0].y);
}
__device__ __forceinline__ void xchg_a_read(const float2* lds, float2 (&v)[16], int t) {
    // issue order = layer-1 consumption order (slots q0, q0+4, q0+8, q0+12 feed the q0-th DFT4)
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0)
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) v[q0 + 4 * q1] = lds[(q0 + 4 * q1) * kBcHalf + t];
}
// role B thread: u = 16*k0 + n0; slot = n1
__device__ __forceinline__ void xchg_b_write(float2* lds, const float2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, n0 = u & 15;
    float2* base = lds + k0 * kBcHalf + n0 * 2 + p;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) base[n1 * 32] = make_float2(v[n1].x, v[n1].y);
}
__device__ __forceinline__ void xchg_b_read(const float2* lds, float2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, n0 = u & 15;
    const float2* base = lds + k0 * kBcHalf + n0 * 2 + p;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = base[n1 * 32];
}
// B<->C image (inside one half wave, fixed k0): idx = k0*kBcHalf + k1*kBcRow + 2*n0 + p with rows of
// 32 complex padded to 34.  Role B (a = n0, slot k1) touches 32 contiguous complex per access; role
// C (a = k1, slot n0) touches 16 rows 34 complex apart, i.e. 16 distinct 16-B bank groups (68*a mod
// 64 = 4*a): conflict free both ways, and every slot is base + immediate offset (no per-slot
// address registers).
__device__ __forceinline__ void xchg_bc_write_b(float2* lds, const float2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, a = u & 15;
    float2* base = lds + k0 * kBcHalf + 2 * a + p;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) base[k1 * kBcRow] = make_float2(v[k1].x, v[k1].y);
}
__device__ __forceinline__ void xchg_bc_read_c(const float2* lds, float2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, a = u & 15;
    const float2* base = lds + k0 * kBcHalf + a * kBcRow + p;
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) v[n0] = base[2 * n0];
}
__device__ __forceinline__ void xchg_bc_write_c(float2* lds, const float2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, a = u & 15;
    float2* base = lds + k0 * kBcHalf + a * kBcRow + p;
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) base[2 * n0] = make_float2(v[n0].x, v[n0].y);
}
__device__ __forceinline__ void xchg_bc_read_b(const float2* lds, float2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, a = u & 15;
    const float2* base = lds + k0 * kBcHalf + 2 * a + p;
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0)
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) v[q0 + 4 * q1] = base[(q0 + 4 * q1) * kBcRow];
}

// ---- second generation of the exchanges (k_win only; the kernels of the unfused path keep the ones above) -------------
// Measured (tools/probe/lds_forms.hip): a store's cost is the transfer of its address and data VGPRs to the LDS --
// ds_write_b64 6 cycles per 512 B, ds_write_addtid_b32 (no address VGPR: address = M0 + offset + 4*lane) 2 cycles per
// 256 B -- and the wave-local exchange, whose reads wait for the wave's own stores, is where that shows (-8.5 % of a
// three-pass transform loop when only it changes).  So the B<->C exchange becomes PLANAR (re plane, im plane) in a
// per-wave region: the writer stores slot s of every lane with two ds_write_addtid_b32 into row s (64 consecutive
// floats), the reader fetches its 16 inputs -- 16 consecutive floats of ONE row, written by the 16 lanes that share
// its upper bits -- with 2 x 4 ds_read_b128.  That needs the exchanged digit in lane bits 0-3 on the writing side:
//     role B   lane = n0 | p << 4 | (k0 & 1) << 5,  wave = k0 >> 1        (t = n0 | p << 4 | k0 << 5)
//     role C   lane = k1 | (k0 & 1) << 4 | p << 5,  wave = k0 >> 1
// (role A keeps t = 2 (16 n1 + n0) + p: sample loads, last radix-2 by DPP and the peak search are untouched).
// Rows are 272 bytes apart and the rows of slots {0-3, 12-15} sit on even positions, the others on odd ones: a
// ds_read_b128 serves lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... together, and with the reader's 16-lane
// source group taken as 2 * bit4 + bit5 of its lane every such set of 16 lanes hits 16 different bank quads (brute
// force over all layouts: tools/model_kwin_lds.py).  Reader and writer of a direction differ exactly by that swap
// of bits 4 and 5, which is why ONE lane map per role serves both directions.
// The barrier image becomes [k0][n1][p][n0] (complex): role B reads / writes 16 consecutive complex per 16-lane
// group and (n0, p) = 32 consecutive per half wave; role A reads conflict free (its lane pairs sit 128 bytes
// apart); only role A's stores of the FORWARD transform are 2-way conflicted (8 of 36 transforms).
constexpr int kLocRow = 272;                   // bytes between rows of the wave-local planar image
constexpr int kLocPlane = 16 * kLocRow;        // re plane, then im plane
constexpr int kLocWave = 2 * kLocPlane;        // 8704 bytes per wave = the wave's two k0 rows of the barrier image
static_assert(kLocWave == 2 * kBcHalf * 8, "the wave-local region is the wave's own part of the exchange image");
__host__ __device__ constexpr int loc_pos(int s) { return s < 4 ? 2 * s : (s >= 12 ? 2 * (s - 8) : 2 * (s - 4) + 1); }

// eight floats = four complex slots S0..S3 of every lane into their rows (M0 = byte address of the wave's region)
template <int S0, int S1, int S2, int S3>
__device__ __forceinline__ void loc_write4(int m0, const float2& a, const float2& b, const float2& c, const float2& d) {
    asm volatile(
        "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
        "ds_write_addtid_b32 %0 offset:%9\n\tds_write_addtid_b32 %1 offset:%10\n\t"
        "ds_write_addtid_b32 %2 offset:%11\n\tds_write_addtid_b32 %3 offset:%12\n\t"
        "ds_write_addtid_b32 %4 offset:%13\n\tds_write_addtid_b32 %5 offset:%14\n\t"
        "ds_write_addtid_b32 %6 offset:%15\n\tds_write_addtid_b32 %7 offset:%16"
        ::"v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "v"(c.x), "v"(c.y), "v"(d.x), "v"(d.y), "s"(m0),
          "n"(loc_pos(S0) * kLocRow), "n"(kLocPlane + loc_pos(S0) * kLocRow), "n"(loc_pos(S1) * kLocRow),
          "n"(kLocPlane + loc_pos(S1) * kLocRow), "n"(loc_pos(S2) * kLocRow), "n"(kLocPlane + loc_pos(S2) * kLocRow),
          "n"(loc_pos(S3) * kLocRow), "n"(kLocPlane + loc_pos(S3) * kLocRow)
        : "memory", "m0");   // (M0 is written here: the compiler must not keep a value of its own in it across the statement)
}
__device__ __forceinline__ void loc_write16(int m0, const float2 (&v)[16]) {
    loc_write4<0, 1, 2, 3>(m0, v[0], v[1], v[2], v[3]);
    loc_write4<4, 5, 6, 7>(m0, v[4], v[5], v[6], v[7]);
    loc_write4<8, 9, 10, 11>(m0, v[8], v[9], v[10], v[11]);
    loc_write4<12, 13, 14, 15>(m0, v[12], v[13], v[14], v[15]);
}
// byte offset, inside the wave's region, of the 16 floats this lane reads (re plane; im plane + kLocPlane)
__device__ __forceinline__ int loc_read_off(int lane) {
    const int d = lane & 15;
    const int pos = d < 4 ? 2 * d : (d >= 12 ? 2 * (d - 8) : 2 * (d - 4) + 1);
    return pos * kLocRow + (((lane >> 4) & 1) * 2 + (lane >> 5)) * 64;
}
// the 16 inputs of this lane: slot q = float q of its run (rd = region + loc_read_off(lane))
__device__ __forceinline__ void loc_read16(const char* rd, float2 (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 re = *reinterpret_cast<const float4*>(rd + 16 * j);
        const float4 im = *reinterpret_cast<const float4*>(rd + kLocPlane + 16 * j);
        v[4 * j] = make_float2(re.x, im.x);
        v[4 * j + 1] = make_float2(re.y, im.y);
        v[4 * j + 2] = make_float2(re.z, im.z);
        v[4 * j + 3] = make_float2(re.w, im.w);
    }
}
// barrier image [k0][n1][p][n0]: complex index of (k0 = 0, n1 = 0) for this thread in role A / role B; slot k0
// (role A) adds k0 * kBcHalf, slot n1 (role B) adds 32 * n1
__device__ __forceinline__ int xa2_base(int t) { const int p = t & 1, u = t >> 1; return 32 * (u >> 4) + 16 * p + (u & 15); }
__device__ __forceinline__ int xb2_base(int t) { return (t >> 5) * kBcHalf + 16 * ((t >> 4) & 1) + (t & 15); }
__device__ __forceinline__ void xchg_a2_write(float2* lds, const float2 (&v)[16], int t) {
    float2* b = lds + xa2_base(t);
#pragma unroll
    for (int k0 = 0; k0 < 16; ++k0) b[k0 * kBcHalf] = make_float2(v[k0].x, v[k0].y);
}
__device__ __forceinline__ void xchg_a2_read(const float2* lds, float2 (&v)[16], int t) {
    const float2* b = lds + xa2_base(t);
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0)
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) v[q0 + 4 * q1] = b[(q0 + 4 * q1) * kBcHalf];
}
__device__ __forceinline__ void xchg_b2_read(const float2* lds, float2 (&v)[16], int t) {
    const float2* b = lds + xb2_base(t);
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = b[n1 * 32];
}

// TW2[a][b] = W_256^(a*b) lives in LDS as 16 rows of 16 complex padded to 18 (144-B rows: the 16
// rows then start on 16 distinct 16-B bank groups, so a ds_read_b128 of one column is conflict free).
constexpr int kTw2RowF2 = 18;
__device__ __forceinline__ void mul_tw2(float2 (&v)[16], const float2* tw2_lds, int a) {
    const float4* row = reinterpret_cast<const float4*>(tw2_lds + a * kTw2RowF2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 w = row[j];
        if (j > 0) v[2 * j] = cmul(v[2 * j], make_float2(w.x, w.y));
        v[2 * j + 1] = cmul(v[2 * j + 1], make_float2(w.z, w.w));
    }
}

__device__ __forceinline__ void mul_tw1(float2 (&v)[16], const float2 (&tw1)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = cmul(v[k], tw1[k]);
}

// x[k] += s * x[k] of lane^1 for 8 registers, one v_fmac_f32_dpp each (quad_perm [1,0,3,2]).  hipcc
// does not form this instruction from the mov_dpp builtin, and inside asm the VALU-write -> DPP-read
// hazard (2 wait states) is ours: one s_nop 1 in front covers all eight (they do not feed each other).
__device__ __forceinline__ void pair_fmac8(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5,
                                           float& x6, float& x7, float s) {
#define RMX_DPPF(n) "v_fmac_f32_dpp %" #n ", %" #n ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" RMX_DPPF(0) RMX_DPPF(1) RMX_DPPF(2) RMX_DPPF(3) RMX_DPPF(4) RMX_DPPF(5) RMX_DPPF(6)
                     RMX_DPPF(7)
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                 : "v"(s));
#undef RMX_DPPF
}

// lane <-> lane^1 exchange (DPP quad_perm [1,0,3,2]); folds into the consuming VOP2 as a DPP operand
__device__ __forceinline__ float dpp_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}

// ---- wave64 reductions on the VALU (DPP row ops + 4 readlanes; no LDS traffic) -------------------
// Each step is one v_max_f32_dpp / v_min_i32_dpp (x = op(x of the partner lane, x)); hipcc expands the
// update_dpp builtin into v_mov + s_nop + v_mov_dpp + op instead, which doubles the length of this
// dependent chain.  The 2 wait states between a VALU write and a DPP read of the same VGPR are ours
// inside asm: s_nop 1 in front of every step.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false); }
#define RMX_DPP_CHAIN(op)                                                                   \
    "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"      \
    "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"      \
    "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"          \
    "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
__device__ __forceinline__ float wave_max_f32(float x) {
    asm volatile(RMX_DPP_CHAIN("v_max_f32_dpp") : "+v"(x));       // every lane: max of its row of 16
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ int wave_min_i32(int x) {
    asm volatile(RMX_DPP_CHAIN("v_min_i32_dpp") : "+v"(x));
    const int r0 = __builtin_amdgcn_readlane(x, 0), r1 = __builtin_amdgcn_readlane(x, 16);
    const int r2 = __builtin_amdgcn_readlane(x, 32), r3 = __builtin_amdgcn_readlane(x, 48);
    return min(min(r0, r1), min(r2, r3));
}

// qa..qd = Q0..Q0+3 where m0..m3 == tmax (else unchanged): four compares into four SGPR pairs, then
// four selects.  hipcc funnels every compare through VCC and pays a wait state between each compare
// and its select; with distinct masks nothing waits.
template <int Q0>
__device__ __forceinline__ void argsel4(int& qa, int& qb, int& qc, int& qd, float m0, float m1, float m2, float m3,
                                        float tmax) {
    unsigned long long k0, k1, k2, k3;
    asm("v_cmp_eq_f32_e64 %4, %8, %12\n\t"
        "v_cmp_eq_f32_e64 %5, %9, %12\n\t"
        "v_cmp_eq_f32_e64 %6, %10, %12\n\t"
        "v_cmp_eq_f32_e64 %7, %11, %12\n\t"
        "v_cndmask_b32_e64 %0, %0, %13, %4\n\t"
        "v_cndmask_b32_e64 %1, %1, %14, %5\n\t"
        "v_cndmask_b32_e64 %2, %2, %15, %6\n\t"
        "v_cndmask_b32_e64 %3, %3, %16, %7"
        : "+v"(qa), "+v"(qb), "+v"(qc), "+v"(qd), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
        : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(tmax), "n"(Q0), "n"(Q0 + 1), "n"(Q0 + 2), "n"(Q0 + 3));
}

}  // namespace rmx
