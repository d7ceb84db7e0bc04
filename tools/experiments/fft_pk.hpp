// fft_pk.hpp -- the radix-16 building blocks of fft_r16.hpp on PACKED fp32 (VOP3P v_pk_add/mul/fma_f32):
// a complex value lives in one even-aligned VGPR pair (re = low dword, im = high dword), so a complex add
// is one instruction, a multiply by -i / +i is the same add with op_sel/neg modifiers, a complex multiply is
// two instructions and a complex multiply-add is two.  The VALU pipe retires a packed instruction in the
// time of two scalar ones, so nothing is gained once enough waves keep the pipe busy -- but a wave issues a
// packed instruction as fast as a scalar one, and the fused window kernel runs at 2 waves per SIMD that
// rarely issue together (profiles/r01e_pmc.json: one VALU instruction per 4.7 cycles and SIMD): half the
// instructions for the same arithmetic is what shortens its issue-bound stretches.
// hipcc does not select these forms with op_sel swizzles from scalar source (its SLP vectoriser inserts
// v_mov shuffles: 1.7x slower), so every primitive is one asm statement.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../radio-mapper_amd/csrc/fft_r16.hpp"

namespace rmx {
namespace pk {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2 mk(float x, float y) { f2 r = {x, y}; return r; }
__device__ __forceinline__ f2 lo2(f4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f2 hi2(f4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// ---- instruction forms (asm text): op_sel[i] / op_sel_hi[i] select the half of source i that feeds the low /
// high result, neg_lo / neg_hi negate it.  hipcc puts an s_nop between an asm statement and a following
// instruction that reads its result (it cannot see that no hazard applies), so a whole butterfly is ONE asm
// statement, with its two independent chains interleaved.
#define PK_ADD(d, a, b) "v_pk_add_f32 " d ", " a ", " b "\n\t"
#define PK_SUB(d, a, b) "v_pk_add_f32 " d ", " a ", " b " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define PK_ADDMI(d, a, b) "v_pk_add_f32 " d ", " a ", " b " op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"   /* a - i b */
#define PK_ADDPI(d, a, b) "v_pk_add_f32 " d ", " a ", " b " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"   /* a + i b */
#define PK_MULB(d, a, w) "v_pk_mul_f32 " d ", " a ", " w " op_sel_hi:[1,0]\n\t"                              /* a * w.x */
#define PK_FMAB(d, a, w, c) "v_pk_fma_f32 " d ", " a ", " w ", " c " op_sel_hi:[1,0,1]\n\t"                  /* c + a * w.x */
#define PK_FMAI(d, a, w, c) \
    "v_pk_fma_f32 " d ", " a ", " w ", " c " op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"             /* c + i a * w.y */
/* the same three for the (im, re)-swapped a */
#define PK_MULBS(d, a, w) "v_pk_mul_f32 " d ", " a ", " w " op_sel:[1,0] op_sel_hi:[0,0]\n\t"
#define PK_FMABS(d, a, w, c) "v_pk_fma_f32 " d ", " a ", " w ", " c " op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
#define PK_FMAIS(d, a, w, c) \
    "v_pk_fma_f32 " d ", " a ", " w ", " c " op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0]\n\t"
#define PK_TWM(d, a, b) "v_pk_fma_f32 " d ", " a ", 2.0, " b " op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"   /* 2a - b */

__device__ __forceinline__ f2 cmul(f2 a, f2 w) {
    f2 d;
    asm(PK_MULB("%0", "%1", "%2") PK_FMAI("%0", "%1", "%2", "%0") : "=&v"(d) : "v"(a), "v"(w));
    return d;
}
// 4-point DFT, natural order in and out, in place (8 instructions)
__device__ __forceinline__ void dft4(f2& a0, f2& a1, f2& a2, f2& a3) {
    f2 t0, t1, t2, t3;
    asm(PK_ADD("%4", "%0", "%2") PK_SUB("%5", "%0", "%2") PK_ADD("%6", "%1", "%3") PK_SUB("%7", "%1", "%3")
        PK_ADD("%0", "%4", "%6") PK_SUB("%2", "%4", "%6") PK_ADDMI("%1", "%5", "%7") PK_ADDPI("%3", "%5", "%7")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
}
// DFT4 of (w0 a0, w1 a1, w2 a2, w3 a3), twiddles in VGPR pairs (14 instructions, 12 when w0 == 1);
// SWAPPED: of (w0 swap(a0), ...), swap(a) = (a.y, a.x): the (im, re)-swapped spectrum costs nothing
template <bool W0_IS_ONE, bool SWAPPED = false>
__device__ __forceinline__ void dft4_tw(f2& a0, f2& a1, f2& a2, f2& a3, f2 w0, f2 w1, f2 w2, f2 w3) {
    f2 x0, x1, y0, y1;
    // %0..%3 = a0..a3, %4..%7 = x0 x1 y0 y1, %8..%11 = w0..w3
#define PK_TAIL                                                                                             \
    PK_TWM("%4", "%4", "%5") PK_TWM("%6", "%6", "%7") PK_ADD("%0", "%5", "%7") PK_SUB("%2", "%5", "%7")       \
    PK_ADDMI("%1", "%4", "%6") PK_ADDPI("%3", "%4", "%6")
    if constexpr (!SWAPPED && !W0_IS_ONE)
        asm(PK_MULB("%4", "%0", "%8") PK_MULB("%6", "%1", "%9") PK_FMAI("%4", "%0", "%8", "%4")
            PK_FMAI("%6", "%1", "%9", "%6") PK_FMAB("%5", "%2", "%10", "%4") PK_FMAB("%7", "%3", "%11", "%6")
            PK_FMAI("%5", "%2", "%10", "%5") PK_FMAI("%7", "%3", "%11", "%7") PK_TAIL
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1)
            : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
    else if constexpr (SWAPPED && !W0_IS_ONE)
        asm(PK_MULBS("%4", "%0", "%8") PK_MULBS("%6", "%1", "%9") PK_FMAIS("%4", "%0", "%8", "%4")
            PK_FMAIS("%6", "%1", "%9", "%6") PK_FMABS("%5", "%2", "%10", "%4") PK_FMABS("%7", "%3", "%11", "%6")
            PK_FMAIS("%5", "%2", "%10", "%5") PK_FMAIS("%7", "%3", "%11", "%7") PK_TAIL
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1)
            : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
    else {
        static_assert(!(SWAPPED && W0_IS_ONE), "not used");
        // w0 == 1: A0 = a0 itself; x0 receives t1 = 2 a0 - t0
        asm(PK_MULB("%6", "%1", "%9") PK_FMAB("%5", "%2", "%10", "%0") PK_FMAI("%6", "%1", "%9", "%6")
            PK_FMAI("%5", "%2", "%10", "%5") PK_FMAB("%7", "%3", "%11", "%6") PK_TWM("%4", "%0", "%5")
            PK_FMAI("%7", "%3", "%11", "%7") PK_TWM("%6", "%6", "%7") PK_ADD("%0", "%5", "%7") PK_SUB("%2", "%5", "%7")
            PK_ADDMI("%1", "%4", "%6") PK_ADDPI("%3", "%4", "%6")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1)
            : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
    }
#undef PK_TAIL
}
// the same with w0 == 1 and three compile-time twiddles in SGPR pairs (one constant-bus operand per instruction)
__device__ __forceinline__ void dft4_tw_k(f2& a0, f2& a1, f2& a2, f2& a3, f2 w1, f2 w2, f2 w3) {
    f2 x0, x1, y0, y1;
    asm(PK_MULB("%6", "%1", "%8") PK_FMAB("%5", "%2", "%9", "%0") PK_FMAI("%6", "%1", "%8", "%6")
        PK_FMAI("%5", "%2", "%9", "%5") PK_FMAB("%7", "%3", "%10", "%6") PK_TWM("%4", "%0", "%5")
        PK_FMAI("%7", "%3", "%10", "%7") PK_TWM("%6", "%6", "%7") PK_ADD("%0", "%5", "%7") PK_SUB("%2", "%5", "%7")
        PK_ADDMI("%1", "%4", "%6") PK_ADDPI("%3", "%4", "%6")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1)
        : "s"(w1), "s"(w2), "s"(w3));
}
// group ka = 2 of the second layer: twiddles 1, W8, -i, W8^3 (11 instructions)
__device__ __forceinline__ void dft4_tw_k2(f2& a0, f2& a1, f2& a2, f2& a3, f2 c1, f2 c3) {
    f2 x0, x1, y0, y1;
    asm(PK_MULB("%6", "%1", "%8") PK_ADDMI("%5", "%0", "%2") PK_FMAI("%6", "%1", "%8", "%6") PK_ADDPI("%4", "%0", "%2")
        PK_FMAB("%7", "%3", "%9", "%6") PK_FMAI("%7", "%3", "%9", "%7") PK_TWM("%6", "%6", "%7")
        PK_ADD("%0", "%5", "%7") PK_SUB("%2", "%5", "%7") PK_ADDMI("%1", "%4", "%6") PK_ADDPI("%3", "%4", "%6")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1)
        : "s"(c1), "s"(c3));
}
// x[k] *= w[k], k = 0..3 (8 instructions, chains interleaved)
__device__ __forceinline__ void cmul4(f2& a0, f2& a1, f2& a2, f2& a3, f2 w0, f2 w1, f2 w2, f2 w3) {
    f2 t0, t1, t2, t3;
    asm(PK_MULB("%4", "%0", "%8") PK_MULB("%5", "%1", "%9") PK_MULB("%6", "%2", "%10") PK_MULB("%7", "%3", "%11")
        PK_FMAI("%0", "%0", "%8", "%4") PK_FMAI("%1", "%1", "%9", "%5") PK_FMAI("%2", "%2", "%10", "%6")
        PK_FMAI("%3", "%3", "%11", "%7")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
}
// the same with the twiddles in SGPR pairs (used under the odd-lane branch: in place, no result copies)
__device__ __forceinline__ void cmul4_k(f2& a0, f2& a1, f2& a2, f2& a3, f2 w0, f2 w1, f2 w2, f2 w3) {
    f2 t0, t1, t2, t3;
    asm volatile(PK_MULB("%4", "%0", "%8") PK_MULB("%5", "%1", "%9") PK_MULB("%6", "%2", "%10") PK_MULB("%7", "%3", "%11")
                 PK_FMAI("%0", "%0", "%8", "%4") PK_FMAI("%1", "%1", "%9", "%5") PK_FMAI("%2", "%2", "%10", "%6")
                 PK_FMAI("%3", "%3", "%11", "%7")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                 : "s"(w0), "s"(w1), "s"(w2), "s"(w3));
}
__device__ __forceinline__ void swap2(f2& a, f2& b) { const f2 t = a; a = b; b = t; }

// group ka of the second radix-4 layer (constant inner twiddles W16^(q0 ka) merged), in place on v[4ka..4ka+3]
template <int KA>
__device__ __forceinline__ void layer2_group(f2 (&v)[16]) {
    if constexpr (KA == 0) {
        dft4(v[0], v[1], v[2], v[3]);
    } else if constexpr (KA == 1) {
        dft4_tw_k(v[4], v[5], v[6], v[7], mk(RMX_C1, -RMX_S1), mk(RMX_RH, -RMX_RH), mk(RMX_S1, -RMX_C1));   // W1 W2 W3
    } else if constexpr (KA == 2) {                                                                         // W2 W4 W6
        dft4_tw_k2(v[8], v[9], v[10], v[11], mk(RMX_RH, -RMX_RH), mk(-RMX_RH, -RMX_RH));
    } else {
        dft4_tw_k(v[12], v[13], v[14], v[15], mk(RMX_S1, -RMX_C1), mk(-RMX_RH, -RMX_RH), mk(-RMX_C1, RMX_S1));  // W3 W6 W9
    }
}
__device__ __forceinline__ void dft16_layer2(f2 (&v)[16]) {
    layer2_group<0>(v);
    layer2_group<1>(v);
    layer2_group<2>(v);
    layer2_group<3>(v);
    // un-transpose (register renaming): X[ka + 4kb] sits in v[4ka + kb]
    swap2(v[1], v[4]);
    swap2(v[2], v[8]);
    swap2(v[3], v[12]);
    swap2(v[6], v[9]);
    swap2(v[7], v[13]);
    swap2(v[11], v[14]);
}
// the same layer, handing each group's outputs X[ka], X[ka+4], X[ka+8], X[ka+12] to `emit` as they exist
template <class F>
__device__ __forceinline__ void dft16_layer2_emit(f2 (&v)[16], F&& emit) {
    layer2_group<0>(v);
    emit(std::integral_constant<int, 0>{}, v[0], v[1], v[2], v[3]);
    __builtin_amdgcn_sched_barrier(0);
    layer2_group<1>(v);
    emit(std::integral_constant<int, 1>{}, v[4], v[5], v[6], v[7]);
    __builtin_amdgcn_sched_barrier(0);
    layer2_group<2>(v);
    emit(std::integral_constant<int, 2>{}, v[8], v[9], v[10], v[11]);
    __builtin_amdgcn_sched_barrier(0);
    layer2_group<3>(v);
    emit(std::integral_constant<int, 3>{}, v[12], v[13], v[14], v[15]);
    __builtin_amdgcn_sched_barrier(0);
}
// 16-point DFT, natural order in and out (75 instructions)
__device__ __forceinline__ void dft16(f2 (&v)[16]) {
    dft4(v[0], v[4], v[8], v[12]);
    dft4(v[1], v[5], v[9], v[13]);
    dft4(v[2], v[6], v[10], v[14]);
    dft4(v[3], v[7], v[11], v[15]);
    dft16_layer2(v);
}
// layer 1 of the 16-point DFT of (v[q] * w[q]); SWAPPED: of (swap(v[q]) * w[q])
template <bool W0_IS_ONE, bool SWAPPED = false>
__device__ __forceinline__ void dft16_tw_l1(f2 (&v)[16], const f2 (&w)[16]) {
    dft4_tw<W0_IS_ONE, SWAPPED>(v[0], v[4], v[8], v[12], w[0], w[4], w[8], w[12]);
    dft4_tw<false, SWAPPED>(v[1], v[5], v[9], v[13], w[1], w[5], w[9], w[13]);
    dft4_tw<false, SWAPPED>(v[2], v[6], v[10], v[14], w[2], w[6], w[10], w[14]);
    dft4_tw<false, SWAPPED>(v[3], v[7], v[11], v[15], w[3], w[7], w[11], w[15]);
}
// layer 1 with the twiddle row fetched just in time from LDS (row layout: dft16_tw_row_l1 of fft_r16.hpp)
__device__ __forceinline__ void dft16_tw_row_l1(f2 (&v)[16], const f4* row, f4 f0, f4 f1) {
    dft4_tw<true>(v[0], v[4], v[8], v[12], mk(1.0f, 0.0f), lo2(f0), hi2(f0), lo2(f1));
    const f4 g2 = row[2], g3 = row[3];
    dft4_tw<false>(v[1], v[5], v[9], v[13], hi2(f1), lo2(g2), hi2(g2), lo2(g3));
    const f4 g4 = row[4], g5 = row[5];
    dft4_tw<false>(v[2], v[6], v[10], v[14], hi2(g3), lo2(g4), hi2(g4), lo2(g5));
    const f4 g6 = row[6], g7 = row[7];
    dft4_tw<false>(v[3], v[7], v[11], v[15], hi2(g5), lo2(g6), hi2(g6), lo2(g7));
}
__device__ __forceinline__ void mul_tw1(f2 (&v)[16], const f2 (&tw1)[16]) {
#pragma unroll
    for (int k = 0; k < 16; k += 4)
        cmul4(v[k], v[k + 1], v[k + 2], v[k + 3], tw1[k], tw1[k + 1], tw1[k + 2], tw1[k + 3]);
}

// ---- LDS exchanges: the images of fft_r16.hpp, 8 bytes per access -------------------------------------
__device__ __forceinline__ void xchg_a_write(f2* lds, const f2 (&v)[16], int t) {
#pragma unroll
    for (int k0 = 0; k0 < 16; ++k0) lds[k0 * kBcHalf + t] = v[k0];
}
__device__ __forceinline__ void xchg_a_read(const f2* lds, f2 (&v)[16], int t) {
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0)
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) v[q0 + 4 * q1] = lds[(q0 + 4 * q1) * kBcHalf + t];
}
__device__ __forceinline__ void xchg_b_read(const f2* lds, f2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, n0 = u & 15;
    const f2* base = lds + k0 * kBcHalf + n0 * 2 + p;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = base[n1 * 32];
}
__device__ __forceinline__ void xchg_bc_write_b(f2* lds, const f2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, a = u & 15;
    f2* base = lds + k0 * kBcHalf + 2 * a + p;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) base[k1 * kBcRow] = v[k1];
}
__device__ __forceinline__ void xchg_bc_read_c(const f2* lds, f2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, a = u & 15;
    const f2* base = lds + k0 * kBcHalf + a * kBcRow + p;
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) v[n0] = base[2 * n0];
}
__device__ __forceinline__ void xchg_bc_read_b(const f2* lds, f2 (&v)[16], int t) {
    const int p = t & 1, u = t >> 1, k0 = u >> 4, a = u & 15;
    const f2* base = lds + k0 * kBcHalf + 2 * a + p;
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0)
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) v[q0 + 4 * q1] = base[(q0 + 4 * q1) * kBcRow];
}

}  // namespace pk
}  // namespace rmx
