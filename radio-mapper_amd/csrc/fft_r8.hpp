// fft_r8.hpp -- the plain 8-point DFT in registers (52 instructions), used by the generic path's LDS transforms for passes
// of three radix-2 stages (generic_path.hpp: fused pass of RAD == 8).  Split out of the k_win8 experiment (now
// tools/experiments/win8.hpp, -DRMX_EXPERIMENTS builds only), which keeps the twiddle-merged variants for itself.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_r16.hpp"

namespace rmx {
namespace w8 {

// ---- radix-8 butterflies --------------------------------------------------------------------------
// odd half of the radix-2 split, W8^q merged: (u0..u3) -> X[1], X[3], X[5], X[7]  (20 instructions)
__device__ __forceinline__ void dft4_w8(float2& u0, float2& u1, float2& u2, float2& u3) {
    const float b1x = u1.x + u1.y, b1y = u1.y - u1.x;   // sqrt(2) u1 W8
    const float p3 = u3.x + u3.y, q3 = u3.y - u3.x;     // sqrt(2) u3 W8^3 = (q3, -p3)
    const float sx = b1x + q3, sy = b1y - p3;
    const float dx = b1x - q3, dy = b1y + p3;
    const float t0x = u0.x + u2.y, t0y = u0.y - u2.x;   // u0 + (-i) u2
    const float t1x = u0.x - u2.y, t1y = u0.y + u2.x;
    u0 = make_float2(fmaf(RMX_RH, sx, t0x), fmaf(RMX_RH, sy, t0y));
    u2 = make_float2(fmaf(-RMX_RH, sx, t0x), fmaf(-RMX_RH, sy, t0y));
    u1 = make_float2(fmaf(RMX_RH, dy, t1x), fmaf(-RMX_RH, dx, t1y));
    u3 = make_float2(fmaf(-RMX_RH, dy, t1x), fmaf(RMX_RH, dx, t1y));
}
// after the two half transforms v holds X[0], X[2], X[4], X[6], X[1], X[3], X[5], X[7]: rename to natural order
__device__ __forceinline__ void dft8_unshuffle(float2 (&v)[8]) {
    float tx = v[1].x, ty = v[1].y;            // natural[1] <- v4, [4] <- v2, [2] <- v1
    v[1].x = v[4].x; v[1].y = v[4].y;
    v[4].x = v[2].x; v[4].y = v[2].y;
    v[2].x = tx; v[2].y = ty;
    tx = v[3].x; ty = v[3].y;                  // natural[3] <- v5, [5] <- v6, [6] <- v3
    v[3].x = v[5].x; v[3].y = v[5].y;
    v[5].x = v[6].x; v[5].y = v[6].y;
    v[6].x = tx; v[6].y = ty;
}
__device__ __forceinline__ void dft8_stage_a(float2 (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2 a = v[q], b = v[q + 4];
        v[q] = cadd(a, b);
        v[q + 4] = csub(a, b);
    }
}
// 8-point DFT, X[k] = sum_q v[q] W8^(qk), natural order in and out (52 instructions)
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    dft8_stage_a(v);
    dft4(v[0], v[1], v[2], v[3]);
    dft4_w8(v[4], v[5], v[6], v[7]);
    dft8_unshuffle(v);
}
}  // namespace w8
}  // namespace rmx
