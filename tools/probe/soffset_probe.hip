// Minimal probe for the buffer_store + SGPR soffset question: the same 8 x 16-byte stores per thread
// through (a) voffset only, (b) SGPR soffset with a loop-uniform offset, (c) soffset while re-using the data VGPRs
// immediately (the pattern of store_spec in k_win).  Prints the number of wrong floats per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, int nb, int iters) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, nb * 8 * 512 * 16, 0x00020000);
    const int t = threadIdx.x, soff = t * 16;
    for (int it = 0; it < iters; ++it)
    for (int b = 0; b < nb; ++b) {
        float v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = (float)(b * 100000 + q * 1000 + t) + 0.25f * it;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float e0 = v[4 * j], e1 = v[4 * j + 1], e2 = v[4 * j + 2], e3 = v[4 * j + 3];
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
            const u32x4 w = {__float_as_uint(e0), __float_as_uint(e1), __float_as_uint(e2), __float_as_uint(e3)};
            if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(w, rs, soff + (b * 8 + j) * (512 * 16), 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(w, rs, soff, (b * 8 + j) * (512 * 16), 0);
        }
    }
}
int main() {
    const int nb = 8, n = nb * 8 * 512 * 4;
    float* d; hipMalloc(&d, n * 4);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, n * 4);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(512), 0, 0, d, nb, 3);
        else hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, d, nb, 3);
        hipDeviceSynchronize();
        std::vector<float> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int b = 0; b < nb; ++b) for (int j = 0; j < 8; ++j) for (int t = 0; t < 512; ++t) for (int c = 0; c < 4; ++c) {
            const float want = (float)(b * 100000 + (4 * j + c) * 1000 + t) + 0.5f;
            if (h[((b * 8 + j) * 512 + t) * 4 + c] != want) ++bad;
        }
        printf("mode %d (%s): %ld wrong of %d\n", mode, mode ? "SGPR soffset" : "voffset only", bad, n);
    }
    return 0;
}
