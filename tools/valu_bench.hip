// VALU issue-rate probe for gfx950: scalar vs packed f32 ops at 1/2/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_bench valu_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float b = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f;
    f2 pb = {b, b}, pc = {c, c};
    float a[8];
    f2 pa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; pa[i] = f2{a[i], a[i] + 1}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[i]) : "v"(pb), "v"(pc));
                if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pc));
                if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb));
                if (MODE == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 7) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + pa[i].x + pa[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int flops_per_lane_instr) {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {
        dim3 grid(256 * wps), block(256);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<grid, block>>>(out, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE><<<grid, block>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)grid.x * 4 /*waves*/ * iters * 32.0;          // wave-instructions
        double per_simd = winstr / (256.0 * 4.0);                             // per SIMD
        double ns_per = ms * 1e6 / per_simd;
        double tflops = winstr * 64.0 * flops_per_lane_instr / (ms * 1e-3) / 1e12;
        printf("%-16s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)  %.1f TFLOP/s\n",
               name, wps, ms, ns_per, ns_per * 2.4, tflops);
    }
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32", 2);
    run<1>("v_pk_fma_f32", 4);
    run<2>("v_add_f32", 1);
    run<3>("v_pk_add_f32", 2);
    run<4>("v_mul_f32", 1);
    run<5>("v_pk_mul_f32", 2);
    run<6>("v_fmac_f32", 2);
    run<7>("v_add_f32_dpp", 1);
    return 0;
}
