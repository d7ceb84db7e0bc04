// valu_forms.hip -- what does one VALU instruction cost a SIMD of gfx950 at k_win's occupancy (two waves per SIMD)?
//
// The fused kernel's pieces run at 6.5-7.5 cycles per VALU instruction and wave (tools/probe/kwin_bench.hip
// -DRMX_KWIN_STAMPS), i.e. one instruction per ~3.4 cycles and SIMD.  Against the 2 cycles a SIMD-32 needs for a
// wave64 instruction that looks like 60 % utilisation; this probe measures what two waves CAN issue, per encoding:
// one workgroup of 128 / 256 / 512 / 1024 threads per CU (0.5 / 1 / 2 / 4 waves per SIMD), 8 independent
// accumulators per lane, straight-line blocks of 64 instructions.
//   build: hipcc --offload-arch=gfx950 -O3 -o valu_forms valu_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float bb, float cc) {
    extern __shared__ char pad[];   // 100 KiB: one workgroup per CU
    float b = bb + threadIdx.x * 1e-9f, c = cc, d = cc * 0.5f;
    const float sb = bb, sc = cc;   // kernel arguments: SGPRs
    unsigned long long m = threadIdx.x & 1 ? ~0ull : 0x5555555555555555ull;
    m = __builtin_amdgcn_readfirstlane((int)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32);
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const unsigned lds_addr8 = threadIdx.x * 8, lds_addr16 = threadIdx.x * 16, lds_addr4 = threadIdx.x * 4;
    const int m0v = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 256);
    f4 q4 = {b, c, b, c};
    int sacc = 0;
    float a[8];
    f2 pa[8], pb = {b, b}, pc = {c, c};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; pa[i] = f2{a[i], a[i] + 1}; }
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 50 && MODE < 70) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 1) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 4) asm volatile("v_fma_f32 %0, %0, 2.0, -%1" : "+v"(a[i]) : "v"(c));
                if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
                if (MODE == 6) asm volatile("v_fmamk_f32 %0, %0, 0x3f3504f3, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 7) asm volatile("v_mul_f32_e32 %0, 0x3f3504f3, %0" : "+v"(a[i]));
                if (MODE == 8) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 9) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(m));
                if (MODE == 10) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
                if (MODE == 11) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(d));
                if (MODE == 12) {   // the butterflies' mix: fmac, fmac, VOP3 fma with 2.0, add, sub
                    if ((i & 3) == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    if ((i & 3) == 1) asm volatile("v_fma_f32 %0, %0, 2.0, -%1" : "+v"(a[i]) : "v"(c));
                    if ((i & 3) == 2) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                    if ((i & 3) == 3) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
                }
                if (MODE == 13) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "s"(sb));   // one VGPR read
                if (MODE == 14) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 15) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
                if (MODE == 16) asm volatile("v_fmac_f32_e32 %0, 0x3f3504f3, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 17) asm volatile("v_cmp_eq_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(c), "v"(b) : "vcc");
                if (MODE == 18) { unsigned long long kk; asm volatile("v_cmp_eq_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %3, %1" : "+v"(a[i]), "=&s"(kk) : "v"(c), "v"(b)); }
                if (MODE == 19) { unsigned long long kk; asm volatile("v_cmp_eq_f32_e64 %0, %1, %2" : "=s"(kk) : "v"(a[i]), "v"(c)); }
                if (MODE == 20) asm volatile("v_cmp_eq_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
                if (MODE == 21) { int rr; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(rr) : "v"(a[i])); }
                if (MODE == 22) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 23) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
                if (MODE == 24) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
                if (MODE == 25) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
                if (MODE >= 30 && MODE < 40) {   // one instruction of a slow class among three plain v_add_f32_e32
                    if ((i & 3) != 0) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                    else if (MODE == 30) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
                    else if (MODE == 31) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
                    else if (MODE == 32) { unsigned long long kk; asm volatile("v_cmp_eq_f32_e64 %0, %1, %2" : "=s"(kk) : "v"(a[i]), "v"(c)); }
                    else if (MODE == 33) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    else if (MODE == 34) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    else if (MODE == 35) asm volatile("v_fmamk_f32 %0, %0, 0x3f3504f3, %1" : "+v"(a[i]) : "v"(c));
                    else if (MODE == 36) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]));
                    else if (MODE == 37) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]));
                    else if (MODE == 38) asm volatile("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
                    else if (MODE == 39) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
                }
                if (MODE >= 40 && MODE < 50) {   // every second instruction
                    if ((i & 1) != 0) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                    else if (MODE == 40) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
                    else if (MODE == 41) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
                    else if (MODE == 42) { unsigned long long kk; asm volatile("v_cmp_eq_f32_e64 %0, %1, %2" : "=s"(kk) : "v"(a[i]), "v"(c)); }
                    else if (MODE == 43) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    else if (MODE == 44) asm volatile("v_fmamk_f32 %0, %0, 0x3f3504f3, %1" : "+v"(a[i]) : "v"(c));
                }
                if (MODE >= 50 && MODE < 70) {   // one non-VALU instruction among seven v_add_f32_e32
                    if ((i & 7) != 0) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                    else if (MODE == 50) asm volatile("ds_write_b64 %0, %1 offset:0" : : "v"(lds_addr8), "v"(pa[r & 7]) : "memory");
                    else if (MODE == 51) asm volatile("ds_write_b128 %0, %1 offset:0" : : "v"(lds_addr16), "v"(q4) : "memory");
                    else if (MODE == 52) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:0" : : "v"(a[1]), "s"(m0v) : "memory", "m0");
                    else if (MODE == 53) asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(pa[r & 7]) : "v"(lds_addr8) : "memory");
                    else if (MODE == 54) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(q4) : "v"(lds_addr16) : "memory");
                    else if (MODE == 55) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) : : "scc");
                    else if (MODE == 56) asm volatile("s_nop 0");
                    else if (MODE == 57) asm volatile("ds_write_b32 %0, %1 offset:0" : : "v"(lds_addr4), "v"(a[1]) : "memory");
                    else if (MODE == 58) asm volatile("s_waitcnt lgkmcnt(15)");
                    else if (MODE == 59) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                }
                if (MODE == 26) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[i]) : "v"(pb), "v"(pc));
                if (MODE == 27) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + pa[i].x + pa[i].y;
    s += q4.x + q4.y + q4.z + q4.w + (float)sacc;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 1024 + threadIdx.x] = s + sc;
}

template <int MODE>
void run(const char* name) {
    static float* out = nullptr;
    if (!out) CHECK(hipMalloc(&out, sizeof(float) * 1024 * 256));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    const int iters = 2048;
    std::printf("%-46s", name);
    for (int threads : {128, 256, 512, 1024}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 100 * 1024, 0, out, 64, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 100 * 1024, 0, out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double waves_per_simd = threads / 256.0;
        const double instr_per_simd = (waves_per_simd < 1 ? 1 : waves_per_simd) * iters * 64.0;   // below 1: two SIMDs idle
        std::printf("  %5.2f ns", ms * 1e6 / instr_per_simd);
    }
    std::printf("\n");
}

int main() {
    std::printf("ns per wave-instruction and SIMD at %-16s    0.5        1        2        4  waves per SIMD\n", "");
    run<0>("v_add_f32_e32 (4 bytes)");
    run<10>("v_sub_f32_e32");
    run<14>("v_mov_b32_e32");
    run<1>("v_add_f32_e64 (VOP3 encoding, 2 sources)");
    run<2>("v_fmac_f32_e32 (3 reads, 4 bytes)");
    run<3>("v_fma_f32 v, v, v (VOP3)");
    run<11>("v_fma_f32 -v, v, v");
    run<4>("v_fma_f32 v, 2.0, -v");
    run<5>("v_fma_f32 v, s, v");
    run<13>("v_fma_f32 v, s, s");
    run<6>("v_fmamk_f32 (literal)");
    run<7>("v_mul_f32_e32 literal");
    run<8>("v_fmac_f32_dpp quad_perm");
    run<9>("v_cndmask_b32_e64 (SGPR mask)");
    run<12>("mix fmac / fma 2.0 / add / sub");
    run<15>("v_mul_f32_e32 v, s, v");
    run<25>("v_add_f32_e32 v, s, v");
    run<16>("v_fmac_f32_e32 v, literal, v");
    run<22>("v_max3_f32");
    run<24>("v_mov_b32_dpp quad_perm");
    run<19>("v_cmp_eq_f32_e64 -> SGPR pair");
    run<20>("v_cmp_eq_f32_e32 -> vcc");
    run<23>("v_cndmask_b32_e32 (vcc)");
    run<21>("v_readlane_b32");
    std::printf("(two instructions per line below: ns per PAIR)\n");
    run<18>("v_cmp_eq_f32_e64 s + v_cndmask_b32_e64 s");
    run<17>("v_cmp_eq_f32_e32 vcc + v_cndmask_b32_e32 vcc");
    std::printf("(one in four instructions of the named form, the rest v_add_f32_e32: ns per instruction)\n");
    run<30>("1/4 v_mul_f32 v, s, v");
    run<31>("1/4 v_fmac_f32_dpp");
    run<32>("1/4 v_cmp_eq_f32_e64 -> SGPR");
    run<33>("1/4 v_max3_f32");
    run<34>("1/4 v_fma_f32 v, v, v");
    run<35>("1/4 v_fmamk_f32 literal");
    run<36>("1/4 v_permlane32_swap_b32");
    run<37>("1/4 v_permlane16_swap_b32");
    run<38>("1/4 v_add_f32_dpp row_mirror");
    run<39>("1/4 v_mov_b32_dpp quad_perm");
    std::printf("(one in two)\n");
    run<40>("1/2 v_mul_f32 v, s, v");
    run<41>("1/2 v_fmac_f32_dpp");
    run<42>("1/2 v_cmp_eq_f32_e64 -> SGPR");
    run<43>("1/2 v_fma_f32 v, v, v");
    run<44>("1/2 v_fmamk_f32 literal");
    std::printf("(one in EIGHT instructions of the named form, the rest v_add_f32_e32: ns per instruction slot)\n");
    run<59>("1/8 v_add_f32_e32 (all VALU)");
    run<55>("1/8 s_add_u32");
    run<56>("1/8 s_nop 0");
    run<58>("1/8 s_waitcnt lgkmcnt(15)");
    run<57>("1/8 ds_write_b32");
    run<50>("1/8 ds_write_b64");
    run<51>("1/8 ds_write_b128");
    run<52>("1/8 s_mov m0 + s_nop + ds_write_addtid_b32");
    run<53>("1/8 ds_read_b64");
    run<54>("1/8 ds_read_b128");
    std::printf("(packed: two results per instruction)\n");
    run<26>("v_pk_fma_f32");
    run<27>("v_pk_mul_f32");
    return 0;
}
