// xcd_l2_probe.hip -- can a four-step product stay in ONE XCD's L2 between the row phase and the column phase?
//
// VERDICT r02 item 5 asks for the row and column phases of the four-step inverse back to back on one XCD behind an
// XCD-wide barrier, so that the product X_j conj(X_i) (row-transformed) never goes to HBM.  This probe measures the one
// thing that design stands on, without any transform: the 32 workgroups of an XCD WRITE a buffer of S bytes row-wise
// (the row pass's stores: whole rows, 16 bytes per lane), meet at an XCD-wide barrier, READ it column-wise (the column
// pass's loads: 128-byte segments, one per row, 16 columns per tile), meet again, and repeat on the SAME buffer (the
// product buffer of the next pair).  S = 4 MiB is cfg5 / cfg1 (2^19 complex64), 16 MiB is cfg2.
//
//   build:  hipcc --offload-arch=gfx950 -O3 -o xcd_l2_probe xcd_l2_probe.hip
//   run:    ./xcd_l2_probe            (table: S per XCD, ms per round trip, GB/s chip-wide, stale reads)
//           rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ... -- ./xcd_l2_probe   (one dispatch per S: counters per row)
//
// Every wait is bounded (a workgroup that does not see its XCD's counter move within ~2 s sets an error flag and
// leaves), so a wrong assumption about residency or the workgroup -> XCD map ends in an error line, not in a hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kThreads = 512;
constexpr int kWgPerXcd = 32;
constexpr int kRows = 512;              // rows of the product matrix (cfg5: 512 x 1024)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15; }   // HW_REG_XCC_ID[3:0]

// XCD-wide barrier: one counter per XCD on a line of its own, monotonically increasing
__device__ inline bool xcd_barrier(unsigned* ctr, unsigned target, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        for (long spin = 0; spin < 2000000L; ++spin) {
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!good) *err = 1;
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// mode 0: write rows, barrier, read columns, barrier  (same buffer every iteration)
// mode 1: the same without the barriers' data dependence across workgroups: every workgroup reads back ITS OWN rows
//         (what a single-CU round trip through L2 would cost)
__global__ __launch_bounds__(kThreads) void xcd_probe(char* __restrict__ base, long bytes_per_xcd, int iters, unsigned* ctrs,
                                                      int mode, int* err, unsigned* stale, int* xcc_seen) {
    extern __shared__ char lds_pad[];     // sized by the host so that only one workgroup fits a CU
    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc_id();
    unsigned* ctr = ctrs + xcd * 32;      // 128 bytes apart
    char* buf = base + (long)xcd * bytes_per_xcd;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes_per_xcd, 0x00020000);
    const long row_bytes = bytes_per_xcd / kRows;
    const int rows_per_wg = kRows / kWgPerXcd;                  // 16 rows per workgroup in the row phase
    const int tiles = (int)(row_bytes / 128);                   // 128-byte column tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned bad = 0, acc = 0;
    unsigned phase = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned tag = 0x1000u * (unsigned)(it + 1);
        // ---- row phase: rows [rank*16, rank*16+16), whole rows, 16 bytes per lane, plain stores (stay in L2)
        for (int r = 0; r < rows_per_wg; ++r) {
            const int row = rank * rows_per_wg + r;
            for (long off = (long)threadIdx.x * 16; off < row_bytes; off += kThreads * 16) {
                const unsigned w = tag + (unsigned)row;
                const u32x4 v = {w, w ^ (unsigned)off, w, w};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(row * row_bytes + off), 0, 0);
            }
        }
        if (!xcd_barrier(ctr, (++phase) * kWgPerXcd, err)) return;
        // ---- column phase: tiles rank, rank+32, ...: per tile 512 rows x 128 bytes; a wave-load covers 8 rows
        if (mode == 0) {
            for (int tile = rank; tile < tiles; tile += kWgPerXcd) {
                for (int r0 = wave * 8; r0 < kRows; r0 += 8 * 8) {
                    const int row = r0 + (lane >> 3);
                    const int off = (int)(row * row_bytes + (long)tile * 128 + (lane & 7) * 16);
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16 /* sc1: past L1 */);
                    bad += (v.x != tag + (unsigned)row);
                    acc += v.y + v.z + v.w;
                }
            }
        } else {
            for (int r = 0; r < rows_per_wg; ++r) {
                const int row = rank * rows_per_wg + r;
                for (long off = (long)threadIdx.x * 16; off < row_bytes; off += kThreads * 16) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(row * row_bytes + off), 0, 16);
                    bad += (v.x != tag + (unsigned)row);
                    acc += v.y + v.z + v.w;
                }
            }
        }
        if (!xcd_barrier(ctr, (++phase) * kWgPerXcd, err)) return;
    }
    if (acc == 0x12345u) bad += 1u << 30;    // keep acc alive
    if (bad) atomicAdd(stale, bad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 200;
    const int only_mb = argc > 2 ? std::atoi(argv[2]) : 0;      // one size only (for counter runs)
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    if (prop.multiProcessorCount != 256) { std::printf("expected 256 CUs, have %d: not running\n", prop.multiProcessorCount); return 0; }
    const long max_bytes = 64L << 20;
    char* buf; unsigned* ctrs; int* err; unsigned* stale; int* xcc;
    CHECK(hipMalloc(&buf, 8 * max_bytes));
    CHECK(hipMalloc(&ctrs, 8 * 128));
    CHECK(hipMalloc(&err, 4));
    CHECK(hipMalloc(&stale, 4));
    CHECK(hipMalloc(&xcc, 256 * 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xcd_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::printf("%8s %5s %10s %12s %10s %s\n", "S/XCD", "mode", "ms/iter", "GB/s (w+r)", "stale", "");
    const int sizes_mb[] = {1, 2, 3, 4, 6, 8, 16, 64};
    for (int mode = 0; mode < 2; ++mode)
        for (int mb : sizes_mb) {
            if (only_mb && mb != only_mb) continue;
            const long S = (long)mb << 20;
            const int it = mb >= 16 ? iters / 4 + 1 : iters;
            CHECK(hipMemset(ctrs, 0, 8 * 128));
            CHECK(hipMemset(err, 0, 4));
            CHECK(hipMemset(stale, 0, 4));
            // warm-up launch + timed launch
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemset(ctrs, 0, 8 * 128));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(xcd_probe, dim3(256), dim3(kThreads), 100 * 1024, 0, buf, S, it, ctrs, mode, err, stale, xcc);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
            }
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            int herr; unsigned hstale; std::vector<int> hx(256);
            CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&hstale, stale, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
            int mism = 0;
            for (int b = 0; b < 256; ++b) mism += (hx[b] != (b & 7));
            const double gbs = 2.0 * 8 * (double)S * it / (ms * 1e-3) / 1e9;
            std::printf("%6d MiB %5d %10.4f %12.0f %10u %s%s\n", mb, mode, ms / it, gbs, hstale, herr ? "BARRIER TIMEOUT " : "",
                        mism ? "workgroup->XCD map is not blockIdx & 7" : "");
            if (herr) return 1;
        }
    return 0;
}
