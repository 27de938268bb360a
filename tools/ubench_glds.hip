// How fast can a CU pull operand tiles through the global -> LDS path (global_load_lds_dwordx4)?  The NT GEMM main loop moves 73.7 KiB per
// 320x256x64 K tile and measures ~19 B/clk/CU; rounds 1-3 called that "the stream limit" without ever measuring the path alone.  This
// program measures it alone: no MFMAs, no fragment reads, only direct-to-LDS loads with counted waits, for
//   * the three access shapes the GEMMs use (16 rows x 64 B, 8 rows x 128 B, 1 KiB linear per wave-instruction),
//   * 8 waves x 1 workgroup/CU and 4 waves x 2 workgroups/CU,
//   * working sets resident in L2 (2 MB), in the Infinity Cache (96 MB) and in HBM (2 GB),
//   * 8 / 12 / 24 wave-instructions in flight per wave.
//   make build/ubench_glds && ./build/ubench_glds > profiles/r04_ubench_glds.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LDS_PTR(T) __attribute__((address_space(3))) T*
#define GLB_PTR(T) __attribute__((address_space(1))) T*

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((GLB_PTR(const uint32_t))gsrc, (LDS_PTR(uint32_t))lds_wave_base, 16, 0, 0);
}

// SHAPE 0: 16 rows x 64 B   SHAPE 1: 8 rows x 128 B   SHAPE 2: 1 KiB linear.   INFL: wave-instructions kept in flight (batch of INFL/2 issued,
// then wait until INFL/2 remain).  REG = 1: plain global_load_dwordx4 into registers instead (the same addresses), for comparison.
template <int SHAPE, int INFL, int REG>
__global__ void stream_kernel(const char* __restrict__ src, long ld, long rows, int ksteps, int iters, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    char* ring = smem + wave * 16384;
    constexpr int RPI = SHAPE == 0 ? 16 : (SHAPE == 1 ? 8 : 0);     // rows per wave-instruction
    constexpr int BPR = SHAPE == 0 ? 64 : 128;                       // bytes per row piece
    long rowblk = ((long)blockIdx.x * nw + wave);
    const long nblk = SHAPE == 2 ? (rows * ld) / (1024L * ksteps) : rows / RPI;
    const long lane_off = SHAPE == 0 ? (long)(lane >> 2) * ld + (lane & 3) * 16 : (SHAPE == 1 ? (long)(lane >> 3) * ld + (lane & 7) * 16 : (long)lane * 16);
    constexpr int H = INFL / 2;
    uint4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
        rowblk %= nblk;
        const char* base = SHAPE == 2 ? src + rowblk * 1024 * (long)ksteps : src + rowblk * RPI * ld;
        for (int k = 0; k < ksteps; k += H) {
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const char* g = base + (SHAPE == 2 ? (long)(k + j) * 1024 : (long)(k + j) * BPR) + lane_off;
                if (REG) { const uint4 v = *reinterpret_cast<const uint4*>(g); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
                else glds16(g, ring + ((slot + j) & 15) * 1024);
            }
            slot = (slot + H) & 15;
            if (!REG) {
                if (H == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (H == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            }
        }
        rowblk += (long)gridDim.x * nw;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
    if (REG && acc.x == 0x12345678u && acc.y == 1u) sink[0] = (float)acc.z + (float)acc.w;
}

template <int SHAPE, int INFL, int REG>
static void run(const char* name, const char* src, size_t ws_bytes, long ld, int waves, int wg_per_cu, long long* dcyc, float* sink) {
    const int grid = 256 * wg_per_cu, threads = waves * 64, lds = waves * 16384;
    const int ksteps = SHAPE == 2 ? 24 : (int)(ld / (SHAPE == 0 ? 64 : 128));   // 24 / 12 / 24: multiples of every batch size used
    const long rows = (long)(ws_bytes / ld) - 64;
    const long per_iter = (long)ksteps * 1024;                       // bytes per wave and iteration
    int iters = (int)((64L << 20) / per_iter / 8);                   // ~8 MB per wave ... scaled below
    iters = iters < 8 ? 8 : iters;
    auto k = stream_kernel<SHAPE, INFL, REG>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, src, ld, rows, ksteps, iters / 4 + 1, dcyc, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, src, ld, rows, ksteps, iters, dcyc, sink);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    static long long h[2048]; CK(hipMemcpy(h, dcyc, grid * 8, hipMemcpyDeviceToHost));
    double mc = 0; for (int i = 0; i < grid; ++i) mc += (double)h[i]; mc /= grid;
    const double bytes = (double)grid * waves * iters * per_iter;
    printf("%-10s %-22s %dw x %d/CU infl %2d ws %7.1f MB: %8.1f us  %6.2f TB/s  %6.1f GB/s/CU  %5.1f B/clk/CU (readcyclecounter: %.0f ticks)\n", name, REG ? "global_load->VGPR" : "global_load_lds", waves, wg_per_cu, INFL,
           ws_bytes / 1048576.0, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256, bytes / 256 / mc, mc);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const size_t cap = 2048UL << 20;
    char* src; CK(hipMalloc(&src, cap)); CK(hipMemset(src, 1, cap));
    long long* dcyc; CK(hipMalloc(&dcyc, 2048 * 8)); float* sink; CK(hipMalloc(&sink, 64));
    const long ld = 1536;                                            // a K = 768 bf16 row
    const size_t sets[3] = {2UL << 20, 96UL << 20, 2048UL << 20};
    for (int s = 0; s < 3; ++s) {
        printf("---- working set %.0f MB (%s)\n", sets[s] / 1048576.0, s == 0 ? "L2-resident" : (s == 1 ? "Infinity-Cache-resident" : "HBM"));
        run<0, 12, 0>("16r x 64B", src, sets[s], ld, 8, 1, dcyc, sink);
        run<0, 24, 0>("16r x 64B", src, sets[s], ld, 8, 1, dcyc, sink);
        run<0, 8, 0>("16r x 64B", src, sets[s], ld, 8, 1, dcyc, sink);
        run<0, 12, 0>("16r x 64B", src, sets[s], ld, 4, 2, dcyc, sink);
        run<0, 12, 0>("16r x 64B", src, sets[s], ld, 4, 1, dcyc, sink);
        run<1, 12, 0>("8r x 128B", src, sets[s], ld, 8, 1, dcyc, sink);
        run<1, 12, 0>("8r x 128B", src, sets[s], ld, 4, 2, dcyc, sink);
        run<2, 12, 0>("1KiB lin", src, sets[s], ld, 8, 1, dcyc, sink);
        run<2, 12, 0>("1KiB lin", src, sets[s], ld, 4, 2, dcyc, sink);
        run<0, 12, 1>("16r x 64B", src, sets[s], ld, 8, 1, dcyc, sink);
        run<1, 12, 1>("8r x 128B", src, sets[s], ld, 8, 1, dcyc, sink);
        run<2, 12, 1>("1KiB lin", src, sets[s], ld, 8, 1, dcyc, sink);
    }
    return 0;
}
