// Ablations of the 256 x 256 weight-gradient GEMM (gemm_bf16.hip, tn256_body): which part of a 64-token stage does the loop wait for?
//   make build/ubench_tn_ab && ./build/ubench_tn_ab > profiles/r04_ubench_tn_ab.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include "../tcow_amd/csrc/gemm_bf16.hip"
void tcow_set_error(const char*, ...) {}
void tcow_ensure_lds(const void* k, int bytes) { (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
bool tcow_gemm_nt_c2_ok(const tcow_gemm_args*) { return false; }
int tcow_gemm_nt_bf16_c2(hipStream_t, const tcow_gemm_args*) { return -1; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int AB, int SCHED>
__global__ __launch_bounds__(512, 2) void tn_ab_kernel(TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn256_body<AB, SCHED>(p, xcd_remap(blockIdx.x, gridDim.x), smem);
}

template <int AB, int SCHED = 2>
static void run(const char* what, const TnParams& p, int grid, double flops) {
    auto k = tn_ab_kernel<AB, SCHED>;
    tcow_ensure_lds((const void*)k, T2_LDS);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(512), T2_LDS, 0, p);
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(512), T2_LDS, 0, p);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-66s %7.1f us  %5.0f TFLOP/s\n", what, ms * 100, flops / (ms * 1e-4) / 1e12);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int M = 27090, N = 3072, K = 768, nzr = 7;
    std::vector<uint16_t> h((size_t)M * N);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) { const float v = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; memcpy(&u, &v, 4); h[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
    bf16_t *dY, *X; float *slab, *bpart;
    CK(hipMalloc(&dY, (size_t)M * N * 2)); CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&slab, (size_t)(nzr + 1) * N * K * 4)); CK(hipMalloc(&bpart, (size_t)64 * 24 * 2 * N * 4));
    CK(hipMemcpy(dY, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(X, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    TnParams p; p.M = M; p.N = N; p.K = K; p.dY = dY; p.ldy = N; p.X = X; p.ldx = K; p.slab = slab;
    p.tiles_n = (N + 255) / 256; p.tiles_k = (K + 255) / 256;
    int mps = (M + nzr - 1) / nzr; mps = ((mps + 63) / 64) * 64; p.mps = mps; p.nz = (M + mps - 1) / mps;
    p.bias_part = bpart; p.rows_per_pk = (64 + p.tiles_k - 1) / p.tiles_k;
    const int grid = p.nz * p.tiles_n * p.tiles_k;
    const double fl = 2.0 * M * N * K;
    printf("TN 256 tile  dW[%d x %d] over %d token rows, %d slices = %d workgroups (%.2f rounds of 256 CUs), %d stages of 64 tokens each\n", N, K, M, p.nz, grid, grid / 256.0, mps / 64);
    run<0>("as shipped (reads between the MFMAs in every k-step)", p, grid, fl);
    run<0>("  (again)", p, grid, fl);
    run<0, 1>("requests + reads in one burst behind the barrier", p, grid, fl);
    run<0, 0>("round-3 order (wait + barrier at the stage end)", p, grid, fl);
    run<8 + 64, 0>("round-3 order, no slab store, no column sums", p, grid, fl);
    run<64>("no bias column sums", p, grid, fl);
    run<8>("no slab store", p, grid, fl);
    run<8 + 64>("no slab store, no column sums", p, grid, fl);
    run<8 + 64 + 2>("  ... no loads after the first stage", p, grid, fl);
    run<8 + 64 + 4>("  ... no barriers (racy: timing only)", p, grid, fl);
    run<8 + 64 + 2 + 4>("  ... no loads, no barriers", p, grid, fl);
    run<8 + 64 + 2 + 4 + 128, 1>("  ... no loads, no barriers, MFMAs on constant operands (reads still issued, burst)", p, grid, fl);
    run<8 + 64 + 2 + 4 + 128, 2>("  ... no loads, no barriers, MFMAs on constant operands (reads between MFMAs)", p, grid, fl);
    run<8 + 64 + 2 + 4, 2>("  ... no loads, no barriers (reads between MFMAs)", p, grid, fl);
    run<8 + 64 + 2 + 4 + 512, 2>("  ... no loads, no barriers, MFMAs on 12 distinct constant operands (reads into the usual registers)", p, grid, fl);
    run<8 + 64 + 2 + 4 + 512 + 16, 2>("  ... the same without the reads", p, grid, fl);
    run<8 + 64 + 2 + 4 + 512 + 1024, 2>("  ... distinct constant operands holding RANDOM values in (-1, 1), reads issued", p, grid, fl);
    run<8 + 64 + 2 + 4 + 512 + 1024 + 16, 2>("  ... the same without the reads (MFMAs only, random operand values)", p, grid, fl);
    run<8 + 64 + 2 + 4 + 256, 2>("  ... no loads, no barriers, no waits for the reads (racy: timing only)", p, grid, fl);
    run<8 + 64 + 2 + 4 + 16, 2>("  ... MFMAs only (reads between MFMAs)", p, grid, fl);
    run<8 + 64 + 2 + 4 + 32, 2>("  ... reads only (reads between MFMAs)", p, grid, fl);
    run<8 + 64 + 16>("  ... no transpose reads", p, grid, fl);
    run<8 + 64 + 32>("  ... no MFMAs (loads + reads + barriers)", p, grid, fl);
    run<8 + 64 + 16 + 32>("  ... loads + barriers only", p, grid, fl);
    run<8 + 64 + 16 + 32 + 4>("  ... loads only, no barriers", p, grid, fl);
    run<8 + 64 + 2 + 32>("  ... transpose reads + barriers only", p, grid, fl);
    run<8 + 64 + 2 + 16>("  ... MFMAs + barriers only", p, grid, fl);
    run<8 + 64 + 2 + 16 + 4>("  ... MFMAs only", p, grid, fl);
    run<8 + 64 + 2 + 4 + 32>("  ... transpose reads only", p, grid, fl);
    return 0;
}
