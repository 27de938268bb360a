// Phase timeline of the shipped weight-gradient loop (gemm_bf16.hip built with TCOW_TN_DBG).   make ubench_tn && ./build/ubench_tn
// CAVEAT: every stamp is an s_memtime whose result comes back through lgkmcnt, i.e. it drains the wave's LDS queue -- this loop keeps the next
// k-step's transpose reads in flight under the MFMAs, so the stamped build runs ~40 % slower and its read phases are inflated.  Good for
// spotting phases that are long for OTHER reasons (it pointed at the load-address arithmetic and the bias column sums), not for budgets.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#define TCOW_TN_DBG 1
#include "../tcow_amd/csrc/gemm_bf16.hip"
void tcow_set_error(const char*, ...) {}
void tcow_ensure_lds(const void* k, int bytes) { (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
    const int M = 27090, N = 3072, K = 768, nzr = 7;
    std::vector<uint16_t> h((size_t)M * N);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) { const float v = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; memcpy(&u, &v, 4); h[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
    bf16_t *dY, *X; float *slab, *bpart; long long* dbg;
    CK(hipMalloc(&dY, (size_t)M * N * 2)); CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&slab, (size_t)(nzr + 1) * N * K * 4)); CK(hipMalloc(&bpart, (size_t)64 * 24 * 2 * N * 4));
    CK(hipMemcpy(dY, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(X, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    TnParams p; p.M = M; p.N = N; p.K = K; p.dY = dY; p.ldy = N; p.X = X; p.ldx = K; p.slab = slab;
    p.tiles_n = (N + 255) / 256; p.tiles_k = (K + 255) / 256;
    int mps = (M + nzr - 1) / nzr; mps = ((mps + 63) / 64) * 64; p.mps = mps; p.nz = (M + mps - 1) / mps;
    p.bias_part = bpart; p.rows_per_pk = (64 + p.tiles_k - 1) / p.tiles_k;
    const int grid = p.nz * p.tiles_n * p.tiles_k;
    CK(hipMalloc(&dbg, (size_t)grid * 8 * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 8 * 16 * 8));
    int it = -1; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tn_dbg), &dbg, sizeof(dbg))); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tn_dbg_it), &it, sizeof(it)));
    tcow_ensure_lds((const void*)gemm_tn_bf16_256_kernel, T2_LDS);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gemm_tn_bf16_256_kernel, dim3(grid), dim3(512), T2_LDS, 0, p);
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(gemm_tn_bf16_256_kernel, dim3(grid), dim3(512), T2_LDS, 0, p);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("TN 256 tile  %d x %d x %d, %d slices (%d workgroups): %.1f us, %.0f TFLOP/s (no stamps)\n", M, N, K, p.nz, grid, ms * 100, 2.0 * M * N * K / (ms * 1e-4) / 1e12);
    it = (mps / 64) / 2; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tn_dbg_it), &it, sizeof(it)));
    hipLaunchKernelGGL(gemm_tn_bf16_256_kernel, dim3(grid), dim3(512), T2_LDS, 0, p); CK(hipDeviceSynchronize());
    std::vector<long long> t((size_t)grid * 8 * 16); CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
    const char* nm[9] = {"issue 8 loads of the next stage", "reads k-step 1 + wait k-step 0", "MFMA 0 + reads 2 + wait 1", "MFMA 1 + reads 3 + wait 2", "MFMA 2 + wait reads 3",
                         "MFMA 3 (issue)", "bias column sums", "drain loads + LDS", "barrier"};
    for (int w = 0; w < 8; w += 4) {
        double acc[9] = {0}; int n = 0; double tot = 0;
        for (int b = 0; b < grid; ++b) {
            const long long* s = &t[(size_t)(b * 8 + w) * 16];
            if (!s[9] || !s[0]) continue;
            ++n; for (int i = 0; i < 9; ++i) acc[i] += (double)(s[i + 1] - s[i]);
            tot += (double)(s[9] - s[0]);
        }
        printf("  wave %d, stage %d of %d, mean over %d workgroups: %.0f cycles per 64-token stage (32 MFMAs = 1024 matrix-pipe cycles per wave)\n", w, it, mps / 64, n, tot / n);
        for (int i = 0; i < 9; ++i) printf("    %-36s %7.0f\n", nm[i], acc[i] / n);
    }
    return 0;
}
