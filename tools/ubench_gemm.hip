// Phase timeline of the phase-structured NT main loop (gemm_p8.hip built with TCOW_P8_DBG): shader-clock stamps of one K tile in steady state,
// per wave, averaged over the workgroups.   make ubench_gemm && ./build/ubench_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define TCOW_P8_DBG 1
#include "gemm_p8.hip"
void tcow_set_error(const char*, ...) {}
void tcow_ensure_lds(const void* k, int bytes) { (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BM>
static void run(int M, int N, int K) {
    std::vector<uint16_t> h((size_t)(M > N ? M : N) * K);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) { const float v = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; memcpy(&u, &v, 4); h[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
    bf16_t *A, *W, *C; long long* dbg;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    P8Params p; p.M = M; p.N = N; p.K = K; p.A = A; p.lda = K; p.W = W; p.ldw = K; p.C = C; p.ldc = N; p.out_f32 = 0; p.bias = nullptr;
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + 255) / 256;
    const int nblk = p.tiles_m * p.tiles_n, grid = 8 * ((nblk + 7) / 8);
    CK(hipMalloc(&dbg, (size_t)grid * 8 * 32 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 8 * 32 * 8));
    p.dbg = dbg; p.dbg_kt = -1;
    const int lds = 2 * 2 * (BM / 16 + 16) * 1024;
    tcow_ensure_lds((const void*)gemm_nt_p8_kernel<BM>, lds);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gemm_nt_p8_kernel<BM>, dim3(grid), dim3(512), lds, 0, p);
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(gemm_nt_p8_kernel<BM>, dim3(grid), dim3(512), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("P8 BM=%d  %d x %d x %d: %.1f us, %.0f TFLOP/s (no stamps)\n", BM, M, K, N, ms * 100, 2.0 * M * N * K / (ms * 1e-4) / 1e12);
    p.dbg_kt = (K / 64) / 2;
    hipLaunchKernelGGL(gemm_nt_p8_kernel<BM>, dim3(grid), dim3(512), lds, 0, p); CK(hipDeviceSynchronize());
    std::vector<long long> t((size_t)grid * 8 * 32); CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
    const char* nm[23] = {"p1 reads+loads issue", "p1 barrier 1", "p1 wait LDS", "p1 MFMA block issue", "p1 barrier 2",
                          "p2 reads+loads issue", "p2 wait loads (vmcnt)", "p2 barrier 1", "p2 wait LDS", "p2 MFMA block issue", "p2 barrier 2",
                          "p3 reads+loads issue", "p3 barrier 1", "p3 wait LDS", "p3 MFMA block issue", "p3 barrier 2",
                          "p4 reads+loads issue", "p4 wait loads (vmcnt)", "p4 barrier 1", "p4 wait LDS", "p4 MFMA block issue", "p4 barrier 2", ""};
    for (int w = 0; w < 8; w += 4) {
        double acc[22] = {0}; int n = 0; double tot = 0;
        for (int b = 0; b < grid; ++b) {
            const long long* s = &t[(size_t)(b * 8 + w) * 32];
            if (!s[22] || !s[0]) continue;
            ++n; for (int i = 0; i < 22; ++i) acc[i] += (double)(s[i + 1] - s[i]);
            tot += (double)(s[22] - s[0]);
        }
        printf("  wave %d (wave row %d), K tile %d of %d, mean over %d workgroups: %.0f cycles per K tile\n", w, w >> 2, p.dbg_kt, K / 64, n, tot / n);
        for (int i = 0; i < 22; ++i) printf("    %-24s %7.0f\n", nm[i], acc[i] / n);
    }
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(dbg));
}
int main() {
    run<256>(4096, 4096, 4096);
    run<320>(27090, 768, 3072);
    run<320>(27090, 2304, 768);
    return 0;
}
