// Ablations of the 160 x 256 two-per-CU NT kernel (gemm_nt_c2.hip): what does its main loop wait for?
//   make build/ubench_c2 && ./build/ubench_c2 > profiles/r04_ubench_c2.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../tcow_amd/csrc/gemm_nt_c2.hip"
void tcow_set_error(const char*, ...) {}
void tcow_ensure_lds(const void* k, int bytes) { (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int AB>
static void run(const char* what, int M, int N, int K, const bf16_t* A, const bf16_t* W, bf16_t* C, int lds, float skew_us) {
    NtParams p; memset(&p, 0, sizeof(p));
    p.M = M; p.N = N; p.K = K; static const int pad = getenv("LD_PAD") ? atoi(getenv("LD_PAD")) : 0;
    p.A = A; p.lda = K + pad; p.W = W; p.ldw = K + pad; p.C = C; p.ldc = N; p.out_f32 = 0;
    p.tiles_m = (M + D_BM - 1) / D_BM; p.tiles_n = (N + D_BN - 1) / D_BN;
    p.skew = (int)(skew_us * 100.f * (K / 64)); p.skew_mode = 1;
    auto k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_NONE, 0>, AB>;
    tcow_ensure_lds((const void*)k, 140 * 1024);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = p.tiles_m * p.tiles_n;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, p);
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-58s %dx%dx%d  LDS %3d KiB (%d/CU) skew %.2f: %7.1f us  %5.0f TFLOP/s\n", what, M, K, N, lds / 1024, lds > 81920 ? 1 : 2, skew_us, ms * 100, 2.0 * M * N * K / (ms * 1e-4) / 1e12);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int M = 27090;
    std::vector<uint16_t> h((size_t)M * 3072);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) { const float v = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; memcpy(&u, &v, 4); h[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
    bf16_t *A, *W, *C;
    CK(hipMalloc(&A, (size_t)M * 4096 * 2)); CK(hipMalloc(&W, (size_t)3072 * 4096 * 2)); CK(hipMalloc(&C, (size_t)M * 3072 * 2));
    CK(hipMemcpy(A, h.data(), (size_t)M * 3072 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), (size_t)3072 * 3072 * 2, hipMemcpyHostToDevice));
    const int shapes[3][2] = {{768, 3072}, {2304, 768}, {768, 768}};
    const int nshapes = getenv("C2_ALL") ? 3 : 1;      // N, K
    for (int s = 0; s < nshapes; ++s) {
        const int N = shapes[s][0], K = shapes[s][1];
        printf("---- N = %d, K = %d\n", N, K);
        if (getenv("C2_SHORT")) {
            run<8>("no epilogue", M, N, K, A, W, C, D_LDS, 0.f);
            run<8 + 16 + 32>("no epilogue, loads only", M, N, K, A, W, C, D_LDS, 0.f);
            run<8 + 16 + 32 + 64>("no epilogue, A loads only", M, N, K, A, W, C, D_LDS, 0.f);
            run<8 + 16 + 32 + 128>("no epilogue, W loads only", M, N, K, A, W, C, D_LDS, 0.f);
            continue;
        }
        run<0>("as shipped", M, N, K, A, W, C, D_LDS, 0.45f);
        run<0>("as shipped, no start skew", M, N, K, A, W, C, D_LDS, 0.f);
        run<0>("one workgroup per CU (LDS request raised)", M, N, K, A, W, C, 100 * 1024, 0.f);
        run<8>("no epilogue", M, N, K, A, W, C, D_LDS, 0.f);
        run<8>("no epilogue, one workgroup per CU", M, N, K, A, W, C, 100 * 1024, 0.f);
        run<9>("no epilogue, operands L2-resident (every WG tile 0,0)", M, N, K, A, W, C, D_LDS, 0.f);
        run<9>("no epilogue, L2-resident, one workgroup per CU", M, N, K, A, W, C, 100 * 1024, 0.f);
        run<10>("no epilogue, no loads in the loop", M, N, K, A, W, C, D_LDS, 0.f);
        run<10>("no epilogue, no loads, one workgroup per CU", M, N, K, A, W, C, 100 * 1024, 0.f);
        run<14>("no epilogue, no loads, no barriers", M, N, K, A, W, C, D_LDS, 0.f);
        run<14>("no epilogue, no loads, no barriers, one WG per CU", M, N, K, A, W, C, 100 * 1024, 0.f);
        run<12>("no epilogue, loads, no barriers (racy: timing only)", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 16>("no epilogue, loads + MFMAs, NO fragment reads", M, N, K, A, W, C, D_LDS, 0.f);
        run<9 + 16>("no epilogue, L2-resident loads + MFMAs, NO fragment reads", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 32>("no epilogue, loads + fragment reads, NO MFMAs", M, N, K, A, W, C, D_LDS, 0.f);
        run<9 + 32>("no epilogue, L2-resident loads + reads, NO MFMAs", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 16 + 32>("no epilogue, loads only", M, N, K, A, W, C, D_LDS, 0.f);
        run<9 + 16 + 32>("no epilogue, L2-resident loads only", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 16 + 32 + 4>("no epilogue, loads only, no barriers", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 16 + 32 + 64>("no epilogue, A loads only", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 16 + 32 + 128>("no epilogue, W loads only", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 16 + 32 + 64 + 4>("no epilogue, A loads only, no barriers", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 16 + 32 + 128 + 4>("no epilogue, W loads only, no barriers", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 64>("no epilogue, everything but the W loads", M, N, K, A, W, C, D_LDS, 0.f);
        run<8 + 128>("no epilogue, everything but the A loads", M, N, K, A, W, C, D_LDS, 0.f);
        run<10 + 32>("no epilogue, fragment reads only", M, N, K, A, W, C, D_LDS, 0.f);
        run<10 + 16>("no epilogue, MFMAs only", M, N, K, A, W, C, D_LDS, 0.f);
    }
    return 0;
}
