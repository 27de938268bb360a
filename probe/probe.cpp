// Hardware-semantics probe for gfx950: validates the MFMA fragment layouts, the LDS
// transpose-read and a few cross-lane primitives that the tcow_amd kernels rely on.
// Build: hipcc --offload-arch=gfx950 -O2 -o probe/probe probe/probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// A: [M][K] row-major bf16 bits, B: [K][N] row-major; assumed layouts from the CDNA4 guide.
__global__ void k_mfma_32x32x16_bf16(const uint16_t* A, const uint16_t* B, float* C) {
    int l = threadIdx.x; int i = l & 31, hi = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        uint16_t av = A[i * 16 + 8 * hi + j]; uint16_t bv = B[(8 * hi + j) * 32 + i];
        a[j] = __builtin_bit_cast(__bf16, av); b[j] = __builtin_bit_cast(__bf16, bv);
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * hi; C[row * 32 + i] = c[r]; }
}
__global__ void k_mfma_16x16x32_bf16(const uint16_t* A, const uint16_t* B, float* C) {
    int l = threadIdx.x; int i = l & 15, g = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        uint16_t av = A[i * 32 + 8 * g + j]; uint16_t bv = B[(8 * g + j) * 16 + i];
        a[j] = __builtin_bit_cast(__bf16, av); b[j] = __builtin_bit_cast(__bf16, bv);
    }
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { int row = 4 * g + r; C[row * 16 + i] = c[r]; }
}
__global__ void k_mfma_32x32x2_f32(const float* A, const float* B, float* C) {
    int l = threadIdx.x; int i = l & 31, hi = l >> 5;
    float a = A[i * 2 + hi], b = B[hi * 32 + i];
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * hi; C[row * 32 + i] = c[r]; }
}
__global__ void k_mfma_16x16x4_f32(const float* A, const float* B, float* C) {
    int l = threadIdx.x; int i = l & 15, g = l >> 4;
    float a = A[i * 4 + g], b = B[g * 16 + i];
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { int row = 4 * g + r; C[row * 16 + i] = c[r]; }
}

// LDS transpose read: lds[e] = e (u16). mode 0: lane address = lane*8 B (linear).
// mode 1: lane i (in 16-group g) supplies row (i&3), quad (i>>2) of a [4][16] block at g*128 B.
__global__ void k_trread(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    int l = threadIdx.x;
    for (int e = l; e < 1024; e += 64) lds[e] = (uint16_t)e;
    __syncthreads();
    int g = l >> 4, i = l & 15;
    int elem_off;
    if (mode == 0) elem_off = l * 4;                                        // linear: lane*8 B
    else if (mode == 1) elem_off = g * 64 + (i & 3) * 16 + (i >> 2) * 4;    // lane 4c+r -> row r quad c
    else elem_off = g * 8 + (i >> 2) * 40 + (i & 3) * 4 + 256;              // rows of stride 40 elems (80 B), group g at col 8g... offset
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    lds_ptr p = (lds_ptr)(&lds[elem_off]);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    out[l * 4 + 0] = (uint16_t)(bits & 0xffff); out[l * 4 + 1] = (uint16_t)((bits >> 16) & 0xffff);
    out[l * 4 + 2] = (uint16_t)((bits >> 32) & 0xffff); out[l * 4 + 3] = (uint16_t)((bits >> 48) & 0xffff);
}

__global__ void k_permlane32_swap(unsigned* out) {
    int l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l * 2 + 0] = r[0]; out[l * 2 + 1] = r[1];
}

// global_load_lds 16 B: each lane copies 16 B from src + lane*16 to lds base + lane*16 (wave-uniform base).
__global__ void k_glds(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * 4];
    int l = threadIdx.x; int w = l >> 6;
    typedef __attribute__((address_space(3))) uint32_t* lds_ptr_t;
    typedef __attribute__((address_space(1))) const uint32_t* g_ptr_t;
    // per-wave region of 1 KiB
    __builtin_amdgcn_global_load_lds((g_ptr_t)(src + l * 4), (lds_ptr_t)(&lds[w * 256]), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = lds[l * 4 + j];
}

// bandwidth: float4 copy
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}

template <typename TA> static double maxdiff(const std::vector<float>& ref, const float* got, int n) { double m = 0; for (int i = 0; i < n; ++i) m = fmax(m, fabs(ref[i] - got[i])); return m; }

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s arch=%s CUs=%d LDS/block=%zu regs/block=%d clock=%d MHz mem=%.1f GB L2=%d\n", p.name, p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, p.regsPerBlock, p.clockRate / 1000, p.totalGlobalMem / 1e9, p.l2CacheSize);
    printf("maxSharedMemoryPerMultiProcessor=%zu warpSize=%d\n", p.maxSharedMemoryPerMultiProcessor, p.warpSize);
    srand(1);
    // ---- MFMA bf16 32x32x16
    {
        int M = 32, N = 32, K = 16; std::vector<uint16_t> A(M * K), B(K * N); std::vector<float> ref(M * N, 0.f);
        for (auto& x : A) x = f2bf((rand() % 17 - 8) / 4.f); for (auto& x : B) x = f2bf((rand() % 13 - 6) / 2.f);
        for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { float s = 0; for (int k = 0; k < K; ++k) s += bf2f(A[i * K + k]) * bf2f(B[k * N + j]); ref[i * N + j] = s; }
        uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, M * N * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma_32x32x16_bf16<<<1, 64>>>(dA, dB, dC); std::vector<float> C(M * N); CK(hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
        printf("mfma_f32_32x32x16_bf16 layout check: maxdiff=%g\n", maxdiff<float>(ref, C.data(), M * N));
    }
    {
        int M = 16, N = 16, K = 32; std::vector<uint16_t> A(M * K), B(K * N); std::vector<float> ref(M * N, 0.f);
        for (auto& x : A) x = f2bf((rand() % 17 - 8) / 4.f); for (auto& x : B) x = f2bf((rand() % 13 - 6) / 2.f);
        for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { float s = 0; for (int k = 0; k < K; ++k) s += bf2f(A[i * K + k]) * bf2f(B[k * N + j]); ref[i * N + j] = s; }
        uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, M * N * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma_16x16x32_bf16<<<1, 64>>>(dA, dB, dC); std::vector<float> C(M * N); CK(hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
        printf("mfma_f32_16x16x32_bf16 layout check: maxdiff=%g\n", maxdiff<float>(ref, C.data(), M * N));
    }
    {
        int M = 32, N = 32, K = 2; std::vector<float> A(M * K), B(K * N), ref(M * N, 0.f);
        for (auto& x : A) x = (rand() % 17 - 8) / 4.f; for (auto& x : B) x = (rand() % 13 - 6) / 2.f;
        for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { float s = 0; for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * N + j]; ref[i * N + j] = s; }
        float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, M * N * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        k_mfma_32x32x2_f32<<<1, 64>>>(dA, dB, dC); std::vector<float> C(M * N); CK(hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
        printf("mfma_f32_32x32x2f32 layout check: maxdiff=%g\n", maxdiff<float>(ref, C.data(), M * N));
    }
    {
        int M = 16, N = 16, K = 4; std::vector<float> A(M * K), B(K * N), ref(M * N, 0.f);
        for (auto& x : A) x = (rand() % 17 - 8) / 4.f; for (auto& x : B) x = (rand() % 13 - 6) / 2.f;
        for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { float s = 0; for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * N + j]; ref[i * N + j] = s; }
        float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, M * N * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        k_mfma_16x16x4_f32<<<1, 64>>>(dA, dB, dC); std::vector<float> C(M * N); CK(hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
        printf("mfma_f32_16x16x4f32 layout check: maxdiff=%g\n", maxdiff<float>(ref, C.data(), M * N));
    }
    // ---- tr read
    for (int mode = 0; mode < 3; ++mode) {
        uint16_t* d; CK(hipMalloc(&d, 256 * 2)); k_trread<<<1, 64>>>(d, mode); std::vector<uint16_t> h(256); CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
        printf("ds_read_tr16_b64 mode %d (lds[e]=e; result per lane, 4 elems):\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); }
    }
    {
        unsigned* d; CK(hipMalloc(&d, 128 * 4)); k_permlane32_swap<<<1, 64>>>(d); std::vector<unsigned> h(128); CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
        printf("permlane32_swap(a=1000+l, b=2000+l): lane0 r=(%u,%u) lane5 r=(%u,%u) lane32 r=(%u,%u) lane37 r=(%u,%u)\n", h[0], h[1], h[10], h[11], h[64], h[65], h[74], h[75]);
    }
    {
        uint32_t *s, *d; CK(hipMalloc(&s, 4096)); CK(hipMalloc(&d, 4096)); std::vector<uint32_t> h(1024); for (int i = 0; i < 1024; ++i) h[i] = i * 3 + 1; CK(hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice));
        k_glds<<<1, 256>>>(s, d); std::vector<uint32_t> o(1024); CK(hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost)); int bad = 0; for (int i = 0; i < 1024; ++i) bad += (o[i] != h[i]);
        printf("global_load_lds 16B linear copy: mismatches=%d\n", bad);
    }
    {
        size_t bytes = 1ull << 30; float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int it = 0; it < 3; ++it) k_copy<<<2048, 256>>>(a, b, bytes / 16);
        CK(hipEventRecord(e0)); for (int it = 0; it < 10; ++it) k_copy<<<2048, 256>>>(a, b, bytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("float4 copy 1 GiB: %.1f us/iter, %.2f TB/s (read+write)\n", ms * 100, 2.0 * bytes * 10 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
