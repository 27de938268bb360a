// What the MFMA pipe of an MI355X sustains, by instruction shape and by operand VALUES: every wave runs independent accumulate chains on operands
// it keeps in registers (no memory traffic in the loop).  make build/ubench_mfma && ./build/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ u32x4 operand(int seed, int mode) {
    u32x4 v;
    for (int c = 0; c < 4; ++c) {
        const uint32_t h = (uint32_t)(threadIdx.x * 97 + blockIdx.x * 13 + seed * 7 + c) * 2654435761u;
        // mode 0: tiny constants (near-zero bf16 pairs); 1: random values in (-1, 1) (random sign, 7 mantissa bits, exponents 126 / 127 -> [0.5, 2) scaled); 2: random normal-ish spread of exponents
        if (mode == 0) v[c] = 0x00010002u + c;
        else if (mode == 1) v[c] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
        else v[c] = (h & 0x87ff87ffu) | 0x38003800u;                      // exponents 112..127: values over five decades
    }
    return v;
}

template <int SHAPE, int MODE>
__global__ __launch_bounds__(512) void mfma_kernel(int iters, float* out) {
    u32x4 a[4], b[2];
    for (int i = 0; i < 4; ++i) { a[i] = operand(i, MODE); asm volatile("" : "+v"(a[i])); }
    for (int j = 0; j < 2; ++j) { b[j] = operand(10 + j, MODE); asm volatile("" : "+v"(b[j])); }
    float sink = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) sink += acc[i][j][0];
    } else {
        f32x4 acc[4][2][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)       // (two 16x16x32 are the flops of one 32x32x16: this loop body is 2 x the other one's work)
                        acc[i][j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(i + q) & 3]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j][q], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) sink += acc[i][j][q][0];
    }
    if (sink == 123.456f) out[0] = sink;
}

template <int SHAPE, int MODE> static int run(const char* what, float* out) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((mfma_kernel<SHAPE, MODE>), dim3(grid), dim3(512), 0, 0, iters, out);       // warm
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((mfma_kernel<SHAPE, MODE>), dim3(grid), dim3(512), 0, 0, iters, out);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = 10.0 * grid * 8 /*waves*/ * (double)iters * (SHAPE == 32 ? 8 * 2.0 * 32 * 32 * 16 : 32 * 2.0 * 16 * 16 * 32);
        printf("  %-78s %s %7.1f ms  %6.0f TFLOP/s\n", what, rep ? "(again)" : "       ", ms, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* out; CK(hipMalloc(&out, 64));
    printf("MFMA-only streams, 256 workgroups x 8 waves (2 per SIMD), 8 independent accumulate chains per wave, 10 launches of ~20 ms each\n");
    run<32, 0>("v_mfma_f32_32x32x16_bf16, operands = tiny constants", out);
    run<32, 1>("v_mfma_f32_32x32x16_bf16, operands = random bf16 in (-1, 1)", out);
    run<32, 2>("v_mfma_f32_32x32x16_bf16, operands = random bf16 over five decades", out);
    run<16, 0>("v_mfma_f32_16x16x32_bf16, operands = tiny constants", out);
    run<16, 1>("v_mfma_f32_16x16x32_bf16, operands = random bf16 in (-1, 1)", out);
    run<16, 2>("v_mfma_f32_16x16x32_bf16, operands = random bf16 over five decades", out);
    run<32, 0>("v_mfma_f32_32x32x16_bf16, tiny constants (after the chip is warm)", out);
    return 0;
}
