// Microbenchmarks that price the softmax arithmetic of the attention kernels on gfx950.
//
//   A  VALU issue cost: cycles per wave64 instruction on ONE SIMD with 1 / 2 / 4 waves resident on it, independent streams (throughput)
//      and one dependent chain (latency): v_fma_f32, v_add_f32, v_mul_f32, v_max3_f32, v_exp_f32, v_cvt_pk_bf16_f32, v_pk_mul_f32,
//      v_permlane32_swap.
//   B  the same instructions as FILLERS between v_mfma_f32_32x32x16_bf16 (4 independent accumulators, so the matrix pipe is the only
//      limit at 0 fillers): cycles per MFMA for n = 0 .. 16 fillers per MFMA gap, 1 and 2 waves per SIMD.
//   C  the shipped forward tile function (attention_bf16.hip::fwd_tile) on LDS-resident K / V tiles, no global traffic inside the
//      timed loop: cycles per (32 query x 32 key) step and per SIMD with 1 .. 4 waves per SIMD -> what the step ARITHMETIC costs when
//      nothing waits for memory, i.e. the MFMA utilisation ceiling of this formulation (8 MFMAs = 256 matrix-pipe cycles per step).
//
// Cycles are s_memtime ticks (= shader cycles, MI355X_MICROARCH.md "Per-instruction cycle constants"), taken per wave around the
// timed loop; one workgroup on one CU, so "per SIMD" = the longest wave of that SIMD.   build: make ubench   run: build/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <string>

#define UBENCH_ATTN 1
#include "../tcow_amd/csrc/attention_bf16.hip"
#include "attn_fwd_variants.inc"

// the library's error plumbing, not linked here
void tcow_set_error(const char*, ...) {}
void tcow_ensure_lds(const void*, int) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA, ADD, MUL, MAX3, EXP, CVTPK, PKMUL, SWAP, NOPS };
static const char* op_name[NOPS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_max3_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_pk_mul_f32", "v_permlane32_swap"};

template <int OP>
__device__ __forceinline__ void one_op(float& x, float& y, float c) {
    if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    else if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    else if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    else if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(y));
    else if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if (OP == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    else if (OP == SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void pk_mul(f32x2& x, f32x2 c) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c)); }

// ---- A: plain VALU streams.  DEP = 1: one dependent chain; DEP = 0: 8 independent registers round-robin
template <int OP, int DEP>
__global__ __launch_bounds__(1024) void valu_kernel(int iters, float seed, long long* cyc, float* sink) {
    float x[8], y[8];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = seed + i + threadIdx.x * 1e-3f; y[i] = seed * 0.5f + i; p[i] = f32x2{x[i], y[i]}; }
    const float c = seed * 1.0001f;
    const f32x2 c2 = {c, c};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int r = DEP ? 0 : (u & 7);
            if (OP == PKMUL) pk_mul(p[r], c2); else one_op<OP>(x[r], y[r], c);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += x[i] + y[i] + p[i][0] + p[i][1];
    if (acc == 123.456f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// ---- B: fillers between MFMAs
template <int OP, int NF>
__global__ __launch_bounds__(512) void mfma_fill_kernel(int iters, float seed, long long* cyc, float* sink) {
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 fa, fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa[e] = (tcow_h16)(seed + e * 0.01f); fb[e] = (tcow_h16)(seed * 0.5f - e * 0.02f); }
    float x[8], y[8];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = seed + i; y[i] = seed * 0.5f + i; p[i] = f32x2{x[i], y[i]}; }
    const float c = seed * 1.0001f;
    const f32x2 c2 = {c, c};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[a]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                const int r = (a * NF + u) & 7;
                if (OP == PKMUL) pk_mul(p[r], c2); else one_op<OP>(x[r], y[r], c);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][15];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + y[i] + p[i][0];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// ---- B2: the instruction MIX of one softmax step (8 MFMAs, 16 v_exp_f32, 16 v_add_f32, 8 v_cvt_pk_bf16_f32, 8 v_max3_f32, 8 other VALU),
// all independent, issued either clustered (8 MFMAs, then the VALU) or interleaved (1 MFMA : 7 VALU): what in-wave software pipelining
// of the attention step could reach
template <int INTERLEAVE>
__global__ __launch_bounds__(768) void mix_kernel(int iters, float seed, long long* cyc, float* sink) {
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 fa, fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa[e] = (tcow_h16)(seed + e * 0.01f); fb[e] = (tcow_h16)(seed * 0.5f - e * 0.02f); }
    float x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = seed + i; y[i] = seed * 0.5f + i; }
    const float c = seed * 1.0001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (INTERLEAVE) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k & 3]) : "v"(fa), "v"(fb));
                one_op<EXP>(x[0], y[0], c); one_op<ADD>(x[1], y[1], c); one_op<MAX3>(x[2], y[2], c); one_op<EXP>(x[3], y[3], c);
                one_op<ADD>(x[4], y[4], c); one_op<CVTPK>(x[5], y[5], c); one_op<FMA>(x[6], y[6], c);
            }
        }
        if (!INTERLEAVE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k & 3]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                one_op<EXP>(x[0], y[0], c); one_op<ADD>(x[1], y[1], c); one_op<MAX3>(x[2], y[2], c); one_op<EXP>(x[3], y[3], c);
                one_op<ADD>(x[4], y[4], c); one_op<CVTPK>(x[5], y[5], c); one_op<FMA>(x[6], y[6], c);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][15];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// ---- C: the shipped forward step on LDS-resident tiles
__global__ __launch_bounds__(1024) void step_kernel(int iters, const bf16_t* __restrict__ src, long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[8 * TILE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int l31 = lane & 31, hi = lane >> 5;
    // 4 K tiles + 4 V tiles of random data, the same swizzled image load_tile builds; rows of 64 elements at stride 64
    if (tid < 256) {
        const int wv = tid >> 6;
        load_tile(src, 64, 32 * wv, 1 << 20, smem + wv * TILE_B, lane);
        load_tile(src + 128 * 64, 64, 32 * wv, 1 << 20, smem + (4 + wv) * TILE_B, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SeqDesc sd; sd.n_outer = 1; sd.n_inner = 1; sd.outer_stride = 0; sd.inner_stride = 0; sd.offset = 0; sd.pos_stride = 1; sd.L = 1 << 20; sd.diag = 1 << 28; sd.heads = 1; sd.D = 64;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = frag_row_global(src + 256 * 64, 64, l31 + 32 * ((tid >> 6) & 3), ks, hi);
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -1e30f, l = 0.f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
        for (int j = 0; j < 4; ++j) fwd_tile(sd, smem + j * TILE_B, smem + (4 + j) * TILE_B, qf, j, 0, l31, l31, hi, lane, m, l, o0, o1);
    const long long t1 = __builtin_readcyclecounter();
    float s = m + l;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += o0[r] + o1[r];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) cyc[tid >> 6] = t1 - t0;
}

// ---- C2: the two-query-tile step of the resident forward kernel (16 MFMAs = 512 matrix-pipe cycles per double step)
__global__ __launch_bounds__(512) void step2_kernel(int iters, const bf16_t* __restrict__ src, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int l31 = lane & 31, hi = lane >> 5;
    if (tid < 256) {
        const int wv = tid >> 6;
        load_tile(src, 64, 32 * wv, 1 << 20, smem + wv * TILE_B, lane);
        load_tile(src + 128 * 64, 64, 32 * wv, 1 << 20, smem + RES_V0 + wv * TILE_B, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 qa[4], qb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qa[ks] = frag_row_global(src + 256 * 64, 64, l31 + 32 * ((tid >> 6) & 3), ks, hi); qb[ks] = frag_row_global(src + 384 * 64, 64, l31 + 32 * ((tid >> 6) & 3), ks, hi); }
    f32x16 oa0, oa1, ob0, ob1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oa0[r] = 0.f; oa1[r] = 0.f; ob0[r] = 0.f; ob1[r] = 0.f; }
    float ma = -1e30f, la = 0.f, mb = -1e30f, lb = 0.f;
    uint32_t kad[4], vad[4];
    res_offsets(kad, vad, lane);
    const uint32_t s0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) { kad[i] += s0; vad[i] += s0 + RES_V0; }
    u32x4_t k0[4], k1[4];
    const int L = 1 << 20;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        res_read_k<0>(k0, kad);
        RES_STEP2(0, k0, k1, true);
        RES_STEP2(1, k1, k0, true);
        RES_STEP2(2, k0, k1, true);
        RES_STEP2(3, k1, k0, false);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = ma + la + mb + lb;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += oa0[r] + oa1[r] + ob0[r] + ob1[r];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) cyc[tid >> 6] = t1 - t0;
}

// ---- C3: the single-query-tile step of the persistent kernel (p10_step), 8 MFMAs = 256 matrix-pipe cycles per step
template <bool PRE>
__global__ __launch_bounds__(768) void step1_kernel(int iters, const bf16_t* __restrict__ src, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int l31 = lane & 31, hi = lane >> 5;
    if (tid < 256) {
        const int wv = tid >> 6;
        load_tile(src, 64, 32 * wv, 1 << 20, smem + wv * TILE_B, lane);
        load_tile(src + 128 * 64, 64, 32 * wv, 1 << 20, smem + RES_V0 + wv * TILE_B, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 q[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) q[ks] = frag_row_global(src + 256 * 64, 64, l31 + 32 * ((tid >> 6) & 3), ks, hi);
    f32x16 o0, o1, negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; negm[r] = 0.f; }
    float m = PRE ? 0.f : -1e30f, l = 0.f;
    uint32_t kad[4], vad[4];
    res_offsets(kad, vad, lane);
    const uint32_t s0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) { kad[i] += s0; vad[i] += s0 + RES_V0; }
    u32x4_t kf[4];
    const int L = 1 << 20, nt = 99; const bf16_t* qnext = nullptr;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        res_read_k<0>(kf, kad);
        P10_STEP(0, false, true);
        P10_STEP(1, false, true);
        P10_STEP(2, false, true);
        P10_STEP(3, false, false);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = m + l;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += o0[r] + o1[r];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) cyc[tid >> 6] = t1 - t0;
}

static long long* d_cyc; static float* d_sink;
// longest wave of the block (all SIMDs carry the same number of waves), divided by the instructions one SIMD issued
static double run_max(int waves) {
    std::vector<long long> h(16);
    CK(hipMemcpy(h.data(), d_cyc, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = std::max(mx, h[w]);
    return (double)mx;
}

template <int OP>
static void part_a(FILE* f) {
    const int iters = 2000;
    double thr[3], dep = 0;
    const int wps[3] = {1, 2, 4};
    for (int k = 0; k < 3; ++k) {
        const int waves = 4 * wps[k];
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((valu_kernel<OP, 0>), dim3(1), dim3(64 * waves), 0, 0, iters, 1.25f, d_cyc, d_sink); CK(hipDeviceSynchronize()); }
        thr[k] = run_max(waves) / ((double)iters * 64 * wps[k]);
    }
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((valu_kernel<OP, 1>), dim3(1), dim3(256), 0, 0, iters, 1.25f, d_cyc, d_sink); CK(hipDeviceSynchronize()); }
    dep = run_max(4) / ((double)iters * 64);
    fprintf(f, "A  %-20s cycles/instr/SIMD  1 wave %.2f   2 waves %.2f   4 waves %.2f   | dependent chain (1 wave) %.2f\n", op_name[OP], thr[0], thr[1], thr[2], dep);
}

template <int OP, int NF>
static void part_b_one(double* out) {
    const int iters = 2000;
    for (int k = 0; k < 2; ++k) {
        const int wps = k + 1, waves = 4 * wps;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((mfma_fill_kernel<OP, NF>), dim3(1), dim3(64 * waves), 0, 0, iters, 1.25f, d_cyc, d_sink); CK(hipDeviceSynchronize()); }
        out[k] = run_max(waves) / ((double)iters * 4 * wps);     // cycles per MFMA per SIMD
    }
}
template <int OP>
static void part_b(FILE* f) {
    double r[8][2];
    part_b_one<OP, 0>(r[0]); part_b_one<OP, 1>(r[1]); part_b_one<OP, 2>(r[2]); part_b_one<OP, 4>(r[3]);
    part_b_one<OP, 6>(r[4]); part_b_one<OP, 8>(r[5]); part_b_one<OP, 12>(r[6]); part_b_one<OP, 16>(r[7]);
    const int nf[8] = {0, 1, 2, 4, 6, 8, 12, 16};
    fprintf(f, "B  %-20s cycles per MFMA per SIMD, n fillers per gap (1 wave | 2 waves per SIMD):", op_name[OP]);
    for (int i = 0; i < 8; ++i) fprintf(f, "  n=%d %.1f|%.1f", nf[i], r[i][0], r[i][1]);
    fprintf(f, "\n");
}

int main(int argc, char** argv) {
    FILE* f = stdout;
    CK(hipMalloc(&d_cyc, 16 * sizeof(long long))); CK(hipMalloc(&d_sink, 64));
    fprintf(f, "# tools/ubench_valu.hip on gfx950: one workgroup on one CU, s_memtime cycles\n");
    part_a<FMA>(f); part_a<ADD>(f); part_a<MUL>(f); part_a<MAX3>(f); part_a<EXP>(f); part_a<CVTPK>(f); part_a<PKMUL>(f); part_a<SWAP>(f);
    part_b<FMA>(f); part_b<EXP>(f); part_b<CVTPK>(f); part_b<MAX3>(f); part_b<PKMUL>(f);
    for (int il = 0; il < 2; ++il) {
        const int iters = 1000;
        fprintf(f, "B2 softmax-step instruction mix (8 MFMA + 56 VALU incl. 16 v_exp_f32), %s:", il ? "interleaved 1 MFMA : 7 VALU" : "clustered (8 MFMAs, then 56 VALU)");
        for (int wps = 1; wps <= 3; ++wps) {
            const int waves = 4 * wps;
            for (int rep = 0; rep < 2; ++rep) { if (il) hipLaunchKernelGGL(mix_kernel<1>, dim3(1), dim3(64 * waves), 0, 0, iters, 1.25f, d_cyc, d_sink); else hipLaunchKernelGGL(mix_kernel<0>, dim3(1), dim3(64 * waves), 0, 0, iters, 1.25f, d_cyc, d_sink); CK(hipDeviceSynchronize()); }
            fprintf(f, "  %d wave(s)/SIMD %.0f cycles per step per SIMD", wps, run_max(waves) / ((double)iters * wps));
        }
        fprintf(f, "\n");
    }
    // C
    bf16_t* d_src;
    {
        const size_t n = 512 * 64;
        std::vector<uint16_t> h(n);
        srand(1);
        for (size_t i = 0; i < n; ++i) { const float v = (rand() / (float)RAND_MAX - 0.5f) * 4.0f; uint32_t u; memcpy(&u, &v, 4);
#ifdef TCOW_FP16
            _Float16 hv = (_Float16)v; memcpy(&h[i], &hv, 2);
#else
            h[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
#endif
        }
        CK(hipMalloc(&d_src, n * 2)); CK(hipMemcpy(d_src, h.data(), n * 2, hipMemcpyHostToDevice));
        const int iters = 500;
        fprintf(f, "C  fwd_tile (shipped forward step, 8 MFMAs = 256 matrix-pipe cycles), LDS-resident K/V, no memory waits:\n");
        for (int wps = 1; wps <= 4; ++wps) {
            const int waves = 4 * wps;
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(step_kernel, dim3(1), dim3(64 * waves), 0, 0, iters, d_src, d_cyc, d_sink); CK(hipDeviceSynchronize()); }
            const double per_wave = run_max(waves) / (iters * 4.0);
            fprintf(f, "C  %d wave(s)/SIMD: %.0f cycles per step per wave, %.0f per step per SIMD -> MFMA pipe busy %.1f %%\n", wps, per_wave, per_wave / wps, 100.0 * 256.0 * wps / per_wave);
        }
    }
    {
        const int iters = 500;
        CK(hipFuncSetAttribute((const void*)step2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS));
        fprintf(f, "C2 res_step2 (two query tiles per wave, 16 MFMAs = 512 matrix-pipe cycles per double step), LDS-resident K/V:\n");
        for (int wps = 1; wps <= 2; ++wps) {
            const int waves = 4 * wps;
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(step2_kernel, dim3(1), dim3(64 * waves), RES_LDS, 0, iters, d_src, d_cyc, d_sink); CK(hipDeviceSynchronize()); }
            const double per_wave = run_max(waves) / (iters * 4.0);
            fprintf(f, "C2 %d wave(s)/SIMD: %.0f cycles per double step per wave, %.0f per SIMD -> MFMA pipe busy %.1f %%\n", wps, per_wave, per_wave / wps, 100.0 * 512.0 * wps / per_wave);
        }
    }
    {
        const int iters = 500;
        for (int pre = 0; pre < 2; ++pre) {
            auto kern = pre ? step1_kernel<true> : step1_kernel<false>;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS));
            fprintf(f, "C3 p10_step<PRE=%d> (one query tile per wave, hand-placed LDS reads, 8 MFMAs = 256 matrix-pipe cycles per step):\n", pre);
            for (int wps = 1; wps <= 3; ++wps) {
                const int waves = 4 * wps;
                for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), RES_LDS, 0, iters, d_src, d_cyc, d_sink); CK(hipDeviceSynchronize()); }
                const double per_wave = run_max(waves) / (iters * 4.0);
                fprintf(f, "C3 %d wave(s)/SIMD: %.0f cycles per step per wave, %.0f per SIMD -> MFMA pipe busy %.1f %%\n", wps, per_wave, per_wave / wps, 100.0 * 256.0 * wps / per_wave);
            }
        }
    }
    // D: phase timeline of the resident forward kernel at the benchmark shape (B*Qs = 3, T = 30, S = 301, 12 heads)
    {
        const int B = 3, T = 30, S = 301, heads = 12, D = 768; const long M = (long)B * T * S;
        bf16_t *qkv, *o; float* lse; long long* dbg;
        CK(hipMalloc(&qkv, M * 3 * D * 2)); CK(hipMalloc(&o, M * D * 2)); CK(hipMalloc(&lse, M * heads * 4));
        {
            std::vector<uint16_t> h((size_t)M * 3 * D);
            for (size_t i = 0; i < h.size(); ++i) { const float v = (rand() / (float)RAND_MAX - 0.5f) * 4.0f; uint32_t u; memcpy(&u, &v, 4); h[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
            CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        const int pairs = B * T * heads;
        CK(hipMalloc(&dbg, (size_t)pairs * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)pairs * 16 * 8));
        SeqDesc sd; sd.n_outer = B * T; sd.n_inner = 1; sd.outer_stride = S; sd.inner_stride = 0; sd.offset = 0; sd.pos_stride = 1; sd.L = S; sd.diag = 1 << 28; sd.heads = heads; sd.D = D;
        CK(hipFuncSetAttribute((const void*)attn_fwd_res, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(attn_fwd_res, dim3(pairs), dim3(256), RES_LDS, 0, sd, 10, qkv, o, lse, (long long*)nullptr);
        CK(hipEventRecord(e0)); for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(attn_fwd_res, dim3(pairs), dim3(256), RES_LDS, 0, sd, 10, qkv, o, lse, (long long*)nullptr);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        fprintf(f, "D  attn_fwd_res at B*Qs=3 T=30 S=301 h=12: %.1f us per launch (20 back-to-back launches)\n", ms * 1000 / 20);
        {
            CK(hipFuncSetAttribute((const void*)attn_fwd_p10<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RES_LDS));
            for (int grid = 256; grid >= 216; grid -= 40) {
                for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(attn_fwd_p10<false>, dim3(grid), dim3(640), 2 * RES_LDS, 0, sd, 10, pairs, qkv, o, lse, (long long*)nullptr);
                CK(hipEventRecord(e0)); for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(attn_fwd_p10<false>, dim3(grid), dim3(640), 2 * RES_LDS, 0, sd, 10, pairs, qkv, o, lse, (long long*)nullptr);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
                fprintf(f, "D  attn_fwd_p10<false> grid %d: %.1f us per launch\n", grid, ms * 1000 / 20);
            }
            {   // the Q third of every row times log2(e) / 8, as the caller of the PRE variant delivers it
                std::vector<uint16_t> h((size_t)M * 3 * D);
                CK(hipMemcpy(h.data(), qkv, h.size() * 2, hipMemcpyDeviceToHost));
                for (long r = 0; r < M; ++r) for (int c = 0; c < D; ++c) { uint32_t u = (uint32_t)h[(size_t)r * 3 * D + c] << 16; float v; memcpy(&v, &u, 4); v *= 0.18033688f; memcpy(&u, &v, 4); h[(size_t)r * 3 * D + c] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
                CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
            }
            CK(hipFuncSetAttribute((const void*)attn_fwd_p10<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RES_LDS));
            for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(attn_fwd_p10<true>, dim3(256), dim3(640), 2 * RES_LDS, 0, sd, 10, pairs, qkv, o, lse, (long long*)nullptr);
            CK(hipEventRecord(e0)); for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(attn_fwd_p10<true>, dim3(256), dim3(640), 2 * RES_LDS, 0, sd, 10, pairs, qkv, o, lse, (long long*)nullptr);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
            fprintf(f, "D  attn_fwd_p10<true> (Q pre-scaled, timing only) grid 256: %.1f us per launch\n", ms * 1000 / 20);
        }
        {
            long long* dbg2; CK(hipMalloc(&dbg2, 256 * 10 * 8 * 8)); CK(hipMemset(dbg2, 0, 256 * 10 * 8 * 8));
            hipLaunchKernelGGL(attn_fwd_p10<false>, dim3(256), dim3(640), 2 * RES_LDS, 0, sd, 10, pairs, qkv, o, lse, dbg2); CK(hipDeviceSynchronize());
            std::vector<long long> h2(256 * 10 * 8); CK(hipMemcpy(h2.data(), dbg2, h2.size() * 8, hipMemcpyDeviceToHost));
            const char* nm2[6] = {"steps 0-3", "steps 4-7", "steps 8-9", "wait own loads", "barrier", "store"};
            fprintf(f, "D  attn_fwd_p10 second pair of every workgroup, mean shader cycles per phase, per wave index:\n");
            for (int i = 0; i < 6; ++i) {
                fprintf(f, "D    %-16s", nm2[i]);
                for (int w = 0; w < 10; ++w) { double sm = 0; for (int b = 0; b < 256; ++b) { const long long* t = &h2[(size_t)(b * 10 + w) * 8]; sm += (double)(t[i + 1] - t[i]); } fprintf(f, " %7.0f", sm / 256); }
                fprintf(f, "\n");
            }
        }
        // E: phase timeline of the shipped streaming backward kernels (wave 0..3 of every workgroup that owns four live tiles)
        {
            bf16_t *dout, *dqkv; float2* ldt; long long* dbg3;
            CK(hipMalloc(&dout, M * D * 2)); CK(hipMalloc(&dqkv, M * 3 * D * 2)); CK(hipMalloc(&ldt, (size_t)pairs * 320 * 8));
            CK(hipMemcpy(dout, qkv, M * D * 2, hipMemcpyDeviceToDevice));
            const int nchunk = 3, grid = stream_grid(pairs, nchunk);
            CK(hipMalloc(&dbg3, (size_t)grid * 4 * 24 * 8));
            hipLaunchKernelGGL(attn_fwd_stream<5>, dim3(grid), dim3(256), 0, 0, sd, 10, qkv, o, lse); CK(hipDeviceSynchronize());
            for (int which = 0; which < 4; ++which) {
                auto launch = [&](long long* d) {
                    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), &d, sizeof(d)));
                    if (which == 0) hipLaunchKernelGGL(attn_bwd_dq_stream<4>, dim3(grid), dim3(256), 0, 0, sd, 10, qkv, o, dout, lse, ldt, dqkv);
                    else if (which == 1) hipLaunchKernelGGL(attn_bwd_dkv_stream<4>, dim3(grid), dim3(256), 0, 0, sd, 10, qkv, dout, ldt, dqkv);
                    else if (which == 2) hipLaunchKernelGGL(attn_bwd_dq_stream<5>, dim3(grid), dim3(256), 0, 0, sd, 10, qkv, o, dout, lse, ldt, dqkv);
                    else hipLaunchKernelGGL(attn_bwd_dkv_stream<5>, dim3(grid), dim3(256), 0, 0, sd, 10, qkv, dout, ldt, dqkv);
                };
                for (int rep = 0; rep < 3; ++rep) launch(nullptr);
                CK(hipEventRecord(e0)); for (int rep = 0; rep < 20; ++rep) launch(nullptr);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
                fprintf(f, "E  %s at B*Qs=3 T=30 S=301 h=12: %.1f us per launch\n", (which & 1) ? (which > 1 ? "attn_bwd_dkv_stream<5>" : "attn_bwd_dkv_stream<4>") : (which > 1 ? "attn_bwd_dq_stream<5>" : "attn_bwd_dq_stream<4>"), ms * 1000 / 20);
                CK(hipMemset(dbg3, 0, (size_t)grid * 4 * 24 * 8));
                launch(dbg3); CK(hipDeviceSynchronize());
                std::vector<long long> h3((size_t)grid * 4 * 24); CK(hipMemcpy(h3.data(), dbg3, h3.size() * 8, hipMemcpyDeviceToHost));
                const char* ph[5] = {"barrier in", "issue loads", "loads land", "barrier", "tile steps"};
                const int nch = which > 1 ? 2 : 3;
                for (int w = 0; w < 4; w += 3) {
                    double acc[3][5] = {{0}}; double store = 0, total = 0; int n = 0;
                    for (int b = 0; b < grid; ++b) {
                        const int k = b >> 3, ch = k % nchunk; if (ch == 2) continue;            // (the workgroup of tiles 8, 9 has two idle waves)
                        const long long* t = &h3[(size_t)(b * 4 + w) * 24]; if (!t[21]) continue;
                        ++n; store += (double)(t[21] - t[5 * nch]); total += (double)(t[21] - t[0]);
                        for (int c = 0; c < nch; ++c) for (int i = 0; i < 5; ++i) { const int a = 1 + 5 * c + i; acc[c][i] += (double)(t[a] - t[a - 1]); }
                    }
                    fprintf(f, "E    wave %d (%d workgroups): total %.0f cycles; issue of the first chunk's loads %.0f, fragment loads (+ delta) %.0f; store %.0f\n", w, n, total / n, acc[0][0] / n, acc[0][1] / n, store / n);
                    for (int c = 0; c < nch; ++c) {
                        fprintf(f, "E      chunk %d:", c);
                        for (int i = (c ? 0 : 2); i < 5; ++i) fprintf(f, "  %s %.0f", ph[i], acc[c][i] / n);
                        fprintf(f, "\n");
                    }
                }
            }
        }
        // F: phase timeline of the one-kernel spatial backward (round 4): per wave index, mean shader cycles
        {
            bf16_t *dout, *dqkv; long long* dbg4;
            CK(hipMalloc(&dout, M * D * 2)); CK(hipMalloc(&dqkv, M * 3 * D * 2));
            CK(hipMemcpy(dout, qkv, M * D * 2, hipMemcpyDeviceToDevice));
            CK(hipMalloc(&dbg4, (size_t)pairs * 12 * 40 * 8));
            CK(hipFuncSetAttribute((const void*)attn_bwd_one_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ONE_LDS));
            auto launch = [&](long long* d) {
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), &d, sizeof(d)));
                hipLaunchKernelGGL(attn_bwd_one_kernel, dim3(pairs), dim3(768), ONE_LDS, 0, sd, 10, qkv, o, dout, lse, dqkv);
            };
            for (int rep = 0; rep < 3; ++rep) launch(nullptr);
            CK(hipEventRecord(e0)); for (int rep = 0; rep < 20; ++rep) launch(nullptr);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
            fprintf(f, "F  attn_bwd_one_kernel at B*Qs=3 T=30 S=301 h=12: %.1f us per launch\n", ms * 1000 / 20);
            CK(hipMemset(dbg4, 0, (size_t)pairs * 12 * 40 * 8));
            launch(dbg4); CK(hipDeviceSynchronize());
            std::vector<long long> h4((size_t)pairs * 12 * 40); CK(hipMemcpy(h4.data(), dbg4, h4.size() * 8, hipMemcpyDeviceToHost));
            fprintf(f, "F  mean shader cycles per phase over %d workgroups, per wave index 0..11 (10, 11 = the dQ chain waves):\n", pairs);
            auto row = [&](const char* name, auto get) {
                fprintf(f, "F    %-44s", name);
                for (int w = 0; w < 12; ++w) { double sm = 0; for (int b = 0; b < pairs; ++b) sm += get(&h4[((size_t)b * 12 + w) * 40]); fprintf(f, " %7.0f", sm / pairs); }
                fprintf(f, "\n");
            };
            row("prologue (loads, table, first barrier)", [](const long long* t) { return (double)(t[1] - t[0]); });
            row("step: tile arithmetic + strip write (mean of 10)", [](const long long* t) { double s = 0; for (int i = 0; i < 10; ++i) s += (double)(t[2 + 3 * i] - (i ? t[4 + 3 * (i - 1)] : t[1])); return s / 10; });
            row("step: wait for own loads + barrier (mean of 10)", [](const long long* t) { double s = 0; for (int i = 0; i < 10; ++i) s += (double)(t[3 + 3 * i] - t[2 + 3 * i]); return s / 10; });
            row("step: requests + dQ chains / dQ stores (mean of 10)", [](const long long* t) { double s = 0; for (int i = 0; i < 10; ++i) s += (double)(t[4 + 3 * i] - t[3 + 3 * i]); return s / 10; });
            row("final barrier", [](const long long* t) { return (double)(t[32] - t[31]); });
            row("last dQ tile + dK / dV stores", [](const long long* t) { return (double)(t[33] - t[32]); });
            row("workgroup lifetime", [](const long long* t) { return (double)(t[33] - t[0]); });
        }
        hipLaunchKernelGGL(attn_fwd_res, dim3(pairs), dim3(256), RES_LDS, 0, sd, 10, qkv, o, lse, dbg); CK(hipDeviceSynchronize());
        std::vector<long long> h((size_t)pairs * 16); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        const char* nm[12] = {"issue loads", "wait group 0 (Q + tiles 0-3)", "steps 0-3", "wait group 1", "steps 4-7", "issue qx", "wait group 2", "steps 8-9", "store 2 tiles", "5 single steps", "barrier", "merge + store"};
        double sum[12] = {0}; long long w0 = 1LL << 62, w1 = 0; double life = 0;
        for (int b = 0; b < pairs; ++b) {
            const long long* t = &h[(size_t)b * 16];
            for (int i = 0; i < 11; ++i) sum[i] += (double)(t[i + 2] - t[i + 1]);
            w0 = std::min(w0, t[0]); w1 = std::max(w1, t[13]); life += (double)(t[13] - t[0]);
        }
        fprintf(f, "D  timeline of wave 0, mean shader cycles per phase over %d workgroups:\n", pairs);
        for (int i = 0; i < 11; ++i) fprintf(f, "D    %-32s %8.0f\n", nm[i], sum[i] / pairs);
        fprintf(f, "D  kernel span %.1f us (100 MHz wall clock), mean workgroup lifetime %.1f us\n", (w1 - w0) / 100.0, life / pairs / 100.0);
        // how many workgroups are alive over time (0.5 us buckets)
        const int nbk = (int)((w1 - w0) / 50) + 1; std::vector<int> alive(nbk, 0);
        for (int b = 0; b < pairs; ++b) { const long long* t = &h[(size_t)b * 16]; for (long long x = (t[0] - w0) / 50; x <= (t[13] - w0) / 50 && x < nbk; ++x) alive[x]++; }
        fprintf(f, "D  resident workgroups per 0.5 us bucket:"); for (int i = 0; i < nbk; ++i) fprintf(f, " %d", alive[i]); fprintf(f, "\n");
    }
    return 0;
}
