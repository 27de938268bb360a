// NT GEMM main loop on the phase structure of the MI355X guide's "256^2 8-phase template" (cdna_hip_programming.md), written for this path's
// shapes: C[M,N] = A[M,K] . W[N,K]^T (+ bias), 16-bit operands, f32 accumulation, (BM x 256) output tile with BM = 320 or 256, K 64 at a time.
//
//  * 8 waves as 2 wave rows x 4 wave columns, (BM/2) x 64 outputs per wave, v_mfma_f32_16x16x32 (issued as (W fragment, A fragment): a lane
//    owns output row m = lane & 15 of a 16-row block and four consecutive columns per accumulator).
//  * LDS image of a K tile = four PLANES (A k 0..31, W k 0..31, A k 32..63, W k 32..63); a plane is a stack of 1 KiB subtiles (16 rows x 32 k)
//    stored row-major (64 B per row) with chunk c of row r at position c ^ ((r >> 2) & 3): one global_load_lds_dwordx4 fills one subtile with
//    four consecutive lanes on one contiguous 64-byte piece of a row, and a ds_read_b128 fragment read (lane -> row lane & 15, k chunk
//    lane >> 4) is conflict-free; every fragment address is `per-lane constant + immediate`.
//  * A K tile is four phases (row half of the wave's tile x k half): 4-5 A fragments (+ 4 W fragments in the first phase of a k half),
//    2-3 direct-to-LDS loads of ONE plane of the NEXT K tile, barrier, 16-20 MFMAs under s_setprio(1), barrier.  Counted vmcnt waits only in
//    phases 2 and 4 (the planes a phase pair reads were requested >= 3 phases earlier).  Wave row 1 runs one barrier behind wave row 0, so
//    on every SIMD (which hosts wave w and w + 4) one wave is in its MFMA block while the other is in its read / load part.
//  * Plain epilogue only (bias, 16-bit or f32 output) in this version; the fused-epilogue GEMMs stay on gemm_bf16.hip.
//  * STATUS (round 3): the same loop with BM = 320 is now the DEFAULT main loop of gemm_nt_bf16_320_kernel (gemm_bf16.hip, template argument
//    ML = 1) in front of that kernel's LDS-staged fused epilogues -- every path shape 3-13 % faster than the two-stage loop it replaced.  This
//    file keeps the stand-alone plain-epilogue form (BM = 320 or 256, direct 8-byte stores: slower than the shipped kernel at K = 768) for
//    measurements on square problems (profiles/r03_gemm_cube.txt: 1106-1188 TFLOP/s at 4096^3) and as the reference for the layout and the
//    ordering argument.  TCOW_GEMM_P8=1 (or tile = 8320 / 8256 in tcow_gemm_args) routes plain-epilogue problems here.
#include <stdint.h>
#include <stdlib.h>

#include "../tcow_amd/csrc/common.h"

namespace {

#ifdef TCOW_FP16
#define TCOW_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
#define TCOW_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

struct P8Params {
    int M, N, K;
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    void* C; long ldc; int out_f32;
    const float* bias;
    int tiles_m, tiles_n;
#ifdef TCOW_P8_DBG
    long long* dbg; int dbg_kt;      // tools/ubench_gemm.hip: shader-clock stamps of one K tile, 32 per wave
#endif
};

__device__ __forceinline__ int p8_xcd_remap(int bid, int nblk) {
    // workgroups are dispatched round-robin over the 8 XCDs: give each XCD a contiguous range of tiles (bijective for any nblk; the grid is
    // padded to a multiple of 8 and the surplus workgroups get -1)
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    if (idx >= (xcd < r ? q + 1 : q)) return -1;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void p8_glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((GLB_PTR(const uint32_t))gsrc, (LDS_PTR(uint32_t))lds_wave_base, 16, 0, 0);
}

typedef uint32_t p8_u32x4 __attribute__((ext_vector_type(4)));
#ifdef TCOW_P8_DBG
#define P8_STAMP(i) do { if (kt == p.dbg_kt && lane == 0) p.dbg[(blockIdx.x * 8 + wave) * 32 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define P8_STAMP(i) do { } while (0)
#endif

template <int BM>
__global__ __launch_bounds__(512, 2) void gemm_nt_p8_kernel(P8Params p) {
    constexpr int ARB = BM / 16;                  // 16-row blocks of the A tile
    constexpr int RBH = ARB / 4;                  // row blocks per wave and phase (4 at BM = 256, 5 at BM = 320)
    constexpr int A_PLANE = ARB * 1024, B_PLANE = 16 * 1024;
    constexpr int KH = A_PLANE + B_PLANE;         // one k half: A plane, W plane
    constexpr int KTILE = 2 * KH;
    constexpr int NA = (ARB + 7) / 8;             // direct-to-LDS loads per wave and A plane (the last one only for waves < ARB - 8 * (NA - 1))
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nblk = p.tiles_m * p.tiles_n;
    const int pid = p8_xcd_remap(blockIdx.x, nblk);
    if (pid < 0) return;
    const int pm = pid / p.tiles_n, pn = pid - pm * p.tiles_n;
    const int m0 = pm * BM, n0 = pn * 256;

    // source offsets (elements, relative to the tile's first row) of this lane for the subtiles this wave loads.  A subtile is row-major in LDS
    // (16 rows x 64 B; a load instruction writes lane * 16, i.e. row lane >> 2, position lane & 3) with the 16-byte chunk c of row r at position
    // c ^ ((r >> 2) & 3) -- applied on the SOURCE address; four consecutive lanes still fetch one contiguous 64-byte piece of a row
    const bf16_t* a_base = p.A + (size_t)m0 * p.lda;
    const bf16_t* w_base = p.W + (size_t)n0 * p.ldw;
    uint32_t a_src[NA], w_src[2];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int r = (wave + 8 * j) * 16 + (lane >> 2);
        const int rr = m0 + r < p.M ? r : p.M - 1 - m0;
        a_src[j] = (uint32_t)(rr * p.lda + ((lane & 3) ^ ((lane >> 4) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 8 * j) * 16 + (lane >> 2);
        const int rr = n0 + r < p.N ? r : p.N - 1 - n0;
        w_src[j] = (uint32_t)(rr * p.ldw + ((lane & 3) ^ ((lane >> 4) & 3)) * 8);
    }
    const bool a_last = wave + 8 * (NA - 1) < ARB;        // does this wave own a subtile in the last round of an A plane
    const int n_a = (NA - 1) + (a_last ? 1 : 0);           // loads per A plane of this wave; W planes: always 2
    auto load_a = [&](int kt, int kh) {                   // A plane kh of K tile kt into buffer kt & 1
        char* dst = smem + (kt & 1) * KTILE + kh * KH;
        const bf16_t* g = a_base + (size_t)kt * 64 + kh * 32;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j < NA - 1 || a_last) p8_glds16(g + a_src[j], dst + (wave + 8 * j) * 1024);
    };
    auto load_w = [&](int kt, int kh) {
        char* dst = smem + (kt & 1) * KTILE + kh * KH + A_PLANE;
        const bf16_t* g = w_base + (size_t)kt * 64 + kh * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) p8_glds16(g + w_src[j], dst + (wave + 8 * j) * 1024);
    };

    f32x4 acc[2 * RBH][4];
#pragma unroll
    for (int i = 0; i < 2 * RBH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
    const uint32_t frag_off = (uint32_t)((lane & 15) * 64 + (((lane >> 4) ^ (((lane & 15) >> 2) & 3)) << 4));    // row lane & 15, k chunk lane >> 4: conflict-free b128
    const uint32_t a_ad = lds0 + (wr * (ARB / 2)) * 1024 + frag_off;            // + (ri * RBH + i) * 1024 + kh * KH + buffer
    const uint32_t w_ad = lds0 + A_PLANE + (wc * 4) * 1024 + frag_off;          // + j * 1024 + kh * KH + buffer
    p8_u32x4 fa[2][RBH], fw[2][4];
#define P8_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
    auto read_a = [&](p8_u32x4 (&f)[RBH], uint32_t base, int ri) {
        if (ri == 0) {
            P8_DSR(f[0], base, 0); P8_DSR(f[1], base, 1024); P8_DSR(f[2], base, 2048); P8_DSR(f[3], base, 3072);
            if constexpr (RBH == 5) P8_DSR(f[4], base, 4096);
        } else {
            P8_DSR(f[0], base, RBH * 1024); P8_DSR(f[1], base, RBH * 1024 + 1024); P8_DSR(f[2], base, RBH * 1024 + 2048); P8_DSR(f[3], base, RBH * 1024 + 3072);
            if constexpr (RBH == 5) P8_DSR(f[4], base, RBH * 1024 + 4096);
        }
    };
    auto read_w = [&](p8_u32x4 (&f)[4], uint32_t base) {
        P8_DSR(f[0], base, 0); P8_DSR(f[1], base, 1024); P8_DSR(f[2], base, 2048); P8_DSR(f[3], base, 3072);
    };
    auto mfma_block = [&](const p8_u32x4 (&fA)[RBH], const p8_u32x4 (&fW)[4], int ri) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < RBH; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[ri * RBH + i][j] = TCOW_MFMA_16x16x32_H16(__builtin_bit_cast(bf16x8, fW[j]), __builtin_bit_cast(bf16x8, fA[i]), acc[ri * RBH + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // wait until at most `n` of this wave's direct-to-LDS loads are still in flight (n in {0, 4, 5})
    auto wait_vm = [&](bool all) {
        if (all) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n_a == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n_a == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };
#define P8_PHASE_TAIL(set_a, set_w, ri, st)                                                               \
    do {                                                                                                  \
        P8_STAMP(st);                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                     \
        P8_STAMP(st + 1);                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        P8_STAMP(st + 2);                                                                                 \
        mfma_block(fa[set_a], fw[set_w], ri);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        P8_STAMP(st + 3);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                     \
        P8_STAMP(st + 4);                                                                                 \
    } while (0)

    const int nk = p.K / 64;
    load_a(0, 0); load_w(0, 0); load_a(0, 1); load_w(0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();            // wave row 1 runs one barrier behind wave row 0
    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t bo = (uint32_t)(kt & 1) * KTILE;
        const bool more = kt + 1 < nk;
        P8_STAMP(0);
        // phase 1: row half 0, k half 0
        read_w(fw[0], w_ad + bo); read_a(fa[0], a_ad + bo, 0);
        if (more) load_a(kt + 1, 0);
        P8_PHASE_TAIL(0, 0, 0, 1);
        // phase 2: row half 1, k half 0; afterwards the k-half-1 planes of this K tile must be complete
        read_a(fa[1], a_ad + bo, 1);
        if (more) load_w(kt + 1, 0);
        P8_STAMP(6);
        wait_vm(!more);
        P8_PHASE_TAIL(1, 0, 1, 7);
        // phase 3: row half 0, k half 1
        read_w(fw[1], w_ad + bo + KH); read_a(fa[0], a_ad + bo + KH, 0);
        if (more) load_a(kt + 1, 1);
        P8_PHASE_TAIL(0, 1, 0, 12);
        // phase 4: row half 1, k half 1; afterwards the k-half-0 planes of the next K tile must be complete
        read_a(fa[1], a_ad + bo + KH, 1);
        if (more) load_w(kt + 1, 1);
        P8_STAMP(17);
        wait_vm(!more);
        P8_PHASE_TAIL(1, 1, 1, 18);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef P8_DSR
#undef P8_PHASE_TAIL

    // ---- epilogue (plain): lane owns row lane & 15 of each 16-row block and columns 4 * (lane >> 4) .. + 3 of each 16-column block
    const int mrow = m0 + wr * (BM / 2) + (lane & 15);
    const int ncol = n0 + wc * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = ncol + j * 16;
        if (n >= p.N) continue;
        const float4 b4 = p.bias ? ld4(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 2 * RBH; ++i) {
            const int m = mrow + i * 16;
            if (m >= p.M) continue;
            const float4 v = make_float4(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w);
            if (p.out_f32) st4(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, v);
            else st4(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n, v);
        }
    }
}

}  // namespace

void tcow_ensure_lds(const void* kernel, int bytes);
void tcow_set_error(const char* fmt, ...);

// true when this problem is one the phase-structured kernel takes (plain epilogue, shapes aligned for the 16-byte paths)
bool tcow_gemm_nt_p8_ok(const tcow_gemm_args* a) {
    return a->dtype == TCOW_BF16 && a->act == TCOW_ACT_NONE && !a->row_scale && !a->resid && !a->bias2 && !a->aux && a->K % 64 == 0 && a->K >= 128 && a->N % 4 == 0 &&
           a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldc % 4 == 0 && a->M >= 256 && a->N >= 256;
}

int tcow_gemm_nt_p8(hipStream_t stream, const tcow_gemm_args* a, int bm) {
    P8Params p;
    p.M = a->M; p.N = a->N; p.K = a->K; p.A = (const bf16_t*)a->A; p.lda = a->lda; p.W = (const bf16_t*)a->W; p.ldw = a->ldw;
    p.C = a->C; p.ldc = a->ldc; p.out_f32 = a->out_f32; p.bias = a->bias;
    p.tiles_n = (a->N + 255) / 256;
    if (bm == 320) {
        p.tiles_m = (a->M + 319) / 320;
        const int lds = 2 * 2 * (20 + 16) * 1024;
        tcow_ensure_lds(reinterpret_cast<const void*>(gemm_nt_p8_kernel<320>), lds);
        hipLaunchKernelGGL(gemm_nt_p8_kernel<320>, dim3(8 * ((p.tiles_m * p.tiles_n + 7) / 8)), dim3(512), lds, stream, p);
    } else {
        p.tiles_m = (a->M + 255) / 256;
        const int lds = 2 * 2 * (16 + 16) * 1024;
        tcow_ensure_lds(reinterpret_cast<const void*>(gemm_nt_p8_kernel<256>), lds);
        hipLaunchKernelGGL(gemm_nt_p8_kernel<256>, dim3(8 * ((p.tiles_m * p.tiles_n + 7) / 8)), dim3(512), lds, stream, p);
    }
    if (hipGetLastError() != hipSuccess) { tcow_set_error("tcow_gemm_nt_p8: launch failed"); return TCOW_ERR_LAUNCH; }
    return TCOW_OK;
}
