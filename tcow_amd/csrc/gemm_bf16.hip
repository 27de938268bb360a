// bf16 MFMA GEMMs for the Seeker hot path (gfx950).
//
//   gemm_nt_bf16 : C[M,N] = epi(A[M,K] . W[N,K]^T)  -- every nn.Linear forward and input-gradient on the path
//                  (vit.py:50-61,74-76,111,146; mask_tracker.py:113) with bias / DropPath row-scale / GELU /
//                  GELU' / residual fused in the epilogue.
//   gemm_tn_bf16 : dW[N,K] += dY[M,N]^T . X[M,K]    -- weight gradients, token dimension split across
//                  workgroups, operands transposed on the fly with ds_read_b64_tr_b16.
//
// Structure of gemm_nt (per 256-thread workgroup = 4 waves as 2x2):
//   128x128 output tile, K walked in 64-element (128-byte) slices, two LDS stages filled with direct-to-LDS
//   loads (global_load_lds_dwordx4, no VGPR round trip).  Each 16-byte chunk c of tile row r is stored at chunk
//   position c ^ ((r>>1)&7) (applied on the per-lane SOURCE address, the LDS image of a wave-load stays
//   lane-linear), which makes the ds_read_b128 fragment reads of the 32x32x16 MFMA conflict-free.  Accumulators
//   are staged through LDS as f32 so that bias / residual loads and the C stores are full-row coalesced.
//   blockIdx is remapped so that each XCD (private L2) owns a contiguous band of row tiles.
#include <stdlib.h>

#include "common.h"
#include "gemm_nt_common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int BK = 64;                    // bf16 elements per k-slice (128 bytes per tile row)
constexpr int TILE_BYTES = BM * 128;      // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int NT_LDS_BYTES = 2 * STAGE_BYTES;  // 64 KiB (also holds the 128x128 f32 epilogue tile)

__device__ __forceinline__ void nt_epilogue(const NtParams& p, char* smem, f32x16 (&acc)[2][2], float4 b4, int tid, int wm, int wn, int l31, int hi, int m0, int n0) {
    const int c4 = (tid & 31) * 4, gn = n0 + c4;
    const bool col_ok = gn < p.N;                 // N % 4 == 0: a thread's 4 columns are all in or all out
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + crow32(r, hi);
                const int col = wn * 64 + j * 32 + l31;
                ct[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    if (!col_ok) return;
    epi_rows(p, ct, BN, b4, m0 + (tid >> 5), tid >> 5, 8, c4, gn);
}

// bias for this thread's 4 epilogue columns, fetched at kernel start (its latency hides behind the whole main loop)
__device__ __forceinline__ float4 epi_bias(const NtParams& p, int tid, int n0) {
    const int gn = n0 + (tid & 31) * 4;
    return (p.bias && gn < p.N) ? ld4(p.bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- 256 x 256 tile, 8 waves (2 x 4, 128 x 64 each).  PMC on the 128-tile kernel (profiles/r01_pmc_gemm_units.txt): zero LDS bank
// conflicts, LdsUtil ~22 %, MFMA pipe busy 41-48 %, half of all wave cycles parked in the vmcnt/barrier wait and the texture-address
// path 58-68 % busy -- the 128 x 128 x 64 step pulls 32 KiB per workgroup per 2.1 MFLOP through the global->LDS path and is bound
// by it.  The 256-square tile halves the bytes per FLOP (64 KiB per 8.4 MFLOP; MFMA busy 55 % at 8192^3) and needs 6 instead of 8
// fragment reads per 8 MFMAs.  128 KiB LDS (two stages), one workgroup per CU.
constexpr int B_BM = 256, B_BN = 256;
constexpr int B_TILE = 256 * 128;              // 32 KiB per operand per stage
constexpr int B_STAGE = 2 * B_TILE;
constexpr int B_LDS = 2 * B_STAGE;             // 128 KiB

__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_256_kernel(NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nblk = p.tiles_m * p.tiles_n;
    const int pid = xcd_remap(blockIdx.x, nblk);
    const int pm = pid / p.tiles_n, pn = pid - pm * p.tiles_n;
    const int m0 = pm * B_BM, n0 = pn * B_BN;

    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gm = m0 + r; gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + r; gn = gn < p.N ? gn : p.N - 1;
        a_src[j] = p.A + (size_t)gm * p.lda + c * 8;
        w_src[j] = p.W + (size_t)gn * p.ldw + c * 8;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * B_STAGE;
        char* sw = sa + B_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(a_src[j] + (size_t)kt * BK, sa + (wave * 4 + j) * 1024);
            glds16(w_src[j] + (size_t)kt * BK, sw + (wave * 4 + j) * 1024);
        }
    };
    // Fragment reads are software-pipelined by hand: left to hipcc, the loop keeps ONE fragment register set and waits lgkmcnt(0)
    // before every group of four MFMAs, exposing the LDS latency eight times per K-slice.  Here the six ds_read_b128 of k-step ks+1
    // are issued (inline asm, so the compiler neither merges nor reorders them) before the eight MFMAs of k-step ks, and a counted
    // s_waitcnt lgkmcnt(6) retires exactly the previous k-step's reads.  All four A (two W) fragments of a k-step share one address
    // register: rows 32 apart have the same swizzle, so they differ by an immediate offset.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
    uint32_t a_ad[4], w_ad[4];
    {
        const int ra = wm * 128 + l31, rw = wn * 64 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a_ad[ks] = lds0 + ra * 128 + (((2 * ks + hi) ^ ((ra >> 1) & 7)) << 4);
            w_ad[ks] = lds0 + B_TILE + rw * 128 + (((2 * ks + hi) ^ ((rw >> 1) & 7)) << 4);
        }
    }
    u32x4 fa[2][4], fw[2][2];
#define TCOW_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define TCOW_READ_FRAGS(buf, ks, so)                                            \
    do {                                                                        \
        const uint32_t aa = a_ad[ks] + (so), ww = w_ad[ks] + (so);              \
        TCOW_DSR(fw[buf][0], ww, 0); TCOW_DSR(fw[buf][1], ww, 4096);            \
        TCOW_DSR(fa[buf][0], aa, 0); TCOW_DSR(fa[buf][1], aa, 4096);            \
        TCOW_DSR(fa[buf][2], aa, 8192); TCOW_DSR(fa[buf][3], aa, 12288);        \
    } while (0)
#define TCOW_MFMA8(buf)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
            acc[i][j] = TCOW_MFMA_32x32x16_H16(__builtin_bit_cast(bf16x8, fa[buf][i]), __builtin_bit_cast(bf16x8, fw[buf][j]), acc[i][j], 0, 0, 0)

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TCOW_READ_FRAGS(0, 0, 0u);
    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t so = (uint32_t)(kt & 1) * B_STAGE;
        if (kt + 1 < nk) issue(kt + 1, (kt & 1) ^ 1);
        TCOW_READ_FRAGS(1, 1, so);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TCOW_MFMA8(0);
        TCOW_READ_FRAGS(0, 2, so);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TCOW_MFMA8(1);
        TCOW_READ_FRAGS(1, 3, so);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TCOW_MFMA8(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TCOW_MFMA8(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) TCOW_READ_FRAGS(0, 0, so ^ (uint32_t)B_STAGE);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef TCOW_DSR
#undef TCOW_READ_FRAGS
#undef TCOW_MFMA8

    // ---- epilogue: every wave stages its own 128 x 64 tile through a private 16 KiB LDS region, 64 rows at a time, and writes
    // full output rows (128 B bf16 / 256 B f32 per row).  No workgroup barrier: a wave's LDS operations execute in order.
    float* ct = reinterpret_cast<float*>(smem + wave * 16384);
    const int c4 = (lane & 15) * 4;
    const int gn = n0 + wn * 64 + c4;
    const bool col_ok = gn < p.N;
    const float4 b4 = (p.bias && col_ok) ? ld4(p.bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int mrow0 = m0 + wm * 128 + pass * 64;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(ii * 32 + crow32(r, hi)) * 64 + j * 32 + l31] = acc[pass * 2 + ii][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (col_ok) epi_rows(p, ct, 64, b4, mrow0 + (lane >> 4), lane >> 4, 4, c4, gn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads of this pass done before the next pass overwrites the region
    }
}

// ---- 320 x 256 tile, 8 waves (2 x 4, 160 x 64 each), one workgroup per CU.  Why this odd shape: the global->LDS stream bounds
// these GEMMs, and (a) 320 x 256 moves (320+256)/(320*256) = 1/142 B per FLOP (256-square 1/128, 128-square 1/64), (b) the path's
// M = 27 090 token rows are 84.7 x 320, so N = 768 / 2304 / 3072 give 255 / 765 / 1020 tiles = 0.996 of 1 / 3 / 4 full rounds
// over the 256 CUs (256-square: 318 tiles = 1.24 rounds for N = 768, which is why those GEMMs had to stay on the 128-square
// kernel).  LDS: (320 + 256) rows x 128 B x 2 stages = 144 KiB.  160 accumulator registers per lane + two fragment sets.
// (A 4-wave version with 160 x 128 per wave needs 320 accumulator registers: hipcc then shuttles accumulators between AGPRs and
// VGPRs around every MFMA -- 480 v_accvgpr moves per K-slice.)
constexpr int C_BM = 320, C_BN = 256;
constexpr int C_ATILE = C_BM * 128;             // 40 KiB
constexpr int C_WTILE = C_BN * 128;             // 32 KiB
constexpr int C_STAGE = C_ATILE + C_WTILE;      // 72 KiB
constexpr int C_LDS = 2 * C_STAGE;              // 144 KiB

// The main loop was first built stand-alone in tools/gemm_p8.hip (16x16x32 MFMAs on 1 KiB subtiles, four phases per K tile, wave rows staggered by a
// barrier): the accumulators sit as [10 row blocks of 16][4 column blocks of 16] (ML = 1 form of the shared epilogue; the round-2 two-stage loop on
// 32x32x16 MFMAs -- ML = 0, 7-12 % slower on every shape, profiles/r03_gemm_shapes.txt -- is gone).
template <typename E, int ML = 1>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_320_kernel(NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nblk = p.tiles_m * p.tiles_n;
    const int pid = xcd_remap(blockIdx.x, nblk);
    int pm, pn;
    nt_tile_of(pid, p.tiles_m, p.tiles_n, p.band, pm, pn);
    const int m0 = pm * C_BM, n0 = pn * C_BN;
    f32x16 acc[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 acc16[10][4];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_assert(ML == 1, "one main loop");
    {
        // (see tools/gemm_p8.hip for the layout and the ordering argument; BM = 320: 20 row blocks, 5 per wave and phase)
        constexpr int ARB = 20, RBH = 5, A_PLANE = ARB * 1024, KH = A_PLANE + 16 * 1024, KTILE = 2 * KH, NA = 3;
        const int wr = wm, wc = wn;
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
        const bool a_last = wave + 16 < ARB;
        auto load_a = [&](int kt, int kh) {
            char* dst = smem + (kt & 1) * KTILE + kh * KH;
            const bf16_t* g = a_base + (size_t)kt * 64 + kh * 32;
            glds16(g + a_src[0], dst + wave * 1024); glds16(g + a_src[1], dst + (wave + 8) * 1024);
            if (a_last) glds16(g + a_src[2], dst + (wave + 16) * 1024);
        };
        auto load_w = [&](int kt, int kh) {
            char* dst = smem + (kt & 1) * KTILE + kh * KH + A_PLANE;
            const bf16_t* g = w_base + (size_t)kt * 64 + kh * 32;
            glds16(g + w_src[0], dst + wave * 1024); glds16(g + w_src[1], dst + (wave + 8) * 1024);
        };
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        const uint32_t lds0_ = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
        const uint32_t frag_off = (uint32_t)((lane & 15) * 64 + (((lane >> 4) ^ (((lane & 15) >> 2) & 3)) << 4));
        const uint32_t a_ad = lds0_ + (wr * (ARB / 2)) * 1024 + frag_off;
        const uint32_t w_ad = lds0_ + A_PLANE + (wc * 4) * 1024 + frag_off;
        u32x4_ fa[2][RBH], fw[2][4];
#define P8_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
        auto read_a = [&](u32x4_ (&f)[RBH], uint32_t base, int ri) {
            if (ri == 0) { P8_DSR(f[0], base, 0); P8_DSR(f[1], base, 1024); P8_DSR(f[2], base, 2048); P8_DSR(f[3], base, 3072); P8_DSR(f[4], base, 4096); }
            else { P8_DSR(f[0], base, 5120); P8_DSR(f[1], base, 6144); P8_DSR(f[2], base, 7168); P8_DSR(f[3], base, 8192); P8_DSR(f[4], base, 9216); }
        };
        auto read_w = [&](u32x4_ (&f)[4], uint32_t base) { P8_DSR(f[0], base, 0); P8_DSR(f[1], base, 1024); P8_DSR(f[2], base, 2048); P8_DSR(f[3], base, 3072); };
        auto mfma_block = [&](const u32x4_ (&fA)[RBH], const u32x4_ (&fW)[4], int ri) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < RBH; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc16[ri * RBH + i][j] = TCOW_MFMA_16x16x32_H16(__builtin_bit_cast(bf16x8, fW[j]), __builtin_bit_cast(bf16x8, fA[i]), acc16[ri * RBH + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        auto wait_vm = [&](bool all) {
            if (all) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (a_last) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        };
#define P8_PHASE_TAIL(set_a, set_w, ri)                                                                   \
        do {                                                                                              \
            __builtin_amdgcn_s_barrier();                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            mfma_block(fa[set_a], fw[set_w], ri);                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            __builtin_amdgcn_s_barrier();                                                                 \
        } while (0)
        const int nk = p.K / 64;
        load_a(0, 0); load_w(0, 0); load_a(0, 1); load_w(0, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const uint32_t bo = (uint32_t)(kt & 1) * KTILE;
            const bool more = kt + 1 < nk;
            read_w(fw[0], w_ad + bo); read_a(fa[0], a_ad + bo, 0);
            if (more) load_a(kt + 1, 0);
            P8_PHASE_TAIL(0, 0, 0);
            read_a(fa[1], a_ad + bo, 1);
            if (more) load_w(kt + 1, 0);
            wait_vm(!more);
            P8_PHASE_TAIL(1, 0, 1);
            read_w(fw[1], w_ad + bo + KH); read_a(fa[0], a_ad + bo + KH, 0);
            if (more) load_a(kt + 1, 1);
            P8_PHASE_TAIL(0, 1, 0);
            read_a(fa[1], a_ad + bo + KH, 1);
            if (more) load_w(kt + 1, 1);
            wait_vm(!more);
            P8_PHASE_TAIL(1, 1, 1);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
#undef P8_DSR
#undef P8_PHASE_TAIL
    }
    wave_tile_epilogue_160x64<E, ML>(p, smem + wave * (64 * 68 * 4), acc, acc16, lane, m0 + wm * 160, n0 + wn * 64);
}

__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_kernel(NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int nblk = p.tiles_m * p.tiles_n;
    const int pid = xcd_remap(blockIdx.x, nblk);
    const int pm = pid / p.tiles_n, pn = pid - pm * p.tiles_n;
    const int m0 = pm * BM, n0 = pn * BN;
    const float4 b4 = epi_bias(p, tid, n0);

    // ---- per-lane source pointers for the direct-to-LDS loads: wave w issues wave-loads 4w..4w+3 per operand,
    // each covering 8 tile rows x 128 B; lane -> (row r = 8*q + (lane>>3), LDS chunk position lane&7).
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gm = m0 + r; gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + r; gn = gn < p.N ? gn : p.N - 1;
        a_src[j] = p.A + (size_t)gm * p.lda + c * 8;
        w_src[j] = p.W + (size_t)gn * p.ldw + c * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(a_src[j] + (size_t)kt * BK, sa + (wave * 4 + j) * 1024);
            glds16(w_src[j] + (size_t)kt * BK, sw + (wave * 4 + j) * 1024);
        }
    };

    // fragment byte offsets inside a tile (row-dependent part), constant over k
    int a_off[2], w_off[2], a_sw[2], w_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31, rw = wn * 64 + i * 32 + l31;
        a_off[i] = ra * 128; a_sw[i] = (ra >> 1) & 7;
        w_off[i] = rw * 128; w_sw[i] = (rw >> 1) & 7;
    }

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, stage ^ 1);
        const char* sa = smem + stage * STAGE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = 2 * ks + hi;
            bf16x8 fa[2], fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + a_off[i] + ((c ^ a_sw[i]) << 4)));
                fw[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sw + w_off[i] + ((c ^ w_sw[i]) << 4)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = TCOW_MFMA_32x32x16_H16(fa[i], fw[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    nt_epilogue(p, smem, acc, b4, tid, wm, wn, l31, hi, m0, n0);
}

}  // namespace

bool tcow_gemm_nt_c2_ok(const tcow_gemm_args* a);
int tcow_gemm_nt_bf16_c2(hipStream_t stream, const tcow_gemm_args* a);

// Tile order of the wide-output GEMMs (fc1, fc2's input gradient: N = 3072 at K = 768).  Row-major order gives an XCD a band of row tiles with all of W:
// 4.7 MB of W do not stay in a 4 MB L2 beside the A stream, every round of workgroups re-fetches them (FETCH_SIZE 4.4x the algorithmic bytes on the 320
// tile, 7.2x on the 160 tile).  Column bands (nt_tile_of) halve what an XCD keeps of W: measured per launch (profiles/r06_pmc_band.txt) 205 -> 172 MB with
// bands of 6 tiles on the 320 tile, 341 -> 227 MB with bands of 4 on the 160 tile -- and the same time within +-1 % (the refills come from the Infinity
// Cache and the kernel is not bound by them); narrower bands make more XCDs read the same A rows and fetch MORE (band 1: 516 MB).  Kept for the traffic.
int nt_band_for(const tcow_gemm_args* a, int tiles_n, int tile) {
    if (a->N < 3072) return 0;
    const int band = tile == 320 ? 6 : 4;
    return (tiles_n % band == 0 && tiles_n > band) ? band : 0;
}

int tcow_gemm_nt_bf16(hipStream_t stream, const tcow_gemm_args* a) {
    TCOW_CHECK_ARG(a->K % BK == 0, "tcow_gemm_nt(bf16): K=%d must be a multiple of %d", a->K, BK);
    TCOW_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0, "tcow_gemm_nt(bf16): lda/ldw must be multiples of 8 elements");
    TCOW_CHECK_ARG(a->ldc % 4 == 0 && a->N % 4 == 0, "tcow_gemm_nt(bf16): N and ldc must be multiples of 4");
    TCOW_CHECK_ARG((!a->resid || a->ldr % 4 == 0) && (!a->aux || a->ldaux % 4 == 0), "tcow_gemm_nt(bf16): ldr / ldaux must be multiples of 4");
    NtParams p = nt_params_from_args(a);
    p.tiles_m = cdiv(a->M, BM); p.tiles_n = cdiv(a->N, BN);
    TCOW_CHECK_ARG(a->tile == 0 || a->tile == 128 || a->tile == 160 || a->tile == 256 || a->tile == 320, "tcow_gemm_nt(bf16): tile must be 0, 128, 160, 256 or 320 (got %d)", a->tile);
    {
        // the 320 x 256 tile runs one workgroup per CU: take it when its tiles fill whole rounds of the 256 CUs
        const long t320 = (long)cdiv(a->M, C_BM) * cdiv(a->N, C_BN);
        const long rounds = (t320 + 255) / 256;
        const bool fills = t320 * 100 >= rounds * 256 * 80;   // (measured: still ahead of the 256 / 128 tiles at 88 % -- configs[3], configs[4])
        // the 160 x 256 tile at two workgroups per CU (gemm_nt_c2.hip): same shapes (its tiles are the wave rows of the 320 tile)
        // It takes the shapes where it measured ahead of the 320 tile at M = 27 090 (profiles/r04_gemm_c2.txt): short-K GEMMs with an f32 residual
        // epilogue (the HBM-bound epilogue hides under the co-resident workgroup's main loop) and plain short-K GEMMs of three rounds.  Same-box A/B
        // of the training step in round 4: 28.05 / 28.10 ms with this routing, 28.24 / 28.30 ms without the kernel.
        // ... and, since round 5, the x GELU' epilogue (fc2's input gradient): with its aux tile read non-temporally the pair fc2-gradient -> fc1-gradient takes
        // 250 us on the 160 tile, 260 on the 320 tile (269 before; profiles/r05_nontemporal.txt)
        const bool c2_pick = a->K <= 1024 && ((a->out_f32 && a->resid) || (a->act == TCOW_ACT_NONE && !a->row_scale && !a->resid && !a->bias2 && a->N >= 2304 && a->N < 3072) ||
                                               (a->act == TCOW_ACT_MUL_AUX && !a->out_f32 && !a->row_scale && !a->resid && !a->bias2));
        if (a->tile == 160 || (a->tile == 0 && c2_pick && fills && t320 >= 200 && tcow_gemm_nt_c2_ok(a))) {
            TCOW_CHECK_ARG(tcow_gemm_nt_c2_ok(a), "tcow_gemm_nt(bf16): tile 160 needs K %% 128 == 0 and operands below 2 GiB");
            return tcow_gemm_nt_bf16_c2(stream, a);
        }
        if (a->tile == 320 || (a->tile == 0 && fills && t320 >= 200)) {
            p.tiles_m = cdiv(a->M, C_BM); p.tiles_n = cdiv(a->N, C_BN);
            p.band = nt_band_for(a, p.tiles_n, 320);
            // epilogue specialisations for the combinations the path uses; anything else takes the run-time-configured kernel
            typedef void (*Kern)(NtParams);
            const int rows = (a->row_scale ? 1 : 0) | (a->resid ? 2 : 0) | (a->bias2 ? 4 : 0);
            const bool vec8 = a->N % 8 == 0 && a->ldc % 8 == 0 && a->ldr % 8 == 0 && a->ldaux % 8 == 0;   // the row-operand epilogues move 8 columns per lane
#define TCOW_PICK(R)                                                                                                                         \
            do {                                                                                                                             \
                k = gemm_nt_bf16_320_kernel<EpiAny, R>;                                                                                      \
                if (!vec8) { /* run-time configured kernel */ }                                                                              \
                else if (a->act == TCOW_ACT_NONE && rows == 0) k = gemm_nt_bf16_320_kernel<EpiCfg<TCOW_ACT_NONE, 0>, R>;                     \
                else if (a->act == TCOW_ACT_NONE && rows == 1) k = gemm_nt_bf16_320_kernel<EpiCfg<TCOW_ACT_NONE, 1>, R>;                     \
                else if (a->act == TCOW_ACT_NONE && rows == 2) k = gemm_nt_bf16_320_kernel<EpiCfg<TCOW_ACT_NONE, 2>, R>;                     \
                else if (a->act == TCOW_ACT_NONE && rows == 3) k = gemm_nt_bf16_320_kernel<EpiCfg<TCOW_ACT_NONE, 3>, R>;                     \
                else if (a->act == TCOW_ACT_NONE && rows == 7) k = gemm_nt_bf16_320_kernel<EpiCfg<TCOW_ACT_NONE, 7>, R>;                     \
                else if (a->act == TCOW_ACT_GELU_DSAVE && rows == 0) k = gemm_nt_bf16_320_kernel<EpiCfg<TCOW_ACT_GELU_DSAVE, 0>, R>;         \
                else if (a->act == TCOW_ACT_MUL_AUX && rows == 0) k = gemm_nt_bf16_320_kernel<EpiCfg<TCOW_ACT_MUL_AUX, 0>, R>;               \
                else if (a->act == TCOW_ACT_GELU && rows == 0) k = gemm_nt_bf16_320_kernel<EpiCfg<TCOW_ACT_GELU, 0>, R>;                     \
            } while (0)
            Kern k;
            TCOW_PICK(1);                        // (K % 64 == 0: checked above)
#undef TCOW_PICK
            tcow_ensure_lds(reinterpret_cast<const void*>(k), C_LDS);
            hipLaunchKernelGGL(k, dim3(p.tiles_m * p.tiles_n), dim3(512), C_LDS, stream, p);
            TCOW_CHECK_LAUNCH();
            return TCOW_OK;
        }
    }
    // the 256-square tile runs one workgroup per CU: it only pays when there are several full rounds of tiles (>= ~2.7 per CU)
    if (a->tile == 256 || (a->tile == 0 && (long)cdiv(a->M, B_BM) * cdiv(a->N, B_BN) >= 700)) {
        p.tiles_m = cdiv(a->M, B_BM); p.tiles_n = cdiv(a->N, B_BN);
        tcow_ensure_lds(reinterpret_cast<const void*>(gemm_nt_bf16_256_kernel), B_LDS);
        hipLaunchKernelGGL(gemm_nt_bf16_256_kernel, dim3(p.tiles_m * p.tiles_n), dim3(512), B_LDS, stream, p);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    // (tried and dropped on this tile: a 4-deep BK = 32 ring -- profiles/r01_gemm_variants.txt --, a register-direct epilogue with the swapped MFMA
    // orientation, BK = 32 two-stage and BK = 64 single-stage variants with 4 workgroups per CU: -5 ... -30 %)
    tcow_ensure_lds(reinterpret_cast<const void*>(gemm_nt_bf16_kernel), NT_LDS_BYTES);
    hipLaunchKernelGGL(gemm_nt_bf16_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), NT_LDS_BYTES, stream, p);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// =====================================================================================================
// Weight-gradient GEMM: slab[z][N][K] = sum over token rows of slice z of dY[m][n] * X[m][k].
// Both operands have the contraction index (token row m) as their slow dimension, so the MFMA fragments
// (8 consecutive m for one column) are gathered with the LDS transpose read ds_read_b64_tr_b16:
// within a 16-lane group lane 4r+c supplies the address of row r / 4-element column quad c of a [4][16]
// block and receives column (lane&15), rows 0..3 (verified on hardware, profiles/r01_hw_probe.txt).
// LDS rows are 256 B (128 columns); 16-byte chunk c of row r sits at chunk position c ^ ((r&3)<<2) so the four
// rows a half-wave touches per read fall into four different 64-byte bank segments.
namespace {

constexpr int TN_T = 128;                 // output tile edge (n and k)
// token rows per LDS stage: template parameter MC (64: 64 KiB of LDS, 2 workgroups/CU; 32: 32 KiB, 4 workgroups/CU)

__device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};

struct TnParams {
    int M, N, K;
    const bf16_t* dY; long ldy;
    const bf16_t* X; long ldx;
    float* slab;
    int tiles_n, tiles_k, mps, nz;   // mps: token rows per slice (multiple of TN_MC); nz slices
    float* bias_part;                // optional [nz][tiles_k][2][N] partial column sums of dY (bias gradient)
    int rows_per_pk;                 // LDS rows of each 64-row stage summed by the workgroup with k-tile index pk
};

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int off0) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off0 + 4 * 256));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int MC>
__global__ __launch_bounds__(256, MC == 32 ? 4 : 2) void gemm_tn_bf16_kernel(TnParams p) {
    constexpr int TILE_BYTES_ = MC * 256, STAGE_BYTES_ = 2 * TILE_BYTES_;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware placement: blocks are dispatched round-robin over the 8 XCDs (private L2s).  All output tiles of one token
    // slice z read the same dY / X rows, so slice z is pinned to XCD z % 8: its rows are fetched from HBM once per XCD-resident
    // slice instead of once per XCD (measured: FETCH_SIZE 4-8x the algorithmic bytes with the naive order).
    const int ntile = p.tiles_n * p.tiles_k;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int z = xcd + 8 * (idx / ntile), tile = idx % ntile;
    if (z >= p.nz) return;
    const int pn = tile / p.tiles_k, pk = tile - pn * p.tiles_k;
    const int n0 = pn * TN_T, k0 = pk * TN_T;
    const int mbeg = z * p.mps;
    const int mend = (mbeg + p.mps < p.M) ? mbeg + p.mps : p.M;

    // direct-to-LDS loads: wave-load q (16 per operand per stage) covers tile rows 4q..4q+3 x 256 B.
    const int lrow = lane >> 4;
    const int schunk = (lane & 15) ^ (lrow << 2);
    int ncol = n0 + schunk * 8; const bool n_ok = ncol < p.N;     // N, K multiples of 8 -> whole chunk in or out
    int kcol = k0 + schunk * 8; const bool k_ok = kcol < p.K;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(&g_zero16);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue = [&](int mt, int stage) {
        char* sy = smem + stage * STAGE_BYTES_;
        char* sx = sy + TILE_BYTES_;
#pragma unroll
        for (int j = 0; j < MC / 16; ++j) {
            const int q = wave * (MC / 16) + j;
            const int gm = mt + q * 4 + lrow;
            const bool ok = gm < mend;
            const bf16_t* ys = (ok && n_ok) ? p.dY + (size_t)gm * p.ldy + ncol : zero;
            const bf16_t* xs = (ok && k_ok) ? p.X + (size_t)gm * p.ldx + kcol : zero;
            glds16(ys, sy + q * 1024);
            glds16(xs, sx + q * 1024);
        }
    };

    // transpose-read addressing (constant over the loop)
    const int q16 = lane & 15, g16 = (lane >> 4) & 1, hi = lane >> 5;
    int y_off[2], x_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cy = wm * 64 + i * 32 + 16 * g16 + 4 * (q16 & 3);
        const int cx = wn * 64 + i * 32 + 16 * g16 + 4 * (q16 & 3);
        const int rr = 8 * hi + (q16 >> 2);
        y_off[i] = rr * 256 + ((((cy >> 3) ^ ((q16 >> 2) << 2))) << 4) + (cy & 7) * 2;
        x_off[i] = rr * 256 + ((((cx >> 3) ^ ((q16 >> 2) << 2))) << 4) + (cx & 7) * 2;
    }

    // column-sum duty of this thread: column cs_col of the dY tile, LDS rows [cs_r0, cs_r1) of every stage
    const int cs_col = tid & 127;
    const int cs_lo = pk * p.rows_per_pk, cs_hi = (cs_lo + p.rows_per_pk < MC) ? cs_lo + p.rows_per_pk : MC;
    const int cs_mid = cs_lo + (cs_hi - cs_lo + 1) / 2;
    const int cs_r0 = (tid >> 7) ? cs_mid : cs_lo, cs_r1 = (tid >> 7) ? cs_hi : (cs_mid < cs_hi ? cs_mid : cs_hi);
    float colsum = 0.f;

    const int nmt = (mend - mbeg + MC - 1) / MC;
    if (nmt > 0) {
        issue(mbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int it = 0; it < nmt; ++it) {
        const int stage = it & 1;
        if (it + 1 < nmt) issue(mbeg + (it + 1) * MC, stage ^ 1);
        const char* sy = smem + stage * STAGE_BYTES_;
        const char* sx = sy + TILE_BYTES_;
#pragma unroll
        for (int ks = 0; ks < MC / 16; ++ks) {
            bf16x8 fy[2], fx[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fy[i] = tr_frag(sy, y_off[i] + ks * 16 * 256);
                fx[i] = tr_frag(sx, x_off[i] + ks * 16 * 256);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = TCOW_MFMA_32x32x16_H16(fy[i], fx[j], acc[i][j], 0, 0, 0);
        }
        if (p.bias_part) {
            // bias gradient = column sums of dY: the dY stage is already in LDS; the tiles_k workgroups that share it split
            // its 64 rows between them (and each between its two thread halves), so the extra work is a few LDS reads each.
#pragma unroll 4
            for (int r = cs_r0; r < cs_r1; ++r) {
                const int off = r * 256 + ((((cs_col >> 3) ^ ((r & 3) << 2))) << 4) + (cs_col & 7) * 2;
                colsum += bf2f(*reinterpret_cast<const bf16_t*>(sy + off));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (p.bias_part && n0 + cs_col < p.N)
        p.bias_part[(((size_t)z * p.tiles_k + pk) * 2 + (tid >> 7)) * p.N + n0 + cs_col] = colsum;

    float* out = p.slab + (size_t)z * p.N * p.K;
    const int l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gk = k0 + wn * 64 + j * 32 + l31;
            if (gk >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gn = n0 + wm * 64 + i * 32 + crow32(r, hi);
                if (gn < p.N) out[(size_t)gn * p.K + gk] = acc[i][j][r];
            }
        }
}

}  // namespace

// ---- 256 x 256 output tile, 8 waves (2 x 4, 128 x 64 each), 64 token rows per stage, two stages = 128 KiB, one workgroup per CU.
// The 128-tile kernel above is bound by the global->LDS stream (a loads-only build takes 70 % of its time,
// profiles/r01_gemm_variants.txt); this tile moves half the bytes per FLOP.  Workgroups = tiles x slices <= 256 (one round):
// workgroup ids are handed out so that each XCD owns a contiguous range of (slice, tile) pairs, i.e. at most two token slices.
constexpr int T2 = 256;
constexpr int T2_MC = 64;
constexpr int T2_ROWB = T2 * 2;                 // 512 B per LDS row
constexpr int T2_TILE = T2_MC * T2_ROWB;        // 32 KiB per operand per stage
constexpr int T2_STAGE = 2 * T2_TILE;
constexpr int T2_LDS = 2 * T2_STAGE;            // 128 KiB

__device__ __forceinline__ bf16x8 tr_frag512(const char* tile, int off0) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off0 + 4 * T2_ROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

namespace {
// one workgroup of the 256-tile weight-gradient GEMM `p`: pid = slice * tiles + tile
// AB: ablation switches of tools/ubench_tn_ab.hip (0 in the library): 2 = no loads after the first stage, 4 = no barriers in the loop, 8 = no slab
// store (accumulators kept alive), 16 = no transpose reads in the loop, 32 = no MFMAs, 64 = no bias column sums.
// SCHED = 1 (round 4): the stage's ONE wait + barrier sits between the third and the fourth k-step instead of at the stage end: the fourth
// k-step's fragments are in registers by then, so its MFMAs run right behind the barrier while the NEXT stage's first fragments are read and the
// stage after next is requested into the buffer this stage has just released -- no stage boundary at which all eight waves wait for the barrier,
// then for their first transpose reads, with the matrix pipe idle.  (The round-3 order -- wait + barrier at the stage end -- is gone: 553-566 vs 504 us per block.)
template <int AB = 0, int SCHED = 1>
__device__ __forceinline__ void tn256_body(const TnParams& p, const int pid, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntile = p.tiles_n * p.tiles_k;
    const int z = pid / ntile, tile = pid - z * ntile;
    const int pn = tile / p.tiles_k, pk = tile - pn * p.tiles_k;
    const int n0 = pn * T2, k0 = pk * T2;
    const int mbeg = z * p.mps;
    const int mend = (mbeg + p.mps < p.M) ? mbeg + p.mps : p.M;

    // direct-to-LDS loads: a wave-load covers 2 tile rows x 512 B; wave w issues wave-loads 4w..4w+3 of each operand per stage
    const int lrow = lane >> 5, cl = lane & 31;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(&g_zero16);
    int ycol[2], xcol[2];                                         // source column of this lane for even / odd wave-loads (row & 3 differs)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r3 = (e << 1) | lrow;                            // (tile row) & 3 for wave-load q with q & 1 == e
        const int sc = cl ^ (r3 << 2);
        ycol[e] = n0 + sc * 8; xcol[e] = k0 + sc * 8;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue = [&](int mt, int stage) {
        char* sy = smem + stage * T2_STAGE;
        char* sx = sy + T2_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = wave * 4 + j;
            const int gm = mt + q * 2 + lrow;
            const bool ok = gm < mend;
            const bf16_t* ys = (ok && ycol[j & 1] < p.N) ? p.dY + (size_t)gm * p.ldy + ycol[j & 1] : zero;
            const bf16_t* xs = (ok && xcol[j & 1] < p.K) ? p.X + (size_t)gm * p.ldx + xcol[j & 1] : zero;
            glds16(ys, sy + q * 1024);
            glds16(xs, sx + q * 1024);
        }
    };
    const bool interior = n0 + T2 <= p.N && k0 + T2 <= p.K;
    // The fast path as buffer loads: descriptor + ONE per-lane byte offset per operand and row parity + a scalar row offset -- no vector
    // arithmetic per load, rows past M read as zeros.  (inline asm: hipcc does not count these loads; every wait in the loop is explicit.)
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    auto make_srd = [](const void* base, long bytes) {
        const uint64_t b = (uint64_t)(uintptr_t)base;
        i32x4_ r; r[0] = (int)(uint32_t)b; r[1] = (int)(uint32_t)((b >> 32) & 0xffffu); r[2] = (int)(uint32_t)bytes; r[3] = 0x00020000;
        return r;
    };
    const i32x4_ srd_y = make_srd(p.dY, ((long)(p.M - 1) * p.ldy + p.N) * 2), srd_x = make_srd(p.X, ((long)(p.M - 1) * p.ldx + p.K) * 2);
    const uint32_t yv0 = (uint32_t)(lrow * p.ldy + ycol[0]) * 2u, yv1 = (uint32_t)(lrow * p.ldy + ycol[1]) * 2u;
    const uint32_t xv0 = (uint32_t)(lrow * p.ldx + xcol[0]) * 2u, xv1 = (uint32_t)(lrow * p.ldx + xcol[1]) * 2u;
    const bool small32 = (long)p.M * p.ldy < (1L << 29) && (long)p.M * p.ldx < (1L << 29);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
#define TN_BLD(voff, srd, soff, ldsdst) \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(srd), "s"(soff), "s"(ldsdst) : "memory")
    auto issue_fast = [&](int mt, int stage) {
        const uint32_t dy = lds_base + stage * T2_STAGE + wave * 4096, dx = dy + T2_TILE;
        const uint32_t sy = (uint32_t)((long)(mt + wave * 8) * p.ldy * 2), sx = (uint32_t)((long)(mt + wave * 8) * p.ldx * 2);
        const uint32_t ry = (uint32_t)(2 * p.ldy * 2), rx = (uint32_t)(2 * p.ldx * 2);
        TN_BLD(yv0, srd_y, sy, dy); TN_BLD(xv0, srd_x, sx, dx);
        TN_BLD(yv1, srd_y, sy + ry, dy + 1024); TN_BLD(xv1, srd_x, sx + rx, dx + 1024);
        TN_BLD(yv0, srd_y, sy + 2 * ry, dy + 2048); TN_BLD(xv0, srd_x, sx + 2 * rx, dx + 2048);
        TN_BLD(yv1, srd_y, sy + 3 * ry, dy + 3072); TN_BLD(xv1, srd_x, sx + 3 * rx, dx + 3072);
    };
    // transpose-read addressing (constant over the loop)
    const int q16 = lane & 15, g16 = (lane >> 4) & 1, hi = lane >> 5;
    const int rr = 8 * hi + (q16 >> 2);
    int y_off[4], x_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cy = wm * 128 + i * 32 + 16 * g16 + 4 * (q16 & 3);
        y_off[i] = rr * T2_ROWB + ((((cy >> 3) ^ ((q16 >> 2) << 2))) << 4) + (cy & 7) * 2;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cx = wn * 64 + j * 32 + 16 * g16 + 4 * (q16 & 3);
        x_off[j] = rr * T2_ROWB + ((((cx >> 3) ^ ((q16 >> 2) << 2))) << 4) + (cx & 7) * 2;
    }

    // column-sum duty (bias gradient): column cs_col of the dY tile, rows [cs_r0, cs_r1) of every stage; the tiles_k workgroups
    // that share a dY tile split its 64 rows between them, and each between its two thread halves
    // (thread t sums the eight columns of 16-byte chunk t & 31 over rows cs_lo + (t >> 5), + 16, ...: at most four b128 reads per stage instead
    // of up to 32 two-byte ones; the sixteen row groups are folded through LDS once, after the loop)
    const int cs_col = tid & 255;
    const int cs_lo = pk * p.rows_per_pk, cs_hi = (cs_lo + p.rows_per_pk < T2_MC) ? cs_lo + p.rows_per_pk : T2_MC;
    const int cs_chunk = tid & 31, cs_rg = tid >> 5;
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int nmt = (mend - mbeg + T2_MC - 1) / T2_MC;
    auto issue_stage = [&](int st_, int buf_) {
        const int mt = mbeg + st_ * T2_MC;
        if (interior && small32 && (mt + T2_MC <= mend || mend == p.M)) issue_fast(mt, buf_); else issue(mt, buf_);
    };
    if constexpr (SCHED >= 2) {
        // (every stage through the buffer loads: the host picks this variant for whole tiles and 32-bit offsets only)
        if (nmt > 0) {
            issue_fast(mbeg, 0);
            if (nmt > 1) { issue_fast(mbeg + T2_MC, 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if (nmt > 0) {
        issue(mbeg, 0);
        if (SCHED == 1 && nmt > 1) { issue_stage(1, 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }      // stage 1 (8 loads per wave) stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // Transpose reads software-pipelined by hand, as in the NT kernels: the 12 ds_read_b64_tr_b16 of k-step ks+1 are issued
    // before the 8 MFMAs of k-step ks (inline asm + counted lgkmcnt; hipcc alone waits for each group right before its use).
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
    uint32_t ya[4], xa[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ya[i] = lds0 + (uint32_t)y_off[i];
#pragma unroll
    for (int j = 0; j < 2; ++j) xa[j] = lds0 + (uint32_t)x_off[j];
    u32x2 fyl[2][4], fyh[2][4], fxl[2][2], fxh[2][2];
    u32x4 kfrag = (u32x4){1u, 2u, 3u, 4u};                             // (ablation 128: MFMA operands that do not depend on the reads)
    if (AB & 128) asm volatile("" : "+v"(kfrag));
    u32x4 kfy[2][4], kfx[2][2];                                        // (ablation 512: DISTINCT constant operands, 24 registers as the real ones)
    if (AB & 512) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kfy[b][i] = (u32x4){1u, 2u, 3u, 4u};
                if (AB & 1024) {                                   // random bf16 pairs in (-1, 1) per lane: the data-dependent power of the MFMA pipe
#pragma unroll
                    for (int c = 0; c < 4; ++c) { uint32_t h = (uint32_t)(tid * 97 + b * 31 + i * 7 + c) * 2654435761u; kfy[b][i][c] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u); }
                }
                asm volatile("" : "+v"(kfy[b][i]));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                kfx[b][j] = (u32x4){1u, 2u, 3u, 4u};
                if (AB & 1024) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { uint32_t h = (uint32_t)(tid * 89 + b * 29 + j * 5 + c + 1000) * 2654435761u; kfx[b][j][c] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u); }
                }
                asm volatile("" : "+v"(kfx[b][j]));
            }
        }
    }
#define TCOW_TRR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define TCOW_TN_READ(buf, ks, so)                                                                                          \
    do {                                                                                                                   \
        if (AB & 16) break;                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                    \
            TCOW_TRR(fxl[buf][j], xa[j] + (so), T2_TILE + (ks) * 16 * T2_ROWB);                                            \
            TCOW_TRR(fxh[buf][j], xa[j] + (so), T2_TILE + (ks) * 16 * T2_ROWB + 4 * T2_ROWB);                              \
        }                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
            TCOW_TRR(fyl[buf][i], ya[i] + (so), (ks) * 16 * T2_ROWB);                                                      \
            TCOW_TRR(fyh[buf][i], ya[i] + (so), (ks) * 16 * T2_ROWB + 4 * T2_ROWB);                                        \
        }                                                                                                                  \
    } while (0)
#define TCOW_TN_FRAG(lo, hi) ((AB & 128) ? __builtin_bit_cast(bf16x8, kfrag) : __builtin_bit_cast(bf16x8, (u32x4){(lo).x, (lo).y, (hi).x, (hi).y}))
#define TCOW_TN_MFMA8(buf)                                                                                                 \
    if (!(AB & 32)) _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                      \
            acc[i][j] = TCOW_MFMA_32x32x16_H16(TCOW_TN_FRAG(fyl[buf][i], fyh[buf][i]), TCOW_TN_FRAG(fxl[buf][j], fxh[buf][j]), acc[i][j], 0, 0, 0)

    if (nmt > 0) TCOW_TN_READ(0, 0, 0u);
    if (AB & 16) {                                                     // (ablation: fragment registers defined once)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { fxl[b][j] = (u32x2){1u, 2u}; fxh[b][j] = (u32x2){3u, 4u}; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { fyl[b][i] = (u32x2){5u, 6u}; fyh[b][i] = (u32x2){7u, 8u}; }
        }
    }
    if constexpr (SCHED == 2) {
        // Every k-step as ONE block: its 8 MFMAs with the NEXT k-step's 12 transpose reads behind the first six of them (two each), the waits
        // counted per fragment -- issued as a burst before the MFMAs, the reads of all 8 waves queue up at the LDS while the MFMA pipe idles,
        // and then the LDS idles under the MFMAs: reads-only 30 us + MFMAs-only 60 us = 92 us measured with the loads off, no overlap at all
        // (profiles/r04_ubench_tn_ab.txt).  The fourth k-step also carries the 8 buffer loads of stage it+2, one behind each MFMA.
#define TN_BLDA(...) do { if (!(AB & 2)) TN_BLD(__VA_ARGS__); } while (0)
#define TN_RDA(...) do { if (!(AB & 16)) TCOW_TRR(__VA_ARGS__); } while (0)
#define TN_NOP do { } while (0)
#define TN_MF(cur, i, j) if (!(AB & 32)) acc[i][j] = TCOW_MFMA_32x32x16_H16((AB & 512) ? __builtin_bit_cast(bf16x8, kfy[cur][i]) : TCOW_TN_FRAG(fyl[cur][i], fyh[cur][i]), (AB & 512) ? __builtin_bit_cast(bf16x8, kfx[cur][j]) : TCOW_TN_FRAG(fxl[cur][j], fxh[cur][j]), acc[i][j], 0, 0, 0); \
                    __builtin_amdgcn_sched_barrier(0)
#define TN_BLK(cur, nxt, OFFK, son, W0, B0, B1, B2, B3, B4, B5, B6, B7)                                                                  \
    do {                                                                                                                                 \
        if (!(AB & 256)) asm volatile("s_waitcnt lgkmcnt(" #W0 ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0);                     \
        TN_MF(cur, 0, 0); TN_RDA(fxl[nxt][0], xa[0] + (son), T2_TILE + (OFFK)); TN_RDA(fxh[nxt][0], xa[0] + (son), T2_TILE + (OFFK) + 4 * T2_ROWB); B0; \
        TN_MF(cur, 0, 1); TN_RDA(fxl[nxt][1], xa[1] + (son), T2_TILE + (OFFK)); TN_RDA(fxh[nxt][1], xa[1] + (son), T2_TILE + (OFFK) + 4 * T2_ROWB); B1; \
        TN_MF(cur, 1, 0); TN_RDA(fyl[nxt][0], ya[0] + (son), (OFFK)); TN_RDA(fyh[nxt][0], ya[0] + (son), (OFFK) + 4 * T2_ROWB); B2;              \
        TN_MF(cur, 1, 1); TN_RDA(fyl[nxt][1], ya[1] + (son), (OFFK)); TN_RDA(fyh[nxt][1], ya[1] + (son), (OFFK) + 4 * T2_ROWB); B3;              \
        if (!(AB & 256)) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);    /* the previous block's fy[2] */ \
        TN_MF(cur, 2, 0); TN_RDA(fyl[nxt][2], ya[2] + (son), (OFFK)); TN_RDA(fyh[nxt][2], ya[2] + (son), (OFFK) + 4 * T2_ROWB); B4;              \
        TN_MF(cur, 2, 1); TN_RDA(fyl[nxt][3], ya[3] + (son), (OFFK)); TN_RDA(fyh[nxt][3], ya[3] + (son), (OFFK) + 4 * T2_ROWB); B5;              \
        if (!(AB & 256)) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);    /* ... and its fy[3] */  \
        TN_MF(cur, 3, 0); B6;                                                                                                            \
        TN_MF(cur, 3, 1); B7;                                                                                                            \
    } while (0)
        for (int it = 0; it < nmt; ++it) {
            const int stage = it & 1;
            const uint32_t so = (uint32_t)stage * T2_STAGE;
            const char* sy = smem + stage * T2_STAGE;
            TN_BLK(0, 1, 1 * 16 * T2_ROWB, so, 4, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP);
            TN_BLK(1, 0, 2 * 16 * T2_ROWB, so, 4, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP);
            TN_BLK(0, 1, 3 * 16 * T2_ROWB, so, 4, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP, TN_NOP);
            if (p.bias_part && !(AB & 64)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = cs_lo + cs_rg + 16 * u;
                    if (r < cs_hi) {
                        const uint4 v = *reinterpret_cast<const uint4*>(sy + r * T2_ROWB + ((cs_chunk ^ ((r & 3) << 2)) << 4));
                        csum[0] += bflo(v.x); csum[1] += bfhi(v.x); csum[2] += bflo(v.y); csum[3] += bfhi(v.y);
                        csum[4] += bflo(v.z); csum[5] += bfhi(v.z); csum[6] += bflo(v.w); csum[7] += bfhi(v.w);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (!(AB & 4)) __syncthreads();
            {
                const int mt2 = mbeg + (it + 2) * T2_MC;
                const bool more = it + 2 < nmt;
                const uint32_t dy = lds_base + so + wave * 4096, dx = dy + T2_TILE;
                const uint32_t by = (uint32_t)((long)(mt2 + wave * 8) * p.ldy * 2), bx = (uint32_t)((long)(mt2 + wave * 8) * p.ldx * 2);
                const uint32_t yw0 = more ? yv0 : 0x80000000u, yw1 = more ? yv1 : 0x80000000u;     // (the per-lane offset is the range-checked one)
                const uint32_t xw0 = more ? xv0 : 0x80000000u, xw1 = more ? xv1 : 0x80000000u;
                const uint32_t ry = (uint32_t)(2 * p.ldy * 2), rx = (uint32_t)(2 * p.ldx * 2);
                const uint32_t sn = so ^ (uint32_t)T2_STAGE;
                TN_BLK(1, 0, 0, sn, 0, TN_BLDA(yw0, srd_y, by, dy), TN_BLDA(xw0, srd_x, bx, dx), TN_BLDA(yw1, srd_y, by + ry, dy + 1024),
                       TN_BLDA(xw1, srd_x, bx + rx, dx + 1024), TN_BLDA(yw0, srd_y, by + 2 * ry, dy + 2048), TN_BLDA(xw0, srd_x, bx + 2 * rx, dx + 2048),
                       TN_BLDA(yw1, srd_y, by + 3 * ry, dy + 3072), TN_BLDA(xw1, srd_x, bx + 3 * rx, dx + 3072));
            }
        }
#undef TN_BLK
#undef TN_MF
#undef TN_NOP
#undef TN_BLDA
#undef TN_RDA
    } else
    if constexpr (SCHED == 1) {
        for (int it = 0; it < nmt; ++it) {
            const int stage = it & 1;
            const uint32_t so = (uint32_t)stage * T2_STAGE;
            const char* sy = smem + stage * T2_STAGE;
            TCOW_TN_READ(1, 1, so);
            asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            TCOW_TN_MFMA8(0);
            TCOW_TN_READ(0, 2, so);
            asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            TCOW_TN_MFMA8(1);
            TCOW_TN_READ(1, 3, so);
            asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            TCOW_TN_MFMA8(0);
            if (p.bias_part && !(AB & 64)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = cs_lo + cs_rg + 16 * u;
                    if (r < cs_hi) {
                        const uint4 v = *reinterpret_cast<const uint4*>(sy + r * T2_ROWB + ((cs_chunk ^ ((r & 3) << 2)) << 4));
                        csum[0] += bflo(v.x); csum[1] += bfhi(v.x); csum[2] += bflo(v.y); csum[3] += bfhi(v.y);
                        csum[4] += bflo(v.z); csum[5] += bfhi(v.z); csum[6] += bflo(v.w); csum[7] += bfhi(v.w);
                    }
                }
            }
            // this wave's loads of stage it+1 (requested a whole stage ago) have landed, its reads of this stage are back (the fourth k-step's
            // fragments are in registers): behind the barrier the buffer of this stage is free and stage it+1 is visible
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (!(AB & 4)) __syncthreads();
            if (it + 2 < nmt && !(AB & 2)) issue_stage(it + 2, stage);
            if (it + 1 < nmt) TCOW_TN_READ(0, 0, so ^ (uint32_t)T2_STAGE);
            __builtin_amdgcn_sched_barrier(0);
            TCOW_TN_MFMA8(1);
        }
    }
    // (SCHED = 2: the last stage's fourth k-step has requested fragments of a stage that does not exist into set 0.  The wait re-defines those
    // registers, so that hipcc -- which knows nothing of reads issued by asm statements -- cannot hand them to the code behind the loop before the
    // data is in: see the same note in gemm_nt_c2.hip, where exactly that corrupted tiles)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fxl[0][0]), "+v"(fxl[0][1]), "+v"(fxh[0][0]), "+v"(fxh[0][1]), "+v"(fyl[0][0]), "+v"(fyl[0][1]), "+v"(fyl[0][2]), "+v"(fyl[0][3]),
                   "+v"(fyh[0][0]), "+v"(fyh[0][1]), "+v"(fyh[0][2]), "+v"(fyh[0][3])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
#undef TCOW_TRR
#undef TCOW_TN_READ
#undef TCOW_TN_FRAG
#undef TCOW_TN_MFMA8
    if (p.bias_part) {
        // fold the sixteen row groups (the operand stages are dead: the loop ended on a barrier): red[rg][256 columns]
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[cs_rg * 256 + cs_chunk * 8 + e] = csum[e];
        __syncthreads();
        if (n0 + cs_col < p.N) {
            float t = 0.f;
            if (tid < 256) {
#pragma unroll
                for (int g = 0; g < 16; ++g) t += red[g * 256 + cs_col];
            }
            p.bias_part[(((size_t)z * p.tiles_k + pk) * 2 + (tid >> 8)) * p.N + n0 + cs_col] = t;      // (the second half-row of the table stays zero)
        }
        __syncthreads();
    }

    if (AB & 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[i][j]));
        return;
    }
    float* out = p.slab + (size_t)z * p.N * p.K;
    const int l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gk = k0 + wn * 64 + j * 32 + l31;
            if (gk >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gn = n0 + wm * 128 + i * 32 + crow32(r, hi);
                if (gn < p.N) out[(size_t)gn * p.K + gk] = acc[i][j][r];
            }
        }
}

template <int SCHED>
__global__ __launch_bounds__(512, 2) void gemm_tn_bf16_256_kernel(TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn256_body<0, SCHED>(p, xcd_remap(blockIdx.x, gridDim.x), smem);
}

// Grouped launch: the weight gradients of ONE transformer block (7 Linear layers, 153 tiles of 256 x 256 at ViT-B) as one grid.  Launched
// one by one, a 768 x 768 weight has 9 tiles and needs 28 token slices to fill the chip -- 66 MB of f32 partials written and read back per
// GEMM (11.5 GB per training step), a fold launch each, and a ramp / tail per launch.  Together the tiles fill three rounds with FIVE slices:
// every workgroup walks 5 418 token rows, the partials shrink 5x and one launch replaces seven.
constexpr int TN_GROUP_MAX = 40;          // (five divided space-time blocks: 35-40 problems; TnGroup = 3.7 KiB, below the 4 KiB kernel-argument limit)
struct TnGroup { int n; int first[TN_GROUP_MAX + 1]; TnParams p[TN_GROUP_MAX]; };
template <int SCHED>
__global__ __launch_bounds__(512, 2) void gemm_tn_bf16_256_group_kernel(TnGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pid = xcd_remap(blockIdx.x, gridDim.x);
    int k = 0;
    while (k + 1 < g.n && pid >= g.first[k + 1]) ++k;           // workgroup-uniform
    const TnParams p = g.p[k];
    tn256_body<0, SCHED>(p, pid - g.first[k], smem);
}

}  // namespace

// slices for the 256-tile kernel: as many as fit one round of workgroups (<= 256), at least 256 token rows each
int tcow_tn_splits_256(int M, int N, int K) {
    const int tiles = cdiv(N, T2) * cdiv(K, T2);
    int s = 256 / tiles;
    const int max_s = M / 256;
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}
bool tcow_tn_use_256(int M, int N, int K) {
    const int tiles = cdiv(N, T2) * cdiv(K, T2);
    // (a 768 x 768 weight = 9 tiles x 28 slices still wins 12 % over the 128-tile kernel despite the larger slab fold)
    return M >= 4096 && N >= 256 && K >= 256 && tiles >= 9 && tiles <= 256;
}

// stage loop of the 256-tile weight-gradient kernel (tn256_body): SCHED = 2 -- the transpose reads of the next k-step and the next stage's requests
// spread between the MFMAs of every k-step, the loads as buffer loads with scalar row offsets -- for whole 256-tiles with 32-bit byte offsets
// (tn_whole), SCHED = 1 (the same wait / barrier placement, general addressing) otherwise
static int tn_sched() { return 2; }
static bool tn_whole(int M, int N, int K, long ldy, long ldx) { return N % T2 == 0 && K % T2 == 0 && (long)M * ldy < (1L << 29) && (long)M * ldx < (1L << 29); }

int tcow_gemm_tn_bf16(hipStream_t stream, int M, int N, int K, const bf16_t* dY, long ldy, const bf16_t* X, long ldx, float* slab, int splits,
                      int* nz_out, float* bias_part, int* bias_parts_out) {
    TCOW_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0, "tcow_gemm_tn(bf16): N, K, ldy, ldx must be multiples of 8");
    TnParams p;
    p.M = M; p.N = N; p.K = K; p.dY = dY; p.ldy = ldy; p.X = X; p.ldx = ldx; p.slab = slab;
    p.tiles_n = cdiv(N, TN_T); p.tiles_k = cdiv(K, TN_T);
    constexpr int mc = 32;                      // token rows per stage of the 128-tile kernel: 32 (4 workgroups/CU) measured 5-25 % ahead of 64 (profiles/r01_gemm_tn_ab.txt)
    int mps = cdiv(M, splits); mps = ((mps + 63) / 64) * 64;
    p.mps = mps;
    const int nz = cdiv(M, mps);
    p.nz = nz;
    *nz_out = nz;
    p.bias_part = bias_part;
    p.rows_per_pk = cdiv(mc, p.tiles_k);
    if (bias_parts_out) *bias_parts_out = nz * p.tiles_k * 2;
    if (tcow_tn_use_256(M, N, K)) {
        p.tiles_n = cdiv(N, T2); p.tiles_k = cdiv(K, T2);
        p.rows_per_pk = cdiv(T2_MC, p.tiles_k);
        if (bias_parts_out) *bias_parts_out = nz * p.tiles_k * 2;
        const int sched = (tn_sched() >= 2 && !tn_whole(M, N, K, ldy, ldx)) ? 1 : tn_sched();
#define TN_LAUNCH(S)                                                                                                       \
    do {                                                                                                                   \
        tcow_ensure_lds(reinterpret_cast<const void*>(gemm_tn_bf16_256_kernel<S>), T2_LDS);                                \
        hipLaunchKernelGGL(gemm_tn_bf16_256_kernel<S>, dim3(nz * p.tiles_n * p.tiles_k), dim3(512), T2_LDS, stream, p);    \
    } while (0)
        if (sched == 2) TN_LAUNCH(2); else TN_LAUNCH(1);
#undef TN_LAUNCH
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    const dim3 grid(8 * cdiv(nz, 8) * p.tiles_n * p.tiles_k);
    hipLaunchKernelGGL(gemm_tn_bf16_kernel<32>, grid, dim3(256), 32768, stream, p);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// ---- grouped weight-gradient launch (see gemm_tn_bf16_256_group_kernel).  All problems share M and the slice count nz.
int tcow_tn_group_max(void) { return TN_GROUP_MAX; }
bool tcow_tn_group_ok(int n, const tcow_tn_problem* pr) {
    if (n < 2 || n > TN_GROUP_MAX) return false;
    for (int i = 0; i < n; ++i) {
        if (pr[i].M != pr[0].M || !tcow_tn_use_256(pr[i].M, pr[i].N, pr[i].K)) return false;
        if (pr[i].N % 8 || pr[i].K % 8 || pr[i].ldy % 8 || pr[i].ldx % 8) return false;
    }
    return true;
}
// common slice count: the cheapest one under  cost(s) = 1 / (fill of whole rounds of 256 workgroups) + 0.044 s  -- every slice writes and re-reads
// one f32 image of all the group's weights: slab store + fold measured at 22 % of the loop time with five slices (profiles/r04_ubench_tn_ab.txt,
// r04_step_kernel_stats.txt).  One ViT-B block (153 tiles): 5 slices (765 workgroups = 2.99 rounds); four blocks (612 tiles): 2 slices
// (1224 workgroups = 4.78 rounds, 60 % less slab traffic for 4 % more tail).  >= 256 token rows per slice.
int tcow_tn_group_slices(int n, const tcow_tn_problem* pr) {
    int tiles = 0;
    for (int i = 0; i < n; ++i) tiles += cdiv(pr[i].N, T2) * cdiv(pr[i].K, T2);
    int max_s = pr[0].M / 256; if (max_s > 64) max_s = 64; if (max_s < 1) max_s = 1;
    int best = 1; double best_cost = 1e30;
    for (int s = 1; s <= max_s; ++s) {
        const int wg = s * tiles, rounds = cdiv(wg, 256);
        const double cost = (rounds * 256.0) / (double)wg + 0.044 * s;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}
int tcow_gemm_tn_bf16_group(hipStream_t stream, int n, const tcow_tn_problem* pr, int nz_req, float* const* slabs, float* const* bias_parts, int* nz_out,
                            int* bias_nparts) {
    TnGroup g;
    g.n = n;
    int first = 0, nz = 0;
    for (int i = 0; i < n; ++i) {
        TnParams& p = g.p[i];
        p.M = pr[i].M; p.N = pr[i].N; p.K = pr[i].K; p.dY = (const bf16_t*)pr[i].dY; p.ldy = pr[i].ldy; p.X = (const bf16_t*)pr[i].X; p.ldx = pr[i].ldx;
        p.slab = slabs[i];
        p.tiles_n = cdiv(p.N, T2); p.tiles_k = cdiv(p.K, T2);
        int mps = cdiv(p.M, nz_req); mps = ((mps + 63) / 64) * 64;
        p.mps = mps; p.nz = cdiv(p.M, mps); nz = p.nz;
        p.bias_part = bias_parts[i];
        p.rows_per_pk = cdiv(T2_MC, p.tiles_k);
        bias_nparts[i] = p.nz * p.tiles_k * 2;
        g.first[i] = first;
        first += p.nz * p.tiles_n * p.tiles_k;
    }
    g.first[n] = first;
    for (int i = n + 1; i <= TN_GROUP_MAX; ++i) g.first[i] = first;
    *nz_out = nz;
    int sched = tn_sched();
    for (int i = 0; i < n && sched >= 2; ++i) if (!tn_whole(pr[i].M, pr[i].N, pr[i].K, pr[i].ldy, pr[i].ldx)) sched = 1;
#define TN_LAUNCH(S)                                                                                                       \
    do {                                                                                                                   \
        tcow_ensure_lds(reinterpret_cast<const void*>(gemm_tn_bf16_256_group_kernel<S>), T2_LDS);                          \
        hipLaunchKernelGGL(gemm_tn_bf16_256_group_kernel<S>, dim3(first), dim3(512), T2_LDS, stream, g);                   \
    } while (0)
    if (sched == 2) TN_LAUNCH(2); else TN_LAUNCH(1);
#undef TN_LAUNCH
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
