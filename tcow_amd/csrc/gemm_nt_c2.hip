// bf16 NT GEMM, 160 x 256 tile, FOUR waves, TWO workgroups per CU ("c2" = co-resident pair).
//
// Why a second big-tile kernel.  The 320 x 256 kernel of gemm_bf16.hip owns its CU (8 waves, 144 KiB of LDS, 160 accumulators per lane): while it
// runs its epilogue -- LDS staging, GELU arithmetic, 160-650 KB of row stores and row-operand loads per tile, 8-27 us per round -- the CU's MFMA
// pipe idles, and while it runs its main loop the CU's store path idles.  profiles/r03_gemm_fixed_cost.txt prices that at ~4 ms of a 32 ms step;
// profiles/r04_gemm_skew.txt shows it is a per-CU cost (de-synchronising the CUs of a launch changes nothing), and profiles/r04_ubench_glds.txt
// that the global->LDS path is nowhere near a limit (120 GB/s/CU from L2 against the 39 GB/s/CU the main loop draws).  So: the same wave tile
// (160 x 64 per wave, the same epilogue code) in a workgroup HALF the size, so that two workgroups share a CU and one's epilogue runs under the
// other's main loop.
//
//   * LDS: K is consumed in half tiles of 32 (A plane 10 KiB + W plane 16 KiB = 26 KiB of 1 KiB subtiles, 16 rows x 64 B, chunk c of row r at
//     position c ^ ((r >> 2) & 3) applied on the source address -- the layout of the 320 kernel's phase loop) through a RING OF THREE slots =
//     78 KiB per workgroup, two workgroups = 156 of the CU's 160 KiB.  The epilogue stages through the same 78 KiB (4 x 17 KiB).
//   * One wave per SIMD and workgroup, so the loop is software-pipelined inside the wave instead of across a staggered wave pair: a half step is
//     two blocks of 20 MFMAs (row blocks 0-4, 5-9); the fragments of the NEXT block are requested before the current block's MFMAs are issued
//     (two A sets, two W sets in rotation), and ONE workgroup barrier per half step sits between the two blocks: in front of it every wave has
//     waited (counted vmcnt) for its own loads of half tile j+1 and for its last reads of half tile j, behind it half tile j+1 is read and the
//     loads of half tile j+3 go into the slot half tile j just left -- two half steps (80 MFMAs of this wave plus whatever the co-resident
//     workgroup issues on the same SIMD) before they are needed.
//   * Co-resident workgroups would run in lock step (same work, same start) and reach their epilogues together; the second workgroup of every CU
//     therefore starts late by about half a main loop (NtParams::skew, first dispatch round only) -- from then on one is always ahead.
#include <stdlib.h>

#include "common.h"
#include "gemm_nt_common.h"

namespace {

constexpr int D_BM = 160, D_BN = 256;
constexpr int D_ARB = D_BM / 16;                    // 10 A row blocks
constexpr int D_APLANE = D_ARB * 1024;              // 10 KiB
constexpr int D_WPLANE = (D_BN / 16) * 1024;        // 16 KiB
constexpr int D_SLOT = D_APLANE + D_WPLANE;         // one half tile (k = 32)
constexpr int D_LDS = 3 * D_SLOT;                   // 79 872 B; the epilogue needs 4 x 17 408 = 69 632 B of it

// AB: ablation switches of tools/ubench_c2.hip (0 in the library): 1 = every workgroup loads tile (0, 0) (operands L2-resident), 2 = no loads
// after the prologue, 4 = no workgroup barriers in the loop, 8 = no epilogue (accumulators kept alive), 16 = no fragment reads in the loop, 32 = no MFMAs,
// 64 = no W loads, 128 = no A loads, 256 = whole-line loads (timing only).
template <typename E, int AB = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_c2_kernel(NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = column block of the wave's 160 x 64 tile
    const int nblk = p.tiles_m * p.tiles_n;
    const int pid = xcd_remap(blockIdx.x, nblk);
    const int pm = pid / p.tiles_n, pn = pid - pm * p.tiles_n;
    const int m0 = pm * D_BM, n0 = pn * D_BN;
    if (p.skew > 0) {
        const int b = blockIdx.x;
        const bool late = p.skew_mode == 2 ? (b < 512 && ((b >> 3) & 1)) : (b >= 256 && b < 512);
        if (late) {
            const long long t0 = wall_clock64();
            while (wall_clock64() - t0 < p.skew) __builtin_amdgcn_s_sleep(8);
        }
    }
    f32x16 acc[5][2];                                 // (unused: the epilogue's 32x32x16 form)
    f32x4 acc16[10][4];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- direct-to-LDS loads through BUFFER descriptors (bounds-checked: rows past M / N read as zeros, no per-lane clamping).  Wave w fills A
    // subtiles w, w+4, w+8 and W subtiles w, w+4, w+8, w+12 of every half tile; there are only ten A subtiles, so waves 2 and 3 load subtile
    // w+4 a second time instead of w+8 (same bytes to the same place) -- every wave then has SEVEN loads per half tile in flight and the
    // counted waits need no per-wave branch.  Address = descriptor base + one per-lane VGPR offset (row lane>>2 of a subtile, swizzled 16-byte
    // chunk) + a scalar offset (tile row, subtile row, k): two address registers for all fourteen load shapes, no vector arithmetic per load.
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    auto make_srd = [](const void* base, long bytes) {
        const uint64_t b = (uint64_t)(uintptr_t)base;
        i32x4_ r; r[0] = (int)(uint32_t)b; r[1] = (int)(uint32_t)((b >> 32) & 0xffffu); r[2] = (int)(uint32_t)bytes; r[3] = 0x00020000;
        return r;
    };
    const i32x4_ srd_a = make_srd(p.A, ((long)(p.M - 1) * p.lda + p.K) * 2);
    const i32x4_ srd_w = make_srd(p.W, ((long)(p.N - 1) * p.ldw + p.K) * 2);
    const int csw = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    // (AB & 256, timing only: every load instruction fetches 8 rows x 128 B -- whole cache lines -- instead of 16 rows x 64 B; same byte count)
    const uint32_t a_vo = (AB & 256) ? (uint32_t)((lane >> 3) * p.lda + (lane & 7) * 8) * 2u : (uint32_t)((lane >> 2) * p.lda + csw) * 2u;      // per-lane byte offsets inside a subtile
    const uint32_t w_vo = (AB & 256) ? (uint32_t)((lane >> 3) * p.ldw + (lane & 7) * 8) * 2u : (uint32_t)((lane >> 2) * p.ldw + csw) * 2u;
    const int qa2 = wave >= 2 ? wave + 4 : wave + 8;                       // third A subtile of this wave
    const uint32_t a_t0 = (AB & 1) ? 0u : (uint32_t)((long)m0 * p.lda * 2), w_t0 = (AB & 1) ? 0u : (uint32_t)((long)n0 * p.ldw * 2);
    const uint32_t sA0 = a_t0 + (uint32_t)(wave * 16 * p.lda * 2), sA1 = a_t0 + (uint32_t)((wave + 4) * 16 * p.lda * 2), sA2 = a_t0 + (uint32_t)(qa2 * 16 * p.lda * 2);
    const uint32_t sW0 = w_t0 + (uint32_t)(wave * 16 * p.ldw * 2), sWs = (uint32_t)(64 * p.ldw * 2);     // W subtiles wave + 4 i: scalar stride
    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
    // One load = one statement: M0 (LDS destination, wave-uniform) is written in the statement that reads it.  hipcc does not count these loads
    // (inline asm): every wait for them below is an explicit counted vmcnt.
#define C2_GLDS(voff, srd, soff, ldsdst) \
    if (!((AB & 64) && (&(srd) == &srd_w)) && !((AB & 128) && (&(srd) == &srd_a))) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(srd), "s"(soff), "s"(ldsdst) : "memory")
    const uint32_t dA0 = lds0 + wave * 1024, dA1 = lds0 + (wave + 4) * 1024, dA2 = lds0 + qa2 * 1024;
    const uint32_t dW0 = lds0 + D_APLANE + wave * 1024;             // W subtiles wave, +4, +8, +12
    auto load_half = [&](int j, uint32_t slot_off) {                // the seven loads of one half tile in one go (prologue only)
        const uint32_t kb = (uint32_t)j * 64u;
        C2_GLDS(w_vo, srd_w, sW0 + kb, dW0 + slot_off); C2_GLDS(w_vo, srd_w, sW0 + sWs + kb, dW0 + slot_off + 4096);
        C2_GLDS(a_vo, srd_a, sA0 + kb, dA0 + slot_off); C2_GLDS(a_vo, srd_a, sA1 + kb, dA1 + slot_off); C2_GLDS(a_vo, srd_a, sA2 + kb, dA2 + slot_off);
        C2_GLDS(w_vo, srd_w, sW0 + 2 * sWs + kb, dW0 + slot_off + 8192); C2_GLDS(w_vo, srd_w, sW0 + 3 * sWs + kb, dW0 + slot_off + 12288);
    };

    // ---- fragment reads (inline asm: hipcc must neither merge nor move them; every address is one per-lane constant + an immediate)
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    const uint32_t frag_off = (uint32_t)((lane & 15) * 64 + (((lane >> 4) ^ (((lane & 15) >> 2) & 3)) << 4));
    const uint32_t a_ad = lds0 + frag_off;
    const uint32_t w_ad = lds0 + D_APLANE + (wave * 4) * 1024 + frag_off;
    u32x4_ fa[2][5], fw[2][4];
#define C2_DSR0(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define C2_DSR(dst, addr, off) do { if (!(AB & 16)) C2_DSR0(dst, addr, off); } while (0)
#define C2_PIN __builtin_amdgcn_sched_barrier(0)
#define C2_MF(RI, i, jj, FA, FW) if (!(AB & 32)) acc16[(RI) * 5 + (i)][jj] = TCOW_MFMA_16x16x32_H16(__builtin_bit_cast(bf16x8, FW[jj]), __builtin_bit_cast(bf16x8, FA[i]), acc16[(RI) * 5 + (i)][jj], 0, 0, 0)

    const int nh = p.K / 32;                           // half tiles (K % 64 == 0: even, >= 2)
    uint32_t s_cur = 0, s_nxt = D_SLOT, s_nn = 2 * D_SLOT;            // ring: slot of half tile j, j+1, j+2 (= the slot half tile j+3 will take)
    load_half(0, 0); load_half(1, D_SLOT);
    if (nh > 2) { load_half(2, 2 * D_SLOT); asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    { const uint32_t bw = w_ad, ba = a_ad;
      C2_DSR0(fw[0][0], bw, 0); C2_DSR0(fw[0][1], bw, 1024); C2_DSR0(fw[0][2], bw, 2048); C2_DSR0(fw[0][3], bw, 3072);
      C2_DSR0(fa[0][0], ba, 0); C2_DSR0(fa[0][1], ba, 1024); C2_DSR0(fa[0][2], ba, 2048); C2_DSR0(fa[0][3], ba, 3072); C2_DSR0(fa[0][4], ba, 4096);
      if (AB & 16) { C2_DSR0(fw[1][0], bw, 0); C2_DSR0(fw[1][1], bw, 1024); C2_DSR0(fw[1][2], bw, 2048); C2_DSR0(fw[1][3], bw, 3072);
                     C2_DSR0(fa[1][0], ba, 0); C2_DSR0(fa[1][1], ba, 1024); C2_DSR0(fa[1][2], ba, 2048); C2_DSR0(fa[1][3], ba, 3072); C2_DSR0(fa[1][4], ba, 4096); } }

    // One half step j (slot s_cur; W set WS and fa[0] hold its fragments, requested during the previous block):
    //   block 0: 20 MFMAs on row blocks 0-4, with the five reads of row blocks 5-9 (-> fa[1]) issued between the first of them;
    //   counted vmcnt (this wave's loads of half tile j+1 have landed) + lgkmcnt(0) + ONE barrier: now every wave is done with slot s_cur's
    //   row blocks ... and half tile j+1 is visible;
    //   block 1: 20 MFMAs on row blocks 5-9, with the nine reads of half tile j+1 (W -> the other W set, row blocks 0-4 -> fa[0]) and the seven
    //   loads of half tile j+3 (into the slot half tile j leaves) issued one per MFMA gap -- no burst that leaves the MFMA pipe without work.
    // NEXT: half tile j+1 exists; INFL: half tile j+2 exists (its seven loads stay in flight across the barrier); LOAD: half tile j+3 exists
    // (literal constants at every expansion: the conditions fold at compile time).
#define C2_HALF_STEP(j, WS, NEXT, INFL, LOAD)                                                                                                  \
    do {                                                                                                                                  \
        const uint32_t ba1_ = a_ad + s_cur;                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); C2_PIN;                                                                        \
        C2_MF(0, 0, 0, fa[0], fw[WS]); C2_DSR(fa[1][0], ba1_, 5120); C2_PIN;                                                               \
        C2_MF(0, 0, 1, fa[0], fw[WS]); C2_DSR(fa[1][1], ba1_, 6144); C2_PIN;                                                               \
        C2_MF(0, 0, 2, fa[0], fw[WS]); C2_DSR(fa[1][2], ba1_, 7168); C2_PIN;                                                               \
        C2_MF(0, 0, 3, fa[0], fw[WS]); C2_DSR(fa[1][3], ba1_, 8192); C2_PIN;                                                               \
        C2_MF(0, 1, 0, fa[0], fw[WS]); C2_DSR(fa[1][4], ba1_, 9216); C2_PIN;                                                               \
        C2_MF(0, 1, 1, fa[0], fw[WS]); C2_MF(0, 1, 2, fa[0], fw[WS]); C2_MF(0, 1, 3, fa[0], fw[WS]);                                       \
        C2_MF(0, 2, 0, fa[0], fw[WS]); C2_MF(0, 2, 1, fa[0], fw[WS]); C2_MF(0, 2, 2, fa[0], fw[WS]); C2_MF(0, 2, 3, fa[0], fw[WS]);        \
        C2_MF(0, 3, 0, fa[0], fw[WS]); C2_MF(0, 3, 1, fa[0], fw[WS]); C2_MF(0, 3, 2, fa[0], fw[WS]); C2_MF(0, 3, 3, fa[0], fw[WS]);        \
        C2_MF(0, 4, 0, fa[0], fw[WS]); C2_MF(0, 4, 1, fa[0], fw[WS]); C2_MF(0, 4, 2, fa[0], fw[WS]); C2_MF(0, 4, 3, fa[0], fw[WS]);        \
        C2_PIN;                                                                                                                           \
        if (INFL) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        if (!(AB & 4)) __builtin_amdgcn_s_barrier();                                                                                      \
        C2_PIN;                                                                                                                           \
        const uint32_t bw_ = w_ad + s_nxt, ba0_ = a_ad + s_nxt;                                                                           \
        const uint32_t kb_ = (AB & 256) ? (uint32_t)(((j) + 3) >> 1) * 128u + (uint32_t)((((j) + 3) & 1) * 8 * p.lda * 2) : (uint32_t)((j) + 3) * 64u;   \
        const bool ld_ = (LOAD) && !(AB & 2);                                                                                             \
        C2_MF(1, 0, 0, fa[1], fw[WS]); if (NEXT) C2_DSR(fw[(WS) ^ 1][0], bw_, 0); C2_PIN;                                                            \
        C2_MF(1, 0, 1, fa[1], fw[WS]); if (NEXT) C2_DSR(fw[(WS) ^ 1][1], bw_, 1024); C2_PIN;                                                         \
        C2_MF(1, 0, 2, fa[1], fw[WS]); if (NEXT) C2_DSR(fw[(WS) ^ 1][2], bw_, 2048); C2_PIN;                                                         \
        C2_MF(1, 0, 3, fa[1], fw[WS]); if (NEXT) C2_DSR(fw[(WS) ^ 1][3], bw_, 3072); C2_PIN;                                                         \
        C2_MF(1, 1, 0, fa[1], fw[WS]); if (NEXT) C2_DSR(fa[0][0], ba0_, 0); C2_PIN;                                                                  \
        C2_MF(1, 1, 1, fa[1], fw[WS]); if (NEXT) C2_DSR(fa[0][1], ba0_, 1024); C2_PIN;                                                               \
        C2_MF(1, 1, 2, fa[1], fw[WS]); if (NEXT) C2_DSR(fa[0][2], ba0_, 2048); C2_PIN;                                                               \
        C2_MF(1, 1, 3, fa[1], fw[WS]); if (NEXT) C2_DSR(fa[0][3], ba0_, 3072); C2_PIN;                                                               \
        C2_MF(1, 2, 0, fa[1], fw[WS]); if (NEXT) C2_DSR(fa[0][4], ba0_, 4096); C2_PIN;                                                               \
        C2_MF(1, 2, 1, fa[1], fw[WS]); if (ld_) C2_GLDS(w_vo, srd_w, sW0 + kb_, dW0 + s_cur); C2_PIN;                                            \
        C2_MF(1, 2, 2, fa[1], fw[WS]); if (ld_) C2_GLDS(a_vo, srd_a, sA0 + kb_, dA0 + s_cur); C2_PIN;                                            \
        C2_MF(1, 2, 3, fa[1], fw[WS]); if (ld_) C2_GLDS(w_vo, srd_w, sW0 + sWs + kb_, dW0 + s_cur + 4096); C2_PIN;                                        \
        C2_MF(1, 3, 0, fa[1], fw[WS]); if (ld_) C2_GLDS(a_vo, srd_a, sA1 + kb_, dA1 + s_cur); C2_PIN;                                            \
        C2_MF(1, 3, 1, fa[1], fw[WS]); if (ld_) C2_GLDS(w_vo, srd_w, sW0 + 2 * sWs + kb_, dW0 + s_cur + 8192); C2_PIN;                                        \
        C2_MF(1, 3, 2, fa[1], fw[WS]); if (ld_) C2_GLDS(a_vo, srd_a, sA2 + kb_, dA2 + s_cur); C2_PIN;                                            \
        C2_MF(1, 3, 3, fa[1], fw[WS]); if (ld_) C2_GLDS(w_vo, srd_w, sW0 + 3 * sWs + kb_, dW0 + s_cur + 12288); C2_PIN;                                       \
        C2_MF(1, 4, 0, fa[1], fw[WS]); C2_MF(1, 4, 1, fa[1], fw[WS]); C2_MF(1, 4, 2, fa[1], fw[WS]); C2_MF(1, 4, 3, fa[1], fw[WS]);        \
        C2_PIN;                                                                                                                           \
        { const uint32_t t_ = s_cur; s_cur = s_nxt; s_nxt = s_nn; s_nn = t_; }                                                            \
    } while (0)

    int j = 0;
    for (; j + 4 < nh; j += 2) {                       // steady state: no conditions inside the loop
        C2_HALF_STEP(j, 0, true, true, true);
        C2_HALF_STEP(j + 1, 1, true, true, true);
    }
    if (j + 2 < nh) {                                  // second-to-last K tile: half tile j+4 does not exist
        C2_HALF_STEP(j, 0, true, true, true);
        C2_HALF_STEP(j + 1, 1, true, true, false);
        j += 2;
    }
    C2_HALF_STEP(j, 0, true, false, false);            // last K tile: nothing left to request
    C2_HALF_STEP(j + 1, 1, false, false, false);
#undef C2_HALF_STEP
#undef C2_MF
#undef C2_PIN
#undef C2_DSR
#undef C2_DSR0
#undef C2_GLDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // every wave has finished its fragment reads before any wave's staging writes land in the ring
    __builtin_amdgcn_s_barrier();
    if (AB & 8) {
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc16[i][j]));
        return;
    }
    wave_tile_epilogue_160x64<E, 1>(p, smem + wave * (64 * 68 * 4), acc, acc16, lane, m0, n0 + wave * 64);
}

}  // namespace

bool tcow_gemm_nt_c2_ok(const tcow_gemm_args* a) { return a->K % 64 == 0 && a->K >= 64; }

// launch the 160 x 256 kernel (the caller -- tcow_gemm_nt_bf16 -- has validated the arguments)
int tcow_gemm_nt_bf16_c2(hipStream_t stream, const tcow_gemm_args* a) {
    NtParams p = nt_params_from_args(a);
    p.tiles_m = cdiv(a->M, D_BM); p.tiles_n = cdiv(a->N, D_BN);
    // second workgroup of each CU: late by this many microseconds per 64-wide K tile (0.45 us: about half of a tile's main loop when it runs alone)
    static const float skew_us = [] { const char* e = getenv("TCOW_GEMM_C2_SKEW"); return e ? (float)atof(e) : 0.45f; }();
    static const int skew_mode = [] { const char* e = getenv("TCOW_GEMM_C2_SKEWMODE"); return e ? atoi(e) : 1; }();
    p.skew = (int)(skew_us * 100.f * (float)(a->K / 64)); p.skew_mode = skew_mode;
    typedef void (*Kern)(NtParams);
    const int rows = (a->row_scale ? 1 : 0) | (a->resid ? 2 : 0) | (a->bias2 ? 4 : 0);
    const bool vec8 = a->N % 8 == 0 && a->ldc % 8 == 0 && a->ldr % 8 == 0 && a->ldaux % 8 == 0;   // the row-operand epilogues move 8 columns per lane
    Kern k = gemm_nt_bf16_c2_kernel<EpiAny>;
    if (!vec8) { /* run-time configured epilogue */ }
    else if (a->act == TCOW_ACT_NONE && rows == 0) k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_NONE, 0>>;
    else if (a->act == TCOW_ACT_NONE && rows == 1) k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_NONE, 1>>;
    else if (a->act == TCOW_ACT_NONE && rows == 2) k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_NONE, 2>>;
    else if (a->act == TCOW_ACT_NONE && rows == 3) k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_NONE, 3>>;
    else if (a->act == TCOW_ACT_NONE && rows == 7) k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_NONE, 7>>;
    else if (a->act == TCOW_ACT_GELU_DSAVE && rows == 0) k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_GELU_DSAVE, 0>>;
    else if (a->act == TCOW_ACT_MUL_AUX && rows == 0) k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_MUL_AUX, 0>>;
    else if (a->act == TCOW_ACT_GELU && rows == 0) k = gemm_nt_bf16_c2_kernel<EpiCfg<TCOW_ACT_GELU, 0>>;
    tcow_ensure_lds(reinterpret_cast<const void*>(k), D_LDS);
    hipLaunchKernelGGL(k, dim3(p.tiles_m * p.tiles_n), dim3(256), D_LDS, stream, p);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
