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
//   * LDS (72 KiB per workgroup, two workgroups = 144 of the CU's 160 KiB): K tiles of 64 as 8-row x 128-byte subtiles -- one direct-to-LDS
//     load instruction moves whole cache lines.  A (160 rows, shared by the four waves) is double-buffered; W needs ONE buffer: wave w loads and
//     reads only the 64 W rows of its own columns, and its W fragments of a K tile sit in registers, so its part of the buffer is free for the
//     next tile as soon as it has read them.
//   * One wave per SIMD and workgroup, so the loop is software-pipelined inside the wave instead of across a staggered wave pair: a K tile is four
//     blocks of 20 MFMAs; the next block's fragment reads and the next tile's 13 loads are placed one per MFMA gap (no burst that leaves the MFMA
//     pipe without work), and ONE workgroup barrier per K tile publishes the next A tile.
//   * (A start skew of the second workgroup of each CU -- so that the pair's epilogues do not coincide -- was measured in round 4 and removed: over the
//     nine shapes of the path 851 us without, 870 / 883 / 924 us with 0.45 / 0.9 / 1.3 us per K tile; a late start is lost time.)
#include <stdlib.h>

#include "common.h"
#include "gemm_nt_common.h"

namespace {

constexpr int D_BM = 160, D_BN = 256;
constexpr int D_W = 32 * 1024;                      // W: 256 rows x 128 B (k = 64), ONE buffer: 32 subtiles of 8 rows
constexpr int D_A = 20 * 1024;                      // A: 160 rows x 128 B, TWO buffers
constexpr int D_LDS = D_W + 2 * D_A;                // 73 728 B; the epilogue needs 4 x 17 408 = 69 632 B of it

// AB: ablation switches of tools/ubench_c2.hip (0 in the library): 1 = every workgroup loads tile (0, 0), 2 = no loads after the prologue,
// 4 = no workgroup barriers in the loop, 8 = no epilogue (accumulators kept alive), 16 = no fragment reads in the loop, 32 = no MFMAs,
// 64 = no W loads, 128 = no A loads.
template <typename E, int AB = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_c2_kernel(NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = column block of the wave's 160 x 64 tile
    const int nblk = p.tiles_m * p.tiles_n;
    const int pid = xcd_remap(blockIdx.x, nblk);
    int pm, pn;
    nt_tile_of(pid, p.tiles_m, p.tiles_n, p.band, pm, pn);
    const int m0 = pm * D_BM, n0 = pn * D_BN;
    f32x16 acc[5][2];                                 // (unused: the epilogue's 32x32x16 form)
    f32x4 acc16[10][4];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- direct-to-LDS loads through BUFFER descriptors (bounds-checked: rows past M / N read as zeros -- the tile's row offset travels in the
    // SCALAR offset, and on gfx950 the range check of a raw buffer covers voffset + soffset: tools/probe_soffset.hip,
    // profiles/r05_probe_soffset.txt; a load whose sum passes num_records returns zeros and makes no memory request).  One load instruction moves one
    // SUBTILE = 8 rows x 128 B (k = 64): whole 128-byte cache lines.  (profiles/r04_c2_whole_line_probe.txt: with 16 rows x 64 B pieces --
    // the k = 32 planes of the 320 kernel's loop -- every line crosses the L2 -> L1 path twice, half of it unused each time, and the load
    // stream of a GEMM saturates the L2s at 14 TB/s of useful bytes; whole-line pieces move the same bytes in 0.57 of the time.)
    // LDS row r of a subtile holds its eight 16-byte chunks at positions c ^ (r & 6) (applied on the source address): the ds_read_b128 of a
    // 16-row x 32-k MFMA fragment -- lane l: row l & 15, chunk 4 kh + (l >> 4) -- then touches sixteen different 16-byte bank slots per
    // 16-lane service group.
    // W subtiles 8 w .. 8 w + 7 (= the 64 W rows of wave w's own columns) are loaded AND read by wave w only: W needs no workgroup barrier
    // and -- its fragments live in registers for a whole K tile -- only ONE buffer.  A subtile s is loaded by wave s & 3 and read by all.
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    auto make_srd = [](const void* base, long bytes) {
        const uint64_t b = (uint64_t)(uintptr_t)base;
        i32x4_ r; r[0] = (int)(uint32_t)b; r[1] = (int)(uint32_t)((b >> 32) & 0xffffu); r[2] = (int)(uint32_t)bytes; r[3] = 0x00020000;
        return r;
    };
    const i32x4_ srd_a = make_srd(p.A, ((long)(p.M - 1) * p.lda + p.K) * 2);
    const i32x4_ srd_w = make_srd(p.W, ((long)(p.N - 1) * p.ldw + p.K) * 2);
    const int lr = lane >> 3, lc = ((lane & 7) ^ (lr & 6)) * 8;
    const uint32_t a_vo = (uint32_t)(lr * p.lda + lc) * 2u;               // per-lane byte offsets inside a subtile
    const uint32_t w_vo = (uint32_t)(lr * p.ldw + lc) * 2u;
    const uint32_t a_t0 = (AB & 1) ? 0u : (uint32_t)((long)m0 * p.lda * 2), w_t0 = (AB & 1) ? 0u : (uint32_t)((long)n0 * p.ldw * 2);
    const uint32_t sA0 = a_t0 + (uint32_t)(wave * 8 * p.lda * 2), sAs = (uint32_t)(32 * p.lda * 2);          // A subtiles wave + 4 q: 32 rows apart
    const uint32_t sW0 = w_t0 + (uint32_t)(wave * 64 * p.ldw * 2), sWs = (uint32_t)(8 * p.ldw * 2);          // W subtiles 8 wave + q: 8 rows apart
    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
    const uint32_t dW0 = lds0 + wave * 8192, dA0 = lds0 + D_W + wave * 1024;
    // One load = one statement: M0 (LDS destination, wave-uniform) is written in the statement that reads it.  hipcc does not count these loads
    // (inline asm): every wait for them below is an explicit vmcnt.
#define C2_GLDS(voff, srd, soff, ldsdst) \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(srd), "s"(soff), "s"(ldsdst) : "memory")
#define C2_LDWV(q, kb, vo) do { if (!(AB & 64)) C2_GLDS(vo, srd_w, sW0 + (q) * sWs + (kb), dW0 + (q) * 1024); } while (0)
#define C2_LDAV(q, kb, bo, vo) do { if (!(AB & 128)) C2_GLDS(vo, srd_a, sA0 + (q) * sAs + (kb), dA0 + (bo) + (q) * 4096); } while (0)
#define C2_LDW(q, kb) C2_LDWV(q, kb, w_vo)
#define C2_LDA(q, kb, bo) C2_LDAV(q, kb, bo, a_vo)

    // ---- fragment reads (inline asm: hipcc must neither merge nor move them; every address is one per-lane constant + an immediate)
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    const int fr = lane & 15, fk = lane >> 4;
    const uint32_t fo0 = (uint32_t)((fr >> 3) * 1024 + (fr & 7) * 128 + ((fk ^ (fr & 6)) << 4));             // k half 0: chunks 0-3
    const uint32_t fo1 = (uint32_t)((fr >> 3) * 1024 + (fr & 7) * 128 + (((4 + fk) ^ (fr & 6)) << 4));       // k half 1: chunks 4-7
    const uint32_t w_ad0 = lds0 + wave * 8192 + fo0, w_ad1 = lds0 + wave * 8192 + fo1;                       // W block jj of this wave: + jj * 2048
    const uint32_t a_adb0 = lds0 + D_W + fo0, a_adb1 = lds0 + D_W + fo1;                                     // A block i: + buffer + i * 2048
    u32x4_ fa[2][5], fw[2][4];
#define C2_DSR0(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define C2_DSR(dst, addr, off) do { if (!(AB & 16)) C2_DSR0(dst, addr, off); } while (0)
#define C2_PIN __builtin_amdgcn_sched_barrier(0)
#define C2_MF(RI, i, jj, FA, FW) if (!(AB & 32)) acc16[(RI) * 5 + (i)][jj] = TCOW_MFMA_16x16x32_H16(__builtin_bit_cast(bf16x8, FW[jj]), __builtin_bit_cast(bf16x8, FA[i]), acc16[(RI) * 5 + (i)][jj], 0, 0, 0)
    // 20 MFMAs of one (row half RI, k half) block; F0 .. F12 are placed one per gap after the first thirteen MFMAs (empty arguments allowed)
#define C2_BLOCK(RI, FA, FW, F0, F1, F2, F3, F4, F5, F6, F7, F8, F9, F10, F11, F12)                                                       \
    do {                                                                                                                                  \
        C2_MF(RI, 0, 0, FA, FW); F0; C2_PIN; C2_MF(RI, 0, 1, FA, FW); F1; C2_PIN; C2_MF(RI, 0, 2, FA, FW); F2; C2_PIN;                     \
        C2_MF(RI, 0, 3, FA, FW); F3; C2_PIN; C2_MF(RI, 1, 0, FA, FW); F4; C2_PIN; C2_MF(RI, 1, 1, FA, FW); F5; C2_PIN;                     \
        C2_MF(RI, 1, 2, FA, FW); F6; C2_PIN; C2_MF(RI, 1, 3, FA, FW); F7; C2_PIN; C2_MF(RI, 2, 0, FA, FW); F8; C2_PIN;                     \
        C2_MF(RI, 2, 1, FA, FW); F9; C2_PIN; C2_MF(RI, 2, 2, FA, FW); F10; C2_PIN; C2_MF(RI, 2, 3, FA, FW); F11; C2_PIN;                   \
        C2_MF(RI, 3, 0, FA, FW); F12; C2_PIN; C2_MF(RI, 3, 1, FA, FW); C2_MF(RI, 3, 2, FA, FW); C2_MF(RI, 3, 3, FA, FW);                   \
        C2_MF(RI, 4, 0, FA, FW); C2_MF(RI, 4, 1, FA, FW); C2_MF(RI, 4, 2, FA, FW); C2_MF(RI, 4, 3, FA, FW); C2_PIN;                        \
    } while (0)
#define C2_NONE ((void)0)

    // ---- prologue: tile 0 in, its W k-half-0 and A rows 0-79 k-half-0 fragments requested
    const int nk = p.K / 64;
#pragma unroll
    for (int q = 0; q < 8; ++q) C2_LDW(q, 0u);
#pragma unroll
    for (int q = 0; q < 5; ++q) C2_LDA(q, 0u, 0u);
#pragma unroll
    for (int q = 0; q < 5; ++q) C2_LDA(q, 128u, (uint32_t)D_A);      // A of tile 1 (nk >= 2): stays in flight
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    C2_DSR0(fw[0][0], w_ad0, 0); C2_DSR0(fw[0][1], w_ad0, 2048); C2_DSR0(fw[0][2], w_ad0, 4096); C2_DSR0(fw[0][3], w_ad0, 6144);
    C2_DSR0(fa[0][0], a_adb0, 0); C2_DSR0(fa[0][1], a_adb0, 2048); C2_DSR0(fa[0][2], a_adb0, 4096); C2_DSR0(fa[0][3], a_adb0, 6144); C2_DSR0(fa[0][4], a_adb0, 8192);
    if (AB & 16) {
        C2_DSR0(fw[1][0], w_ad1, 0); C2_DSR0(fw[1][1], w_ad1, 2048); C2_DSR0(fw[1][2], w_ad1, 4096); C2_DSR0(fw[1][3], w_ad1, 6144);
        C2_DSR0(fa[1][0], a_adb0, 10240); C2_DSR0(fa[1][1], a_adb0, 12288); C2_DSR0(fa[1][2], a_adb0, 14336); C2_DSR0(fa[1][3], a_adb0, 16384); C2_DSR0(fa[1][4], a_adb0, 18432);
    }

    // One K tile kt (A in buffer kt & 1 at byte offset bo_; on entry fw[0] = W k-half 0, fa[0] = A rows 0-79 k-half 0, both requested during the
    // previous tile's last block).  Four blocks of 20 MFMAs -- (rows 0-79, k0), (rows 80-159, k0), (rows 0-79, k1), (rows 80-159, k1) -- each with
    // the NEXT block's fragment reads and a share of tile kt+1's loads placed one per MFMA gap:
    //   block 0: W k-half-1 fragments -> fw[1], A rows 80-159 k0 -> fa[1]; once those W reads are back (counted lgkmcnt) the wave's eight W loads
    //            of tile kt+1 go into the W buffer it has just finished reading;
    //   block 1: A rows 0-79 k1 -> fa[0];
    //   block 2: A rows 80-159 k1 -> fa[1]; then vmcnt(0) (A of tile kt+1 was requested a whole tile ago, W two and a half blocks ago) +
    //            lgkmcnt(0) + THE barrier of this K tile: every wave's A loads of tile kt+1 have landed, every wave is done reading A buffer kt & 1
    //            ... except block 3's fragments, which are in registers already;
    //   block 3: tile kt+1's W k-half-0 fragments -> fw[0] and A rows 0-79 k0 -> fa[0]; the wave's five A loads of tile kt+2 into the A buffer
    //            this tile has just left.
    // LOAD / LOAD2: tile kt+1 / kt+2 exists; PAR = kt & 1 (a literal constant at each expansion).
#define C2_TILE(kt, LOAD, LOAD2, PAR)                                                                                                           \
    do {                                                                                                                                  \
        const uint32_t bo_ = (PAR) * D_A;                                                                                                 \
        const uint32_t kb_ = (uint32_t)((kt) + 1) * 128u, kb2_ = (uint32_t)((kt) + 2) * 128u;                                             \
        /* no next tile: the per-lane offset is sent past the descriptor = zeros, no traffic */                                           \
        const uint32_t wv_ = (LOAD) ? w_vo : 0x80000000u, av_ = (LOAD2) ? a_vo : 0x80000000u;                                             \
        const bool ld_ = !(AB & 2);                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); C2_PIN;                                                                        \
        C2_BLOCK(0, fa[0], fw[0],                                                                                                         \
                 C2_DSR(fw[1][0], w_ad1, 0), C2_DSR(fw[1][1], w_ad1, 2048), C2_DSR(fw[1][2], w_ad1, 4096), C2_DSR(fw[1][3], w_ad1, 6144),   \
                 C2_DSR(fa[1][0], a_adb0, (PAR) * D_A + 10240), C2_DSR(fa[1][1], a_adb0, (PAR) * D_A + 12288),                              \
                 C2_DSR(fa[1][2], a_adb0, (PAR) * D_A + 14336), C2_DSR(fa[1][3], a_adb0, (PAR) * D_A + 16384),                              \
                 C2_DSR(fa[1][4], a_adb0, (PAR) * D_A + 18432); asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"),                         \
                 if (ld_) { C2_LDWV(0, kb_, wv_); C2_LDWV(1, kb_, wv_); }, if (ld_) { C2_LDWV(2, kb_, wv_); C2_LDWV(3, kb_, wv_); },                                \
                 if (ld_) { C2_LDWV(4, kb_, wv_); C2_LDWV(5, kb_, wv_); }, if (ld_) { C2_LDWV(6, kb_, wv_); C2_LDWV(7, kb_, wv_); });                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); C2_PIN;                                                                        \
        C2_BLOCK(1, fa[1], fw[0],                                                                                                         \
                 C2_DSR(fa[0][0], a_adb1, (PAR) * D_A), C2_DSR(fa[0][1], a_adb1, (PAR) * D_A + 2048), C2_DSR(fa[0][2], a_adb1, (PAR) * D_A + 4096), \
                 C2_DSR(fa[0][3], a_adb1, (PAR) * D_A + 6144), C2_DSR(fa[0][4], a_adb1, (PAR) * D_A + 8192),                                \
                 C2_NONE, C2_NONE, C2_NONE, C2_NONE, C2_NONE, C2_NONE, C2_NONE, C2_NONE);                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); C2_PIN;                                                                        \
        C2_BLOCK(0, fa[0], fw[1],                                                                                                         \
                 C2_DSR(fa[1][0], a_adb1, (PAR) * D_A + 10240), C2_DSR(fa[1][1], a_adb1, (PAR) * D_A + 12288),                              \
                 C2_DSR(fa[1][2], a_adb1, (PAR) * D_A + 14336), C2_DSR(fa[1][3], a_adb1, (PAR) * D_A + 16384),                              \
                 C2_DSR(fa[1][4], a_adb1, (PAR) * D_A + 18432), C2_NONE, C2_NONE, C2_NONE, C2_NONE, C2_NONE, C2_NONE, C2_NONE, C2_NONE);    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                                      \
        if (!(AB & 4)) __builtin_amdgcn_s_barrier();                                                                                      \
        C2_PIN;                                                                                                                           \
        C2_BLOCK(1, fa[1], fw[1],                                                                                                         \
                 C2_DSR(fw[0][0], w_ad0, 0), C2_DSR(fw[0][1], w_ad0, 2048), C2_DSR(fw[0][2], w_ad0, 4096),   \
                 C2_DSR(fw[0][3], w_ad0, 6144), C2_DSR(fa[0][0], a_adb0, (1 - (PAR)) * D_A),                           \
                 C2_DSR(fa[0][1], a_adb0, (1 - (PAR)) * D_A + 2048), C2_DSR(fa[0][2], a_adb0, (1 - (PAR)) * D_A + 4096), \
                 C2_DSR(fa[0][3], a_adb0, (1 - (PAR)) * D_A + 6144), C2_DSR(fa[0][4], a_adb0, (1 - (PAR)) * D_A + 8192), \
                 if (ld_) { C2_LDAV(0, kb2_, bo_, av_); C2_LDAV(1, kb2_, bo_, av_); }, if (ld_) { C2_LDAV(2, kb2_, bo_, av_); C2_LDAV(3, kb2_, bo_, av_); },        \
                 if (ld_) C2_LDAV(4, kb2_, bo_, av_), C2_NONE);                                                                                 \
    } while (0)

    // (two tiles per iteration: the A buffer of a tile is a literal, so every fragment address is ONE per-lane register + an immediate)
    // (nk is even -- tcow_gemm_nt_c2_ok; behind the last tile the loads are sent out of range -- zeros into free buffers, no memory traffic -- and
    // the fragment reads fetch stale bytes nobody uses: one straight-line loop body, no tail copies for hipcc's register allocator to trip over)
    for (int kt = 0; kt < nk; kt += 2) {
        const bool more = kt + 2 < nk;
        C2_TILE(kt, true, more, 0);
        C2_TILE(kt + 1, more, more, 1);
    }
#undef C2_TILE
#undef C2_BLOCK
#undef C2_NONE
#undef C2_MF
#undef C2_PIN
#undef C2_DSR
#undef C2_DSR0
#undef C2_LDA
#undef C2_LDAV
#undef C2_LDWV
#undef C2_LDW
#undef C2_GLDS
    // The last tile's block 3 has requested the fragments of a tile that does not exist (stale bytes nobody uses) into fa[0] / fw[0].  hipcc knows
    // nothing of reads issued by asm statements: to it those registers are dead behind the loop, free for the epilogue's values -- and register-only
    // instructions may be scheduled ABOVE a wait that clobbers nothing but memory, where the late data then lands on top of them (round 5: whole wave
    // tiles of the row-scale epilogue came out corrupted in 7 of 10 launches after an unrelated edit had shifted the register allocation).  The
    // wait therefore re-defines the nine registers: nothing can take them before the data is in.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[0][4]), "+v"(fw[0][0]), "+v"(fw[0][1]), "+v"(fw[0][2]), "+v"(fw[0][3])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // every wave has finished its fragment reads before any wave's staging writes land in the buffers
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

int nt_band_for(const tcow_gemm_args* a, int tiles_n, int tile);
bool tcow_gemm_nt_c2_ok(const tcow_gemm_args* a) { return a->K % 128 == 0 && (long)a->M * a->lda < (1L << 30) && (long)a->N * a->ldw < (1L << 30); }   // (32-bit byte offsets below 2^31 + the out-of-range marker)

// launch the 160 x 256 kernel (the caller -- tcow_gemm_nt_bf16 -- has validated the arguments)
int tcow_gemm_nt_bf16_c2(hipStream_t stream, const tcow_gemm_args* a) {
    NtParams p = nt_params_from_args(a);
    p.tiles_m = cdiv(a->M, D_BM); p.tiles_n = cdiv(a->N, D_BN);
    p.band = nt_band_for(a, p.tiles_n, 160);
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
