// Spatial attention forward, round 5: FOUR waves, one per SIMD, persistent over (frame, head) pairs ("p4").
//
// Replaces softmax(q k^T / 8) v of Attention.forward (vit.py:88-109) for workgroup-shared sequences without a causal mask (spatial attention,
// vit.py:184-186 / 206-208, and the joint sequences of vit.py:159-161).  The streaming kernel it succeeds (attn_fwd_stream, attention_bf16.hip)
// gives every wave ONE 32-query tile and relies on 2.6 co-resident waves per SIMD to overlap its serial chain S -> softmax -> PV: 56 us at
// BASELINE configs[1], MfmaUtil 19 %, and half of a workgroup's short life (10 tile steps) is prologue / epilogue (profiles/r04_pmc_attn.txt).
//
// Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure"):
//   * one workgroup per CU, 4 waves, each the only wave of its SIMD (up to 512 registers); 256 workgroups walk the work items
//     (pair, chunk of <= 12 query tiles) in a fixed order -- one WHOLE (frame, head) per item at S = 301, so K / V leave HBM exactly once;
//   * a wave owns NQ <= 3 query tiles of the item (10 tiles -> 3 / 3 / 2 / 2) and runs their S -> softmax -> PV chains INTERLEAVED: the
//     instruction stream alternates one MFMA with ~7 VALU instructions of another tile's softmax, in an order fixed at compile time
//     (sched_barrier pins every MFMA gap).  Per key tile and query tile: 8 MFMAs (32x32x16) and ~54 VALU instructions;
//   * every K / V fragment read from LDS feeds the MFMAs of all NQ tiles (a third of the LDS traffic of one tile per wave);
//   * K / V tiles stream through a ring of R slots filled by LDS-DMA (buffer_load ... lds, one 1 KiB piece per wave and tile), requested
//     R - 2 tiles ahead ACROSS item seams, one workgroup barrier per key tile, counted vmcnt;
//   * Q tiles of the NEXT item are requested at the start of the current one into a wave-private staging area; O leaves as whole 128-byte
//     rows through a wave-private staging tile (store_tile_staged);
//   * VALU diet: the running maximum lives in the MFMA's C operand (S' = K Q'^T - m comes out of the matrix pipe: no subtraction per element),
//     Q is multiplied by 0.125 log2(e) once per item (PRE = false) or arrives pre-scaled (PRE = true), so p = exp2(S') is ONE instruction;
//     the maximum is lazy (rescale only when a row grew by more than 2^8).
#include <stdlib.h>

#include "../tcow_amd/csrc/attention_tiles.h"

namespace {

// NW = waves per workgroup.  4: one wave per SIMD, up to three query tiles per wave (the whole register file per wave).  8: two waves per
// SIMD (w and w + 4), at most two query tiles per wave -- 256 registers each, everything in architectural VGPRs.
template <int NW> struct P4Cfg {
    static constexpr int R = NW == 4 ? 6 : 5;                 // ring slots (key tiles in LDS)
    static constexpr int NQ = NW == 4 ? 3 : 2;                // query tiles per wave, at most
    static constexpr int PIECES = 8 / NW;                     // 1 KiB LDS-DMA pieces per wave and key tile (a K / V tile pair is eight)
    // the running maximum as the C operand of the first S MFMA (S - m straight out of the matrix pipe) costs 16 registers per query tile for the
    // splat of -m: taken where the register file has room (NW = 4); with 256 registers per wave the softmax subtracts m itself
    static constexpr bool CT = NW == 4;
    static constexpr int SLOT = 2 * TILE_B;                   // K tile + V tile
    static constexpr int RING = R * SLOT;
    static constexpr int QST = NQ * TILE_B;                   // Q staging per wave
    static constexpr int LDS = RING + NW * QST + NW * TILE_B; // + O staging per wave: 114 688 B (NW = 4) / 139 264 B (NW = 8)
};
constexpr int P4_ITEM_TILES = 12;                         // query tiles per work item, at most (both configurations)
constexpr uint32_t P4_OOB = 0x80000000u;                  // per-lane buffer offset that fails the descriptor's range check: zeros, no traffic

typedef int p4_i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t p4_u32x4 __attribute__((ext_vector_type(4)));

// one query tile of a wave
struct P4Tile {
    bf16x8 qf[4];            // Q fragments (B operand of S^T = K Q^T), pre-scaled by 0.125 log2(e)
    f32x16 o0, o1;           // O^T accumulators (channels 0-31 / 32-63)
    f32x16 negm;             // -m in all 16 registers: the C operand of the first S MFMA
    f32x16 s;                // S' = scores - m (log2 units) of the tile in flight
    float m, l;              // reference maximum, lane-local partial row sum
    float t[5], mx;          // maximum tree
    float pa, pb;            // partial sums of the tile in flight
    uint32_t pk[8];          // P as packed 16-bit pairs: pk[0..3] = keys crow32(0..7), pk[4..7] = keys crow32(8..15)
};

struct P4Item { int valid; long base; int head; int q0; int n; };

// item v of the launch: XCD x = v & 7 takes pairs x, x + 8, ... and the chunks of one pair back to back (v and v + 8 are chunks of the same pair when
// nchunk > 1: they run at the same time on the same XCD and share K / V through its L2)
__device__ __forceinline__ P4Item p4_item(const SeqDesc& sd, int v, int pairs, int nchunk, int per, int nqt) {
    const int x = v & 7, k = v >> 3;
    const int i = k / nchunk, chunk = k - i * nchunk;
    const int pair = 8 * i + x;
    P4Item it;
    it.valid = pair < pairs;
    const int item = pair / sd.heads;
    it.head = pair - item * sd.heads;
    it.base = seq_base(sd, it.valid ? item : 0);
    it.q0 = chunk * per;
    it.n = (nqt - it.q0 < per) ? nqt - it.q0 : per;
    if (it.n < 0) { it.n = 0; it.valid = 0; }
    return it;
}

__device__ __forceinline__ p4_i32x4 p4_srd(const bf16_t* base, uint32_t bytes) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    p4_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((b >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// one LDS-DMA piece: 64 lanes x 16 B -> 1 KiB at LDS address `dst` (wave-uniform).  hipcc does not count these loads: every wait is explicit.
// (the descriptor words pass through readfirstlane at the point of use: a descriptor that lives across the item loop can end up in vector registers,
// and an "s" constraint does not move it back -- the assembler then rejects the instruction)
__device__ __forceinline__ p4_i32x4 p4_uniform(const p4_i32x4& d) {
    p4_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane(d[0]); r[1] = __builtin_amdgcn_readfirstlane(d[1]);
    r[2] = __builtin_amdgcn_readfirstlane(d[2]); r[3] = __builtin_amdgcn_readfirstlane(d[3]);
    return r;
}
#define P4_DMA(voff, srd, soff, dst)                                                                                                       \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(srd),                         \
                 "s"(__builtin_amdgcn_readfirstlane((int)(soff))), "s"(__builtin_amdgcn_readfirstlane((int)(dst))) : "memory")

__device__ __forceinline__ float p4_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// ablation switches of tools/ubench_p4.hip (0 in the library): 1 = no softmax arithmetic, 2 = no S MFMAs, 4 = no PV MFMAs, 8 = no K / V fragment reads
#ifndef P4_AB
#define P4_AB 0
#endif
#ifndef P4_ASM_ROUNDS
#define P4_ASM_ROUNDS 1      // NW = 8: the rounds of attn_round_asm.inc (0: the C++ rounds, compiler-scheduled)
#endif

// ---- the softmax of one tile step in eight pieces (one per MFMA gap).  G = 0 .. 3 run beside the PV MFMAs of the previous tile in the round,
// G = 4 .. 7 beside the S MFMAs of the next one.
template <int G, bool FIRST, bool LASTPAD, bool AG, bool CT>
__device__ __forceinline__ void p4_softmax_piece(P4Tile& t, int lr, int hi) {
    if constexpr ((P4_AB & 1) != 0) return;
    if constexpr (G == 0) {
        if constexpr (!CT && !FIRST) {                                   // (CT: the MFMA's C operand has done this)
#pragma unroll
            for (int r = 0; r < 16; ++r) t.s[r] -= t.m;
        }
        if constexpr (LASTPAD) {                                         // key padding of the sequence's last tile (keys >= L): exp2 underflows to 0
#pragma unroll
            for (int r = 0; r < 16; ++r) t.s[r] = (crow32(r, hi) >= lr) ? -1e30f : t.s[r];
        }
        t.t[0] = p4_max3(t.s[0], t.s[1], t.s[2]); t.t[1] = p4_max3(t.s[3], t.s[4], t.s[5]); t.t[2] = p4_max3(t.s[6], t.s[7], t.s[8]);
        t.t[3] = p4_max3(t.s[9], t.s[10], t.s[11]); t.t[4] = p4_max3(t.s[12], t.s[13], t.s[14]);
    } else if constexpr (G == 1) {
        const float u0 = p4_max3(t.t[0], t.t[1], t.t[2]), u1 = p4_max3(t.t[3], t.t[4], t.s[15]);
        t.mx = half_max(fmaxf(u0, u1));                                  // row maximum over the tile's 32 keys, relative to m
    } else if constexpr (G == 2) {
        // lazy maximum: the reference moves only when some row of the wave grew by more than 2^8 (probabilities then stay <= 256); the first
        // tile of an item always sets it.  Everything still at the old reference is rescaled exactly once: O, l, and this tile's scores.
        if (FIRST || __any(t.mx > 8.0f)) {
            TCOW_NO_IFCVT();
            const float delta = FIRST ? t.mx : fmaxf(t.mx, 0.0f);
            t.m += delta;
            if (!FIRST) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                t.l *= alpha;
                // (AG: the accumulators are re-defined by an empty asm statement INSIDE the cold block, so that hipcc's speculative hoisting
                // cannot lift their 32 accumulation-register reads into the hot path above the branch -- it did: 136 copies per key tile)
                if constexpr (AG) asm volatile("" : "+a"(t.o0), "+a"(t.o1));
                if constexpr (AG) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");      // (O was written by asm MFMAs: see p4_acc_fence)
#pragma unroll
                for (int r = 0; r < 16; ++r) { t.o0[r] *= alpha; t.o1[r] *= alpha; }
                // (the rescaled accumulators go back to the accumulation registers INSIDE this cold block: both values that meet behind the branch
                // are then of that class, and the hot path keeps O where its MFMAs want it instead of copying 32 registers per tile step)
                if constexpr (AG) asm volatile("" : "+a"(t.o0), "+a"(t.o1));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { t.s[r] -= delta; if constexpr (CT) t.negm[r] = -t.m; }
        }
        const float p0 = __builtin_amdgcn_exp2f(t.s[0]), p1 = __builtin_amdgcn_exp2f(t.s[1]);
        t.pa = p0; t.pb = p1; t.pk[0] = pack_bf2(p0, p1);
    } else {
        // pairs (2i, 2i + 1): G = 3: 1, 2;  4: 3;  5: 4, 5;  6: 6;  7: 7 + the row sum
        constexpr int i0 = G == 3 ? 1 : G == 4 ? 3 : G == 5 ? 4 : G == 6 ? 6 : 7;
        constexpr int i1 = G == 3 ? 3 : G == 4 ? 4 : G == 5 ? 6 : G == 6 ? 7 : 8;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const float p0 = __builtin_amdgcn_exp2f(t.s[2 * i]), p1 = __builtin_amdgcn_exp2f(t.s[2 * i + 1]);
            t.pa += p0; t.pb += p1; t.pk[i] = pack_bf2(p0, p1);
        }
        if constexpr (G == 7) t.l += t.pa + t.pb;
    }
}

__device__ __forceinline__ bf16x8 p4_pfrag(const P4Tile& t, int h) {
    const p4_u32x4 w = {t.pk[4 * h], t.pk[4 * h + 1], t.pk[4 * h + 2], t.pk[4 * h + 3]};
    return __builtin_bit_cast(bf16x8, w);
}

// what a wave needs to walk the ring
template <int NW> struct P4Ctx {
    uint32_t lds0;                      // LDS address of the ring
    uint32_t koff[4];                   // lane offsets of the four K row fragments inside a tile
    uint32_t voff[2][2];                // lane offsets of the V transpose reads: [dt][row block 0 / +8], key half s adds 2048
    uint32_t vo[P4Cfg<NW>::PIECES], vo_last[P4Cfg<NW>::PIECES];   // per-lane buffer offsets of this wave's pieces (8 rows each) of a tile; the sequence's last tile (padding rows out of range)
    uint32_t tile_stride;               // bytes between consecutive 32-row tiles in global memory
    int wave, lane, hi, l31;
    int nt, lr;                         // key tiles per sequence, valid rows of the last one
    // producer
    int pg, total_g, pj;                // next tile to request (global count), tiles this workgroup consumes in all, its index inside its item
    p4_i32x4 srd_kv;                    // descriptor of this wave's operand (K or V) of the item being requested
    int pv;                             // its item index
    // consumer
    int g;                              // tile being consumed (global count)
#ifdef P4_STAMPS
    long long t_wait = 0, t_pro = 0, t_rounds = 0, t_epi = 0, t_vm = 0;
#endif
};

enum { P4_FIRST = 0, P4_STEADY = 1, P4_DRAIN = 2 };

// phase stamps of tools/ubench_p4.hip (never in the library): shader cycles per wave spent waiting at the key-tile barriers, in the item
// prologues (Q fragments, accumulator reset), in the rounds and in the stores
#ifdef P4_STAMPS
__device__ long long* g_p4_dbg;
#define P4_NOW() ((long long)__builtin_readcyclecounter())
#define P4_STAMP(...) __VA_ARGS__
#else
#define P4_STAMP(...)
#endif

// ---- request the next tile of the stream into its ring slot (this wave's pieces) and advance the producer.  Piece i of a tile pair: i < 4 rows
// 8 i .. 8 i + 7 of the K tile, i >= 4 of the V tile; wave w brings pieces w PIECES .. (w + 1) PIECES - 1 (all of one operand).
template <int NW, typename F>
__device__ __forceinline__ void p4_produce(P4Ctx<NW>& c, F&& next_item) {
    typedef P4Cfg<NW> C;
    if (c.pg >= c.total_g) return;
    const uint32_t dst = c.lds0 + (uint32_t)(c.pg % C::R) * C::SLOT + (uint32_t)(c.wave * C::PIECES) * 1024u;
    const uint32_t soff = (uint32_t)c.pj * c.tile_stride;
    const bool last = c.pj == c.nt - 1;
    const p4_i32x4 srd = p4_uniform(c.srd_kv);
#pragma unroll
    for (int i = 0; i < C::PIECES; ++i) {
        const uint32_t vo = last ? c.vo_last[i] : c.vo[i];
        P4_DMA(vo, srd, soff, dst + (uint32_t)i * 1024u);
    }
    ++c.pg;
    if (++c.pj == c.nt) { c.pj = 0; next_item(c); }
}

// ---- end of a consumed tile: wait for this wave's pieces of tile g + 1, meet the others, hand the freed slot to the producer.  The wait counts
// LOADS younger than tile g + 1 (PIECES per requested tile: tiles g + 2 .. pg - 1); Q requests and O stores issued in between only lengthen it.
template <int NW, typename F>
__device__ __forceinline__ void p4_advance(P4Ctx<NW>& c, F&& next_item) {
    typedef P4Cfg<NW> C;
    if (c.g + 1 >= c.total_g) { ++c.g; return; }
    P4_STAMP(const long long t0_ = P4_NOW();)
    if (c.pg - c.g - 2 >= C::R - 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::PIECES * (C::R - 3)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P4_STAMP(const long long t1_ = P4_NOW(); c.t_vm += t1_ - t0_;)
    __builtin_amdgcn_s_barrier();
    P4_STAMP(c.t_wait += P4_NOW() - t1_;)
    p4_produce(c, next_item);
    ++c.g;
}

template <int NW>
__device__ __forceinline__ void p4_read_k(const P4Ctx<NW>& c, int g, bf16x8 (&kf)[4]) {
    const uint32_t b = c.lds0 + (uint32_t)(g % P4Cfg<NW>::R) * P4Cfg<NW>::SLOT;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = __builtin_bit_cast(bf16x8, *(LDS_PTR(const uint4))(uintptr_t)(b + c.koff[ks]));
}

// V fragment (key half s, channel half dt) of the tile in ring slot g: two transpose reads
template <int NW>
__device__ __forceinline__ bf16x8 p4_read_v(const P4Ctx<NW>& c, int g, int s, int dt) {
    const uint32_t b = c.lds0 + (uint32_t)(g % P4Cfg<NW>::R) * P4Cfg<NW>::SLOT + TILE_B + (uint32_t)s * 2048u;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(uintptr_t)(b + c.voff[dt][0]));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(uintptr_t)(b + c.voff[dt][1]));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
}

// O^T += V^T P^T.  AG (the four-wave configuration): the accumulators live in the ACCUMULATION registers -- the file is compiled with
// -amdgpu-mfma-vgpr-form (S comes out of the matrix pipe in architectural VGPRs, where the softmax reads it), and a wave with three query tiles
// needs more than the 256 architectural registers, so the 96 O registers (touched by nothing but these MFMAs, a rare rescale and the final store) are
// pinned to the other half of the file through the asm constraint.  hipcc does not know an asm statement is an MFMA: NOPS = wait states in front of
// it (the packed P it reads was written by VALU instructions shortly before).
template <bool AG>
__device__ __forceinline__ void p4_pv(f32x16& o, const bf16x8& a, const bf16x8& b, bool nops) {
    if constexpr (AG) {
        if (nops) asm volatile("s_nop 1\n\t" TCOW_MFMA_32x32x16_H16_ASM " %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));
        else asm volatile(TCOW_MFMA_32x32x16_H16_ASM " %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));
    } else {
        o = TCOW_MFMA_32x32x16_H16(a, b, o, 0, 0, 0);
    }
}
// VALU reads of accumulators written by asm MFMAs: the 16-pass result is complete 18 wait states after issue
template <bool AG> __device__ __forceinline__ void p4_acc_fence() { if constexpr (AG) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory"); }

// ---- one key tile for the wave's NQ query tiles.  The round over the tiles x = 0 .. NQ-1 is two slots of four MFMAs per tile:
//   slot 2x     S_x (this key tile)                          beside softmax pieces 4-7 of tile x-1   (x = 0: of tile NQ-1, PREVIOUS key tile)
//   slot 2x + 1 PV of tile x-1 (x = 0: tile NQ-1, previous)  beside softmax pieces 0-3 of tile x
// so every tile's own chain S -> softmax -> PV is serial, the chains are staggered by two slots, and both pipes always have work.  The only
// branch inside a round is each tile's rescale decision (piece 2), so a scheduling region runs from one decision to the next: 8 MFMAs and one tile
// step's ~54 VALU instructions, which the sched_group_barrier sequence behind every slot deals out as MFMA : 5 VALU : 2 exponentials.
// MODE: FIRST = first key tile of the item (nothing pending, S starts from C = 0, the maximum is set), DRAIN = behind the last one (only the pending
// half of tile NQ-1).  LASTPAD (compile time: the round of the sequence's last key tile is its own instantiation) masks the padding keys.
// V fragments of key tile j are read in slot 2 (free after slot 1's PV of the previous key tile, needed from slot 3), K fragments of key tile
// j + 1 behind the barrier in the last slot.  NQ = 1 has nothing to interleave with: S, softmax, PV in sequence (such a wave shares its SIMD
// with another wave, or waits for the workgroup's three-tile waves anyway).
#ifndef P4_USE_PACE
#define P4_USE_PACE (NW == 4)
#endif
#define P4_PACE()                                                                                                        \
    do {                                                                                                                 \
        if (!(P4_USE_PACE)) break;                                                                                                                 \
        _Pragma("unroll") for (int pace_ = 0; pace_ < 4; ++pace_) {                                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                                                           \
        }                                                                                                                \
    } while (0)
template <int NW, int NQ, int MODE, bool LASTPAD, bool AG, typename F>
__device__ __forceinline__ void p4_round(P4Ctx<NW>& c, P4Tile (&t)[NQ], bf16x8 (&kf)[4], bf16x8 (&vf)[2][2], F&& next_item) {
    constexpr bool CUR = MODE != P4_DRAIN, PREV = MODE != P4_FIRST, FIRST = MODE == P4_FIRST, CT = P4Cfg<NW>::CT;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int g = c.g;
    if constexpr (NQ == 1) {
        if constexpr (CUR) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[0].s = TCOW_MFMA_32x32x16_H16(kf[k], t[0].qf[k], k == 0 ? ((FIRST || !CT) ? zero : t[0].negm) : t[0].s, 0, 0, 0);
                vf[k & 1][k >> 1] = p4_read_v(c, g, k & 1, k >> 1);
            }
            p4_softmax_piece<0, FIRST, LASTPAD, AG, CT>(t[0], c.lr, c.hi); p4_softmax_piece<1, FIRST, LASTPAD, AG, CT>(t[0], c.lr, c.hi);
            p4_softmax_piece<2, FIRST, LASTPAD, AG, CT>(t[0], c.lr, c.hi); p4_softmax_piece<3, FIRST, LASTPAD, AG, CT>(t[0], c.lr, c.hi);
            p4_softmax_piece<4, FIRST, LASTPAD, AG, CT>(t[0], c.lr, c.hi); p4_softmax_piece<5, FIRST, LASTPAD, AG, CT>(t[0], c.lr, c.hi);
            p4_softmax_piece<6, FIRST, LASTPAD, AG, CT>(t[0], c.lr, c.hi); p4_softmax_piece<7, FIRST, LASTPAD, AG, CT>(t[0], c.lr, c.hi);
            const bf16x8 pb0 = p4_pfrag(t[0], 0), pb1 = p4_pfrag(t[0], 1);
            p4_pv<AG>(t[0].o0, vf[0][0], pb0, true); p4_pv<AG>(t[0].o1, vf[0][1], pb0, false);
            p4_pv<AG>(t[0].o0, vf[1][0], pb1, false); p4_pv<AG>(t[0].o1, vf[1][1], pb1, false);
            p4_advance(c, next_item);
            if (c.g < c.total_g) p4_read_k(c, c.g, kf);
        }
    } else {
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
            const int xp = x == 0 ? NQ - 1 : x - 1;                      // the tile whose second half / PV runs in this pair of slots
            const bool have_p = x == 0 ? PREV : CUR;                     // ... and whether it has something pending
            if (!CUR && x > 0) break;
            // ---- slot 2x: S_x
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (CUR && !(P4_AB & 2)) t[x].s = TCOW_MFMA_32x32x16_H16(kf[k], t[x].qf[k], k == 0 ? ((FIRST || !CT) ? zero : t[x].negm) : t[x].s, 0, 0, 0);
                if (have_p) {
                    if (k == 0) p4_softmax_piece<4, false, false, AG, CT>(t[xp], c.lr, c.hi);
                    if (k == 1) p4_softmax_piece<5, false, false, AG, CT>(t[xp], c.lr, c.hi);
                    if (k == 2) p4_softmax_piece<6, false, false, AG, CT>(t[xp], c.lr, c.hi);
                    if (k == 3) p4_softmax_piece<7, false, false, AG, CT>(t[xp], c.lr, c.hi);
                }
                if (CUR && x == 1 && !(P4_AB & 8)) vf[k & 1][k >> 1] = p4_read_v(c, g, k & 1, k >> 1);
            }
            P4_PACE();
            // ---- slot 2x + 1: PV of tile xp, softmax pieces 0-3 of tile x
            bf16x8 pb0, pb1;
            if (have_p) { pb0 = p4_pfrag(t[xp], 0); pb1 = p4_pfrag(t[xp], 1); }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (have_p && !(P4_AB & 4)) {
                    if (k == 0) p4_pv<AG>(t[xp].o0, vf[0][0], pb0, true);
                    if (k == 1) p4_pv<AG>(t[xp].o1, vf[0][1], pb0, false);
                    if (k == 2) p4_pv<AG>(t[xp].o0, vf[1][0], pb1, false);
                    if (k == 3) p4_pv<AG>(t[xp].o1, vf[1][1], pb1, false);
                }
                if (CUR) {
                    if (k == 0) p4_softmax_piece<0, FIRST, LASTPAD, AG, CT>(t[x], c.lr, c.hi);
                    if (k == 1) p4_softmax_piece<1, FIRST, LASTPAD, AG, CT>(t[x], c.lr, c.hi);
                    if (k == 2) p4_softmax_piece<2, FIRST, LASTPAD, AG, CT>(t[x], c.lr, c.hi);
                    if (k == 3) p4_softmax_piece<3, FIRST, LASTPAD, AG, CT>(t[x], c.lr, c.hi);
                    if (x == NQ - 1 && k == 2) p4_advance(c, next_item);                   // barrier of key tile j + 1, request of the tile R - 2 ahead
                    if (x == NQ - 1 && k == 3 && c.g < c.total_g && !(P4_AB & 8)) p4_read_k(c, c.g, kf);   // (kf was last used by S of tile NQ-1, slot 2 NQ - 2)
                }
            }
            P4_PACE();
        }
    }
}

// ---- the same rounds with every hot instruction as its own asm volatile statement: tools/gen_attn_round.py -> attn_round_asm.inc (source order =
// issue order; NW = 8 form: S from C = 0, m subtracted by the softmax; the compiler keeps register allocation and the LDS-read waits)
#ifndef P4A_DBG_MFMA
#define P4A_DBG_MFMA 0
#endif
#ifdef TCOW_FP16
#define PA_CVT_ASM "v_cvt_pk_f16_f32"
#else
#define PA_CVT_ASM "v_cvt_pk_bf16_f32"
#endif
#if P4A_DBG_MFMA
#define PA_MFMA_Z(d, a, b) d = TCOW_MFMA_32x32x16_H16(a, b, (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
#define PA_MFMA(d, a, b) d = TCOW_MFMA_32x32x16_H16(a, b, d, 0, 0, 0)
#define PA_PV_FIRST(d, a, b) d = TCOW_MFMA_32x32x16_H16(a, b, d, 0, 0, 0)
#else
#define PA_MFMA_Z(d, a, b) asm volatile(TCOW_MFMA_32x32x16_H16_ASM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b))
#define PA_MFMA(d, a, b) asm volatile(TCOW_MFMA_32x32x16_H16_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#define PA_PV_FIRST(d, a, b) asm volatile("s_nop 1\n\t" TCOW_MFMA_32x32x16_H16_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#endif
#ifndef P4A_DBG
#define P4A_DBG 0          // bisecting aid: 1 = half-wave maximum in C++, 2 = every VALU statement in C++ (compiler-ordered), 4 = MFMAs as builtins
#endif
#if P4A_DBG & 2
#define PA_MAX3(d, a, b, c_) d = fmaxf(fmaxf(a, b), c_)
#define PA_MAX(d, a, b) d = fmaxf(a, b)
#define PA_SUB(d, a, b) d = (a) - (b)
#define PA_ADD(d, a, b) d = (a) + (b)
#define PA_EXP(d, a) d = __builtin_amdgcn_exp2f(a)
#define PA_CVT(d, a, b) d = pack_bf2(a, b)
#else
#define PA_MAX3(d, a, b, c_) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c_))
#define PA_MAX(d, a, b) asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define PA_SUB(d, a, b) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define PA_ADD(d, a, b) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define PA_EXP(d, a) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a))
#define PA_CVT(d, a, b) asm volatile(PA_CVT_ASM " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#endif
#if P4A_DBG & 3
#define PA_HALFMAX(x) x = half_max(x)
#else
#define PA_HALFMAX(x) do { float hm_; asm volatile("v_mov_b32 %1, %0\n\tv_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1" : "+v"(x), "=&v"(hm_)); } while (0)
#endif
#define PA_NOP1() asm volatile("s_nop 1")
#define PA_NOP11() asm volatile("s_nop 10")
#define PA_PFRAG(T, X) do { X.f0 = p4_pfrag(t[T], 0); X.f1 = p4_pfrag(t[T], 1); } while (0)
#define PA_VREAD(s_, dt_) vf[s_][dt_] = p4_read_v(c, g_, s_, dt_)
#define PA_KREAD() do { if (c.g < c.total_g) p4_read_k(c, c.g, kf); } while (0)
#define PA_ADVANCE() p4_advance(c, next_item)
#define DEC(T, X) do { if (P4A_FIRST || __any(X.mxr > 8.0f)) { TCOW_NO_IFCVT(); p4a_rescale<P4A_FIRST>(t[T], X.mxr); } } while (0)

struct P4AScr {                      // softmax temporaries of one query tile (registers; the second half of tile B's step lives across rounds)
    float t0, t1, t2, t3, t4, u0, u1, mx, mxr, pa, pb;
    float d[16], p[16], sm[16];
    bf16x8 f0, f1;
};

template <bool FIRST>
__device__ __forceinline__ void p4a_rescale(P4Tile& t, float mxr) {
    const float delta = FIRST ? mxr : fmaxf(mxr, 0.0f);
    t.m += delta;
    if (!FIRST) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        t.l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { t.o0[r] *= alpha; t.o1[r] *= alpha; }
    }
}

#include "attn_round_asm.inc"

template <int NW, int NQ, int MODE, bool PAD, typename F>
__device__ __forceinline__ void p4a_round(P4Ctx<NW>& c, P4Tile (&t)[NQ], bf16x8 (&kf)[4], bf16x8 (&vf)[2][2], P4AScr& A, P4AScr& B, const float (&padadd)[16], F&& next_item) {
    constexpr bool P4A_FIRST = MODE == P4_FIRST;
    const int g_ = c.g;
    (void)g_; (void)padadd;
    if constexpr (NQ == 2) {
        if constexpr (MODE == P4_FIRST) P4A_ROUND_NQ2_FIRST();
        else if constexpr (MODE == P4_DRAIN) P4A_ROUND_NQ2_DRAIN();
        else if constexpr (PAD) P4A_ROUND_NQ2_STEADY_PAD();
        else P4A_ROUND_NQ2_STEADY();
    } else {
        static_assert(NQ == 1, "asm rounds: one or two query tiles per wave");
        if constexpr (MODE == P4_FIRST) P4A_ROUND_NQ1_FIRST();
        else if constexpr (PAD) P4A_ROUND_NQ1_STEADY_PAD();
        else P4A_ROUND_NQ1_STEADY();
    }
}

// query tiles of an item that wave w owns: NW = 4: tiles w, w + 4, w + 8 of the chunk; NW = 8: waves 0-3 take tiles w, w + 4, waves 4-7 tile 8 + (w - 4)
// (the SIMD of waves w and w + 4 then carries 3 / 3 / 2 / 2 of ten tiles either way)
template <int NW> __device__ __forceinline__ int p4_nqw(int n, int wave) {
    if (NW == 4) return (n - wave + 3) >> 2;
    if (wave < 4) { const int k = (n - wave + 3) >> 2; return k > 2 ? 2 : (k < 0 ? 0 : k); }
    return n > 4 + wave ? 1 : 0;
}
template <int NW> __device__ __forceinline__ int p4_qtile(int q0, int wave, int x) { return (NW == 8 && wave >= 4) ? q0 + 4 + wave : q0 + wave + 4 * x; }

// request the query tiles of an item into the wave's staging area (4 pieces of 8 rows per tile)
template <int NW>
__device__ __forceinline__ void p4_request_q(const P4Ctx<NW>& c, const SeqDesc& sd, const P4Item& it, const p4_i32x4& srd_q, long pse, uint32_t qst) {
    const int nqw = p4_nqw<NW>(it.n, c.wave);
    const p4_i32x4 srd = p4_uniform(srd_q);
    for (int x = 0; x < nqw; ++x) {
        const int qt = p4_qtile<NW>(it.q0, c.wave, x);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 8 * q + (c.lane >> 3);
            const int ch = (c.lane & 7) ^ swz_g(r);
            const uint32_t vo = (32 * qt + r < sd.L) ? (uint32_t)(((long)r * pse + ch * 8) * 2) : P4_OOB;
            P4_DMA(vo, srd, (uint32_t)qt * c.tile_stride, qst + (uint32_t)x * TILE_B + (uint32_t)q * 1024u);
        }
    }
}

template <int NQ, bool PRE>
__device__ __forceinline__ void p4_load_q(P4Tile (&t)[NQ], uint32_t qst, int l31, int hi) {
#pragma unroll
    for (int x = 0; x < NQ; ++x) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cidx = (2 * ks + hi) ^ swz_g(l31);
            bf16x8 f = __builtin_bit_cast(bf16x8, *(LDS_PTR(const uint4))(uintptr_t)(qst + (uint32_t)x * TILE_B + (uint32_t)(l31 * 128 + (cidx << 4))));
            if (!PRE) {
                typedef __attribute__((ext_vector_type(8))) float f32x8;
                f = __builtin_convertvector(__builtin_convertvector(f, f32x8) * (kScale * kLog2e), bf16x8);
            }
            t[x].qf[ks] = f;
        }
    }
}

// ---- all key tiles of one item for a wave with NQ query tiles, then the stores
template <int NW, int NQ, bool PRE, bool AG, typename F>
__device__ __forceinline__ void p4_run_item(P4Ctx<NW>& c, const SeqDesc& sd, const P4Item& it, const P4Item& nx, const p4_i32x4& srd_qn, long pse, uint32_t qst, uint32_t ost,
                                            bf16x8 (&kf)[4], bf16_t* __restrict__ out, float* __restrict__ lse, F&& next_item) {
    const bool lastpad_seq = c.lr < 32;
    if constexpr (NQ == 0) {
        if (nx.valid) p4_request_q(c, sd, nx, srd_qn, pse, qst);
        for (int j = 0; j < c.nt; ++j) p4_advance(c, next_item);
        if (c.g < c.total_g) p4_read_k(c, c.g, kf);                     // (keeps the wave's view of kf uniform with the others; unused)
    } else {
        P4_STAMP(const long long ti0_ = P4_NOW();)
        P4Tile t[NQ];
        // (this item's Q tiles were requested a whole item ago: with at least R key tiles per item the counted waits of the rounds in between have
        // covered them -- they are older than everything those waits leave in flight; shorter sequences wait here)
        if (c.nt < P4Cfg<NW>::R) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p4_load_q<NQ, PRE>(t, qst, c.l31, c.hi);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the fragments are in registers: the staging area may take the next item's tiles
        if (nx.valid) p4_request_q(c, sd, nx, srd_qn, pse, qst);
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
            t[x].m = 0.f; t[x].l = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[x].o0[r] = 0.f; t[x].o1[r] = 0.f; if constexpr (P4Cfg<NW>::CT) t[x].negm[r] = 0.f; }
        }
        bf16x8 vf[2][2];
        P4_STAMP(const long long ti1_ = P4_NOW(); c.t_pro += ti1_ - ti0_;)
        // (sequences of at least two key tiles: tcow_attn_fwd_p4_ok; the last key tile's round masks the padding keys when L % 32 != 0)
        if constexpr (NW == 8 && P4_ASM_ROUNDS) {
            P4AScr A, B;
            float padadd[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) padadd[r] = (crow32(r, c.hi) >= c.lr) ? -1e30f : 0.0f;
            p4a_round<NW, NQ, P4_FIRST, false>(c, t, kf, vf, A, B, padadd, next_item);
#pragma unroll 1
            for (int j = 1; j < c.nt - 1; ++j) p4a_round<NW, NQ, P4_STEADY, false>(c, t, kf, vf, A, B, padadd, next_item);
            if (lastpad_seq) p4a_round<NW, NQ, P4_STEADY, true>(c, t, kf, vf, A, B, padadd, next_item);
            else p4a_round<NW, NQ, P4_STEADY, false>(c, t, kf, vf, A, B, padadd, next_item);
            if constexpr (NQ > 1) p4a_round<NW, NQ, P4_DRAIN, false>(c, t, kf, vf, A, B, padadd, next_item);
        } else {
            p4_round<NW, NQ, P4_FIRST, false, AG>(c, t, kf, vf, next_item);
#pragma unroll 1
            for (int j = 1; j < c.nt - 1; ++j) p4_round<NW, NQ, P4_STEADY, false, AG>(c, t, kf, vf, next_item);
            if (lastpad_seq) p4_round<NW, NQ, P4_STEADY, true, AG>(c, t, kf, vf, next_item);
            else p4_round<NW, NQ, P4_STEADY, false, AG>(c, t, kf, vf, next_item);
            if constexpr (NQ > 1) p4_round<NW, NQ, P4_DRAIN, false, AG>(c, t, kf, vf, next_item);
        }
        // ---- normalise, store whole rows through the staging tile, log-sum-exp
        P4_STAMP(const long long ti2_ = P4_NOW(); c.t_rounds += ti2_ - ti1_;)
        p4_acc_fence<AG>();
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
            const int q0 = 32 * p4_qtile<NW>(it.q0, c.wave, x);
            const float l = half_sum(t[x].l);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            store_tile_staged(ost, c.lane, 1.0f / l, t[x].o0, t[x].o1, out + (it.base + (long)q0 * sd.pos_stride) * sd.D + it.head * ATT_HD, sd.pos_stride * (long)sd.D, sd.L - q0);
            const int q = q0 + c.l31;
            if (lse && c.hi == 0 && q < sd.L) lse[(it.base + (long)q * sd.pos_stride) * sd.heads + it.head] = (t[x].m + log2f(l)) * 0.6931471805599453f;
        }
        P4_STAMP(c.t_epi += P4_NOW() - ti2_;)
    }
}

template <int NW, bool PRE>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_fwd_p4(SeqDesc sd, int nt, int nchunk, int per, const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse) {
    typedef P4Cfg<NW> C;
    constexpr bool AG = NW == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    P4_STAMP(const long long tk0_ = P4_NOW();)
    const int pairs = sd.n_outer * sd.n_inner * sd.heads;
    const int vtot = 8 * ((pairs + 7) / 8) * nchunk;
    const int G = gridDim.x;
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const uint32_t seq_bytes = (uint32_t)((((long)sd.L - 1) * pse + ATT_HD) * 2);

    // this workgroup's items: v = blockIdx.x, + G, ... (the invalid ones -- pair index past the end -- are skipped by everybody alike)
    int n_items = 0;
    for (int v = blockIdx.x; v < vtot; v += G) n_items += p4_item(sd, v, pairs, nchunk, per, nt).valid;
    if (n_items == 0) return;

    P4Ctx<NW> c;
    c.lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
    c.wave = wave; c.lane = lane; c.hi = lane >> 5; c.l31 = lane & 31;
    c.nt = nt; c.lr = sd.L - 32 * (nt - 1);
    c.tile_stride = (uint32_t)(32 * pse * 2);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) c.koff[ks] = (uint32_t)(c.l31 * 128 + (((2 * ks + c.hi) ^ swz_g(c.l31)) << 4));
    {
        const int q16 = lane & 15, g16 = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int chunk = 4 * dt + 2 * g16 + ((q16 & 3) >> 1);
            const int r0 = 4 * c.hi + (q16 >> 2), r1 = r0 + 8;
            c.voff[dt][0] = (uint32_t)(r0 * 128 + ((chunk ^ swz_g(r0)) << 4) + (q16 & 1) * 8);
            c.voff[dt][1] = (uint32_t)(r1 * 128 + ((chunk ^ swz_g(r1)) << 4) + (q16 & 1) * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < C::PIECES; ++i) {
        const int piece = wave * C::PIECES + i;                         // 0-3: K rows 8 piece .. + 7, 4-7: V rows
        const int r = 8 * (piece & 3) + (lane >> 3);
        const int ch = (lane & 7) ^ swz_g(r);
        c.vo[i] = (uint32_t)(((long)r * pse + ch * 8) * 2);
        c.vo_last[i] = (r < c.lr) ? c.vo[i] : P4_OOB;
    }
    const bool wave_loads_v = wave * C::PIECES >= 4;
    const uint32_t qst = c.lds0 + C::RING + (uint32_t)wave * C::QST;
    const uint32_t ost = c.lds0 + C::RING + NW * C::QST + (uint32_t)wave * TILE_B;

    // the producer walks the same item list as the consumer, R - 2 tiles ahead
    auto set_producer = [&](P4Ctx<NW>& cc) {
        const P4Item it = p4_item(sd, cc.pv, pairs, nchunk, per, nt);
        const bf16_t* qh = qkv + it.base * ld3 + it.head * ATT_HD;
        cc.srd_kv = p4_srd(qh + (wave_loads_v ? 2 : 1) * sd.D, seq_bytes);
    };
    auto next_valid = [&](int v) {
        v += G;
        while (v < vtot && !p4_item(sd, v, pairs, nchunk, per, nt).valid) v += G;
        return v;
    };
    auto next_item = [&](P4Ctx<NW>& cc) {
        cc.pv = next_valid(cc.pv);
        if (cc.pv < vtot) set_producer(cc);
    };
    int cv = blockIdx.x;
    if (!p4_item(sd, cv, pairs, nchunk, per, nt).valid) cv = next_valid(cv);
    c.pv = cv; c.pj = 0; c.pg = 0; c.g = 0; c.total_g = n_items * nt;
    set_producer(c);

    // ---- prologue: Q of the first item, then the first R - 1 tiles of the stream; the first barrier publishes tile 0
    P4Item it = p4_item(sd, cv, pairs, nchunk, per, nt);
    {
        const p4_i32x4 srd_q = p4_srd(qkv + it.base * ld3 + it.head * ATT_HD, seq_bytes);
        p4_request_q(c, sd, it, srd_q, pse, qst);
    }
#pragma unroll 1
    for (int i = 0; i < C::R - 1; ++i) p4_produce(c, next_item);
    if (c.pg - 1 >= C::R - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::PIECES * (C::R - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 kf[4];
    p4_read_k(c, 0, kf);

    // ---- items
#pragma unroll 1
    for (int k = 0; k < n_items; ++k) {
        const int nv = next_valid(cv);
        P4Item nx; nx.valid = 0; nx.base = 0; nx.head = 0; nx.q0 = 0; nx.n = 0;
        if (nv < vtot) nx = p4_item(sd, nv, pairs, nchunk, per, nt);
        const p4_i32x4 srd_qn = p4_srd(qkv + nx.base * ld3 + nx.head * ATT_HD, seq_bytes);
        const int nqw = p4_nqw<NW>(it.n, wave);
        if (C::NQ >= 3 && nqw >= 3) p4_run_item<NW, C::NQ >= 3 ? 3 : 1, PRE, AG>(c, sd, it, nx, srd_qn, pse, qst, ost, kf, out, lse, next_item);
        else if (nqw == 2) p4_run_item<NW, 2, PRE, AG>(c, sd, it, nx, srd_qn, pse, qst, ost, kf, out, lse, next_item);
        else if (nqw == 1) p4_run_item<NW, 1, PRE, AG>(c, sd, it, nx, srd_qn, pse, qst, ost, kf, out, lse, next_item);
        else p4_run_item<NW, 0, PRE, AG>(c, sd, it, nx, srd_qn, pse, qst, ost, kf, out, lse, next_item);
        it = nx; cv = nv;
    }
#ifdef P4_STAMPS
    if (g_p4_dbg && lane == 0) {
        long long* d = g_p4_dbg + ((long)blockIdx.x * NW + wave) * 8;
        d[0] = P4_NOW() - tk0_; d[1] = c.t_vm; d[2] = c.t_wait; d[3] = c.t_pro; d[4] = c.t_rounds; d[5] = c.t_epi; d[6] = n_items; d[7] = c.total_g;
    }
#endif
}

}  // namespace

// sequences this kernel takes: no causal mask, at least two key tiles, byte offsets inside a sequence below 2^31
bool tcow_attn_fwd_p4_ok(const SeqDesc& d) {
    const int nt = (d.L + 31) / 32;
    const long pse = d.pos_stride * 3L * d.D;
    return d.diag >= (1 << 27) && nt >= 2 && (long)d.L * pse * 2 < (1L << 31);
}

// waves: 4 (one per SIMD, <= 3 query tiles each) or 8 (two per SIMD, <= 2 each); prescaled: the q section of qkv already carries 0.125 log2(e)
int tcow_attn_fwd_p4(hipStream_t st, const SeqDesc& d, const void* qkv, void* out, float* lse, int waves, int prescaled) {
    const int nt = (d.L + 31) / 32;
    const int pairs = d.n_outer * d.n_inner * d.heads;
    const int nchunk = (nt + P4_ITEM_TILES - 1) / P4_ITEM_TILES;
    const int per = (nt + nchunk - 1) / nchunk;
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int vtot = 8 * ((pairs + 7) / 8) * nchunk;
    int grid = (cus / 8) * 8; if (grid < 8) grid = 8;
    if (grid > vtot) grid = vtot;
    typedef void (*Kern)(SeqDesc, int, int, int, const bf16_t*, bf16_t*, float*);
    const Kern k = waves == 8 ? (prescaled ? attn_fwd_p4<8, true> : attn_fwd_p4<8, false>) : (prescaled ? attn_fwd_p4<4, true> : attn_fwd_p4<4, false>);
    const int lds = waves == 8 ? P4Cfg<8>::LDS : P4Cfg<4>::LDS;
    tcow_ensure_lds(reinterpret_cast<const void*>(k), lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(waves == 8 ? 512 : 256), lds, st, d, nt, nchunk, per, (const bf16_t*)qkv, (bf16_t*)out, lse);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
