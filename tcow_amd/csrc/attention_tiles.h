// Tile helpers shared by the 16-bit MFMA attention kernels (attention_bf16.hip, attention_fwd_p4.hip): the [32 rows][64 x 16-bit] LDS tile with its
// universal swizzle, direct-to-LDS tile loads, MFMA operand fragments (row fragments / transpose reads), the half-wave exchange, the forward
// tile step of the streaming kernels and the whole-row store of a transposed accumulator pair.  Everything lives in an anonymous namespace:
// each translation unit gets its own copy.
#pragma once
#include "attention_common.h"

#ifdef TCOW_FP16
#define TCOW_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
#define TCOW_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

namespace {

constexpr int TILE_B = 4096;                 // 32 rows x 128 B
#ifdef UBENCH_ATTN
__device__ long long* g_attn_dbg;
#endif
constexpr float kScale = 0.125f;             // head_dim^-0.5 (vit.py:70)
constexpr float kLog2e = 1.4426950408889634f;

// Combine a value with the other half-wave's (lane ^ 32) without an LDS round trip (__shfl_xor lowers to ds_bpermute): v_permlane32_swap
// exchanges the upper half of its first operand with the lower half of its second, so two copies of v become (lower, lower) and
// (upper, upper).  Inline asm: with identical operands hipcc 7.2 folds the builtin's two results into one (wrong values, caught by the
// oracle tests); the two v_nop are the VALU-write -> permlane-read wait states the compiler does not insert inside an asm statement.
__device__ __forceinline__ void half_swap(float& a, float& b) { asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float half_max(float v) { float a = v, b = v; half_swap(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum(float v) { float a = v, b = v; half_swap(a, b); return a + b; }

// hipcc if-converts a cheap `if (wave_uniform_flag) { selects }` block into unconditional compare / select code -- for the mask evaluation
// of the tile functions below that was ~64 extra VALU instructions per step, half the VALU work of an interior tile on VALU-bound kernels
// (profiles/r02_pmc_attn.txt).  An empty volatile asm statement cannot be speculated, so the block keeps its (scalar) branch.
#define TCOW_NO_IFCVT() asm volatile("" ::: "memory")

__device__ __forceinline__ int swz_g(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// AUX = cache policy bits of the load: 0 plain, LD_NT = non-temporal.  The backward kernels whose workgroup (or wave) is the only reader of its
// (frame, head)'s q / k / v / dO -- attn_bwd_one_kernel, attn_bwd_one_tile -- load them non-temporally: tiles written a whole forward pass ago and never
// read again should not displace what the neighbouring kernels share through the L2s (-0.14 ms per step together with LayerNorm backward's saved
// input, profiles/r05_nontemporal.txt).  The forward kernels keep plain loads: their K / V tiles are shared by the workgroups of a (frame, head).
constexpr int LD_NT = 2;
template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((GLB_PTR(const uint32_t))gsrc, (LDS_PTR(uint32_t))lds_wave_base, 16, 0, AUX);
}

// one wave loads a [32][64] bf16 tile: positions p0..p0+31 (clamped to L-1 so that padding rows hold finite data)
template <int AUX = 0>
__device__ __forceinline__ void load_tile(const bf16_t* src, long pse, int p0, int L, char* tile, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ swz_g(r);
        int pos = p0 + r; pos = pos < L ? pos : L - 1;
        glds16<AUX>(src + (size_t)pos * pse + c * 8, tile + j * 1024);
    }
}

// rows 8c .. 8c+7 of such a tile (a quarter: one direct-to-LDS instruction)
template <int AUX = 0>
__device__ __forceinline__ void load_tile_chunk(const bf16_t* src, long pse, int p0, int L, char* tile, int c, int lane) {
    const int r = 8 * c + (lane >> 3);
    const int ch = (lane & 7) ^ swz_g(r);
    int pos = p0 + r; pos = pos < L ? pos : L - 1;
    glds16<AUX>(src + (size_t)pos * pse + ch * 8, tile + c * 1024);
}

// Streaming kernels: tiles c0 .. c0+CH-1 of two row sources into the LDS tile arrays ta / tb.  Wave w of the 4-wave workgroup brings tile c0+w;
// with CH = 5 the fifth tile comes in quarters, one per wave (nt = 10 -- S = 301 -- then needs two chunk rounds instead of three).
template <int CH>
__device__ __forceinline__ void load_chunk2(const bf16_t* a, long psa, const bf16_t* b, long psb, int c0, int nt, int L, char* ta, char* tb, int wave, int lane) {
    if (c0 + wave < nt) {
        load_tile(a, psa, 32 * (c0 + wave), L, ta + wave * TILE_B, lane);
        load_tile(b, psb, 32 * (c0 + wave), L, tb + wave * TILE_B, lane);
    }
    if (CH == 5 && c0 + 4 < nt) {
        load_tile_chunk(a, psa, 32 * (c0 + 4), L, ta + 4 * TILE_B, wave, lane);
        load_tile_chunk(b, psb, 32 * (c0 + 4), L, tb + 4 * TILE_B, wave, lane);
    }
}

// A/B fragment of a row-major tile for a contraction over d: lane (row = lane&31, hi) gets d = 16*ks + 8*hi .. +7
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int row, int ks, int hi) {
    const int c = (2 * ks + hi) ^ swz_g(row);
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tile + row * 128 + (c << 4)));
}
// the same fragment straight from global memory (rows at stride pse elements)
__device__ __forceinline__ bf16x8 frag_row_global(const bf16_t* src, long pse, int pos, int ks, int hi) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(src + (size_t)pos * pse + 16 * ks + 8 * hi));
}
// fragment for a contraction over the tile's ROWS: lane (col = 32*dt + (lane&31), hi) gets rows crow32(8*s + j, hi), j = 0..7,
// i.e. rows {16s + 4hi + 0..3} and {16s + 8 + 4hi + 0..3}: two transpose reads of [4 rows][16 cols] blocks.
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int s, int dt, int lane) {
    const int q16 = lane & 15, g16 = (lane >> 4) & 1, hi = lane >> 5;
    const int chunk = 4 * dt + 2 * g16 + ((q16 & 3) >> 1);
    const int r0 = 16 * s + 4 * hi + (q16 >> 2), r1 = r0 + 8;
    const int o0 = r0 * 128 + ((chunk ^ swz_g(r0)) << 4) + (q16 & 1) * 8;
    const int o1 = r1 * 128 + ((chunk ^ swz_g(r1)) << 4) + (q16 & 1) * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + o0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + o1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
}
// registers 8s..8s+7 of a C-layout accumulator as a bf16 operand (contraction slot j <-> row crow32(8s+j, hi))
__device__ __forceinline__ bf16x8 pack8(const float* v) {
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    f32x8 t = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    return __builtin_convertvector(t, bf16x8);
}

struct WorkId { int item, head; bool valid; };
template <bool SHARED>
__device__ __forceinline__ WorkId work_id(const SeqDesc& sd, int wave) {
    const int items = sd.n_outer * sd.n_inner;
    const int idx = SHARED ? blockIdx.x : blockIdx.x * 4 + wave;
    WorkId w; w.valid = idx < items * sd.heads; w.item = idx / sd.heads; w.head = idx - w.item * sd.heads;
    return w;
}

// One 32-key tile against one 32-query tile (forward): S^T = K Q^T, online softmax, O^T += V^T P^T.
// VALU budget: the MFMAs of a step take 256 cycles per wave, the softmax arithmetic used to take ~850, so every operation counts:
// the score scale is folded into the exp2 argument (one fma per element), masks are only evaluated on boundary tiles, masked
// elements rely on exp2 underflow (no select), and the running maximum is LAZY: accumulators are rescaled only when some row's
// maximum grew by more than 2^8 since the reference maximum was set (probabilities then stay <= 256, harmless in bf16 / f32, and
// the final O / l and log-sum-exp are independent of the reference) -- after the first key tile that almost never happens.
__device__ __forceinline__ void fwd_tile(const SeqDesc& sd, const char* ktile, const char* vtile, const bf16x8 (&qf)[4], int j, int qt, int q, int l31, int hi, int lane,
                                         float& m, float& l, f32x16& o0, f32x16& o1) {
    const float sc = kScale * kLog2e;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s = TCOW_MFMA_32x32x16_H16(frag_row(ktile, l31, ks, hi), qf[ks], s, 0, 0, 0);
    const bool need_mask = (32 * j + 31 >= sd.L) || ((long)32 * j + 31 > (long)32 * qt + sd.diag);
    if (need_mask) {
        TCOW_NO_IFCVT();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * j + crow32(r, hi);
            if (key >= sd.L || (long)key > (long)q + sd.diag) s[r] = -1e30f;
        }
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = half_max(mx) * sc;                                 // (sc > 0: the maximum commutes with the scale)
    if (__any(mx > m + 8.0f)) {
        const float mn = fmaxf(m, mx);
        const float alpha = exp2f(m - mn);
        l *= alpha;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    float p[16];
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m)); ps += p[r]; }
    if (need_mask) {                                        // a fully masked row (m still at its start value) must contribute nothing
        TCOW_NO_IFCVT();
        ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { if (s[r] <= -1e29f) p[r] = 0.f; ps += p[r]; }
    }
    l += ps;
    const bf16x8 pb0 = pack8(p), pb1 = pack8(p + 8);
    o0 = TCOW_MFMA_32x32x16_H16(frag_tr(vtile, 0, 0, lane), pb0, o0, 0, 0, 0);
    o0 = TCOW_MFMA_32x32x16_H16(frag_tr(vtile, 1, 0, lane), pb1, o0, 0, 0, 0);
    o1 = TCOW_MFMA_32x32x16_H16(frag_tr(vtile, 0, 1, lane), pb0, o1, 0, 0, 0);
    o1 = TCOW_MFMA_32x32x16_H16(frag_tr(vtile, 1, 1, lane), pb1, o1, 0, 0, 0);
}

// normalise and store one query tile's output (+ log-sum-exp)
__device__ __forceinline__ void fwd_store(const SeqDesc& sd, long base, int head, int q, int hi, float m, float l, const f32x16& o0, const f32x16& o1,
                                          bf16_t* __restrict__ out, float* __restrict__ lse) {
    l = half_sum(l);
    if (q < sd.L) {
        const float inv = 1.0f / l;
        const long row = base + (long)q * sd.pos_stride;
        bf16_t* orow = out + row * sd.D + head * ATT_HD;
        // O^T C layout: register r of lane (q, hi) holds d = 32*dt + 8*(r>>2) + 4*hi + (r&3): 4 consecutive d per group
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            st4(orow + 8 * gq + 4 * hi, make_float4(o0[4 * gq] * inv, o0[4 * gq + 1] * inv, o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv));
            st4(orow + 32 + 8 * gq + 4 * hi, make_float4(o1[4 * gq] * inv, o1[4 * gq + 1] * inv, o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv));
        }
        if (lse && hi == 0) lse[row * sd.heads + head] = (m + log2f(l)) * 0.6931471805599453f;   // natural-log LSE of the scaled scores
    }
}

// A 32 x 64 tile held as two TRANSPOSED accumulators (lane (row l31, hi): channels 32 dt + 8 g + 4 hi + 0..3 of register quad g) -> global
// rows, through a wave-private 4 KiB LDS tile: after the round trip 8 lanes hold one row's 128 bytes, i.e. every store instruction
// writes 8 whole cache lines instead of 32 rows x 8 (or 32) bytes.  The row-per-lane stores kept the memory pipe busy for thousands of
// cycles per tile (timeline in profiles/r03_ubench_valu.txt part D) and the next loads queued behind them.  One v_permlane32_swap per
// dword first pairs the half-waves' 8-byte pieces into 16-byte ones (guide T21); LDS chunk c of row r sits at c ^ (r & 7); inline-asm
// LDS accesses (hipcc would wait vmcnt(0) for direct-to-LDS loads that may still be in flight elsewhere in the kernel).
typedef uint32_t stg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_tile_staged(uint32_t stage, int lane, float mul, const f32x16& a0, const f32x16& a1, bf16_t* __restrict__ dst, long row_stride,
                                                  int rows_valid) {
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const f32x16& o = dt ? a1 : a0;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const int g = 2 * gp;
            uint32_t x0 = pack_bf2(o[4 * g] * mul, o[4 * g + 1] * mul), x1 = pack_bf2(o[4 * g + 2] * mul, o[4 * g + 3] * mul);
            uint32_t y0 = pack_bf2(o[4 * g + 4] * mul, o[4 * g + 5] * mul), y1 = pack_bf2(o[4 * g + 6] * mul, o[4 * g + 7] * mul);
            asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(y0));
            asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x1), "+v"(y1));
            const stg_u32x4 w = {x0, x1, y0, y1};                    // channels 32 dt + 16 gp + 8 hi + 0..7 of row l31
            const uint32_t a = stage + l31 * 128 + (((4 * dt + 2 * gp + hi) ^ (l31 & 7)) << 4);
            asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(w) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stg_u32x4 rd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3);
        const uint32_t a = stage + r * 128 + (((lane & 7) ^ (r & 7)) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(rd[i]) : "v"(a) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rd[0]), "+v"(rd[1]), "+v"(rd[2]), "+v"(rd[3]) :: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3);
        if (r < rows_valid) *reinterpret_cast<stg_u32x4*>(dst + (long)r * row_stride + (lane & 7) * 8) = rd[i];
    }
}

// normalise + store one query tile (+ log-sum-exp) as whole rows through the wave-private LDS tile `stage`
__device__ __forceinline__ void fwd_store_rows(const SeqDesc& sd, long base, int head, int q0, int lane, float m, float l, const f32x16& o0, const f32x16& o1,
                                               uint32_t stage, bf16_t* __restrict__ out, float* __restrict__ lse) {
    const int l31 = lane & 31, hi = lane >> 5;
    l = half_sum(l);
    store_tile_staged(stage, lane, 1.0f / l, o0, o1, out + (base + (long)q0 * sd.pos_stride) * sd.D + head * ATT_HD, sd.pos_stride * (long)sd.D, sd.L - q0);
    const int q = q0 + l31;
    if (lse && hi == 0 && q < sd.L) lse[(base + (long)q * sd.pos_stride) * sd.heads + head] = (m + log2f(l)) * 0.6931471805599453f;
}

}  // namespace
