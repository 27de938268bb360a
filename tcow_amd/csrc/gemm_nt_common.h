// Shared pieces of the bf16 NT GEMM kernels (gemm_bf16.hip: 128 / 256 / 320-row tiles, one workgroup per CU for the big ones; gemm_nt_c2.hip: the
// 160 x 256 tile that runs two workgroups per CU): launch parameters, the XCD-aware tile order, the direct-to-LDS load, the fused epilogue
// arithmetic (bias / DropPath row scale / GELU / GELU' / residual / second masked bias; vit.py:50-61,74-76,111,146) and the epilogue of one
// wave's 160 x 64 accumulator tile.
#pragma once
#include "common.h"

#ifdef TCOW_FP16
#define TCOW_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
#define TCOW_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

namespace {

struct NtParams {
    int M, N, K;
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    void* C; long ldc; int out_f32;
    const float* bias;
    const float* row_scale;
    const float* resid; long ldr;
    int act;
    bf16_t* aux; long ldaux;
    int tiles_m, tiles_n;
    const float* bias2; const float* row_scale2;      // second bias with its own row scale (the folded temporal projection), or NULL
    int band;                                         // tile order: 0 = row-major over (row tile, column tile); b > 0 = column bands of b tiles (nt_tile_of)
};

// the launch parameters of a validated tcow_gemm_nt call (tile counts are filled by the kernel's launcher)
static inline NtParams nt_params_from_args(const tcow_gemm_args* a) {
    NtParams p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = (const bf16_t*)a->A; p.lda = a->lda; p.W = (const bf16_t*)a->W; p.ldw = a->ldw;
    p.C = a->C; p.ldc = a->ldc; p.out_f32 = a->out_f32; p.bias = a->bias; p.row_scale = a->row_scale;
    p.resid = a->resid; p.ldr = a->ldr; p.act = a->act; p.aux = (bf16_t*)a->aux; p.ldaux = a->ldaux;
    p.bias2 = a->bias2; p.row_scale2 = a->row_scale2;
    p.tiles_m = p.tiles_n = 0; p.band = 0;
    return p;
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // Blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous chunk of tile ids.
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// Tile of a (remapped) workgroup id.  band = 0: row-major -- an XCD's contiguous range of ids is a band of row tiles with ALL column tiles: its A rows
// are fetched once, all of W passes through its L2 for every few row tiles (fine while W fits beside the A stream: N <= 2304 at K = 768).  band = b > 0:
// the column tiles are walked in bands of b (tiles_n % b == 0): ids run over (band, row tile, column inside the band), so an XCD's range is a row
// range x one band -- it keeps b / tiles_n of W resident and tiles_n / b XCDs read the same A rows (from the Infinity Cache after the first).
__device__ __forceinline__ void nt_tile_of(int pid, int tiles_m, int tiles_n, int band, int& pm, int& pn) {
    if (band <= 0) { pm = pid / tiles_n; pn = pid - pm * tiles_n; return; }
    const int per = tiles_m * band, j = pid / per, rem = pid - j * per;
    pm = rem / band; pn = j * band + (rem - pm * band);
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((GLB_PTR(const uint32_t))gsrc, (LDS_PTR(uint32_t))lds_wave_base, 16, 0, 0);
}

// Epilogue: accumulators -> LDS (f32) -> coalesced row-wise stores.  The row loop is deliberately NOT unrolled and the erf-based
// activations are out-of-line calls: a fully unrolled epilogue with erff inlined 128x is ~160 KB of straight-line code per kernel
// and runs at instruction-fetch speed (measured: 80 us of a 230 us GEMM).  Row-dependent operands (row scale, residual or GELU'
// input) of row group it+1 are requested before row group it is processed so their latency overlaps.
// bf16 mode: Phi(x) = (1 + erf(x / sqrt 2)) / 2 with Abramowitz & Stegun 7.1.26, erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),
// t = 1 / (1 + p z), |error| <= 1.5e-7 -- four orders of magnitude below the bf16 rounding of the result, and the exponential
// is exp(-x^2 / 2), i.e. the normal density GELU' needs anyway.  ~20 VALU operations per element where erff + expf take ~75
// (measured: the erff epilogues added 120 us (GELU) and 200 us (GELU') to a 170 us fc1-shaped GEMM).
// The arithmetic runs on PAIRS of elements with gfx950's packed-f32 instructions (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth of f32 work per issue
// slot): per pair 14 packed operations + 2 x (|x| scale, rcp, exp2, sign insert) -- 16 issue slots per element instead of 22.  Same operation sequence
// per element as the scalar form (each fma / mul / add is the same IEEE operation): identical results.  (Round 5 measured this as time-neutral: the
// GELU GEMMs are bound by their store streams, not by VALU issue -- see the cache policy note below; kept for the lower instruction count.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_set(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& pdf) {
    const f32x2 z = {fabsf(x.x) * 0.70710678118654752440f, fabsf(x.y) * 0.70710678118654752440f};
    const f32x2 den = pk_fma(pk_set(0.3275911f), z, pk_set(1.0f));
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f32x2 nzz = (-z) * z;
    const f32x2 e = {__expf(nzz.x), __expf(nzz.y)};
    const f32x2 poly = t * pk_fma(t, pk_fma(t, pk_fma(t, pk_fma(t, pk_set(1.061405429f), pk_set(-1.453152027f)), pk_set(1.421413741f)), pk_set(-0.284496736f)), pk_set(0.254829592f));
    const f32x2 erfabs = pk_fma(-poly, e, pk_set(1.0f));
    const f32x2 erfs = {copysignf(erfabs.x, x.x), copysignf(erfabs.y, x.y)};
    cdf = pk_fma(pk_set(0.5f), erfs, pk_set(0.5f));
    pdf = pk_set(0.39894228040143267794f) * e;
}
__device__ __forceinline__ float4 gelu4(float4 v) {
    f32x2 c0, q0, c1, q1; const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    gelu_parts2(a, c0, q0); gelu_parts2(b, c1, q1);
    const f32x2 g0 = a * c0, g1 = b * c1;
    return make_float4(g0.x, g0.y, g1.x, g1.y);
}
__device__ __forceinline__ float4 dgelu4(float4 v, float4 x) {
    f32x2 c0, q0, c1, q1; const f32x2 a = {x.x, x.y}, b = {x.z, x.w};
    gelu_parts2(a, c0, q0); gelu_parts2(b, c1, q1);
    const f32x2 d0 = (f32x2){v.x, v.y} * pk_fma(a, q0, c0), d1 = (f32x2){v.z, v.w} * pk_fma(b, q1, c1);
    return make_float4(d0.x, d0.y, d1.x, d1.y);
}
// GELU and GELU' of the same argument (one erf / exp for both)
__device__ __forceinline__ void gelu_both4(float4 v, float4& g, float4& d) {
    f32x2 c0, q0, c1, q1; const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    gelu_parts2(a, c0, q0); gelu_parts2(b, c1, q1);
    const f32x2 g0 = a * c0, g1 = b * c1, d0 = pk_fma(a, q0, c0), d1 = pk_fma(b, q1, c1);
    g = make_float4(g0.x, g0.y, g1.x, g1.y); d = make_float4(d0.x, d0.y, d1.x, d1.y);
}

// Cache policy.  The GELU' tile fc1's forward epilogue saves (166 MB at configs[1]) is read exactly once, a whole backward pass later: it is written with
// NON-TEMPORAL stores and read (x GELU' epilogue) with non-temporal loads, so that it does not push the operand tiles the workgroups of a GEMM share out
// of the 4 MB L2s: fc1 -> fc2 pair 354 -> 343 us, fc2-gradient -> fc1-gradient pair 269 -> 260 us.  Every other stream of the epilogues (16-bit output
// tiles by epilogue kind, f32 outputs, residual loads) measured neutral or slower that way -- the next kernel re-reads them through the Infinity
// Cache (profiles/r05_nontemporal.txt).

struct EpiRow { float4 ext; float rs, rs2; };

// Epilogue configuration: ACT / ROWS < 0 = decided at run time from NtParams (the generic kernels); >= 0 = compile-time constants
// (ROWS bit 0 = row_scale present, bit 1 = resid present, bit 2 = bias2 / row_scale2 present).  The specialised instantiations keep the epilogue of the 320-tile
// kernel small: with every activation inlined behind run-time branches its unrolled row loops were ~100 KB of code.
template <int ACT, int ROWS> struct EpiCfg {
    static constexpr bool kStatic = ACT >= 0 && ROWS >= 0;
    static constexpr bool kRowOps = ROWS > 0 || ACT == TCOW_ACT_DGELU || ACT == TCOW_ACT_MUL_AUX;
    static constexpr bool kAuxOnly = ROWS == 0 && ACT == TCOW_ACT_MUL_AUX;     // the only row operand is the 16-bit aux tile (fc2's input gradient x GELU')
    static __device__ __forceinline__ int act(const NtParams& p) { return ACT < 0 ? p.act : ACT; }
    static __device__ __forceinline__ bool rs(const NtParams& p) { return ROWS < 0 ? p.row_scale != nullptr : (ROWS & 1) != 0; }
    static __device__ __forceinline__ bool res(const NtParams& p) { return ROWS < 0 ? p.resid != nullptr : (ROWS & 2) != 0; }
    static __device__ __forceinline__ bool b2(const NtParams& p) { return ROWS < 0 ? p.bias2 != nullptr : (ROWS & 4) != 0; }
};
typedef EpiCfg<-1, -1> EpiAny;

template <typename E = EpiAny>
__device__ __forceinline__ EpiRow epi_row_fetch(const NtParams& p, int gm, int gn, bool ok) {
    EpiRow o; o.ext = make_float4(0.f, 0.f, 0.f, 0.f); o.rs = 1.0f; o.rs2 = 1.0f;
    if (ok && gm < p.M) {
        if (E::rs(p)) o.rs = p.row_scale[gm];
        if (E::b2(p) && p.row_scale2) o.rs2 = p.row_scale2[gm];
        if (E::res(p)) o.ext = ld4(p.resid + (size_t)gm * p.ldr + gn);
        else if (E::act(p) == TCOW_ACT_DGELU || E::act(p) == TCOW_ACT_MUL_AUX) o.ext = ld4(p.aux + (size_t)gm * p.ldaux + gn);
    }
    return o;
}

template <typename E = EpiAny>
__device__ __forceinline__ void epi_row_apply(const NtParams& p, const EpiRow& o, float4 v, float4 b4, int gm, int gn) {
    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    if (E::rs(p)) { v.x *= o.rs; v.y *= o.rs; v.z *= o.rs; v.w *= o.rs; }
    const int act = E::act(p);
    if (act == TCOW_ACT_GELU) {
        if (p.aux) st4(p.aux + (size_t)gm * p.ldaux + gn, v);
        v = gelu4(v);
    } else if (act == TCOW_ACT_DGELU) {
        v = dgelu4(v, o.ext);
    } else if (act == TCOW_ACT_GELU_DSAVE) {
        float4 g, d; gelu_both4(v, g, d);
        st4(p.aux + (size_t)gm * p.ldaux + gn, d);
        v = g;
    } else if (act == TCOW_ACT_MUL_AUX) {
        v.x *= o.ext.x; v.y *= o.ext.y; v.z *= o.ext.z; v.w *= o.ext.w;
    }
    if (E::b2(p)) { const float4 c4 = ld4(p.bias2 + gn); v.x = fmaf(o.rs2, c4.x, v.x); v.y = fmaf(o.rs2, c4.y, v.y); v.z = fmaf(o.rs2, c4.z, v.z); v.w = fmaf(o.rs2, c4.w, v.w); }
    if (E::res(p)) { v.x += o.ext.x; v.y += o.ext.y; v.z += o.ext.z; v.w += o.ext.w; }
    if (p.out_f32) st4(reinterpret_cast<float*>(p.C) + (size_t)gm * p.ldc + gn, v);
    else st4(reinterpret_cast<bf16_t*>(p.C) + (size_t)gm * p.ldc + gn, v);
}

// Store loop over the 16 row groups a thread owns.  gfx9's vmcnt counts stores as well as loads, and hipcc waits vmcnt(0) for a
// load result that sits behind younger stores -- a loop that mixes "fetch next row operands" with "store this row" therefore
// waits for every store to complete before the next one (measured: 27 us per 256x256 tile).  So: without row operands the loop
// contains no loads at all; with row operands ALL of them are fetched up front and the loop only stores.
template <int NIT = 16, typename E = EpiAny>
__device__ __forceinline__ void epi_rows(const NtParams& p, const float* ct, int ct_ld, float4 b4, int gm_first, int row_first, int row_step, int c4, int gn) {
    const bool rowops = E::rs(p) || E::res(p) || E::b2(p) || E::act(p) == TCOW_ACT_DGELU || E::act(p) == TCOW_ACT_MUL_AUX;
    if (!rowops) {
        EpiRow o; o.ext = make_float4(0.f, 0.f, 0.f, 0.f); o.rs = 1.0f; o.rs2 = 1.0f;
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int gm = gm_first + it * row_step;
            if (gm >= p.M) break;
            epi_row_apply<E>(p, o, *reinterpret_cast<const float4*>(ct + (row_first + it * row_step) * ct_ld + c4), b4, gm, gn);
        }
    } else {
        EpiRow o[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) o[it] = epi_row_fetch<E>(p, gm_first + it * row_step, gn, true);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int gm = gm_first + it * row_step;
            if (gm < p.M) epi_row_apply<E>(p, o[it], *reinterpret_cast<const float4*>(ct + (row_first + it * row_step) * ct_ld + c4), b4, gm, gn);
        }
    }
}

// Epilogue of one wave's 160 x 64 accumulator tile (rows row0 .. row0+159, columns col0 .. col0+63 of C), staged through the wave's private
// 17 KiB LDS region `wave_lds` (64 rows x 68 floats).  ML = 1: accumulators in acc16 (16x16x32 MFMAs, [10 row blocks][4 column blocks]); ML = 0:
// in acc (32x32x16, [5 row bands][2 column blocks]).  No workgroup barrier inside: a wave's LDS operations execute in order.
template <typename E, int ML>
__device__ __forceinline__ void wave_tile_epilogue_160x64(const NtParams& p, char* wave_lds, f32x16 (&acc)[5][2], f32x4 (&acc16)[10][4], int lane, int row0, int col0) {
    const int l31 = lane & 31, hi = lane >> 5;
    // ---- epilogue: every wave stages its 160 x 64 tile through a private 16 KiB LDS region, 64 rows at a time (the last pass 32),
    // and writes full 64-column row segments.  No workgroup barrier: a wave's LDS operations execute in order.
    // (explicit passes: a loop over the pass index that the optimizer declines to unroll would index acc[] dynamically -> scratch)
    // The MFMAs were issued as (W fragment, A fragment), i.e. the accumulators hold C^T: lane (l31, hi) owns output ROW l31 of
    // each 32-row band and, per register quad, four consecutive COLUMNS 8g + 4hi .. +3 -- a 16-byte LDS store per quad (40 per
    // lane and tile instead of 160 four-byte ones).  LDS rows are padded to 68 floats: conflict-free for these writes and for
    // the 8-columns-per-lane row reads below.
    constexpr int CT_LD = 68;
    float* ct = reinterpret_cast<float*>(wave_lds);
    const int c4 = (lane & 15) * 4;
    const int gn = col0 + c4;
    const bool col_ok = gn < p.N;
    const float4 b4 = (p.bias && col_ok) ? ld4(p.bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
    // ML = 1: band b = row blocks 2b, 2b+1; lane (l & 15, l >> 4) owns row l & 15 of a block and columns 16 cb + 4 (l >> 4) .. + 3
#define TCOW_STAGE(b, ii) do { if constexpr (ML == 1) stage_band16(b, ii); else stage_band(acc[b], ii); } while (0)
    auto stage_band16 = [&](int b, int ii) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const f32x4 v = acc16[2 * b + rb][cb];
                *reinterpret_cast<float4*>(ct + (ii * 32 + rb * 16 + (lane & 15)) * CT_LD + cb * 16 + 4 * (lane >> 4)) = make_float4(v[0], v[1], v[2], v[3]);
            }
    };
    auto stage_band = [&](const f32x16 (&a)[2], int ii) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(ct + (ii * 32 + l31) * CT_LD + j * 32 + 8 * g + 4 * hi) = make_float4(a[j][4 * g], a[j][4 * g + 1], a[j][4 * g + 2], a[j][4 * g + 3]);
    };
    const int mrow = row0 + (lane >> 4);
    const int mrow8 = row0 + (lane >> 3);
    constexpr bool kRowOps = E::kStatic && E::kRowOps;
    if constexpr (kRowOps) {
        // Row operands (residual / GELU' / row scale) of band b+1 are requested BEFORE band b is stored: their latency overlaps the
        // LDS staging and the stores of the band in front, and -- vmcnt retiring in order -- no load ever queues behind a store.
        // Pipeline unit = one 32-row accumulator band (8 row groups per lane), two operand sets and the two halves of the LDS region
        // in rotation; registers at the peak: 128 accumulators + 2 x 8 row operands.
        // Each lane handles 8 consecutive columns of a row (16-byte bf16 / 2 x 16-byte f32 accesses): half as many global
        // instructions per byte as the 4-column mapping, i.e. twice the bytes in flight for these latency-bound operand reads.
        struct Row8 { float4 e0, e1; float rs, rs2; };
        const int c8 = (lane & 7) * 8, r8 = lane >> 3;                       // 8 lanes x 8 columns = 64 columns, 8 rows per pass
        const int gn8 = col0 + c8;
        const bool ok8 = gn8 < p.N;                                          // N % 8 == 0 in bf16 mode
        float4 b40 = make_float4(0.f, 0.f, 0.f, 0.f), b41 = b40;
        if (p.bias && ok8) { b40 = ld4(p.bias + gn8); b41 = ld4(p.bias + gn8 + 4); }
        float4 c40 = make_float4(0.f, 0.f, 0.f, 0.f), c41 = c40;
        if (E::b2(p) && ok8) { c40 = ld4(p.bias2 + gn8); c41 = ld4(p.bias2 + gn8 + 4); }
        Row8 oa[4], ob[4];
        auto fetch = [&](Row8* o, int u) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int gm = mrow8 + u * 32 + it * 8;
                Row8 r; r.e0 = make_float4(0.f, 0.f, 0.f, 0.f); r.e1 = r.e0; r.rs = 1.0f; r.rs2 = 1.0f;
                if (ok8 && gm < p.M) {
                    if (E::rs(p)) r.rs = p.row_scale[gm];
                    if (E::b2(p) && p.row_scale2) r.rs2 = p.row_scale2[gm];
                    if (E::res(p)) { const float* q = p.resid + (size_t)gm * p.ldr + gn8; r.e0 = ld4(q); r.e1 = ld4(q + 4); }
                    else if (E::act(p) == TCOW_ACT_MUL_AUX || E::act(p) == TCOW_ACT_DGELU) {
                        const uint4 u4 = *reinterpret_cast<const uint4*>(p.aux + (size_t)gm * p.ldaux + gn8);
                        r.e0 = make_float4(bflo(u4.x), bfhi(u4.x), bflo(u4.y), bfhi(u4.y)); r.e1 = make_float4(bflo(u4.z), bfhi(u4.z), bflo(u4.w), bfhi(u4.w));
                    }
                }
                o[it] = r;
            }
        };
        auto fin = [&](float4 v, float4 bb, float4 cc, float4 e, float rs, float rs2) {
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            if (E::rs(p)) { v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
            if (E::act(p) == TCOW_ACT_MUL_AUX) { v.x *= e.x; v.y *= e.y; v.z *= e.z; v.w *= e.w; }
            else if (E::act(p) == TCOW_ACT_DGELU) v = dgelu4(v, e);
            if (E::b2(p)) { v.x = fmaf(rs2, cc.x, v.x); v.y = fmaf(rs2, cc.y, v.y); v.z = fmaf(rs2, cc.z, v.z); v.w = fmaf(rs2, cc.w, v.w); }
            if (E::res(p)) { v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
            return v;
        };
        auto apply = [&](const Row8* o, int u, int half) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int gm = mrow8 + u * 32 + it * 8;
                if (!(ok8 && gm < p.M)) continue;
                const float* cr = ct + (half * 32 + r8 + it * 8) * CT_LD + c8;
                const float4 v0 = fin(*reinterpret_cast<const float4*>(cr), b40, c40, o[it].e0, o[it].rs, o[it].rs2);
                const float4 v1 = fin(*reinterpret_cast<const float4*>(cr + 4), b41, c41, o[it].e1, o[it].rs, o[it].rs2);
                if (p.out_f32) {
                    float* d = reinterpret_cast<float*>(p.C) + (size_t)gm * p.ldc + gn8;
                    st4(d, v0); st4(d + 4, v1);
                } else {
                    uint4 w; w.x = pack_bf2(v0.x, v0.y); w.y = pack_bf2(v0.z, v0.w); w.z = pack_bf2(v1.x, v1.y); w.w = pack_bf2(v1.z, v1.w);
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)gm * p.ldc + gn8) = w;
                }
            }
        };
        if constexpr (E::kAuxOnly) {
            // The aux tile is requested THREE bands ahead, 16 bytes (8 packed columns) per row group and lane (48 registers; all five bands = 80
            // spilled).  It is a first-touch HBM stream (166 MB per launch): one band ahead, every band waited for its own miss latency behind
            // the previous band's stores.
            uint4 ax[3][4];
            auto fetch_ax = [&](uint4 (&a)[4], int u) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int gm = mrow8 + u * 32 + it * 8;
                    a[it] = (ok8 && gm < p.M) ? ld16c<true>(p.aux + (size_t)gm * p.ldaux + gn8) : make_uint4(0u, 0u, 0u, 0u);
                }
            };
            fetch_ax(ax[0], 0); fetch_ax(ax[1], 1); fetch_ax(ax[2], 2);
            auto apply_ax = [&](const uint4 (&a)[4], int u, int half) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int gm = mrow8 + u * 32 + it * 8;
                    if (!(ok8 && gm < p.M)) continue;
                    const float* cr = ct + (half * 32 + r8 + it * 8) * CT_LD + c8;
                    float4 v0 = *reinterpret_cast<const float4*>(cr), v1 = *reinterpret_cast<const float4*>(cr + 4);
                    const uint4 e = a[it];
                    v0.x = (v0.x + b40.x) * bflo(e.x); v0.y = (v0.y + b40.y) * bfhi(e.x); v0.z = (v0.z + b40.z) * bflo(e.y); v0.w = (v0.w + b40.w) * bfhi(e.y);
                    v1.x = (v1.x + b41.x) * bflo(e.z); v1.y = (v1.y + b41.y) * bfhi(e.z); v1.z = (v1.z + b41.z) * bflo(e.w); v1.w = (v1.w + b41.w) * bfhi(e.w);
                    if (p.out_f32) {
                        float* d = reinterpret_cast<float*>(p.C) + (size_t)gm * p.ldc + gn8;
                        st4(d, v0); st4(d + 4, v1);
                    } else {
                        uint4 w; w.x = pack_bf2(v0.x, v0.y); w.y = pack_bf2(v0.z, v0.w); w.z = pack_bf2(v1.x, v1.y); w.w = pack_bf2(v1.z, v1.w);
                        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)gm * p.ldc + gn8) = w;
                    }
                }
            };
            TCOW_STAGE(0, 0); TCOW_STAGE(1, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            apply_ax(ax[0], 0, 0);
            fetch_ax(ax[0], 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TCOW_STAGE(2, 0);
            apply_ax(ax[1], 1, 1);
            fetch_ax(ax[1], 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TCOW_STAGE(3, 1);
            apply_ax(ax[2], 2, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TCOW_STAGE(4, 0);
            apply_ax(ax[0], 3, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            apply_ax(ax[1], 4, 0);
            return;
        }
        fetch(oa, 0); TCOW_STAGE(0, 0);
        fetch(ob, 1); TCOW_STAGE(1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        apply(oa, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fetch(oa, 2); TCOW_STAGE(2, 0);
        apply(ob, 1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fetch(ob, 3); TCOW_STAGE(3, 1);
        apply(oa, 2, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fetch(oa, 4); TCOW_STAGE(4, 0);
        apply(ob, 3, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        apply(oa, 4, 0);
    } else if constexpr (E::kStatic) {
        // no row operands: a rolled, load-free loop (gfx9 counts stores in vmcnt), again 8 columns per lane: one 16-byte bf16 store
        // per row piece (two for f32), plus the GELU' store of TCOW_ACT_GELU_DSAVE
        const int c8 = (lane & 7) * 8, r8 = lane >> 3;
        const int gn8 = col0 + c8;
        const bool ok8 = gn8 < p.N;
        float4 b40 = make_float4(0.f, 0.f, 0.f, 0.f), b41 = b40;
        if (p.bias && ok8) { b40 = ld4(p.bias + gn8); b41 = ld4(p.bias + gn8 + 4); }
        auto rows8 = [&](int row0, int nit) {
#pragma unroll 1
            for (int it = 0; it < nit; ++it) {
                const int gm = mrow8 + row0 + it * 8;
                if (!(ok8 && gm < p.M)) break;
                const float* cr = ct + (r8 + it * 8) * CT_LD + c8;
                float4 v0 = *reinterpret_cast<const float4*>(cr), v1 = *reinterpret_cast<const float4*>(cr + 4);
                v0.x += b40.x; v0.y += b40.y; v0.z += b40.z; v0.w += b40.w; v1.x += b41.x; v1.y += b41.y; v1.z += b41.z; v1.w += b41.w;
                if (E::act(p) == TCOW_ACT_GELU_DSAVE) {
                    float4 g0, d0, g1, d1; gelu_both4(v0, g0, d0); gelu_both4(v1, g1, d1);
                    uint4 w; w.x = pack_bf2(d0.x, d0.y); w.y = pack_bf2(d0.z, d0.w); w.z = pack_bf2(d1.x, d1.y); w.w = pack_bf2(d1.z, d1.w);
                    st16c<true>(p.aux + (size_t)gm * p.ldaux + gn8, w);
                    v0 = g0; v1 = g1;
                } else if (E::act(p) == TCOW_ACT_GELU) {
                    if (p.aux) {
                        uint4 w; w.x = pack_bf2(v0.x, v0.y); w.y = pack_bf2(v0.z, v0.w); w.z = pack_bf2(v1.x, v1.y); w.w = pack_bf2(v1.z, v1.w);
                        *reinterpret_cast<uint4*>(p.aux + (size_t)gm * p.ldaux + gn8) = w;
                    }
                    v0 = gelu4(v0); v1 = gelu4(v1);
                }
                if (p.out_f32) {
                    float* d = reinterpret_cast<float*>(p.C) + (size_t)gm * p.ldc + gn8;
                    st4(d, v0); st4(d + 4, v1);
                } else {
                    uint4 w; w.x = pack_bf2(v0.x, v0.y); w.y = pack_bf2(v0.z, v0.w); w.z = pack_bf2(v1.x, v1.y); w.w = pack_bf2(v1.z, v1.w);
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)gm * p.ldc + gn8) = w;
                }
            }
        };
        TCOW_STAGE(0, 0); TCOW_STAGE(1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rows8(0, 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TCOW_STAGE(2, 0); TCOW_STAGE(3, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rows8(64, 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TCOW_STAGE(4, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rows8(128, 4);
    } else {
        TCOW_STAGE(0, 0); TCOW_STAGE(1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (col_ok) epi_rows<16, E>(p, ct, CT_LD, b4, mrow, lane >> 4, 4, c4, gn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TCOW_STAGE(2, 0); TCOW_STAGE(3, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (col_ok) epi_rows<16, E>(p, ct, CT_LD, b4, mrow + 64, lane >> 4, 4, c4, gn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TCOW_STAGE(4, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (col_ok) epi_rows<8, E>(p, ct, CT_LD, b4, mrow + 128, lane >> 4, 4, c4, gn);
    }
#undef TCOW_STAGE
}

}  // namespace
