// Flash attention of the f32-storage mode precision='bf16x3' on the bf16 matrix cores: softmax(q k^T * d^-0.5 [causal mask]) v (vit.py:88-109) with every
// matrix product formed as THREE bf16 MFMAs on hi / lo operand splits, exactly as gemm_x3.hip does for the Linear layers:
//   x = hi + lo, hi = bf16(x) (round to nearest even), lo = bf16(x - hi), |x - hi - lo| <= 2^-18 |x|;   a.b ~ a_lo*b_hi + a_hi*b_lo + a_hi*b_hi (f32 accumulate).
// Until round 6 the mode ran its attention on the exact-f32 MFMA (attention_f32.hip: 1/16 of the bf16 rate): 1 499 us per spatial forward at configs[1], a fifth
// of that mode's step.  Same contract as attention_f32.hip (f32 qkv [rows, 3D] in, f32 out / dqkv, natural-log LSE, delta workspace in the lse layout).
//
// Structure = the streaming kernels of attention_bf16.hip: a 256-thread workgroup owns 4 query (forward, dQ) or 4 key (dK / dV) tiles of one (sequence, head),
// one per wave, and walks the other side in chunks of CH tiles.  A chunk is read as f32 by all 256 threads (one 8-element piece of a row per thread and
// tile: two 16-byte loads), split ONCE in registers and stored as two [32][64] bf16 LDS tiles (hi, lo) in the layout of attention_tiles.h (16-byte chunk c of row
// r at c ^ g(r): conflict-free ds_read_b128 row fragments and ds_read_b64_tr_b16 transpose reads).  The wave's own rows (Q, dO or K, V) are split when
// their fragments are loaded; probabilities / dS are split in registers before they become MFMA B operands.  Softmax statistics, masks, the lazy running
// maximum and the accumulation order are those of attention_bf16.hip's fwd_tile / dq_tile / dkv_tile.
// MFMAs per (query tile, key tile) pair: forward 24, dQ 36, dK / dV 48 (bf16 mode: 8 / 12 / 16).
#include <stdlib.h>

#include "attention_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 xb8;       // (always bfloat16, whatever 16-bit format the library is built for: bf16 keeps the f32 exponent range)
typedef __attribute__((ext_vector_type(2))) __bf16 xb2;
typedef uint32_t xu4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short xs4;
typedef __attribute__((ext_vector_type(8))) short xs8;

constexpr int XTILE = 4096;                  // [32 rows][64 bf16]
constexpr float xScale = 0.125f;             // head_dim^-0.5 (vit.py:70)
constexpr float xLog2e = 1.4426950408889634f;
constexpr float xLn2 = 0.6931471805599453f;

#define X3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define X3_NO_IFCVT() asm volatile("" ::: "memory")

__device__ __forceinline__ int xswz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ uint32_t xpack(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xb2));
}
__device__ __forceinline__ void xsplit2(float x0, float x1, uint32_t& h, uint32_t& l) {
    h = xpack(x0, x1);
    l = xpack(x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xffff0000u));     // exact differences (hi shares the leading bits of x)
}
// eight f32 -> the hi and lo bf16x8 operands
__device__ __forceinline__ void xsplit8u(const float* v, xu4& hh, xu4& ll) {
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    xsplit2(v[0], v[1], h0, l0); xsplit2(v[2], v[3], h1, l1); xsplit2(v[4], v[5], h2, l2); xsplit2(v[6], v[7], h3, l3);
    hh = (xu4){h0, h1, h2, h3}; ll = (xu4){l0, l1, l2, l3};
}
__device__ __forceinline__ void xsplit8(const float* v, xb8& h, xb8& l) {
    xu4 hh, ll;
    xsplit8u(v, hh, ll);
    h = __builtin_bit_cast(xb8, hh); l = __builtin_bit_cast(xb8, ll);
}
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }

// acc += a . b with both operands split: small terms first
__device__ __forceinline__ f32x16 x3_mm(const xb8& ah, const xb8& al, const xb8& bh, const xb8& bl, f32x16 acc) {
    acc = X3_MFMA(al, bh, acc);
    acc = X3_MFMA(ah, bl, acc);
    return X3_MFMA(ah, bh, acc);
}

// ---- staging: tile = rows p0 .. p0+31 (clamped to L-1: padding rows hold finite data, masks / zero probabilities keep them out) of a 64-column f32 block.
// Thread t of the 256: row t >> 3, columns 8 (t & 7) .. + 7.
struct XItem { f32x4 a, b; };
__device__ __forceinline__ XItem x_load(const float* __restrict__ src, long stride, int p0, int L, int tid) {
    int pos = p0 + (tid >> 3); pos = pos < L ? pos : L - 1;
    const float* p = src + (size_t)pos * stride + 8 * (tid & 7);
    XItem it; it.a = *reinterpret_cast<const f32x4*>(p); it.b = *reinterpret_cast<const f32x4*>(p + 4);
    return it;
}
__device__ __forceinline__ void x_store_rc(const XItem& it, char* hi_tile, char* lo_tile, int r, int c);
__device__ __forceinline__ void x_store(const XItem& it, char* hi_tile, char* lo_tile, int tid) { x_store_rc(it, hi_tile, lo_tile, tid >> 3, tid & 7); }
__device__ __forceinline__ void x_store_rc(const XItem& it, char* hi_tile, char* lo_tile, int r, int c) {
    xu4 h, l;
    const float v[8] = {it.a[0], it.a[1], it.a[2], it.a[3], it.b[0], it.b[1], it.b[2], it.b[3]};
    xsplit8u(v, h, l);
    const int off = r * 128 + ((c ^ xswz(r)) << 4);
    *reinterpret_cast<xu4*>(hi_tile + off) = h;
    *reinterpret_cast<xu4*>(lo_tile + off) = l;
}
// tiles c0 .. c0+CH-1 (those below `nt_end`) of two row sources a, b -> LDS arrays [a hi][a lo][b hi][b lo], CH tiles each.  All loads are issued before the
// first split (16 CH registers in flight).
template <int CH>
__device__ __forceinline__ void x_stage(const float* __restrict__ a, long sa, const float* __restrict__ b, long sb, int c0, int nt_end, int L, char* smem, int tid) {
    XItem ia[CH], ib[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t)
        if (c0 + t < nt_end) { ia[t] = x_load(a, sa, 32 * (c0 + t), L, tid); ib[t] = x_load(b, sb, 32 * (c0 + t), L, tid); }
#pragma unroll
    for (int t = 0; t < CH; ++t)
        if (c0 + t < nt_end) {
            x_store(ia[t], smem + t * XTILE, smem + (CH + t) * XTILE, tid);
            x_store(ib[t], smem + (2 * CH + t) * XTILE, smem + (3 * CH + t) * XTILE, tid);
        }
}

// SOLO kernels (one-tile sequences, temporal attention): every WAVE owns its own (sequence, head) and stages the one tile of each source itself -- row (lane >> 3) + 8 pass,
// columns 8 (lane & 7) .. + 7, four passes per tile -- into wave-private LDS; no workgroup barrier (a wave's LDS operations execute in order).
__device__ __forceinline__ void x_stage_wave(const float* __restrict__ a, long sa, const float* __restrict__ b, long sb, int L, char* wsm, int lane) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {               // two rounds of 16 rows: 32 registers of loads in flight instead of 64 (the SOLO kernels hold their own fragments too)
        XItem ia[2], ib[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) { ia[ps] = x_load(a, sa, 16 * half + 8 * ps, L, lane); ib[ps] = x_load(b, sb, 16 * half + 8 * ps, L, lane); }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            x_store_rc(ia[ps], wsm, wsm + XTILE, 16 * half + 8 * ps + (lane >> 3), lane & 7);
            x_store_rc(ib[ps], wsm + 2 * XTILE, wsm + 3 * XTILE, 16 * half + 8 * ps + (lane >> 3), lane & 7);
        }
        asm volatile("" ::: "memory");                   // (keeps hipcc from issuing the second round's loads with the first)
    }
}

// A/B fragment of a row-major tile for a contraction over d: lane (row, hi) gets d = 16 ks + 8 hi .. + 7
__device__ __forceinline__ xb8 xfrag_row(const char* tile, int row, int ks, int hi) {
    const int c = (2 * ks + hi) ^ xswz(row);
    return __builtin_bit_cast(xb8, *reinterpret_cast<const xu4*>(tile + row * 128 + (c << 4)));
}
// fragment for a contraction over the tile's ROWS: lane (col = 32 dt + (lane & 31), hi) gets rows crow32(8 s + j, hi), j = 0..7 (two transpose reads)
__device__ __forceinline__ xb8 xfrag_tr(const char* tile, int s, int dt, int lane) {
    const int q16 = lane & 15, g16 = (lane >> 4) & 1, hi = lane >> 5;
    const int chunk = 4 * dt + 2 * g16 + ((q16 & 3) >> 1);
    const int r0 = 16 * s + 4 * hi + (q16 >> 2), r1 = r0 + 8;
    const int o0 = r0 * 128 + ((chunk ^ xswz(r0)) << 4) + (q16 & 1) * 8;
    const int o1 = r1 * 128 + ((chunk ^ xswz(r1)) << 4) + (q16 & 1) * 8;
    const xs4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(xs4))(tile + o0));
    const xs4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(xs4))(tile + o1));
    return __builtin_bit_cast(xb8, (xs8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
}
// the wave's own row (position pos of a 64-column f32 block), split: fragment ks = columns 16 ks + 8 hi .. + 7
__device__ __forceinline__ void xfrag_global(const float* __restrict__ row, int hi, xb8 (&h)[4], xb8 (&l)[4], float* keep = nullptr) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + 16 * ks + 8 * hi), b = *reinterpret_cast<const f32x4*>(row + 16 * ks + 8 * hi + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        xsplit8(v, h[ks], l[ks]);
        if (keep) {
#pragma unroll
            for (int e = 0; e < 8; ++e) keep[8 * ks + e] = v[e];
        }
    }
}

// same placement as the other streaming kernels: the workgroups of one (sequence, head) back to back on one XCD
struct XWork { int pair, chunk; bool valid; };
__device__ __forceinline__ XWork x_work(int pairs, int nchunk) {
    const int b = blockIdx.x, x = b & 7, k = b >> 3;
    const int i = k / nchunk;
    XWork w; w.chunk = k - i * nchunk; w.pair = 8 * i + x; w.valid = w.pair < pairs;
    return w;
}
inline int x_grid(int pairs, int nchunk) { return 8 * ((pairs + 7) / 8) * nchunk; }

// f32 row store of a transposed accumulator pair: lane (row l31, hi) owns channels 32 dt + 8 g + 4 hi + 0..3 of register quad g
__device__ __forceinline__ void x_store_row(float* __restrict__ drow, int hi, const f32x16& a0, const f32x16& a1, float mul) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        st4(drow + 8 * g + 4 * hi, make_float4(a0[4 * g] * mul, a0[4 * g + 1] * mul, a0[4 * g + 2] * mul, a0[4 * g + 3] * mul));
        st4(drow + 32 + 8 * g + 4 * hi, make_float4(a1[4 * g] * mul, a1[4 * g + 1] * mul, a1[4 * g + 2] * mul, a1[4 * g + 3] * mul));
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <int CH, bool SOLO = false>
__global__ __launch_bounds__(256, 2) void attn_x3_fwd(SeqDesc sd, int nt, const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];          // [K hi][K lo][V hi][V lo], CH tiles each (SOLO: one such set per wave)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    XWork w = x_work(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (SOLO) { w.pair = blockIdx.x * 4 + wave; w.chunk = 0; w.valid = w.pair < sd.n_outer * sd.n_inner * sd.heads; }
    if (!w.valid) return;
    char* smem = SOLO ? smem_ + wave * (4 * XTILE) : smem_;
    const int item = w.pair / sd.heads, head = w.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const float* qh = qkv + base * ld3 + head * ATT_HD;
    const int qt = SOLO ? 0 : w.chunk * 4 + wave;
    const bool active = qt < nt;
    const int q = 32 * qt + l31, qc = q < sd.L ? q : sd.L - 1;
    if (SOLO) x_stage_wave(qh + sd.D, pse, qh + 2 * sd.D, pse, sd.L, smem, lane);      // (before the wave's own fragments are loaded: the staging registers are dead by then)
    xb8 qfh[4], qfl[4];
    xfrag_global(qh + (size_t)qc * pse, hi, qfh, qfl);
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -1e30f, l = 0.f;
    const float sc = xScale * xLog2e;
    const long klim = (long)32 * qt + 31 + sd.diag;
    const int kt_end = (!active) ? 0 : (klim >= (long)sd.L - 1 ? nt : (int)(klim / 32) + 1);
    const long klim_wg = (long)32 * (w.chunk * 4 + 3) + 31 + sd.diag;          // last key tile any wave of this workgroup needs
    const int kt_end_wg = klim_wg >= (long)sd.L - 1 ? nt : (int)(klim_wg / 32) + 1;
    for (int c0 = 0; c0 < kt_end_wg; c0 += CH) {
        if (!SOLO) {
            __syncthreads();                                               // previous chunk fully consumed
            x_stage<CH>(qh + sd.D, pse, qh + 2 * sd.D, pse, c0, kt_end_wg, sd.L, smem, tid);
            __syncthreads();
        }
        const int jend = (c0 + CH < kt_end) ? c0 + CH : kt_end;
        for (int j = c0; j < jend; ++j) {
            const char* kh = smem + (j - c0) * XTILE; const char* kl = kh + CH * XTILE;
            const char* vh = kh + 2 * CH * XTILE; const char* vl = kh + 3 * CH * XTILE;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = x3_mm(xfrag_row(kh, l31, ks, hi), xfrag_row(kl, l31, ks, hi), qfh[ks], qfl[ks], s);
            const bool need_mask = (32 * j + 31 >= sd.L) || ((long)32 * j + 31 > (long)32 * qt + sd.diag);
            if (need_mask) {
                X3_NO_IFCVT();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * j + crow32(r, hi);
                    if (key >= sd.L || (long)key > (long)q + sd.diag) s[r] = -1e30f;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = xhalf_max(mx) * sc;                                       // (sc > 0: the maximum commutes with the scale)
            if (__any(mx > m + 8.0f)) {                                    // lazy running maximum (attention_tiles.h fwd_tile)
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
            if (need_mask) {                                               // a fully masked row (m still at its start value) must contribute nothing
                X3_NO_IFCVT();
                ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { if (s[r] <= -1e29f) p[r] = 0.f; ps += p[r]; }
            }
            l += ps;
            xb8 ph0, pl0, ph1, pl1;
            xsplit8(p, ph0, pl0); xsplit8(p + 8, ph1, pl1);
            o0 = x3_mm(xfrag_tr(vh, 0, 0, lane), xfrag_tr(vl, 0, 0, lane), ph0, pl0, o0);
            o1 = x3_mm(xfrag_tr(vh, 0, 1, lane), xfrag_tr(vl, 0, 1, lane), ph0, pl0, o1);
            o0 = x3_mm(xfrag_tr(vh, 1, 0, lane), xfrag_tr(vl, 1, 0, lane), ph1, pl1, o0);
            o1 = x3_mm(xfrag_tr(vh, 1, 1, lane), xfrag_tr(vl, 1, 1, lane), ph1, pl1, o1);
        }
    }
    if (!active) return;
    l = xhalf_sum(l);
    if (q < sd.L) {
        const long row = base + (long)q * sd.pos_stride;
        x_store_row(out + row * sd.D + head * ATT_HD, hi, o0, o1, 1.0f / l);
        if (lse && hi == 0) lse[row * sd.heads + head] = (m + log2f(l)) * xLn2;          // natural-log LSE of the scaled scores
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <int CH, bool SOLO = false>
__global__ __launch_bounds__(256, 2) void attn_x3_bwd_dq(SeqDesc sd, int nt, const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ dout,
                                                         const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];          // [K hi][K lo][V hi][V lo] (SOLO: one set per wave)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    XWork w = x_work(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (SOLO) { w.pair = blockIdx.x * 4 + wave; w.chunk = 0; w.valid = w.pair < sd.n_outer * sd.n_inner * sd.heads; }
    if (!w.valid) return;
    char* smem = SOLO ? smem_ + wave * (4 * XTILE) : smem_;
    const int item = w.pair / sd.heads, head = w.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const float* qh = qkv + base * ld3 + head * ATT_HD;
    const int qt = SOLO ? 0 : w.chunk * 4 + wave;
    const bool active = qt < nt;
    const int q = 32 * qt + l31, qc = q < sd.L ? q : sd.L - 1;
    const long row = base + (long)qc * sd.pos_stride;
    if (SOLO) x_stage_wave(qh + sd.D, pse, qh + 2 * sd.D, pse, sd.L, smem, lane);
    xb8 qfh[4], qfl[4], dfh[4], dfl[4];
    xfrag_global(qh + (size_t)qc * pse, hi, qfh, qfl);
    // delta = rowsum(dO * O) of this lane's query in f32 (this half-wave's 32 channels + the other's): published for the dK / dV kernel, which runs after this one
    float dl;
    {
        float dov[32];
        xfrag_global(dout + row * sd.D + head * ATT_HD, hi, dfh, dfl, dov);
        const float* orow = o + row * sd.D + head * ATT_HD;
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(orow + 16 * ks + 8 * hi), b = *reinterpret_cast<const f32x4*>(orow + 16 * ks + 8 * hi + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { part = fmaf(dov[8 * ks + e], a[e], part); part = fmaf(dov[8 * ks + 4 + e], b[e], part); }
        }
        dl = xhalf_sum(part);
        if (active && q < sd.L && hi == 0) delta[row * sd.heads + head] = dl;
    }
    const float ls = lse[row * sd.heads + head] * xLog2e;
    const float sc = xScale * xLog2e;
    f32x16 dq0, dq1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
    const long klim = (long)32 * qt + 31 + sd.diag;
    const int kt_end = (!active) ? 0 : (klim >= (long)sd.L - 1 ? nt : (int)(klim / 32) + 1);
    const long klim_wg = (long)32 * (w.chunk * 4 + 3) + 31 + sd.diag;
    const int kt_end_wg = klim_wg >= (long)sd.L - 1 ? nt : (int)(klim_wg / 32) + 1;
    for (int c0 = 0; c0 < kt_end_wg; c0 += CH) {
        if (!SOLO) {
            __syncthreads();
            x_stage<CH>(qh + sd.D, pse, qh + 2 * sd.D, pse, c0, kt_end_wg, sd.L, smem, tid);
            __syncthreads();
        }
        const int jend = (c0 + CH < kt_end) ? c0 + CH : kt_end;
        for (int j = c0; j < jend; ++j) {
            const char* kh = smem + (j - c0) * XTILE; const char* kl = kh + CH * XTILE;
            const char* vh = kh + 2 * CH * XTILE; const char* vl = kh + 3 * CH * XTILE;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = x3_mm(xfrag_row(kh, l31, ks, hi), xfrag_row(kl, l31, ks, hi), qfh[ks], qfl[ks], s);            // S^T[key][query]
                dp = x3_mm(xfrag_row(vh, l31, ks, hi), xfrag_row(vl, l31, ks, hi), dfh[ks], dfl[ks], dp);          // dP^T[key][query] = V dO^T
            }
            const bool need_mask = (32 * j + 31 >= sd.L) || (q - l31 + 31 >= sd.L) || ((long)32 * j + 31 > (long)(q - l31) + sd.diag);
            float dsv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -ls));
                dsv[r] = p * (dp[r] - dl);
            }
            if (need_mask) {
                X3_NO_IFCVT();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * j + crow32(r, hi);
                    if (!(q < sd.L && key < sd.L && (long)key <= (long)q + sd.diag)) dsv[r] = 0.f;
                }
            }
            xb8 dh0, dl0, dh1, dl1;
            xsplit8(dsv, dh0, dl0); xsplit8(dsv + 8, dh1, dl1);
            dq0 = x3_mm(xfrag_tr(kh, 0, 0, lane), xfrag_tr(kl, 0, 0, lane), dh0, dl0, dq0);                        // dQ^T += K^T dS^T
            dq1 = x3_mm(xfrag_tr(kh, 0, 1, lane), xfrag_tr(kl, 0, 1, lane), dh0, dl0, dq1);
            dq0 = x3_mm(xfrag_tr(kh, 1, 0, lane), xfrag_tr(kl, 1, 0, lane), dh1, dl1, dq0);
            dq1 = x3_mm(xfrag_tr(kh, 1, 1, lane), xfrag_tr(kl, 1, 1, lane), dh1, dl1, dq1);
        }
    }
    if (!active || q >= sd.L) return;
    x_store_row(dqkv + row * ld3 + head * ATT_HD, hi, dq0, dq1, xScale);        // dS was accumulated without its 1/sqrt(d) factor
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <int CH, bool SOLO = false>
__global__ __launch_bounds__(256, 2) void attn_x3_bwd_dkv(SeqDesc sd, int nt, const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];          // [Q hi][Q lo][dO hi][dO lo], CH tiles each, + the chunk's (lse log2e, delta) table (SOLO: per wave)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    XWork w = x_work(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (SOLO) { w.pair = blockIdx.x * 4 + wave; w.chunk = 0; w.valid = w.pair < sd.n_outer * sd.n_inner * sd.heads; }
    if (!w.valid) return;
    char* smem = SOLO ? smem_ + wave * (4 * XTILE + 256) : smem_;
    float2* tab = reinterpret_cast<float2*>(smem + 4 * CH * XTILE);
    const int item = w.pair / sd.heads, head = w.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const float* qh = qkv + base * ld3 + head * ATT_HD;
    const float* doh = dout + base * sd.D + head * ATT_HD;
    const int jt = SOLO ? 0 : w.chunk * 4 + wave;
    const bool active = jt < nt;
    const int key = 32 * jt + l31, kc = key < sd.L ? key : sd.L - 1;
    if (SOLO) {
        x_stage_wave(qh, pse, doh, pso, sd.L, smem, lane);
        if (lane < 32) {
            const long row = base + (long)(lane < sd.L ? lane : sd.L - 1) * sd.pos_stride;
            tab[lane] = make_float2(lse[row * sd.heads + head] * xLog2e, delta[row * sd.heads + head]);
        }
    }
    xb8 kfh[4], kfl[4], vfh[4], vfl[4];
    xfrag_global(qh + (size_t)kc * pse + sd.D, hi, kfh, kfl);
    xfrag_global(qh + (size_t)kc * pse + 2 * sd.D, hi, vfh, vfl);
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
    const float sc = xScale * xLog2e;
    // first query tile that can see any key of this wave / of this workgroup
    const long qlo = (long)32 * jt - sd.diag;
    const int i0 = qlo > 0 ? (int)(qlo / 32) : 0;
    const long qlo_wg = (long)32 * (w.chunk * 4) - sd.diag;
    const int c_start = qlo_wg > 0 ? ((int)(qlo_wg / 32) / CH) * CH : 0;
    for (int c0 = c_start; c0 < nt; c0 += CH) {
        if (!SOLO) {
            __syncthreads();
            x_stage<CH>(qh, pse, doh, pso, c0, nt, sd.L, smem, tid);
            if (tid < CH * 32) {
                const int qi = 32 * c0 + tid;
                const long row = base + (long)(qi < sd.L ? qi : sd.L - 1) * sd.pos_stride;
                tab[tid] = make_float2(lse[row * sd.heads + head] * xLog2e, delta[row * sd.heads + head]);
            }
            __syncthreads();
        }
        if (!active) continue;
        const int ib = c0 > i0 ? c0 : i0, ie = c0 + CH < nt ? c0 + CH : nt;
        for (int i = ib; i < ie; ++i) {
            const char* qth = smem + (i - c0) * XTILE; const char* qtl = qth + CH * XTILE;
            const char* doth = qth + 2 * CH * XTILE; const char* dotl = qth + 3 * CH * XTILE;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = x3_mm(xfrag_row(qth, l31, ks, hi), xfrag_row(qtl, l31, ks, hi), kfh[ks], kfl[ks], s);          // S[query crow32(r, hi)][key l31]
                dp = x3_mm(xfrag_row(doth, l31, ks, hi), xfrag_row(dotl, l31, ks, hi), vfh[ks], vfl[ks], dp);      // dP[query][key] = dO V^T
            }
            const bool need_mask = (32 * i + 31 >= sd.L) || (key - l31 + 31 >= sd.L) || ((long)(key - l31) + 31 > (long)32 * i + sd.diag);
            float pv[16], dsv[16];
            const float4* t4 = reinterpret_cast<const float4*>(tab + 32 * (i - c0) + 4 * hi);                      // (lse2, delta) of queries 8 gq + 4 hi + e
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 a = t4[4 * gq], b = t4[4 * gq + 1];
                const float lsq[4] = {a.x, a.z, b.x, b.z}, dlq[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * gq + e;
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -lsq[e]));
                    pv[r] = p;
                    dsv[r] = p * (dp[r] - dlq[e]);
                }
            }
            if (need_mask) {
                X3_NO_IFCVT();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = 32 * i + crow32(r, hi);
                    if (!(qi < sd.L && key < sd.L && (long)key <= (long)qi + sd.diag)) { pv[r] = 0.f; dsv[r] = 0.f; }
                }
            }
            xb8 ph0, pl0, ph1, pl1, dh0, dl0, dh1, dl1;
            xsplit8(pv, ph0, pl0); xsplit8(pv + 8, ph1, pl1); xsplit8(dsv, dh0, dl0); xsplit8(dsv + 8, dh1, dl1);
            dv0 = x3_mm(xfrag_tr(doth, 0, 0, lane), xfrag_tr(dotl, 0, 0, lane), ph0, pl0, dv0);                    // dV^T += dO^T P
            dv1 = x3_mm(xfrag_tr(doth, 0, 1, lane), xfrag_tr(dotl, 0, 1, lane), ph0, pl0, dv1);
            dv0 = x3_mm(xfrag_tr(doth, 1, 0, lane), xfrag_tr(dotl, 1, 0, lane), ph1, pl1, dv0);
            dv1 = x3_mm(xfrag_tr(doth, 1, 1, lane), xfrag_tr(dotl, 1, 1, lane), ph1, pl1, dv1);
            dk0 = x3_mm(xfrag_tr(qth, 0, 0, lane), xfrag_tr(qtl, 0, 0, lane), dh0, dl0, dk0);                      // dK^T += Q^T dS
            dk1 = x3_mm(xfrag_tr(qth, 0, 1, lane), xfrag_tr(qtl, 0, 1, lane), dh0, dl0, dk1);
            dk0 = x3_mm(xfrag_tr(qth, 1, 0, lane), xfrag_tr(qtl, 1, 0, lane), dh1, dl1, dk0);
            dk1 = x3_mm(xfrag_tr(qth, 1, 1, lane), xfrag_tr(qtl, 1, 1, lane), dh1, dl1, dk1);
        }
    }
    if (!active || key >= sd.L) return;
    const long row = base + (long)key * sd.pos_stride;
    x_store_row(dqkv + row * ld3 + sd.D + head * ATT_HD, hi, dk0, dk1, xScale);
    x_store_row(dqkv + row * ld3 + 2 * sd.D + head * ATT_HD, hi, dv0, dv1, 1.0f);
}

constexpr int X3_CH = 4;
template <int CH> constexpr int x3_lds() { return 4 * CH * XTILE; }
template <int CH> constexpr int x3_lds_dkv() { return 4 * CH * XTILE + CH * 32 * 8; }

template <int CH>
int x3_fwd_launch(hipStream_t st, const SeqDesc& d, int nt, int grid, const void* qkv, void* out, float* lse) {
    tcow_ensure_lds((const void*)attn_x3_fwd<CH>, x3_lds<CH>());
    hipLaunchKernelGGL(attn_x3_fwd<CH>, dim3(grid), dim3(256), x3_lds<CH>(), st, d, nt, (const float*)qkv, (float*)out, lse);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
template <int CH>
int x3_bwd_launch(hipStream_t st, const SeqDesc& d, int nt, int grid, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv) {
    tcow_ensure_lds((const void*)attn_x3_bwd_dq<CH>, x3_lds<CH>());
    tcow_ensure_lds((const void*)attn_x3_bwd_dkv<CH>, x3_lds_dkv<CH>());
    hipLaunchKernelGGL(attn_x3_bwd_dq<CH>, dim3(grid), dim3(256), x3_lds<CH>(), st, d, nt, (const float*)qkv, (const float*)out, (const float*)dout, lse, delta, (float*)dqkv);
    TCOW_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_x3_bwd_dkv<CH>, dim3(grid), dim3(256), x3_lds_dkv<CH>(), st, d, nt, (const float*)qkv, (const float*)dout, lse, (const float*)delta, (float*)dqkv);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

}  // namespace

// Every length.  One-tile sequences (temporal attention, T <= 32) run the SOLO instantiations: a wave per (sequence, head) with wave-private LDS tiles, no barriers
// (the same kernels with one tile per chunk and one active wave per workgroup measured 84 / 302 us forward / backward at configs[1], attention_f32.hip's wave-private
// exact-f32 kernels 127 / 407 us).
bool tcow_attn_x3_supported(const SeqDesc& d) { (void)d; return true; }

int tcow_attn_x3_fwd(hipStream_t st, const SeqDesc& d, const void* qkv, void* out, float* lse) {
    const int nt = cdiv(d.L, 32), pairs = d.n_outer * d.n_inner * d.heads, grid = x_grid(pairs, cdiv(nt, 4));
    if (nt == 1) {                       // one-tile sequences: a wave per (sequence, head), wave-private 16 KiB
        constexpr int lds = 4 * 4 * XTILE;
        tcow_ensure_lds((const void*)attn_x3_fwd<1, true>, lds);
        hipLaunchKernelGGL((attn_x3_fwd<1, true>), dim3(cdiv(pairs, 4)), dim3(256), lds, st, d, nt, (const float*)qkv, (float*)out, lse);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    return x3_fwd_launch<2>(st, d, nt, grid, qkv, out, lse);      // forward: two tiles per chunk (32 KiB, 146 registers: three workgroups per CU) -- 132-142 us against 158-163 with four
}

// `delta` = rows * heads floats of workspace (the layout of lse)
int tcow_attn_x3_bwd(hipStream_t st, const SeqDesc& d, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv) {
    const int nt = cdiv(d.L, 32), pairs = d.n_outer * d.n_inner * d.heads, grid = x_grid(pairs, cdiv(nt, 4));
    if (nt == 1) {
        constexpr int lds = 4 * 4 * XTILE, lds_dkv = 4 * (4 * XTILE + 256);
        tcow_ensure_lds((const void*)attn_x3_bwd_dq<1, true>, lds);
        tcow_ensure_lds((const void*)attn_x3_bwd_dkv<1, true>, lds_dkv);
        hipLaunchKernelGGL((attn_x3_bwd_dq<1, true>), dim3(cdiv(pairs, 4)), dim3(256), lds, st, d, nt, (const float*)qkv, (const float*)out, (const float*)dout, lse, delta, (float*)dqkv);
        TCOW_CHECK_LAUNCH();
        hipLaunchKernelGGL((attn_x3_bwd_dkv<1, true>), dim3(cdiv(pairs, 4)), dim3(256), lds_dkv, st, d, nt, (const float*)qkv, (const float*)dout, lse, (const float*)delta, (float*)dqkv);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    return x3_bwd_launch<X3_CH>(st, d, nt, grid, qkv, out, dout, lse, delta, dqkv);      // (two tiles per chunk measured 437-449 vs 426-439 us: the backward kernels stay at two waves per SIMD either way)
}
