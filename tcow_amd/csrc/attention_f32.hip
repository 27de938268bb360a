// Exact-f32 flash attention on the f32 matrix instruction (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 256 flop / cycle / CU) -- the
// attention of the f32-storage modes (precision='fp32' and 'bf16x3'), same contract as attention_simple.hip:
//   softmax(q k^T * d^-0.5 [causal mask]) v  (vit.py:88-109) over strided token sequences (SeqDesc), forward + backward, no score matrix.
// The one-thread-per-query VALU kernels of attention_simple.hip reach ~20 TFLOP/s (an LDS read per 4 FMAs); these run the same
// arithmetic as 32 x 32 x 2 outer-product steps on the matrix pipe.
//
// Work unit = 32 queries x 32 keys, d = 64.  A workgroup = 4 waves = 4 consecutive 32-row tiles of ONE (sequence, head); the other
// side's tiles stream through LDS two at a time (f32, rows padded to 65 floats: conflict-free both for "lane = row" and "lane = column"
// reads).  MFMA operand map (A: lane (l31, hi) supplies A[i = l31][k = hi], B: B[k = hi][j = l31], D register r of lane (l31, hi) =
// D[crow32(r, hi)][l31]):
//   forward   S[key][query] = K Q^T             A = K tile (LDS, lane = key), B = Q fragment (registers)  -> lane = query, regs = 16 keys
//             O^T[d][query] += V^T P^T          A = V tile (LDS, lane = d),   B = the lane's own probability registers (the contraction
//                                               slot of step s is key crow32(s, hi) -- exactly the key register s of the lane holds)
//   dQ        S, dP = V dO^T as above; dQ^T[d][query] += K^T dS^T with B = the lane's dS registers
//   dK / dV   lane = key: S[query][key] = Q K^T with A = Q tile (LDS), B = K fragment (registers); dV^T += dO^T P, dK^T += Q^T dS
// so probabilities never leave the lane that computed them and every softmax statistic is a per-lane scalar (+ one half-wave exchange).
#include <stdlib.h>

#include "attention_common.h"

namespace {

constexpr int HD = ATT_HD;
constexpr int TP = 65;                 // LDS row pitch in floats
constexpr int TILE_F = 32 * TP;        // floats per staged 32 x 64 tile
constexpr int NC = 2;                  // tiles per chunk and array
constexpr float kScale = 0.125f;       // 64^-0.5
constexpr float kNeg = -1e30f;

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// rows p0 .. p0+31 of a 64-column block (row stride `stride` floats) -> LDS tile; rows >= L are zero.  All 256 threads.
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, long stride, int p0, int L, float* __restrict__ dst, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + 256 * it, r = i >> 4, c4 = (i & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + r < L) v = ld4(src + (size_t)(p0 + r) * stride + c4);
        float* d = dst + r * TP + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

// acc[r] += sum_d T[l31][d] * f[d]  over the 64 channels: A from the LDS tile (lane = tile row), B from the register fragment
__device__ __forceinline__ f32x16 mm_rows(const float* __restrict__ tile, const float (&f)[32], int l31, int hi, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[l31 * TP + 2 * s + hi], f[s], acc, 0, 0, 0);
    return acc;
}
// acc[r] += sum over the tile's 32 rows of T[row][32 dt + l31] * w[row]: A from the LDS tile (lane = channel), B = per-lane weights in
// C-layout order (register s <-> row crow32(s, hi))
__device__ __forceinline__ f32x16 mm_cols(const float* __restrict__ tile, const f32x16& w, int dt, int l31, int hi, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[crow32(s, hi) * TP + 32 * dt + l31], w[s], acc, 0, 0, 0);
    return acc;
}

// fragment of one row of a [*, 64] block: f[s] = row[2 s + hi]
__device__ __forceinline__ void load_frag(const float* __restrict__ row, int hi, float scale, float (&f)[32]) {
#pragma unroll
    for (int s = 0; s < 32; ++s) f[s] = row[2 * s + hi] * scale;
}

// Workgroups that own different 4-tile chunks of the SAME (sequence, head) stream the same tiles of the other side: run them back to back
// on one XCD (blocks are dispatched round-robin over the 8 XCDs), so the re-reads hit that XCD's L2.
struct Work { int pair, chunk; bool valid; };
__device__ __forceinline__ Work work_of(int pairs, int nchunk) {
    const int b = blockIdx.x, x = b & 7, k = b >> 3;
    const int i = k / nchunk;
    Work w; w.chunk = k - i * nchunk; w.pair = 8 * i + x; w.valid = w.pair < pairs;
    return w;
}
inline int grid_of(int pairs, int nchunk) { return 8 * ((pairs + 7) / 8) * nchunk; }

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 2) void attn_f32_fwd(SeqDesc sd, int nt, const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem; float* Vs = smem + NC * TILE_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const Work w = work_of(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (!w.valid) return;
    const int item = w.pair / sd.heads, head = w.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const float* qh = qkv + base * ld3 + head * HD;
    const int qt = w.chunk * 4 + wave;
    const bool active = qt < nt;
    const int q = 32 * qt + l31, qc = q < sd.L ? q : sd.L - 1;
    float qf[32];
    load_frag(qh + (size_t)qc * pse, hi, kScale, qf);
    f32x16 o0 = zero16(), o1 = zero16();
    float m = kNeg, l = 0.f;
    auto tiles_seen = [&](int t) {                                       // key tiles a query tile can see (causal limit)
        const long klim = (long)32 * t + 31 + sd.diag;
        return klim >= (long)sd.L - 1 ? nt : (int)(klim / 32) + 1;
    };
    const int my_end = active ? tiles_seen(qt) : 0;
    const int last_qt = w.chunk * 4 + 3 < nt ? w.chunk * 4 + 3 : nt - 1;
    const int wg_end = tiles_seen(last_qt);
    for (int c0 = 0; c0 < wg_end; c0 += NC) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NC; ++t)
            if (c0 + t < wg_end) {
                stage_tile(qh + sd.D, pse, 32 * (c0 + t), sd.L, Ks + t * TILE_F, tid);
                stage_tile(qh + 2 * sd.D, pse, 32 * (c0 + t), sd.L, Vs + t * TILE_F, tid);
            }
        __syncthreads();
        for (int j = c0; j < c0 + NC && j < my_end; ++j) {
            const float* kt = Ks + (j - c0) * TILE_F;
            const float* vt = Vs + (j - c0) * TILE_F;
            f32x16 s = mm_rows(kt, qf, l31, hi, zero16());              // s[r] = S[query l31][key 32 j + crow32(r, hi)]
            const bool need_mask = (32 * j + 31 >= sd.L) || ((long)32 * j + 31 > (long)32 * qt + sd.diag);
            if (need_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * j + crow32(r, hi);
                    if (key >= sd.L || (long)key > (long)q + sd.diag) s[r] = kNeg;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);
            const float alpha = __expf(m - mn);                           // key 0 is visible to every query, so m is finite after tile 0
            m = mn; l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            f32x16 p;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __expf(s[r] - m); ps += p[r]; }
            l += ps;
            o0 = mm_cols(vt, p, 0, l31, hi, o0);                          // o[dt][r] = O[query l31][d = 32 dt + crow32(r, hi)]
            o1 = mm_cols(vt, p, 1, l31, hi, o1);
        }
    }
    if (!active) return;
    l += __shfl_xor(l, 32, 64);
    if (q < sd.L) {
        const float inv = 1.0f / l;
        const long row = base + (long)q * sd.pos_stride;
        float* orow = out + row * sd.D + head * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4(orow + 8 * g + 4 * hi, make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
            st4(orow + 32 + 8 * g + 4 * hi, make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
        }
        if (lse && hi == 0) lse[row * sd.heads + head] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
__global__ __launch_bounds__(256, 2) void attn_f32_bwd_dq(SeqDesc sd, int nt, const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem; float* Vs = smem + NC * TILE_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const Work w = work_of(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (!w.valid) return;
    const int item = w.pair / sd.heads, head = w.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const float* qh = qkv + base * ld3 + head * HD;
    const int qt = w.chunk * 4 + wave;
    const bool active = qt < nt;
    const int q = 32 * qt + l31, qc = q < sd.L ? q : sd.L - 1;
    const long row = base + (long)qc * sd.pos_stride;
    float qf[32], dof[32];
    load_frag(qh + (size_t)qc * pse, hi, kScale, qf);
    load_frag(dout + row * sd.D + head * HD, hi, 1.0f, dof);
    // delta = rowsum(dO * O) of this lane's query: published for the dK / dV kernel, which runs after this one
    float dl = 0.f;
    {
        const float* orow = o + row * sd.D + head * HD;
#pragma unroll
        for (int s = 0; s < 32; ++s) dl = fmaf(dof[s], orow[2 * s + hi], dl);
        dl += __shfl_xor(dl, 32, 64);
        if (active && q < sd.L && hi == 0) delta[row * sd.heads + head] = dl;
    }
    const float ls = lse[row * sd.heads + head];
    f32x16 dq0 = zero16(), dq1 = zero16();
    auto tiles_seen = [&](int t) {
        const long klim = (long)32 * t + 31 + sd.diag;
        return klim >= (long)sd.L - 1 ? nt : (int)(klim / 32) + 1;
    };
    const int my_end = active ? tiles_seen(qt) : 0;
    const int last_qt = w.chunk * 4 + 3 < nt ? w.chunk * 4 + 3 : nt - 1;
    const int wg_end = tiles_seen(last_qt);
    for (int c0 = 0; c0 < wg_end; c0 += NC) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NC; ++t)
            if (c0 + t < wg_end) {
                stage_tile(qh + sd.D, pse, 32 * (c0 + t), sd.L, Ks + t * TILE_F, tid);
                stage_tile(qh + 2 * sd.D, pse, 32 * (c0 + t), sd.L, Vs + t * TILE_F, tid);
            }
        __syncthreads();
        for (int j = c0; j < c0 + NC && j < my_end; ++j) {
            const float* kt = Ks + (j - c0) * TILE_F;
            const float* vt = Vs + (j - c0) * TILE_F;
            const f32x16 s = mm_rows(kt, qf, l31, hi, zero16());          // S[query][key], scaled
            const f32x16 dp = mm_rows(vt, dof, l31, hi, zero16());        // dP[query][key] = dO . V[key]
            f32x16 ds;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * j + crow32(r, hi);
                const bool ok = key < sd.L && (long)key <= (long)q + sd.diag;
                const float p = ok ? __expf(s[r] - ls) : 0.f;
                ds[r] = p * (dp[r] - dl);
            }
            dq0 = mm_cols(kt, ds, 0, l31, hi, dq0);                       // dQ[query l31][d = 32 dt + crow32(r, hi)] (before the d^-0.5)
            dq1 = mm_cols(kt, ds, 1, l31, hi, dq1);
        }
    }
    if (!active || q >= sd.L) return;
    float* drow = dqkv + row * ld3 + head * HD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        st4(drow + 8 * g + 4 * hi, make_float4(dq0[4 * g] * kScale, dq0[4 * g + 1] * kScale, dq0[4 * g + 2] * kScale, dq0[4 * g + 3] * kScale));
        st4(drow + 32 + 8 * g + 4 * hi, make_float4(dq1[4 * g] * kScale, dq1[4 * g + 1] * kScale, dq1[4 * g + 2] * kScale, dq1[4 * g + 3] * kScale));
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
__global__ __launch_bounds__(256, 2) void attn_f32_bwd_dkv(SeqDesc sd, int nt, const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
                                                            const float* __restrict__ delta, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem; float* Os = smem + NC * TILE_F;
    float* Ls = smem + 2 * NC * TILE_F; float* Dl = Ls + NC * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const Work w = work_of(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (!w.valid) return;
    const int item = w.pair / sd.heads, head = w.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const float* qh = qkv + base * ld3 + head * HD;
    const float* doh = dout + base * sd.D + head * HD;
    const int jt = w.chunk * 4 + wave;
    const bool active = jt < nt;
    const int key = 32 * jt + l31, kc = key < sd.L ? key : sd.L - 1;
    float kf[32], vf[32];
    load_frag(qh + (size_t)kc * pse + sd.D, hi, 1.0f, kf);
    load_frag(qh + (size_t)kc * pse + 2 * sd.D, hi, 1.0f, vf);
    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
    // first query tile that can see any key of this wave / of this workgroup
    auto first_q = [&](int t) { const long qlo = (long)32 * t - sd.diag; return qlo > 0 ? (int)(qlo / 32) : 0; };
    const int my_i0 = first_q(jt);
    const int c_start = first_q(w.chunk * 4) & ~(NC - 1);
    for (int c0 = c_start; c0 < nt; c0 += NC) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NC; ++t)
            if (c0 + t < nt) {
                stage_tile(qh, pse, 32 * (c0 + t), sd.L, Qs + t * TILE_F, tid);
                stage_tile(doh, pso, 32 * (c0 + t), sd.L, Os + t * TILE_F, tid);
            }
        if (tid < NC * 32) {
            const int qi = 32 * c0 + tid;
            const long row = base + (long)(qi < sd.L ? qi : sd.L - 1) * sd.pos_stride;
            Ls[tid] = lse[row * sd.heads + head];
            Dl[tid] = delta[row * sd.heads + head];
        }
        __syncthreads();
        if (!active) continue;
        for (int i = c0 > my_i0 ? c0 : my_i0; i < c0 + NC && i < nt; ++i) {
            const float* qt_ = Qs + (i - c0) * TILE_F;
            const float* ot_ = Os + (i - c0) * TILE_F;
            const f32x16 s = mm_rows(qt_, kf, l31, hi, zero16());         // S[query 32 i + crow32(r, hi)][key l31], unscaled
            const f32x16 dp = mm_rows(ot_, vf, l31, hi, zero16());        // dP[query][key]
            f32x16 p, ds;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = (i - c0) * 32 + crow32(r, hi), qi = 32 * c0 + qr;
                const bool ok = qi < sd.L && key < sd.L && (long)key <= (long)qi + sd.diag;
                p[r] = ok ? __expf(s[r] * kScale - Ls[qr]) : 0.f;
                ds[r] = p[r] * (dp[r] - Dl[qr]);
            }
            dv0 = mm_cols(ot_, p, 0, l31, hi, dv0);                       // dV[key l31][d = 32 dt + crow32(r, hi)]
            dv1 = mm_cols(ot_, p, 1, l31, hi, dv1);
            dk0 = mm_cols(qt_, ds, 0, l31, hi, dk0);                      // dK before the d^-0.5
            dk1 = mm_cols(qt_, ds, 1, l31, hi, dk1);
        }
    }
    if (!active || key >= sd.L) return;
    const long row = base + (long)key * sd.pos_stride;
    float* dkr = dqkv + row * ld3 + sd.D + head * HD;
    float* dvr = dqkv + row * ld3 + 2 * sd.D + head * HD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        st4(dkr + 8 * g + 4 * hi, make_float4(dk0[4 * g] * kScale, dk0[4 * g + 1] * kScale, dk0[4 * g + 2] * kScale, dk0[4 * g + 3] * kScale));
        st4(dkr + 32 + 8 * g + 4 * hi, make_float4(dk1[4 * g] * kScale, dk1[4 * g + 1] * kScale, dk1[4 * g + 2] * kScale, dk1[4 * g + 3] * kScale));
        st4(dvr + 8 * g + 4 * hi, make_float4(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]));
        st4(dvr + 32 + 8 * g + 4 * hi, make_float4(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]));
    }
}


// ------------------------------------------------------------------------------------------------ one-tile sequences (temporal attention, T <= 32)
// A sequence of <= 32 positions is ONE 32 x 32 work unit: in the kernels above three of a workgroup's four waves would idle.  Here every
// wave owns its own (sequence, head): it stages its K / V (or Q / dO) tile in a private LDS region itself -- no workgroup barrier -- and
// consecutive waves take consecutive heads of one site, so their 256-byte row pieces are neighbours in memory.
__device__ __forceinline__ void stage_tile_wave(const float* __restrict__ src, long stride, int L, float* __restrict__ dst, int lane) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int i = lane + 64 * it, r = i >> 4, c4 = (i & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < L) v = ld4(src + (size_t)r * stride + c4);
        float* d = dst + r * TP + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}
// fragment of tile row l31 read back from the wave's staged tile (per-lane row reads straight from global memory touch 32 cache lines per
// 4-byte instruction: at T = 30 that, not the arithmetic, was the kernel)
__device__ __forceinline__ void frag_from_tile(const float* __restrict__ tile, int l31, int hi, float scale, float (&f)[32]) {
#pragma unroll
    for (int s = 0; s < 32; ++s) f[s] = tile[l31 * TP + 2 * s + hi] * scale;
}
#define TCOW_WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")     /* a wave's LDS operations execute in order; this keeps hipcc from reordering them */

__global__ __launch_bounds__(256, 2) void attn_f32_fwd_solo(SeqDesc sd, const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= sd.n_outer * sd.n_inner * sd.heads) return;
    const int item = pair / sd.heads, head = pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const float* qh = qkv + base * ld3 + head * HD;
    // (Q stays a per-lane read from global memory here: a third staged tile would halve the workgroups per CU, measured 189 vs 121 us --
    // this kernel moves 333 MB and sits at its memory time)
    float* kt = smem + wave * (2 * TILE_F); float* vt = kt + TILE_F;
    stage_tile_wave(qh + sd.D, pse, sd.L, kt, lane);
    stage_tile_wave(qh + 2 * sd.D, pse, sd.L, vt, lane);
    const int q = l31, qc = q < sd.L ? q : sd.L - 1;
    float qf[32];
    load_frag(qh + (size_t)qc * pse, hi, kScale, qf);
    TCOW_WAVE_LDS_FENCE();
    f32x16 s = mm_rows(kt, qf, l31, hi, zero16());
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = crow32(r, hi);
        if (key >= sd.L || (long)key > (long)q + sd.diag) s[r] = kNeg;
    }
    float m = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, s[r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    f32x16 p;
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[r] = __expf(s[r] - m); l += p[r]; }
    l += __shfl_xor(l, 32, 64);
    const f32x16 o0 = mm_cols(vt, p, 0, l31, hi, zero16()), o1 = mm_cols(vt, p, 1, l31, hi, zero16());
    if (q < sd.L) {
        const float inv = 1.0f / l;
        const long row = base + (long)q * sd.pos_stride;
        float* orow = out + row * sd.D + head * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4(orow + 8 * g + 4 * hi, make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
            st4(orow + 32 + 8 * g + 4 * hi, make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
        }
        if (lse && hi == 0) lse[row * sd.heads + head] = m + __logf(l);
    }
}

// dQ, dK and dV of a one-tile sequence in one pass: the wave stages all four tiles (K, V, Q, dO), computes the scores in both
// orientations (lane = query for dQ, lane = key for dK / dV) and needs no delta round trip through memory.
__global__ __launch_bounds__(256, 1) void attn_f32_bwd_solo(SeqDesc sd, const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ dout,
                                                             const float* __restrict__ lse, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= sd.n_outer * sd.n_inner * sd.heads) return;
    const int item = pair / sd.heads, head = pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const float* qh = qkv + base * ld3 + head * HD;
    const float* doh = dout + base * sd.D + head * HD;
    float* kt = smem + wave * (4 * TILE_F + 64); float* vt = kt + TILE_F; float* qt_ = vt + TILE_F; float* ot_ = qt_ + TILE_F;
    float* Ls = ot_ + TILE_F; float* Dl = Ls + 32;
    stage_tile_wave(qh + sd.D, pse, sd.L, kt, lane);
    stage_tile_wave(qh + 2 * sd.D, pse, sd.L, vt, lane);
    stage_tile_wave(qh, pse, sd.L, qt_, lane);
    stage_tile_wave(doh, pso, sd.L, ot_, lane);
    const int pos = l31, pc = pos < sd.L ? pos : sd.L - 1;           // this lane's query (dQ side) and key (dK / dV side) position
    const long row = base + (long)pc * sd.pos_stride;
    // delta[q] = rowsum(dO * O): coalesced (16 lanes x 16 B per row), folded over the 16 lanes of a row
    {
        const float* oh = o + base * sd.D + head * HD;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = lane + 64 * it, r = i >> 4, c4 = (i & 15) * 4;
            float part = 0.f;
            if (r < sd.L) {
                const float4 a = ld4(oh + (size_t)r * pso + c4), b = ld4(doh + (size_t)r * pso + c4);
                part = (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
            }
            part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);
            if ((lane & 15) == 0) Dl[r] = part;
        }
        if (lane < 32) Ls[lane] = lse[(base + (long)(lane < sd.L ? lane : sd.L - 1) * sd.pos_stride) * sd.heads + head];
    }
    TCOW_WAVE_LDS_FENCE();
    float qf[32], dof[32];
    frag_from_tile(qt_, l31, hi, kScale, qf);
    frag_from_tile(ot_, l31, hi, 1.0f, dof);
    const float dl = Dl[l31], ls = Ls[l31];
    // ---- dQ: lane = query
    {
        const f32x16 s = mm_rows(kt, qf, l31, hi, zero16());
        const f32x16 dp = mm_rows(vt, dof, l31, hi, zero16());
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = crow32(r, hi);
            const bool ok = key < sd.L && (long)key <= (long)pos + sd.diag;
            const float p = ok ? __expf(s[r] - ls) : 0.f;
            ds[r] = p * (dp[r] - dl);
        }
        const f32x16 dq0 = mm_cols(kt, ds, 0, l31, hi, zero16()), dq1 = mm_cols(kt, ds, 1, l31, hi, zero16());
        if (pos < sd.L) {
            float* drow = dqkv + row * ld3 + head * HD;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                st4(drow + 8 * g + 4 * hi, make_float4(dq0[4 * g] * kScale, dq0[4 * g + 1] * kScale, dq0[4 * g + 2] * kScale, dq0[4 * g + 3] * kScale));
                st4(drow + 32 + 8 * g + 4 * hi, make_float4(dq1[4 * g] * kScale, dq1[4 * g + 1] * kScale, dq1[4 * g + 2] * kScale, dq1[4 * g + 3] * kScale));
            }
        }
    }
    // ---- dK, dV: lane = key
    {
        float kf[32], vf[32];
        frag_from_tile(kt, l31, hi, 1.0f, kf);
        frag_from_tile(vt, l31, hi, 1.0f, vf);
        const f32x16 s = mm_rows(qt_, kf, l31, hi, zero16());          // S[query crow32(r, hi)][key l31], unscaled
        const f32x16 dp = mm_rows(ot_, vf, l31, hi, zero16());
        f32x16 p, ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = crow32(r, hi);
            const bool ok = qi < sd.L && pos < sd.L && (long)pos <= (long)qi + sd.diag;
            p[r] = ok ? __expf(s[r] * kScale - Ls[qi]) : 0.f;
            ds[r] = p[r] * (dp[r] - Dl[qi]);
        }
        const f32x16 dv0 = mm_cols(ot_, p, 0, l31, hi, zero16()), dv1 = mm_cols(ot_, p, 1, l31, hi, zero16());
        const f32x16 dk0 = mm_cols(qt_, ds, 0, l31, hi, zero16()), dk1 = mm_cols(qt_, ds, 1, l31, hi, zero16());
        if (pos < sd.L) {
            float* dkr = dqkv + row * ld3 + sd.D + head * HD;
            float* dvr = dqkv + row * ld3 + 2 * sd.D + head * HD;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                st4(dkr + 8 * g + 4 * hi, make_float4(dk0[4 * g] * kScale, dk0[4 * g + 1] * kScale, dk0[4 * g + 2] * kScale, dk0[4 * g + 3] * kScale));
                st4(dkr + 32 + 8 * g + 4 * hi, make_float4(dk1[4 * g] * kScale, dk1[4 * g + 1] * kScale, dk1[4 * g + 2] * kScale, dk1[4 * g + 3] * kScale));
                st4(dvr + 8 * g + 4 * hi, make_float4(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]));
                st4(dvr + 32 + 8 * g + 4 * hi, make_float4(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]));
            }
        }
    }
}

constexpr int kLdsFwdSolo = 4 * 2 * TILE_F * 4;
constexpr int kLdsBwdSolo = 4 * (4 * TILE_F + 64) * 4;

constexpr int kLdsFwd = 2 * NC * TILE_F * 4;
constexpr int kLdsDkv = (2 * NC * TILE_F + 2 * NC * 32) * 4;

}  // namespace

int tcow_attn_f32_fwd(hipStream_t st, const SeqDesc& d, const void* qkv, void* out, float* lse) {
    const int nt = cdiv(d.L, 32), grid = grid_of(d.n_outer * d.n_inner * d.heads, cdiv(nt, 4));
    if (nt == 1) {
        tcow_ensure_lds((const void*)attn_f32_fwd_solo, kLdsFwdSolo);
        hipLaunchKernelGGL(attn_f32_fwd_solo, dim3(cdiv(d.n_outer * d.n_inner * d.heads, 4)), dim3(256), kLdsFwdSolo, st, d, (const float*)qkv, (float*)out, lse);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    tcow_ensure_lds((const void*)attn_f32_fwd, kLdsFwd);
    hipLaunchKernelGGL(attn_f32_fwd, dim3(grid), dim3(256), kLdsFwd, st, d, nt, (const float*)qkv, (float*)out, lse);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// `delta` = rows * heads floats of workspace (the layout of lse)
int tcow_attn_f32_bwd(hipStream_t st, const SeqDesc& d, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv) {
    const int nt = cdiv(d.L, 32), grid = grid_of(d.n_outer * d.n_inner * d.heads, cdiv(nt, 4));
    if (nt == 1) {
        tcow_ensure_lds((const void*)attn_f32_bwd_solo, kLdsBwdSolo);
        hipLaunchKernelGGL(attn_f32_bwd_solo, dim3(cdiv(d.n_outer * d.n_inner * d.heads, 4)), dim3(256), kLdsBwdSolo, st, d, (const float*)qkv, (const float*)out, (const float*)dout, lse,
                           (float*)dqkv);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    tcow_ensure_lds((const void*)attn_f32_bwd_dq, kLdsFwd);
    tcow_ensure_lds((const void*)attn_f32_bwd_dkv, kLdsDkv);
    hipLaunchKernelGGL(attn_f32_bwd_dq, dim3(grid), dim3(256), kLdsFwd, st, d, nt, (const float*)qkv, (const float*)out, (const float*)dout, lse, delta, (float*)dqkv);
    TCOW_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_f32_bwd_dkv, dim3(grid), dim3(256), kLdsDkv, st, d, nt, (const float*)qkv, (const float*)dout, lse, (const float*)delta, (float*)dqkv);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
