// bf16 MFMA flash-style attention for the divided space-time blocks (gfx950), forward and backward.
//
// Replaces softmax(q k^T / 8 [causal]) v of Attention.forward (vit.py:88-109) without materialising the
// (T,T) / (S,S) score matrices the reference builds (13 MB + 130 MB f32 per block at the default config).
//
// Work unit = one 32-query tile against 32-key tiles, d = 64, on v_mfma_f32_32x32x16_bf16:
//   S^T = K Q^T    (A = K rows from LDS, B = Q rows)      -> lane (q = lane&31, hi) holds 16 keys of its query, so the
//                                                             softmax row statistics are lane-local (+1 exchange with lane^32)
//   O^T += V^T P^T (A = V gathered with ds_read_b64_tr_b16, B = P straight from the S^T accumulator registers)
// Token rows come straight from the qkv GEMM output [rows, 3D] (128-byte head slices, 1 cache line each) through
// direct-to-LDS loads.  LDS tiles are [32 rows][64 bf16]; 16-byte chunk c of row r is stored at chunk position
// c ^ g(r), g(r) = ((r>>1)&1)<<2 | ((r>>2)&3), which is conflict-free for both the ds_read_b128 row fragments and
// the transpose reads (applied on the SOURCE address: the LDS image of a direct-to-LDS load stays lane-linear).
//
//   streaming (spatial, default; any length): a 256-thread workgroup owns 4 query (or key) tiles of one (clip, frame, head),
//                              one per wave, and walks the other side in chunks of 4 tiles through 32 KiB of LDS
//                              (4 workgroups per CU); also used for temporal sequences longer than 64 frames.
//   SHARED = false (temporal): one wave per (clip, slot, head) with wave-private LDS tiles (T <= 64), no barriers.
//   SHARED = true            : earlier variant with the whole sequence resident in LDS (S <= 320), kept for A/B runs
//                              (TCOW_ATTN_SHARED=1); the streaming kernels are 20-50 % faster (profiles/r01_attention.txt).
//
// Backward (recompute from the saved log-sum-exp, delta = rowsum(dO * O) precomputed):
//   dkv kernel: a wave owns key tile j: S = Q K^T and dP = dO V^T in the (rows = q in registers, cols = key in lanes)
//               orientation, so P and dS feed dV += P^T dO, dK += dS^T Q as A operands without any shuffle.
//   dq  kernel: a wave owns query tile i: S^T, dP^T in the forward orientation, dQ += dS K.
#include <stdlib.h>

#include "attention_common.h"

#include "attention_tiles.h"

namespace {

// Temporal sequences skip slot 0 of every frame (the cls replica: SeqDesc.offset = 1, inner_stride = 1), but the GEMMs that consume the attention
// output / produce dqkv read all rows: the wave that owns slot 1 of a clip also defines the slot-0 rows of its head as zero (`sections` blocks of
// 64 columns, D apart) -- this used to be a separate launch per attention call.
__device__ __forceinline__ void zero_prev_slot(const SeqDesc& sd, const WorkId& w, long base, bf16_t* __restrict__ dst, long ld, int sections, int lane) {
    if (w.item % sd.n_inner != 0) return;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int p0 = 0; p0 < sd.L; p0 += 8) {
        const int p = p0 + (lane >> 3);
        if (p < sd.L)
            for (int sec = 0; sec < sections; ++sec)
                *reinterpret_cast<uint4*>(dst + (base - 1 + (long)p * sd.pos_stride) * ld + (long)sec * sd.D + w.head * ATT_HD + (lane & 7) * 8) = z;
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <bool SHARED>
__global__ __launch_bounds__(256, 2) void attn_fwd_mfma(SeqDesc sd, int nt, const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse, int zero0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const WorkId w = work_id<SHARED>(sd, wave);
    if (!SHARED && !w.valid) return;
    const long base = seq_base(sd, w.item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const bf16_t* qh = qkv + base * ld3 + w.head * ATT_HD;
    if (!SHARED && zero0) zero_prev_slot(sd, w, base, out, sd.D, 1, lane);
    // wave-private variant (temporal): K, V AND Q tiles of the sequence in LDS.  Q used to be fetched as MFMA fragments straight from
    // global memory -- 16 bytes per lane from 32 different rows per instruction, a quarter of every cache line per request -- and the
    // result went out as 8-byte pieces per lane; both now move as whole 128-byte rows (direct-to-LDS loads in, store_tile_staged out
    // through the Q tile's space once its fragments are in registers).
    char* kt = SHARED ? smem : smem + wave * (3 * nt * TILE_B);
    char* vt = kt + nt * TILE_B;
    char* qt_ = vt + nt * TILE_B;
    if (SHARED) {
        for (int t = wave; t < nt; t += 4) {
            load_tile(qh + sd.D, pse, 32 * t, sd.L, kt + t * TILE_B, lane);
            load_tile(qh + 2 * sd.D, pse, 32 * t, sd.L, vt + t * TILE_B, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        for (int t = 0; t < nt; ++t) {
            load_tile(qh, pse, 32 * t, sd.L, qt_ + t * TILE_B, lane);
            load_tile(qh + sd.D, pse, 32 * t, sd.L, kt + t * TILE_B, lane);
            load_tile(qh + 2 * sd.D, pse, 32 * t, sd.L, vt + t * TILE_B, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int qt = SHARED ? wave : 0; qt < nt; qt += SHARED ? 4 : 1) {
        const int q = 32 * qt + l31;
        const int qc = q < sd.L ? q : sd.L - 1;
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = SHARED ? frag_row_global(qh, pse, qc, ks, hi) : frag_row(qt_ + qt * TILE_B, l31, ks, hi);
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        float m = -1e30f, l = 0.f;
        const long klim = (long)32 * qt + 31 + sd.diag;            // last key any query of this tile may see
        const int kt_end = klim >= (long)sd.L - 1 ? nt : (int)(klim / 32) + 1;
        for (int j = 0; j < kt_end; ++j) fwd_tile(sd, kt + j * TILE_B, vt + j * TILE_B, qf, j, qt, q, l31, hi, lane, m, l, o0, o1);
        if (SHARED) fwd_store(sd, base, w.head, q, hi, m, l, o0, o1, out, lse);
        else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fwd_store_rows(sd, base, w.head, 32 * qt, lane, m, l, o0, o1, (uint32_t)(uintptr_t)(LDS_PTR(char))(qt_ + qt * TILE_B), out, lse);
        }
    }
}

// Streaming variant for workgroup-shared sequences of ANY length: a workgroup owns 4 query tiles (one per wave) of one
// (item, head) and walks the keys in chunks of 4 tiles (wave w loads K/V tile 4c+w of chunk c), 32 KiB of LDS -> 4
// workgroups per CU.  K/V are re-read from L2 by the ceil(nt/4) workgroups of a sequence.
// Work placement of the streaming kernels.  The workgroups that own different query (key) chunks of the SAME (sequence, head)
// stream the same K / V (Q / dO) tiles, so they should run at the same time on the same XCD (private L2): workgroups are dispatched
// round-robin over the 8 XCDs, so XCD x takes pairs x, x+8, ... and walks the chunks of one pair back to back.  (With the chunk in
// blockIdx.y the sharers ran a whole grid row apart and every chunk re-read its K / V from HBM: FETCH_SIZE 2.3x the algorithmic bytes.)
struct StreamWork { int pair, chunk; bool valid; };
__device__ __forceinline__ StreamWork stream_work(int pairs, int nchunk) {
    const int b = blockIdx.x, x = b & 7, k = b >> 3;
    const int i = k / nchunk;
    StreamWork w; w.chunk = k - i * nchunk; w.pair = 8 * i + x; w.valid = w.pair < pairs;
    return w;
}
static inline int stream_grid(int pairs, int nchunk) { return 8 * ((pairs + 7) / 8) * nchunk; }

template <int CH>
__global__ __launch_bounds__(256, 2) void attn_fwd_stream(SeqDesc sd, int nt, const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse) {
    __shared__ __attribute__((aligned(16))) char smem[2 * CH * TILE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const StreamWork sw_ = stream_work(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (!sw_.valid) return;
    const int item = sw_.pair / sd.heads, head = sw_.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const bf16_t* qh = qkv + base * ld3 + head * ATT_HD;
    char* kt = smem;
    char* vt = smem + CH * TILE_B;
    // the first K / V chunk is requested before anything else: its flight covers the query-fragment loads below
    load_chunk2<CH>(qh + sd.D, pse, qh + 2 * sd.D, pse, 0, nt, sd.L, kt, vt, wave, lane);
    const int qt = sw_.chunk * 4 + wave;
    const bool active = qt < nt;
    const int q = 32 * qt + l31;
    const int qc = q < sd.L ? q : sd.L - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = frag_row_global(qh, pse, qc, ks, hi);
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -1e30f, l = 0.f;
    const long klim = (long)32 * qt + 31 + sd.diag;
    const int kt_end = (!active) ? 0 : (klim >= (long)sd.L - 1 ? nt : (int)(klim / 32) + 1);
    const long klim_wg = (long)32 * (sw_.chunk * 4 + 3) + 31 + sd.diag;          // last key tile any wave of this workgroup needs
    const int kt_end_wg = klim_wg >= (long)sd.L - 1 ? nt : (int)(klim_wg / 32) + 1;
    for (int c0 = 0; c0 < kt_end_wg; c0 += CH) {
        if (c0) {
            __syncthreads();                               // previous chunk fully consumed
            load_chunk2<CH>(qh + sd.D, pse, qh + 2 * sd.D, pse, c0, nt, sd.L, kt, vt, wave, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int jend = (c0 + CH < kt_end) ? c0 + CH : kt_end;
        for (int j = c0; j < jend; ++j) fwd_tile(sd, kt + (j - c0) * TILE_B, vt + (j - c0) * TILE_B, qf, j, qt, q, l31, hi, lane, m, l, o0, o1);
    }
    if (active) fwd_store(sd, base, head, q, hi, m, l, o0, o1, out, lse);
}

// The streaming forward for sequences WITHOUT a causal mask (spatial / joint attention: every call of the training step), on a VALU diet.  The
// streaming kernel above is VALU-throughput-bound at its 2.6-4 waves per SIMD (profiles/r05_attn_fwd_p4.txt: at d = 64 a tile step's softmax
// costs more issue cycles than its 8 MFMAs), so what counts is the number of vector instructions per step -- 87 in fwd_tile:
//   * Q is multiplied by 0.125 log2(e) ONCE, when its fragments are loaded (16-bit result: the scores see one more rounding of q, the saved
//     log-sum-exp stays consistent with the probabilities the forward used), so the exponent needs no scaling;
//   * the running reference maximum sits in the C operand of the first S MFMA -- S' = K Q'^T - m comes out of the matrix pipe and
//     p = exp2(S') is one instruction per element (fwd_tile: one fma + one exp);
//   * no mask arithmetic except on the sequence's last key tile (padding keys).
// (Walking a chunk's tiles by an unrolled loop -- fragment addresses as lane constant + immediate -- was tried: hipcc then carries the accumulators through
// 50 register copies per step and needs 202 registers, one wave per SIMD less.)
// The lazy maximum is fwd_tile's: the reference moves only when some row grew by more than 2^8, and the first key tile always sets it.
__device__ __forceinline__ void fwd_tile_pre(const char* ktile, const char* vtile, const bf16x8 (&qf)[4], bool first, bool pad, bool half, int lr, int l31, int hi, int lane,
                                             float& m, float& l, f32x16& negm, f32x16& o0, f32x16& o1) {
    f32x16 s = TCOW_MFMA_32x32x16_H16(frag_row(ktile, l31, 0, hi), qf[0], negm, 0, 0, 0);      // (negm = 0 until the first key tile has set the reference)
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) s = TCOW_MFMA_32x32x16_H16(frag_row(ktile, l31, ks, hi), qf[ks], s, 0, 0, 0);
    if (pad) {                                              // the sequence's last key tile: padding keys underflow to probability 0
        TCOW_NO_IFCVT();
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (crow32(r, hi) >= lr) ? -1e30f : s[r];
    }
    float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
    mx = half_max(fmaxf(mx, s[15]));                        // row maximum of this tile, relative to the reference
    if (first || __any(mx > 8.0f)) {
        TCOW_NO_IFCVT();
        const float delta = first ? mx : fmaxf(mx, 0.0f);   // (the first tile SETS the reference, later ones only raise it)
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);       // (first tile: l = O = 0)
        m += delta;
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; s[r] -= delta; negm[r] = -m; }
    }
    // `half`: the sequence's last key tile when at most 16 of its keys exist (S = 301: 13).  Keys 0..15 of a tile are registers 0..7 of the S^T
    // accumulator (crow32(r, hi) = 8 (r >> 2) + 4 hi + (r & 3)) and contraction slots 0..15 of the P V product: registers 8..15 are padding (probability
    // exactly 0), so their exponentials and the second k-step of both P V MFMAs are skipped -- half the softmax arithmetic and 6 instead of 8 MFMAs.
    float p[16];
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int r = 0; r < 8; r += 2) { p[r] = __builtin_amdgcn_exp2f(s[r]); p[r + 1] = __builtin_amdgcn_exp2f(s[r + 1]); pa += p[r]; pb += p[r + 1]; }
    const bf16x8 pb0 = pack8(p);
    o0 = TCOW_MFMA_32x32x16_H16(frag_tr(vtile, 0, 0, lane), pb0, o0, 0, 0, 0);
    o1 = TCOW_MFMA_32x32x16_H16(frag_tr(vtile, 0, 1, lane), pb0, o1, 0, 0, 0);
    if (!half) {
        TCOW_NO_IFCVT();
#pragma unroll
        for (int r = 8; r < 16; r += 2) { p[r] = __builtin_amdgcn_exp2f(s[r]); p[r + 1] = __builtin_amdgcn_exp2f(s[r + 1]); pa += p[r]; pb += p[r + 1]; }
        const bf16x8 pb1 = pack8(p + 8);
        o0 = TCOW_MFMA_32x32x16_H16(frag_tr(vtile, 1, 0, lane), pb1, o0, 0, 0, 0);
        o1 = TCOW_MFMA_32x32x16_H16(frag_tr(vtile, 1, 1, lane), pb1, o1, 0, 0, 0);
    }
    l += pa + pb;
}

// Work placement of attn_fwd_stream_nc.  PACK = false: stream_work (a sequence's ceil(nt / 4) workgroups back to back on one XCD).  PACK = true, for
// nt % 4 == 2 (S = 301: ten query tiles = 4 + 4 + 2): the third workgroup of a (frame, head) would run with two idle waves -- a sixth of the wave
// slots of a kernel whose throughput is set by how many waves a SIMD has to switch between.  Two sequences that follow each other on an XCD (pairs p
// and p + 8) form a group of 2 (nt / 4) + 1 workgroups: the full ones of each, and ONE mixed workgroup whose waves 0-1 take the two remaining query
// tiles of the first sequence and waves 2-3 those of the second; it walks the keys two tiles at a time (wave w stages tile c0 + (w & 1) of ITS
// sequence: the same 32 KiB of LDS).  An odd sequence out at the end of an XCD's list keeps the ordinary mapping.
struct NcWork { int pair, qt; bool mixed, valid; };
template <bool PACK>
__device__ __forceinline__ NcWork nc_work(int pairs, int nt, int wave) {
    NcWork w; w.mixed = false;
    if (!PACK) {
        const StreamWork sw_ = stream_work(pairs, (nt + 3) / 4);
        w.pair = sw_.pair; w.qt = sw_.chunk * 4 + wave; w.valid = sw_.valid;
        return w;
    }
    const int nfull = nt >> 2, G = 2 * nfull + 1;
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int ngr = ((pairs + 7) >> 3) >> 1;                 // groups of two per XCD list
    int lp, chunk;
    if (k < ngr * G) {
        const int g = k / G, slot = k - g * G;
        if (slot < nfull) { lp = 2 * g; chunk = slot; }
        else if (slot < 2 * nfull) { lp = 2 * g + 1; chunk = slot - nfull; }
        else { w.mixed = true; lp = 2 * g + (wave >> 1); chunk = nfull; }
    } else { lp = 2 * ngr; chunk = k - ngr * G; }
    w.pair = 8 * lp + x; w.valid = w.pair < pairs;
    w.qt = w.mixed ? 4 * nfull + (wave & 1) : 4 * chunk + wave;
    return w;
}
static inline int nc_grid(int pairs, int nt, bool pack) {
    if (!pack) return stream_grid(pairs, (nt + 3) / 4);
    const int npl = (pairs + 7) >> 3, nfull = nt >> 2;
    return 8 * ((npl >> 1) * (2 * nfull + 1) + (npl & 1) * (nfull + 1));
}

template <int CH, bool PACK>
__global__ __launch_bounds__(256, 2) void attn_fwd_stream_nc(SeqDesc sd, int nt, const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse) {
    __shared__ __attribute__((aligned(16))) char smem[2 * CH * TILE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const NcWork nw = nc_work<PACK>(sd.n_outer * sd.n_inner * sd.heads, nt, wave);
    if (!nw.mixed && !nw.valid) return;                    // (a mixed workgroup whose second sequence does not exist keeps its waves for the barriers ...
    const int pair = nw.valid ? nw.pair : 0;               //  ... and must not form addresses from a sequence index past the end: it reads sequence 0's query rows, no more)
    const int item = pair / sd.heads, head = pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3;
    const bf16_t* qh = qkv + base * ld3 + head * ATT_HD;
    char* kt = smem;
    char* vt = smem + CH * TILE_B;
    const bool mixed = PACK && nw.mixed;
    auto load_pair_tiles = [&](int c0) {                   // mixed workgroup: K / V tile c0 + (wave & 1) of this wave's sequence into slot `wave`
        if (nw.valid) {
            load_tile(qh + sd.D, pse, 32 * (c0 + (wave & 1)), sd.L, kt + wave * TILE_B, lane);
            load_tile(qh + 2 * sd.D, pse, 32 * (c0 + (wave & 1)), sd.L, vt + wave * TILE_B, lane);
        }
    };
    if (mixed) load_pair_tiles(0);
    else load_chunk2<CH>(qh + sd.D, pse, qh + 2 * sd.D, pse, 0, nt, sd.L, kt, vt, wave, lane);
    const int qt = nw.qt;
    const bool active = nw.valid && qt < nt;
    const int q = 32 * qt + l31;
    const int qc = q < sd.L ? q : sd.L - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        typedef __attribute__((ext_vector_type(8))) float f32x8;
        qf[ks] = __builtin_convertvector(__builtin_convertvector(frag_row_global(qh, pse, qc, ks, hi), f32x8) * (kScale * kLog2e), bf16x8);
    }
    f32x16 o0, o1, negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; negm[r] = 0.f; }
    float m = 0.f, l = 0.f;
    const int lr = sd.L - 32 * (nt - 1);                    // valid keys of the last tile
    const bool pad = lr < 32, half_last = lr <= 16;
    const int step = mixed ? 2 : CH, slot0 = mixed ? (wave & 2) : 0;
    for (int c0 = 0; c0 < nt; c0 += step) {
        if (c0) {
            __syncthreads();                               // previous chunk fully consumed
            if (mixed) load_pair_tiles(c0);
            else load_chunk2<CH>(qh + sd.D, pse, qh + 2 * sd.D, pse, c0, nt, sd.L, kt, vt, wave, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int jend = (c0 + step < nt) ? c0 + step : nt;
        if (active)
            for (int j = c0; j < jend; ++j)
                fwd_tile_pre(kt + (slot0 + j - c0) * TILE_B, vt + (slot0 + j - c0) * TILE_B, qf, j == 0, pad && j == nt - 1, half_last && j == nt - 1, lr, l31, hi, lane, m, l, negm,
                             o0, o1);
    }
    if (active) fwd_store(sd, base, head, q, hi, m, l, o0, o1, out, lse);
}

// ------------------------------------------------------------------------------------------------ backward prep
// ld[(item*heads + h)*Lp + q] = (lse, delta), delta = sum_d dO*O  -- packed per sequence so the kernels read it contiguously
__global__ void attn_bwd_prep_kernel(SeqDesc sd, int Lp, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                     float2* __restrict__ ld) {
    const int items = sd.n_outer * sd.n_inner;
    const long total = (long)items * sd.heads * Lp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % Lp); const long ih = i / Lp; const int h = (int)(ih % sd.heads); const int item = (int)(ih / sd.heads);
        float2 v = make_float2(0.f, 0.f);
        if (q < sd.L) {
            const long row = seq_base(sd, item) + (long)q * sd.pos_stride;
            const bf16_t* a = o + row * sd.D + h * ATT_HD; const bf16_t* b = dout + row * sd.D + h * ATT_HD;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < ATT_HD; d += 4) { const float4 x = ld4(a + d), y = ld4(b + d); s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; }
            v = make_float2(lse[row * sd.heads + h] * kLog2e, s);      // (lse in log2 units: the consumers feed it to exp2)
        }
        ld[i] = v;
    }
}

// One 32-query tile against the wave's 32-key tile (backward, dK / dV side).
__device__ __forceinline__ void dkv_tile(const SeqDesc& sd, const char* qtile, const char* dotile, const float2* __restrict__ ldh, int i, int key,
                                         const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], int l31, int hi, int lane, f32x16& dk0, f32x16& dk1, f32x16& dv0, f32x16& dv1,
                                         bf16x8* ds_out = nullptr) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        s = TCOW_MFMA_32x32x16_H16(frag_row(qtile, l31, ks, hi), kf[ks], s, 0, 0, 0);
        dp = TCOW_MFMA_32x32x16_H16(frag_row(dotile, l31, ks, hi), vf[ks], dp, 0, 0, 0);
    }
    // rows of the accumulators are queries q = 32*i + 8*(r>>2) + 4*hi + (r&3); lse/delta for 4 consecutive q per group.
    // (VALU diet: masks only on boundary tiles -- masked scores are pushed to -1e30 so that exp2 underflows to 0; the 1/sqrt(d)
    // factor of dS is applied once to the finished dK / dQ tiles in dkv_store / dq_store instead of per element here.)
    const bool need_mask = (32 * i + 31 >= sd.L) || (key - l31 + 31 >= sd.L) || ((long)(key - l31) + 31 > (long)32 * i + sd.diag);
    if (need_mask) {
        TCOW_NO_IFCVT();
        if ((32 * i + 31 < sd.L) && ((long)(key - l31) + 31 <= (long)32 * i + sd.diag)) {
            // only key padding (the last key tile of a sequence, every step of its wave): one lane-constant test instead of three per element
            const bool kv = key < sd.L;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = kv ? s[r] : -1e30f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = 32 * i + crow32(r, hi);
                if (!(q < sd.L && key < sd.L && (long)key <= (long)q + sd.diag)) s[r] = -1e30f;
            }
        }
    }
    float pv[16], dsv[16];
    const float4* tab = reinterpret_cast<const float4*>(ldh + 32 * i + 4 * hi);     // (lse2, delta) of queries 8 gq + 4 hi + e: two float4 per group
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const float4 a = tab[4 * gq], b = tab[4 * gq + 1];   // (lse0, d0, lse1, d1), (lse2, d2, lse3, d3)
        const float ls[4] = {a.x, a.z, b.x, b.z}, dl[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * gq + e;
            const float sv = s[r];
            const float p = __builtin_amdgcn_exp2f(fmaf(sv, kScale * kLog2e, -ls[e]));
            pv[r] = p;
            dsv[r] = p * (dp[r] - dl[e]);
        }
    }
    const bf16x8 pa0 = pack8(pv), pa1 = pack8(pv + 8), da0 = pack8(dsv), da1 = pack8(dsv + 8);
    if (ds_out) { ds_out[0] = da0; ds_out[1] = da1; }        // (one-kernel backward: the dS block goes to the dQ strip)
    dv0 = TCOW_MFMA_32x32x16_H16(frag_tr(dotile, 0, 0, lane), pa0, dv0, 0, 0, 0);
    dv0 = TCOW_MFMA_32x32x16_H16(frag_tr(dotile, 1, 0, lane), pa1, dv0, 0, 0, 0);
    dv1 = TCOW_MFMA_32x32x16_H16(frag_tr(dotile, 0, 1, lane), pa0, dv1, 0, 0, 0);
    dv1 = TCOW_MFMA_32x32x16_H16(frag_tr(dotile, 1, 1, lane), pa1, dv1, 0, 0, 0);
    dk0 = TCOW_MFMA_32x32x16_H16(frag_tr(qtile, 0, 0, lane), da0, dk0, 0, 0, 0);
    dk0 = TCOW_MFMA_32x32x16_H16(frag_tr(qtile, 1, 0, lane), da1, dk0, 0, 0, 0);
    dk1 = TCOW_MFMA_32x32x16_H16(frag_tr(qtile, 0, 1, lane), da0, dk1, 0, 0, 0);
    dk1 = TCOW_MFMA_32x32x16_H16(frag_tr(qtile, 1, 1, lane), da1, dk1, 0, 0, 0);
}

// The gradient MFMAs are issued as (transposed-read fragment, P or dS), i.e. they accumulate dV^T / dK^T / dQ^T: lane (l31, hi)
// owns ONE token row and per 32-wide d tile the channels d = 8*(r>>2) + 4*hi + (r&3) -- groups of 4 consecutive channels, each
// an 8-byte store (instead of 64 two-byte stores per lane with the untransposed layout).
__device__ __forceinline__ void store_rowT(bf16_t* drow, int hi, const f32x16& a0, const f32x16& a1) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        st4(drow + 8 * gq + 4 * hi, make_float4(a0[4 * gq], a0[4 * gq + 1], a0[4 * gq + 2], a0[4 * gq + 3]));
        st4(drow + 32 + 8 * gq + 4 * hi, make_float4(a1[4 * gq], a1[4 * gq + 1], a1[4 * gq + 2], a1[4 * gq + 3]));
    }
}
__device__ __forceinline__ void dkv_store(const SeqDesc& sd, long base, long ld3, int head, int j, int l31, int hi, const f32x16& dk0, const f32x16& dk1,
                                          const f32x16& dv0, const f32x16& dv1, bf16_t* __restrict__ dqkv) {
    const int kr = 32 * j + l31;
    if (kr < sd.L) {
        bf16_t* drow = dqkv + (base + (long)kr * sd.pos_stride) * ld3 + head * ATT_HD;
        store_rowT(drow + sd.D, hi, dk0 * kScale, dk1 * kScale);        // dS was accumulated without its 1/sqrt(d) factor
        store_rowT(drow + 2 * sd.D, hi, dv0, dv1);
    }
}

// One 32-key tile against the wave's 32-query tile (backward, dQ side).
__device__ __forceinline__ void dq_tile(const SeqDesc& sd, const char* ktile, const char* vtile, int j, int q, const bf16x8 (&qf)[4], const bf16x8 (&dof)[4],
                                        float ls, float dl, int l31, int hi, int lane, f32x16& dq0, f32x16& dq1) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        s = TCOW_MFMA_32x32x16_H16(frag_row(ktile, l31, ks, hi), qf[ks], s, 0, 0, 0);
        dp = TCOW_MFMA_32x32x16_H16(frag_row(vtile, l31, ks, hi), dof[ks], dp, 0, 0, 0);
    }
    const bool need_mask = (32 * j + 31 >= sd.L) || (q - l31 + 31 >= sd.L) || ((long)32 * j + 31 > (long)(q - l31) + sd.diag);
    if (need_mask) {
        TCOW_NO_IFCVT();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * j + crow32(r, hi);
            if (!(q < sd.L && key < sd.L && (long)key <= (long)q + sd.diag)) s[r] = -1e30f;
        }
    }
    float dsv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float sv = s[r];
        const float p = __builtin_amdgcn_exp2f(fmaf(sv, kScale * kLog2e, -ls));
        dsv[r] = p * (dp[r] - dl);
    }
    const bf16x8 da0 = pack8(dsv), da1 = pack8(dsv + 8);
    dq0 = TCOW_MFMA_32x32x16_H16(frag_tr(ktile, 0, 0, lane), da0, dq0, 0, 0, 0);
    dq0 = TCOW_MFMA_32x32x16_H16(frag_tr(ktile, 1, 0, lane), da1, dq0, 0, 0, 0);
    dq1 = TCOW_MFMA_32x32x16_H16(frag_tr(ktile, 0, 1, lane), da0, dq1, 0, 0, 0);
    dq1 = TCOW_MFMA_32x32x16_H16(frag_tr(ktile, 1, 1, lane), da1, dq1, 0, 0, 0);
}

__device__ __forceinline__ void dq_store(const SeqDesc& sd, long base, long ld3, int head, int qt, int l31, int hi, const f32x16& dq0, const f32x16& dq1,
                                         bf16_t* __restrict__ dqkv) {
    const int qr = 32 * qt + l31;
    if (qr < sd.L) store_rowT(dqkv + (base + (long)qr * sd.pos_stride) * ld3 + head * ATT_HD, hi, dq0 * kScale, dq1 * kScale);
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <bool SHARED>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_mfma(SeqDesc sd, int nt, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                         const float2* __restrict__ ld, bf16_t* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const WorkId w = work_id<SHARED>(sd, wave);
    if (!SHARED && !w.valid) return;
    const long base = seq_base(sd, w.item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const bf16_t* qh = qkv + base * ld3 + w.head * ATT_HD;
    const bf16_t* doh = dout + base * sd.D + w.head * ATT_HD;
    const float2* ldh = ld + ((size_t)w.item * sd.heads + w.head) * (nt * 32);
    char* qt_ = SHARED ? smem : smem + wave * (2 * nt * TILE_B);
    char* dot_ = qt_ + nt * TILE_B;
    if (SHARED) {
        for (int t = wave; t < nt; t += 4) {
            load_tile(qh, pse, 32 * t, sd.L, qt_ + t * TILE_B, lane);
            load_tile(doh, pso, 32 * t, sd.L, dot_ + t * TILE_B, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        for (int t = 0; t < nt; ++t) {
            load_tile(qh, pse, 32 * t, sd.L, qt_ + t * TILE_B, lane);
            load_tile(doh, pso, 32 * t, sd.L, dot_ + t * TILE_B, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int j = SHARED ? wave : 0; j < nt; j += SHARED ? 4 : 1) {
        const int key = 32 * j + l31;
        const int kc = key < sd.L ? key : sd.L - 1;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = frag_row_global(qh + sd.D, pse, kc, ks, hi); vf[ks] = frag_row_global(qh + 2 * sd.D, pse, kc, ks, hi); }
        f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
        // first query tile that can see any key of this tile: q >= 32*j - diag
        const long qlo = (long)32 * j - sd.diag;
        const int i0 = qlo > 0 ? (int)(qlo / 32) : 0;
        for (int i = i0; i < nt; ++i) dkv_tile(sd, qt_ + i * TILE_B, dot_ + i * TILE_B, ldh, i, key, kf, vf, l31, hi, lane, dk0, dk1, dv0, dv1);
        dkv_store(sd, base, ld3, w.head, j, l31, hi, dk0, dk1, dv0, dv1, dqkv);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
template <bool SHARED>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_mfma(SeqDesc sd, int nt, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                        const float2* __restrict__ ld, bf16_t* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const WorkId w = work_id<SHARED>(sd, wave);
    if (!SHARED && !w.valid) return;
    const long base = seq_base(sd, w.item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const bf16_t* qh = qkv + base * ld3 + w.head * ATT_HD;
    const bf16_t* doh = dout + base * sd.D + w.head * ATT_HD;
    const float2* ldh = ld + ((size_t)w.item * sd.heads + w.head) * (nt * 32);
    char* kt = SHARED ? smem : smem + wave * (2 * nt * TILE_B);
    char* vt = kt + nt * TILE_B;
    if (SHARED) {
        for (int t = wave; t < nt; t += 4) {
            load_tile(qh + sd.D, pse, 32 * t, sd.L, kt + t * TILE_B, lane);
            load_tile(qh + 2 * sd.D, pse, 32 * t, sd.L, vt + t * TILE_B, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        for (int t = 0; t < nt; ++t) {
            load_tile(qh + sd.D, pse, 32 * t, sd.L, kt + t * TILE_B, lane);
            load_tile(qh + 2 * sd.D, pse, 32 * t, sd.L, vt + t * TILE_B, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int qt = SHARED ? wave : 0; qt < nt; qt += SHARED ? 4 : 1) {
        const int q = 32 * qt + l31;
        const int qc = q < sd.L ? q : sd.L - 1;
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = frag_row_global(qh, pse, qc, ks, hi); dof[ks] = frag_row_global(doh, pso, qc, ks, hi); }
        const float2 lq = ldh[32 * qt + l31];
        const float ls = lq.x, dl = lq.y;
        f32x16 dq0, dq1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
        const long klim = (long)32 * qt + 31 + sd.diag;
        const int kt_end = klim >= (long)sd.L - 1 ? nt : (int)(klim / 32) + 1;
        for (int j = 0; j < kt_end; ++j) dq_tile(sd, kt + j * TILE_B, vt + j * TILE_B, j, q, qf, dof, ls, dl, l31, hi, lane, dq0, dq1);
        dq_store(sd, base, ld3, w.head, qt, l31, hi, dq0, dq1, dqkv);
    }
}

// ------------------------------------------------------------------------------------------------ backward, one tile (L <= 32)
// Temporal attention at T <= 32: the whole sequence of a (site, head) is one 32-position tile, so one wave produces dQ, dK and dV
// from a single visit of Q, K, V, dO (wave-private LDS tiles): one launch and one read of every operand instead of prep + dK/dV + dQ
// kernels (three launches, Q/K/V/dO read twice).  delta = rowsum(dO * O) is NOT read from O: with the whole row of P in one tile,
// rowsum(dO * O) = sum_j P_ij (dO_i . V_j) = sum_j P_ij dP_ij comes out of the accumulators the dQ pass holds anyway (lane = query: 16
// multiply-adds + one half-wave exchange, in f32) -- the O tensor (a sixth of the kernel's bytes, read as 8-byte pieces per lane) is not touched.
__global__ __launch_bounds__(256, 2) void attn_bwd_one_tile(SeqDesc sd, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                         const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int zero0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const WorkId w = work_id<false>(sd, wave);
    if (!w.valid) return;
    const long base = seq_base(sd, w.item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const bf16_t* qh = qkv + base * ld3 + w.head * ATT_HD;
    const bf16_t* doh = dout + base * sd.D + w.head * ATT_HD;
    if (zero0) zero_prev_slot(sd, w, base, dqkv, ld3, 3, lane);
    char* qt_ = smem + wave * (4 * TILE_B + 256);
    char* kt = qt_ + TILE_B; char* vt = kt + TILE_B; char* dot_ = vt + TILE_B;
    float2* ldw = reinterpret_cast<float2*>(dot_ + TILE_B);
    load_tile<LD_NT>(qh, pse, 0, sd.L, qt_, lane);
    load_tile<LD_NT>(qh + sd.D, pse, 0, sd.L, kt, lane);
    load_tile<LD_NT>(qh + 2 * sd.D, pse, 0, sd.L, vt, lane);
    load_tile<LD_NT>(doh, pso, 0, sd.L, dot_, lane);
    const int qc = l31 < sd.L ? l31 : sd.L - 1;
    const float ls = lse[(base + (long)qc * sd.pos_stride) * sd.heads + w.head] * kLog2e;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- dQ (this wave's queries against its keys), which also yields delta
    {
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = frag_row(qt_, l31, ks, hi); dof[ks] = frag_row(dot_, l31, ks, hi); }
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = TCOW_MFMA_32x32x16_H16(frag_row(kt, l31, ks, hi), qf[ks], s, 0, 0, 0);
            dp = TCOW_MFMA_32x32x16_H16(frag_row(vt, l31, ks, hi), dof[ks], dp, 0, 0, 0);
        }
        // (one tile = the whole sequence: always a boundary tile.  Lane (q = l31, hi) holds the keys crow32(r, hi).)
        float pv[16];
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = crow32(r, hi);
            const bool ok = l31 < sd.L && key < sd.L && (long)key <= (long)l31 + sd.diag;
            const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[r], kScale * kLog2e, -ls)) : 0.f;
            pv[r] = p;
            part = fmaf(p, dp[r], part);
        }
        const float dl = half_sum(part);                     // delta_q = sum over ALL keys of P dP
        if (hi == 0) ldw[l31] = l31 < sd.L ? make_float2(ls, dl) : make_float2(0.f, 0.f);
        float dsv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dsv[r] = pv[r] * (dp[r] - dl);
        const bf16x8 da0 = pack8(dsv), da1 = pack8(dsv + 8);
        f32x16 dq0, dq1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
        dq0 = TCOW_MFMA_32x32x16_H16(frag_tr(kt, 0, 0, lane), da0, dq0, 0, 0, 0);
        dq0 = TCOW_MFMA_32x32x16_H16(frag_tr(kt, 1, 0, lane), da1, dq0, 0, 0, 0);
        dq1 = TCOW_MFMA_32x32x16_H16(frag_tr(kt, 0, 1, lane), da0, dq1, 0, 0, 0);
        dq1 = TCOW_MFMA_32x32x16_H16(frag_tr(kt, 1, 1, lane), da1, dq1, 0, 0, 0);
        // K and V fragments of the second pass are taken BEFORE the gradient tiles go out through the K / V tiles' LDS space (whole-row stores)
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = frag_row(kt, l31, ks, hi); vf[ks] = frag_row(vt, l31, ks, hi); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (also: the (lse, delta) table is in LDS; wave-private, no barrier)
        bf16_t* drow0 = dqkv + base * ld3 + w.head * ATT_HD;
        const uint32_t lds_k = (uint32_t)(uintptr_t)(LDS_PTR(char))kt, lds_v = (uint32_t)(uintptr_t)(LDS_PTR(char))vt;
        store_tile_staged(lds_k, lane, kScale, dq0, dq1, drow0, pse, sd.L);
        // ---- dK, dV (this wave's keys against its queries)
        f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
        dkv_tile(sd, qt_, dot_, ldw, 0, l31, kf, vf, l31, hi, lane, dk0, dk1, dv0, dv1);
        store_tile_staged(lds_k, lane, kScale, dk0, dk1, drow0 + sd.D, pse, sd.L);        // dS was accumulated without its 1/sqrt(d) factor
        store_tile_staged(lds_v, lane, 1.0f, dv0, dv1, drow0 + 2 * sd.D, pse, sd.L);
    }
}

// ---- streaming backward kernels (any sequence length): a workgroup owns 4 key tiles (dK/dV) or 4 query tiles (dQ), one per
// wave, and walks the other side in chunks of 4 tiles staged in 32 KiB of LDS (wave w loads tile 4c+w of the chunk).
template <int CH>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_stream(SeqDesc sd, int nt, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                           const float2* __restrict__ ld, bf16_t* __restrict__ dqkv) {
    __shared__ __attribute__((aligned(16))) char smem[2 * CH * TILE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
#ifdef UBENCH_ATTN          // phase stamps of tools/ubench_valu.hip (part E): compiled into the micro-benchmark only
#define STREAM_STAMP(i) do { if (g_attn_dbg && lane == 0) g_attn_dbg[(blockIdx.x * 4 + wave) * 24 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define STREAM_STAMP(i) do { } while (0)
#endif
    const StreamWork sw_ = stream_work(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (!sw_.valid) return;
    STREAM_STAMP(0);
    const int item = sw_.pair / sd.heads, head = sw_.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const bf16_t* qh = qkv + base * ld3 + head * ATT_HD;
    const bf16_t* doh = dout + base * sd.D + head * ATT_HD;
    const float2* ldh = ld + ((size_t)item * sd.heads + head) * (nt * 32);
    char* qt_ = smem;
    char* dot_ = smem + CH * TILE_B;
    const int j = sw_.chunk * 4 + wave;
    const bool active = j < nt;
    const long qlo = (long)32 * j - sd.diag;                       // first query tile that can see this wave's keys
    const int i0 = qlo > 0 ? (int)(qlo / 32) : 0;
    const long qlo_wg = (long)32 * (sw_.chunk * 4) - sd.diag;     // ... and any key of this workgroup
    const int c_start = qlo_wg > 0 ? ((int)(qlo_wg / 32) / CH) * CH : 0;
    // the first Q / dO chunk is requested before the K / V fragment loads: both latencies run together
    load_chunk2<CH>(qh, pse, doh, pso, c_start, nt, sd.L, qt_, dot_, wave, lane);
    STREAM_STAMP(1);
    const int key = 32 * j + l31;
    const int kc = key < sd.L ? key : sd.L - 1;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kf[ks] = frag_row_global(qh + sd.D, pse, kc, ks, hi); vf[ks] = frag_row_global(qh + 2 * sd.D, pse, kc, ks, hi); }
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
    for (int c0 = c_start; c0 < nt; c0 += CH) {
        const int cidx = (c0 - c_start) / CH;
        const int ck = 1 + 5 * (cidx < 3 ? cidx : 3);
        if (c0 != c_start) {
            __syncthreads();
            STREAM_STAMP(ck);
            load_chunk2<CH>(qh, pse, doh, pso, c0, nt, sd.L, qt_, dot_, wave, lane);
        }
        STREAM_STAMP(ck + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STREAM_STAMP(ck + 2);
        __syncthreads();
        STREAM_STAMP(ck + 3);
        if (active) {
            const int ib = c0 > i0 ? c0 : i0, ie = c0 + CH < nt ? c0 + CH : nt;
            for (int i = ib; i < ie; ++i) dkv_tile(sd, qt_ + (i - c0) * TILE_B, dot_ + (i - c0) * TILE_B, ldh, i, key, kf, vf, l31, hi, lane, dk0, dk1, dv0, dv1);
        }
        STREAM_STAMP(ck + 4);
    }
    if (active) dkv_store(sd, base, ld3, head, j, l31, hi, dk0, dk1, dv0, dv1, dqkv);
    STREAM_STAMP(21);
}

// (This kernel runs FIRST in the streaming backward: every wave owns a query tile, so it also computes delta = rowsum(dO * O) of
// its queries and publishes the packed (lse, delta) table that the dK / dV kernel reads -- no separate preparation launch.)
template <int CH>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_stream(SeqDesc sd, int nt, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ lse, float2* __restrict__ ld, bf16_t* __restrict__ dqkv) {
    __shared__ __attribute__((aligned(16))) char smem[2 * CH * TILE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const StreamWork sw_ = stream_work(sd.n_outer * sd.n_inner * sd.heads, (nt + 3) / 4);
    if (!sw_.valid) return;
    STREAM_STAMP(0);
    const int item = sw_.pair / sd.heads, head = sw_.pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const bf16_t* qh = qkv + base * ld3 + head * ATT_HD;
    const bf16_t* doh = dout + base * sd.D + head * ATT_HD;
    float2* ldh = ld + ((size_t)item * sd.heads + head) * (nt * 32);
    char* kt = smem;
    char* vt = smem + CH * TILE_B;
    // the first K / V chunk is requested before the Q / dO / O fragment loads and the delta sums: both latencies run together
    load_chunk2<CH>(qh + sd.D, pse, qh + 2 * sd.D, pse, 0, nt, sd.L, kt, vt, wave, lane);
    STREAM_STAMP(1);
    const int qt = sw_.chunk * 4 + wave;
    const bool active = qt < nt;
    const int q = 32 * qt + l31;
    const int qc = q < sd.L ? q : sd.L - 1;
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qf[ks] = frag_row_global(qh, pse, qc, ks, hi); dof[ks] = frag_row_global(doh, pso, qc, ks, hi); }
    // delta of row q = sum_d dO * O: the dO fragments are already in registers (this half-wave's 32 of the 64 channels); O is
    // fetched with the same fragment pattern
    const bf16_t* oh = o + base * sd.D + head * ATT_HD;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 of = frag_row_global(oh, pso, qc, ks, hi);
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf((float)of[e], (float)dof[ks][e], part);
    }
    const float dl = half_sum(part);
    const float lsn = lse[(base + (long)qc * sd.pos_stride) * sd.heads + head];
    if (active && hi == 0) ldh[32 * qt + l31] = q < sd.L ? make_float2(lsn * kLog2e, dl) : make_float2(0.f, 0.f);
    const float ls = lsn * kLog2e;
    f32x16 dq0, dq1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
    const long klim = (long)32 * qt + 31 + sd.diag;
    const int kt_end = (!active) ? 0 : (klim >= (long)sd.L - 1 ? nt : (int)(klim / 32) + 1);
    const long klim_wg = (long)32 * (sw_.chunk * 4 + 3) + 31 + sd.diag;
    const int kt_end_wg = klim_wg >= (long)sd.L - 1 ? nt : (int)(klim_wg / 32) + 1;
    for (int c0 = 0; c0 < kt_end_wg; c0 += CH) {
        const int ck = 1 + 5 * ((c0 / CH) < 3 ? (c0 / CH) : 3);
        if (c0) {
            __syncthreads();
            STREAM_STAMP(ck);
            load_chunk2<CH>(qh + sd.D, pse, qh + 2 * sd.D, pse, c0, nt, sd.L, kt, vt, wave, lane);
        }
        STREAM_STAMP(ck + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STREAM_STAMP(ck + 2);
        __syncthreads();
        STREAM_STAMP(ck + 3);
        const int jend = (c0 + CH < kt_end) ? c0 + CH : kt_end;
        for (int j = c0; j < jend; ++j) dq_tile(sd, kt + (j - c0) * TILE_B, vt + (j - c0) * TILE_B, j, q, qf, dof, ls, dl, l31, hi, lane, dq0, dq1);
        STREAM_STAMP(ck + 4);
    }
    if (active) dq_store(sd, base, ld3, head, qt, l31, hi, dq0, dq1, dqkv);
    STREAM_STAMP(21);
#undef STREAM_STAMP
}

// ------------------------------------------------------------------------------------------------ backward, ONE kernel per (frame, head) (S <= 320)
// The two streaming kernels above visit every (query tile, key tile) pair twice -- once for dQ, once for dK / dV: 28 MFMAs per pair, the
// scores and dP recomputed, Q / K / V / dO read twice (HBM floor 82 us at configs[1]).  Here ONE 10-wave workgroup owns a (frame, head):
//   * wave w owns key tile w: K_w, V_w fragments from the LDS copies, dK_w / dV_w in 64 accumulator registers, for the whole kernel;
//   * the query side streams: Q_i / dO_i tiles through a double buffer (8 KiB per step, brought in by waves 0-7 one 1 KiB piece each);
//   * per query tile i every wave runs the dK / dV step (dkv_tile: S, dP, P, dS, dV += P^T dO, dK += dS^T Q -- 16 MFMAs) and writes its
//     32 x 32 dS block (bf16) into a [320 keys][32 queries] strip in LDS; after ONE barrier waves 0-7 form dQ_i^T = K^T dS_i in eight
//     16 x 16 output blocks, each a chain of nt v_mfma_16x16x32 over ALL keys (operands by transpose reads: K from its LDS copy, dS from the
//     strip) -- no partial sums across waves, no atomics; 20 MFMA-equivalents per pair instead of 28, every operand read once (floor 53 us).
//   The strip and the Q / dO buffers are double-buffered, so the dQ phase of step i runs while other waves are already in step i+1: one
//   barrier per step.  delta = rowsum(dO * O) and the log-sum-exp go into an LDS table in the prologue (wave w: query tile w).
// LDS: K 40 + V 40 + Q/dO 16 + strip 40 + table 2.5 = 138.5 KiB, one workgroup per CU; 168 VGPRs (three waves on two of the SIMDs).
constexpr int ONE_MAX_NT = 10;
constexpr int ONE_K = 0, ONE_V = ONE_MAX_NT * TILE_B, ONE_QDO = 2 * ONE_MAX_NT * TILE_B, ONE_STRIP = ONE_QDO + 4 * TILE_B;
constexpr int ONE_STRIP_B = ONE_MAX_NT * 32 * 64;                       // [320 keys][32 queries] bf16
constexpr int ONE_TAB = ONE_STRIP + 2 * ONE_STRIP_B, ONE_DQ = ONE_TAB + ONE_MAX_NT * 32 * 8, ONE_LDS = ONE_DQ + 2 * TILE_B;      // + two dQ staging tiles

__global__ __launch_bounds__(768) void attn_bwd_one_kernel(SeqDesc sd, int nt, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse, bf16_t* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int pair = blockIdx.x;
    const int item = pair / sd.heads, head = pair - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D, pse = sd.pos_stride * ld3, pso = sd.pos_stride * sd.D;
    const bf16_t* qh = qkv + base * ld3 + head * ATT_HD;
    const bf16_t* doh = dout + base * sd.D + head * ATT_HD;
    const bf16_t* oh = o + base * sd.D + head * ATT_HD;
    char* ktiles = smem + ONE_K; char* vtiles = smem + ONE_V;
    float2* tab = reinterpret_cast<float2*>(smem + ONE_TAB);
    const bool owner = wave < nt;
    // (the third tile wave of a SIMD -- waves 8, 9 -- gets the issue slots last and is the one everybody waits for at the step barrier: priorities 2 / 1 / 0 for
    // waves 8-9 / 4-7 / 0-3 and the chain waves)
    if (wave >= 8 && wave < 10) __builtin_amdgcn_s_setprio(2); else if (wave >= 4 && wave < 8) __builtin_amdgcn_s_setprio(1);
#ifdef UBENCH_ATTN          // phase stamps of tools/ubench_valu.hip (part F): compiled into the micro-benchmark only
#define ONE_STAMP(i) do { if (g_attn_dbg && lane == 0) g_attn_dbg[((long)blockIdx.x * 12 + wave) * 40 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define ONE_STAMP(i) do { } while (0)
#endif
    ONE_STAMP(0);
    // ---- prologue: K_w / V_w tiles, the first two Q / dO tiles (a 1 KiB quarter per wave 0-7), the (lse, delta) table of query tile w
    if (owner) {
        load_tile<LD_NT>(qh + sd.D, pse, 32 * wave, sd.L, ktiles + wave * TILE_B, lane);
        load_tile<LD_NT>(qh + 2 * sd.D, pse, 32 * wave, sd.L, vtiles + wave * TILE_B, lane);
    }
    auto load_qdo = [&](int i, int buf) {                     // waves 0-3: quarter `wave` of Q_i, waves 4-7: quarter `wave - 4` of dO_i
        char* dst = smem + ONE_QDO + buf * (2 * TILE_B);
        if (wave < 4) load_tile_chunk<LD_NT>(qh, pse, 32 * i, sd.L, dst, wave, lane);
        else if (wave < 8) load_tile_chunk<LD_NT>(doh, pso, 32 * i, sd.L, dst + TILE_B, wave - 4, lane);
    };
    load_qdo(0, 0);
    if (nt > 1) load_qdo(1, 1);
    if (owner) {
        // delta_q = sum_d dO * O, 8 lanes per row (one 16-byte piece each: every load instruction takes 8 whole rows), three butterfly steps.
        // All nine loads of the wave are issued before the first use (in a loop hipcc waits for each row group's loads in turn: four
        // dependent round trips, 26 000 cycles of the prologue in the first timeline of this kernel).
        uint4 xo[4], yo[4]; float ls4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int q = 32 * wave + 8 * j + (lane >> 3); q = q < sd.L ? q : sd.L - 1;
            xo[j] = *reinterpret_cast<const uint4*>(oh + (long)q * pso + (lane & 7) * 8); yo[j] = *reinterpret_cast<const uint4*>(doh + (long)q * pso + (lane & 7) * 8);
            ls4[j] = lse[(base + (long)q * sd.pos_stride) * sd.heads + head];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 32 * wave + 8 * j + (lane >> 3);
            const uint4 x = xo[j], y = yo[j];
            float part = bflo(x.x) * bflo(y.x) + bfhi(x.x) * bfhi(y.x) + bflo(x.y) * bflo(y.y) + bfhi(x.y) * bfhi(y.y)
                       + bflo(x.z) * bflo(y.z) + bfhi(x.z) * bfhi(y.z) + bflo(x.w) * bflo(y.w) + bfhi(x.w) * bfhi(y.w);
            part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
            if ((lane & 7) == 0) tab[q] = q < sd.L ? make_float2(ls4[j] * kLog2e, part) : make_float2(0.f, 0.f);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    ONE_STAMP(1);

    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
    const int key = 32 * wave + l31;
    // dQ phase (waves 0-7): output block = queries 16 qb .. +15 x channels 16 db .. +15 of the step's tile, as dQ^T (lane: query l & 15,
    // channels 16 db + 4 (l >> 4) .. + 3).  Transpose-read addressing: in its 16-lane group lane 4 r + c supplies row r / 4-element quad c.
    // dQ phase: TWO MORE WAVES (10, 11 -- they land on the two SIMDs that host two key-tile waves, wave w sits on SIMD w % 4) do nothing else: behind
    // the barrier of step i they turn the strip into dQ_i^T while waves 0-9 are already in step i+1 -- the chains no longer sit between two tile
    // steps of the same wave (timeline in profiles/r04_ubench_valu.txt part F: a step cost tile arithmetic 2 700 + chains 2 600 + barrier wait).
    // Chain wave c = wave - 10 owns channel blocks 2c, 2c + 1 (16 channels each) x both query halves: four independent accumulate chains of nt
    // 16x16x32 MFMAs over all key tiles; K and strip fragments by transpose reads (in its 16-lane group lane 4 r + q supplies row r / quad q).
    // Output lane: query l & 15 (+ 16 for the second half), channels 16 db + 4 (l >> 4) .. + 3.
    const bool chain_wave = wave >= 10;
    const int cw = wave - 10;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
    // strip write of this wave's dS block: lane (key l31, hi) holds queries 4 hi + {0..3}, 8 + .., 16 + .., 24 + ..: four 8-byte quads (slots
    // hi, 2 + hi, 4 + hi, 6 + hi of the key's 64-byte row; slot s of key row k sits at s ^ ((k >> 1) & 7): a ds_write_b64 is served in groups of 16
    // consecutive lanes over 32 banks -- rows of equal parity share their banks, so the eight of a group must differ in the slot --, and the chain
    // waves' transpose reads in groups of 32 lanes over 64 banks: rows 8 g4 + tr, g4 = 0 / 1, must differ in bit 2 of the slot.  With k & 7, as in
    // round 4, rows k and k + 8 met on one bank in both: SQ_LDS_BANK_CONFLICT = 17 % of the LDS cycles, profiles/r04_pmc_attn.txt.)
    const uint32_t sw = lds0 + ONE_STRIP + (32 * wave + l31) * 64;
    const int k7 = (l31 >> 1) & 7;

    if (chain_wave) {
        // ---- the chain waves' own loop (a separate one: their 80 registers of K^T fragments must not be live across the tile-step code)
        typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        const int g4 = lane >> 4, tr = (lane & 15) >> 2, tc = lane & 3;
        const int krow = 8 * g4 + tr;                                                        // key row inside a 32-key tile (second read: + 4)
        const int kchunk = 4 * cw + (tc >> 1);                                               // channel block 2 cw; block 2 cw + 1 = chunk + 2 = offset ^ 32
        const uint32_t ko0 = ONE_K + krow * 128 + ((kchunk ^ swz_g(krow)) << 4) + (tc & 1) * 8, ko1 = ONE_K + (krow + 4) * 128 + ((kchunk ^ swz_g(krow + 4)) << 4) + (tc & 1) * 8;
        const uint32_t so0 = ONE_STRIP + krow * 64 + ((tc ^ ((krow >> 1) & 7)) << 3);        // query half 0; half 1 = slot ^ 4 = offset ^ 32
        const uint32_t so1 = ONE_STRIP + (krow + 4) * 64 + ((tc ^ (((krow + 4) >> 1) & 7)) << 3);
        // K^T fragments of ALL key tiles, once: they are the same in every step (80 registers the tile-step waves do not have to spare)
        u32x2_ kfr[ONE_MAX_NT][4];
#pragma unroll
        for (int kt = 0; kt < ONE_MAX_NT; ++kt) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(kfr[kt][0]) : "v"(lds0 + ko0 + kt * TILE_B));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(kfr[kt][1]) : "v"(lds0 + ko1 + kt * TILE_B));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(kfr[kt][2]) : "v"(lds0 + (ko0 ^ 32u) + kt * TILE_B));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(kfr[kt][3]) : "v"(lds0 + (ko1 ^ 32u) + kt * TILE_B));
            if ((kt & 1) == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t stg = lds0 + ONE_DQ + cw * TILE_B;
        for (int i = 0; i < nt; ++i) {
            const int buf = i & 1;
            ONE_STAMP(2 + 3 * i);
            __syncthreads();                                            // barrier of step i: the strip of step i is complete
            ONE_STAMP(3 + 3 * i);
            f32x4 acc[2][2];                                            // [channel block][query half]
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_) acc[a_][b_] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const uint32_t sa0 = lds0 + so0 + buf * ONE_STRIP_B, sa1 = lds0 + so1 + buf * ONE_STRIP_B;
            const uint32_t sb0 = lds0 + (so0 ^ 32u) + buf * ONE_STRIP_B, sb1 = lds0 + (so1 ^ 32u) + buf * ONE_STRIP_B;
            u32x2_ fr[3][4];                                            // per set: strip half 0 (2 reads), half 1 (2)
#define ONE_RD(set, kt_)                                                                                               \
            do {                                                                                                       \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fr[set][0]) : "v"(sa0 + (kt_) * 2048));                \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fr[set][1]) : "v"(sa1 + (kt_) * 2048));                \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fr[set][2]) : "v"(sb0 + (kt_) * 2048));                \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fr[set][3]) : "v"(sb1 + (kt_) * 2048));                \
            } while (0)
#define ONE_KF(kt_, j) __builtin_bit_cast(bf16x8, (u32x4_){kfr[kt_][2 * (j)].x, kfr[kt_][2 * (j)].y, kfr[kt_][2 * (j) + 1].x, kfr[kt_][2 * (j) + 1].y})
#define ONE_SF(set, j) __builtin_bit_cast(bf16x8, (u32x4_){fr[set][2 * (j)].x, fr[set][2 * (j)].y, fr[set][2 * (j) + 1].x, fr[set][2 * (j) + 1].y})
#define ONE_MF(set, kt_)                                                                                               \
            do {                                                                                                       \
                acc[0][0] = TCOW_MFMA_16x16x32_H16(ONE_KF(kt_, 0), ONE_SF(set, 0), acc[0][0], 0, 0, 0);                \
                acc[1][0] = TCOW_MFMA_16x16x32_H16(ONE_KF(kt_, 1), ONE_SF(set, 0), acc[1][0], 0, 0, 0);                \
                acc[0][1] = TCOW_MFMA_16x16x32_H16(ONE_KF(kt_, 0), ONE_SF(set, 1), acc[0][1], 0, 0, 0);                \
                acc[1][1] = TCOW_MFMA_16x16x32_H16(ONE_KF(kt_, 1), ONE_SF(set, 1), acc[1][1], 0, 0, 0);                \
            } while (0)
#define ONE_WAIT(set, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(fr[set][0]), "+v"(fr[set][1]), "+v"(fr[set][2]), "+v"(fr[set][3]) :: "memory")
            // (fully unrolled over the ten key tiles: the K fragments are register arrays; tiles past nt - 1 are skipped)
            ONE_RD(0, 0); ONE_RD(1, 1);
#pragma unroll
            for (int kt = 0; kt < ONE_MAX_NT; ++kt) {
                if (kt < nt) {
                    if (kt % 3 == 0) { ONE_RD(2, kt + 2); ONE_WAIT(0, 8); ONE_MF(0, kt); }
                    else if (kt % 3 == 1) { ONE_RD(0, kt + 2); ONE_WAIT(1, 8); ONE_MF(1, kt); }
                    else { ONE_RD(1, kt + 2); ONE_WAIT(2, 8); ONE_MF(2, kt); }
                }
            }
            // (the last key tiles have requested strip fragments two tiles past the end into the three sets: the wait re-defines them, so that hipcc -- which
            // knows nothing of reads issued by asm statements -- cannot reuse a register the late data will still land on; cf. gemm_nt_c2.hip)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fr[0][0]), "+v"(fr[0][1]), "+v"(fr[0][2]), "+v"(fr[0][3]), "+v"(fr[1][0]), "+v"(fr[1][1]), "+v"(fr[1][2]), "+v"(fr[1][3]),
                           "+v"(fr[2][0]), "+v"(fr[2][1]), "+v"(fr[2][2]), "+v"(fr[2][3])
                         :: "memory");
#undef ONE_WAIT
#undef ONE_RD
#undef ONE_MF
#undef ONE_KF
#undef ONE_SF
            // the wave's half of the dQ tile ([32 q][channels 32 cw .. + 31]) through its PRIVATE staging tile (a wave's LDS operations complete in
            // order: no barrier), then out as 64-byte row pieces: 16-byte chunk c of row q sits at position c ^ (q & 7) of the row's 128 bytes
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_) {
                    const int ql = 16 * b_ + (lane & 15), slot = 4 * (2 * cw + a_) + g4;      // 8-byte slot of the row: channels 4 slot .. + 3
                    const uint32_t da = stg + ql * 128 + (((slot >> 1) ^ (ql & 7)) << 4) + ((slot & 1) << 3);
                    const u32x2_ pk = {pack_bf2(acc[a_][b_][0] * kScale, acc[a_][b_][1] * kScale), pack_bf2(acc[a_][b_][2] * kScale, acc[a_][b_][3] * kScale)};
                    asm volatile("ds_write_b64 %0, %1" :: "v"(da), "v"(pk) : "memory");
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 16 * j + (lane >> 2), c = 4 * cw + (lane & 3);
                u32x4_ v;
                asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(stg + r * 128 + ((c ^ (r & 7)) << 4)) : "memory");
                const int q = 32 * i + r;
                if (q < sd.L) *reinterpret_cast<u32x4_*>(dqkv + (base + (long)q * sd.pos_stride) * ld3 + head * ATT_HD + c * 8) = v;
            }
            ONE_STAMP(4 + 3 * i);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                                // (the owners' final barrier: the K / V tiles become their staging space)
        ONE_STAMP(32); ONE_STAMP(33);
        return;
    }

    for (int i = 0; i < nt; ++i) {
        const int buf = i & 1;
        const char* qtile = smem + ONE_QDO + buf * (2 * TILE_B);
        const char* dotile = qtile + TILE_B;
        if (owner) {
            bf16x8 kf[4], vf[4], dsb[2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { kf[ks] = frag_row(ktiles + wave * TILE_B, l31, ks, hi); vf[ks] = frag_row(vtiles + wave * TILE_B, l31, ks, hi); }
            dkv_tile(sd, qtile, dotile, tab, i, key, kf, vf, l31, hi, lane, dk0, dk1, dv0, dv1, dsb);
            typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
            typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ w0 = __builtin_bit_cast(u32x4_, dsb[0]), w1 = __builtin_bit_cast(u32x4_, dsb[1]);
            const uint32_t sb = sw + buf * ONE_STRIP_B;
            asm volatile("ds_write_b64 %0, %1" :: "v"(sb + ((hi ^ k7) << 3)), "v"((u32x2_){w0.x, w0.y}) : "memory");
            asm volatile("ds_write_b64 %0, %1" :: "v"(sb + (((2 + hi) ^ k7) << 3)), "v"((u32x2_){w0.z, w0.w}) : "memory");
            asm volatile("ds_write_b64 %0, %1" :: "v"(sb + (((4 + hi) ^ k7) << 3)), "v"((u32x2_){w1.x, w1.y}) : "memory");
            asm volatile("ds_write_b64 %0, %1" :: "v"(sb + (((6 + hi) ^ k7) << 3)), "v"((u32x2_){w1.z, w1.w}) : "memory");
        }
        // this wave's piece of tile i+1 has landed, its LDS traffic of this step is done: behind the barrier the strip of step i is complete,
        // tile i+1 is visible and buffer `buf` may take tile i+2
        ONE_STAMP(2 + 3 * i);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        ONE_STAMP(3 + 3 * i);
        if (i + 2 < nt) load_qdo(i + 2, buf);
        ONE_STAMP(4 + 3 * i);
    }
    // dK / dV as whole rows through the waves' own K / V tiles (dead once the chain waves have passed this barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    ONE_STAMP(32);
    if (owner) {
        bf16_t* drow0 = dqkv + (base + (long)(32 * wave) * sd.pos_stride) * ld3 + head * ATT_HD;
        store_tile_staged(lds0 + ONE_K + wave * TILE_B, lane, kScale, dk0, dk1, drow0 + sd.D, pse, sd.L - 32 * wave);      // dS was accumulated without its 1/sqrt(d) factor
        store_tile_staged(lds0 + ONE_V + wave * TILE_B, lane, 1.0f, dv0, dv1, drow0 + 2 * sd.D, pse, sd.L - 32 * wave);
    }
    ONE_STAMP(33);
#undef ONE_STAMP
}

template <typename K>
static void set_lds_attr(K kernel, int bytes) {
    tcow_ensure_lds(reinterpret_cast<const void*>(kernel), bytes);
}

}  // namespace

// nt = number of 32-position tiles.  Workgroup-shared (spatial) sequences and temporal sequences of more than 64 positions take the streaming
// kernels; shorter temporal sequences the wave-private ones.  (The resident / persistent forward variants of round 3 are in
// tools/attn_fwd_variants.inc, the whole-sequence-resident SHARED = true instantiations of the *_mfma kernels are no longer built.)
// streaming backward kernels: tiles per LDS chunk -- 5 when that saves a chunk round (nt = 10: two rounds instead of three), else 4
static bool stream_ch5(int nt) { return (nt + 4) / 5 < (nt + 3) / 4; }

bool tcow_attn_mfma_supported(const SeqDesc& d, bool shared) {
    (void)d; (void)shared;
    return true;      // every length: wave-private tiles for temporal sequences up to T = 64, the streaming kernels otherwise
}

// true when the kernel this shape dispatches to writes the zero rows of the skipped slot 0 itself (wave-private temporal kernels)
bool tcow_attn_mfma_zeroes_slot0(const SeqDesc& d, bool shared, bool backward) {
    const int nt = (d.L + 31) / 32;
    if (shared || d.offset != 1 || d.inner_stride != 1 || d.n_inner < 1) return false;
    return backward ? nt == 1 : nt <= 2;
}

int tcow_attn_mfma_fwd(hipStream_t st, const SeqDesc& d, bool shared, const void* qkv, void* out, float* lse) {
    const int nt = (d.L + 31) / 32;
    const int pairs = d.n_outer * d.n_inner * d.heads;
    if (shared || nt > 2) {
        // (forward: four tiles per chunk -- with five, 40 KiB per workgroup, the fourth workgroup of a CU no longer fits and 56 us become 60)
        if (d.diag >= (1 << 27) && nt >= 2) {
            if (nt % 4 == 2) hipLaunchKernelGGL((attn_fwd_stream_nc<4, true>), dim3(nc_grid(pairs, nt, true)), dim3(256), 0, st, d, nt, (const bf16_t*)qkv, (bf16_t*)out, lse);
            else hipLaunchKernelGGL((attn_fwd_stream_nc<4, false>), dim3(nc_grid(pairs, nt, false)), dim3(256), 0, st, d, nt, (const bf16_t*)qkv, (bf16_t*)out, lse);
        }
        else hipLaunchKernelGGL(attn_fwd_stream<4>, dim3(stream_grid(pairs, cdiv(nt, 4))), dim3(256), 0, st, d, nt, (const bf16_t*)qkv, (bf16_t*)out, lse);
    } else {
        const int lds = 4 * 3 * nt * TILE_B;
        set_lds_attr(attn_fwd_mfma<false>, lds);
        hipLaunchKernelGGL(attn_fwd_mfma<false>, dim3(cdiv(pairs, 4)), dim3(256), lds, st, d, nt, (const bf16_t*)qkv, (bf16_t*)out, lse, tcow_attn_mfma_zeroes_slot0(d, shared, false) ? 1 : 0);
    }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

long tcow_attn_mfma_bwd_workspace_bytes(const SeqDesc& d) {
    const int nt = (d.L + 31) / 32;
    return (long)d.n_outer * d.n_inner * d.heads * nt * 32 * 8;
}

int tcow_attn_mfma_bwd(hipStream_t st, const SeqDesc& d, bool shared, const void* qkv, const void* out, const void* dout, const float* lse, void* ws,
                       void* dqkv) {
    const int nt = (d.L + 31) / 32;
    const int pairs = d.n_outer * d.n_inner * d.heads;
    if (!shared && nt == 1) {
        const int lds = 4 * (4 * TILE_B + 256);
        set_lds_attr(attn_bwd_one_tile, lds);
        hipLaunchKernelGGL(attn_bwd_one_tile, dim3(cdiv(pairs, 4)), dim3(256), lds, st, d, (const bf16_t*)qkv, (const bf16_t*)dout, lse, (bf16_t*)dqkv, tcow_attn_mfma_zeroes_slot0(d, shared, true) ? 1 : 0);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    float2* ld = (float2*)ws;
    const long total = (long)pairs * nt * 32;
    int blocks = cdiv(total, 256); if (blocks > 8192) blocks = 8192;
    // spatial sequences of four to ten tiles: the one-kernel backward (221 -> 150 us at S = 301 against the two streaming kernels, which longer
    // sequences keep)
    if (shared && nt <= ONE_MAX_NT && nt >= 4) {
        set_lds_attr(attn_bwd_one_kernel, ONE_LDS);
        hipLaunchKernelGGL(attn_bwd_one_kernel, dim3(pairs), dim3(768), ONE_LDS, st, d, nt, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    if (shared || nt > 2) {
        const dim3 sg(stream_grid(pairs, cdiv(nt, 4)));
        if (stream_ch5(nt)) {
            hipLaunchKernelGGL(attn_bwd_dq_stream<5>, sg, dim3(256), 0, st, d, nt, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, ld, (bf16_t*)dqkv);
            TCOW_CHECK_LAUNCH();
            hipLaunchKernelGGL(attn_bwd_dkv_stream<5>, sg, dim3(256), 0, st, d, nt, (const bf16_t*)qkv, (const bf16_t*)dout, ld, (bf16_t*)dqkv);
        } else {
            hipLaunchKernelGGL(attn_bwd_dq_stream<4>, sg, dim3(256), 0, st, d, nt, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, ld, (bf16_t*)dqkv);
            TCOW_CHECK_LAUNCH();
            hipLaunchKernelGGL(attn_bwd_dkv_stream<4>, sg, dim3(256), 0, st, d, nt, (const bf16_t*)qkv, (const bf16_t*)dout, ld, (bf16_t*)dqkv);
        }
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(blocks), dim3(256), 0, st, d, nt * 32, (const bf16_t*)out, (const bf16_t*)dout, lse, ld);
    TCOW_CHECK_LAUNCH();
    {
        const int lds = 4 * 2 * nt * TILE_B;
        set_lds_attr(attn_bwd_dkv_mfma<false>, lds); set_lds_attr(attn_bwd_dq_mfma<false>, lds);
        hipLaunchKernelGGL(attn_bwd_dkv_mfma<false>, dim3(cdiv(pairs, 4)), dim3(256), lds, st, d, nt, (const bf16_t*)qkv, (const bf16_t*)dout, ld, (bf16_t*)dqkv);
        TCOW_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_bwd_dq_mfma<false>, dim3(cdiv(pairs, 4)), dim3(256), lds, st, d, nt, (const bf16_t*)qkv, (const bf16_t*)dout, ld, (bf16_t*)dqkv);
    }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
