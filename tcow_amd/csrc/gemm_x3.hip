// f32-storage GEMMs on the bf16 matrix cores: every f32 operand element x is split in registers into two bf16 numbers,
//   hi = bf16(x) (round to nearest even), lo = bf16(x - hi)   =>   |x - hi - lo| <= 2^-18 |x|,
// and a product is accumulated in f32 as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  (three v_mfma_f32_32x32x16_bf16; the dropped
// a_lo*b_lo term is <= 2^-18 of the product).  Per product that is ~1e-5 relative against 4e-3 for plain bf16 operands and 6e-8 for
// the exact-f32 MFMA of gemm_f32.hip, at 3/16 of the exact kernel's matrix-pipe time (v_mfma_f32_32x32x2_f32: 4 096 flop per 16
// passes; the bf16 instruction: 32 768 per 8).  It is the `TCOW_F32X3` arithmetic of tcow_gemm_nt / tcow_gemm_tn: operands, outputs and
// every fused epilogue are exactly those of TCOW_F32 (same parameter block), so the precision='bf16x3' mode of the module is the fp32
// mode with faster GEMMs -- mask logits within ~2e-5 of the reference instead of ~1e-6 (tests/test_gpu_seeker.py), 2x the step rate.
//
// One kernel, two loaders: 128 x 128 output tile, 4 waves (64 x 64 each = 2 x 2 MFMA tiles), k walked 32 at a time.  Global f32
// -> registers (the next slice is requested before the MFMAs of the current one) -> split -> four bf16 LDS planes (A hi / lo, B hi /
// lo; rows of 32 k = 64 B padded to 80 B: conflict-free ds_read_b128 fragment reads).  HBM / L2 traffic per MFMA is 2/3 of the
// bf16 kernel's (4-byte operands, three MFMAs per fragment pair), so the operand stream that bounds the bf16 GEMMs does not bound this one
// harder.  Three loaders (below): float4 along k (NT form: forward / input gradient), float4 along the rows (TN form: weight gradient,
// contraction over token rows), and 4-byte loads for any strides / alignment.
#include "gemm_f32.h"

namespace {

constexpr int XT = 128;              // tile edge
constexpr int XK = 32;               // k-slice
constexpr int XP = 80;               // LDS row pitch in bytes (32 bf16 + 16 B pad)
constexpr int XPLANE = XT * XP;      // 10 KiB per plane

// (always bfloat16 splits, whatever 16-bit format the rest of the library is built for: bf16 keeps the f32 exponent range, so lo never
// underflows)
typedef __attribute__((ext_vector_type(8))) __bf16 x3_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 x3_bf16x2;
__device__ __forceinline__ uint32_t x3_pack(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, x3_bf16x2));
}
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& l) {
    h = x3_pack(x0, x1);
    l = x3_pack(x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xffff0000u));     // exact differences (hi shares the leading bits of x)
}

// Loaders of one 128-row x 32-k operand slice into 16 registers per thread: nothing but loads from clamped addresses -- a guarded load per
// basic block, or a select on the loaded value, makes hipcc wait for the data right behind the load and the prefetch is gone.  Elements
// past the end of the contraction range are zeroed when the slice is split (mask_tail, last slice only).  The host picks per launch:
//   LD_KVEC  contraction index contiguous (NT form), 16-byte aligned rows, K % 4 == 0:   float4 along k
//   LD_RVEC  row index contiguous (TN form), 16-byte aligned k-rows, nrows % 4 == 0:      float4 along the rows, two k per thread item
//   LD_ANY   any strides / alignment: 4-byte loads
enum { LD_ANY = 0, LD_KVEC = 1, LD_RVEC = 2 };

// (the registers of a slice are kept as the four 16-byte vectors the loads return and are only taken apart when the slice is split:
// loop-carried scalars made hipcc allocate the loads of one half of the 2x-unrolled loop to other registers and copy -- behind a full wait)
template <int LD>
__device__ __forceinline__ void load_regs(const float* __restrict__ P, long s_row, long s_k, int row0, int nrows, int k0, int K, f32x4 (&v)[4], int tid) {
    if (LD == LD_KVEC) {
        // thread = (row r + 32 pass, 4 consecutive k): 8 lanes cover a row's 128 B.   v[ps] = k .. k+3 of row r + 32 ps
        const int kq = k0 + (tid & 7) * 4, r = tid >> 3;
        const int kc = kq < K - 4 ? kq : K - 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            int gr = row0 + r + 32 * ps; gr = gr < nrows ? gr : nrows - 1;
            v[ps] = *reinterpret_cast<const f32x4*>(P + (size_t)gr * s_row + kc);
        }
    } else if (LD == LD_RVEC) {
        // thread item = (4 consecutive rows, one k pair); lanes of a wave = 8 row quads x 8 k pairs (128-B global segments; the packed
        // 4-byte LDS writes of a wave then fall on 32 distinct banks twice).   v[2 ps] = rows rq..rq+3 at k, v[2 ps + 1] at k + 1
        const int rq = 4 * ((tid & 7) + 8 * (tid >> 6));
        int gr = row0 + rq; gr = gr < nrows - 4 ? gr : nrows - 4;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int k = k0 + 2 * (((tid >> 3) & 7) + 8 * ps);
            const int kc0 = k < K ? k : K - 1, kc1 = k + 1 < K ? k + 1 : K - 1;
            v[2 * ps] = *reinterpret_cast<const f32x4*>(P + (size_t)kc0 * s_k + gr);
            v[2 * ps + 1] = *reinterpret_cast<const f32x4*>(P + (size_t)kc1 * s_k + gr);
        }
    } else {
        // thread = (rows iq + 32 e, k pairs kp + 8 pass): consecutive lanes read consecutive rows.   v[e] = (k, k+1) of pass 0, of pass 1
        const int iq = tid & 31, kp = tid >> 5;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int gr = row0 + iq + 32 * e; gr = gr < nrows ? gr : nrows - 1;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int k = k0 + 2 * (kp + 8 * ps);
                const int kc0 = k < K ? k : K - 1, kc1 = k + 1 < K ? k + 1 : K - 1;
                v[e][2 * ps] = P[(size_t)gr * s_row + (size_t)kc0 * s_k];
                v[e][2 * ps + 1] = P[(size_t)gr * s_row + (size_t)kc1 * s_k];
            }
        }
    }
}

// zero the elements of a register slice whose contraction index is >= kend (same thread -> element maps as load_regs)
template <int LD>
__device__ __forceinline__ void mask_tail(f32x4 (&v)[4], int k0, int kend, int tid) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (LD == LD_KVEC) {
        const bool live = k0 + (tid & 7) * 4 < kend;               // kend % 4 == 0 on this path: a float4 is inside or outside as a whole
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = live ? v[e] : zero;
    } else if (LD == LD_RVEC) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int k = k0 + 2 * (((tid >> 3) & 7) + 8 * ps);
            v[2 * ps] = k < kend ? v[2 * ps] : zero;
            v[2 * ps + 1] = k + 1 < kend ? v[2 * ps + 1] : zero;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int k = k0 + 2 * ((tid >> 5) + 8 * ps);
                v[e][2 * ps] = k < kend ? v[e][2 * ps] : 0.f; v[e][2 * ps + 1] = k + 1 < kend ? v[e][2 * ps + 1] : 0.f;
            }
    }
}

template <int LD>
__device__ __forceinline__ void store_split(const f32x4 (&v)[4], char* hi_plane, char* lo_plane, int tid) {
    if (LD == LD_KVEC) {
        const int kq = tid & 7, r = tid >> 3;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            uint2 h, l;
            split2(v[ps][0], v[ps][1], h.x, l.x);
            split2(v[ps][2], v[ps][3], h.y, l.y);
            const int off = (r + 32 * ps) * XP + kq * 8;
            *reinterpret_cast<uint2*>(hi_plane + off) = h;
            *reinterpret_cast<uint2*>(lo_plane + off) = l;
        }
    } else if (LD == LD_RVEC) {
        const int rq = 4 * ((tid & 7) + 8 * (tid >> 6));
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int kp = ((tid >> 3) & 7) + 8 * ps;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h, l;
                split2(v[2 * ps][e], v[2 * ps + 1][e], h, l);
                const int off = (rq + e) * XP + kp * 4;
                *reinterpret_cast<uint32_t*>(hi_plane + off) = h;
                *reinterpret_cast<uint32_t*>(lo_plane + off) = l;
            }
        }
    } else {
        const int iq = tid & 31, kp = tid >> 5;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                uint32_t h, l;
                split2(v[e][2 * ps], v[e][2 * ps + 1], h, l);
                const int off = (iq + 32 * e) * XP + (kp + 8 * ps) * 4;
                *reinterpret_cast<uint32_t*>(hi_plane + off) = h;
                *reinterpret_cast<uint32_t*>(lo_plane + off) = l;
            }
    }
}

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ x3_bf16x8 frag(const char* plane, int row, int ks, int hi) {
    return __builtin_bit_cast(x3_bf16x8, *reinterpret_cast<const uint4*>(plane + row * XP + ks * 32 + hi * 16));
}

// (LDA / LDB: the loader of each operand -- the same for both in the NT / TN forms of the module's GEMMs, mixed for the small
// "plain" products A B of tcow_sgemm_x3_batched)
template <int LDA, int LDB>
__device__ __forceinline__ void gemm_x3_body(const F32Params& p, const int bid, const int kz, char* smem) {
    char* a_hi = smem; char* a_lo = smem + XPLANE; char* b_hi = smem + 2 * XPLANE; char* b_lo = smem + 3 * XPLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    // blocks are dispatched round-robin over the 8 XCDs: hand every XCD a contiguous range of tile ids (column tiles fastest), so the
    // workgroups that share an A row-tile run on ONE private L2 instead of fetching it from HBM once per XCD
    const int tiles_n = cdiv_dev(p.N, XT), nblk = tiles_n * cdiv_dev(p.M, XT);
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int pid = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int pm = pid / tiles_n;
    const int m0 = pm * XT, n0 = (pid - pm * tiles_n) * XT;
    const int kbeg = kz * p.kps;
    const int kend = (kbeg + p.kps < p.K) ? kbeg + p.kps : p.K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // two register sets: slices k and k+1 are in flight while slice k-1 is multiplied (one slice ahead left the loads ~400 ns of MFMA
    // time to land: the kernel ran at the memory latency, 5.8 us per slice)
    f32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    auto fetch = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], int k0) {
        load_regs<LDA>(p.A, p.sai, p.sak, m0, p.M, k0, p.K, ra, tid);
        load_regs<LDB>(p.B, p.sbj, p.sbk, n0, p.N, k0, p.K, rb, tid);
    };
    // One slice: split + store the registers of slice kcur, request slice knext into the same registers, multiply.  The body is
    // unconditional on purpose -- loads past the range read clamped (valid) addresses and are masked to zero when they are split, an odd
    // slice count runs one all-zero slice: any branch around a half of the 2x-unrolled loop makes hipcc rotate the register sets with
    // copies on the back edge, i.e. wait for every outstanding load once per iteration.
    auto step = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], int kcur, int knext) {
        if (kcur + XK > kend) { mask_tail<LDA>(ra, kcur, kend, tid); mask_tail<LDB>(rb, kcur, kend, tid); }
        store_split<LDA>(ra, a_hi, a_lo, tid);
        store_split<LDB>(rb, b_hi, b_lo, tid);
        __syncthreads();
        fetch(ra, rb, knext);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x3_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = frag(a_hi, wm * 64 + i * 32 + l31, ks, hi); al[i] = frag(a_lo, wm * 64 + i * 32 + l31, ks, hi);
                bh[i] = frag(b_hi, wn * 64 + i * 32 + l31, ks, hi); bl[i] = frag(b_lo, wn * 64 + i * 32 + l31, ks, hi);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // (B fragment first: the accumulators hold C^T -- lane = output row, register quads = 4 consecutive columns)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    };
    // (the two prologue fetches must stay in this order: s_waitcnt counts are merged over the loop's entry and back edge, and a prologue
    // that interleaves the sets makes every wait in the loop a wait for all 16 loads)
    fetch(ra0, rb0, kbeg);
    __builtin_amdgcn_sched_barrier(0);
    fetch(ra1, rb1, kbeg + XK);
    __builtin_amdgcn_sched_barrier(0);
    for (int k0 = kbeg; k0 < kend; k0 += 2 * XK) {
        step(ra0, rb0, k0, k0 + 2 * XK);
        step(ra1, rb1, k0 + XK, k0 + 3 * XK);
    }
    // ---- epilogue.  Register r of lane (l31, hi) of acc[i][j]: output row l31 of band i, columns j*32 + 8*(r>>2) + 4*hi + (r&3).  Every
    // wave stages one 32-row band of its 64 x 64 tile in a private 8.5 KiB of the (now free) planes -- 16-byte LDS stores, rows padded to
    // 68 floats -- and reads it back row-wise: 16 lanes x 16 B = one 256-B row segment per quarter wave, so residual / aux reads and the
    // C stores are full-line accesses.  (Scalar 4-byte stores straight from the C layout cost 109 us of fixed time at M = 27 090,
    // N = 768: more than the K = 768 main loop.)  No workgroup barrier: a wave's LDS operations execute in order.
    constexpr int CT_LD = 68;
    float* ct = reinterpret_cast<float*>(smem) + wave * (32 * CT_LD);
    const int c4 = (lane & 15) * 4, r4 = lane >> 4;
    const int gn = n0 + wn * 64 + c4;
    const bool v16 = p.slab ? ((reinterpret_cast<uintptr_t>(p.slab) & 15) == 0 && (p.N & 3) == 0)
                            : (((reinterpret_cast<uintptr_t>(p.C) | reinterpret_cast<uintptr_t>(p.resid) | reinterpret_cast<uintptr_t>(p.aux) | reinterpret_cast<uintptr_t>(p.bias2)) & 15) == 0 &&
                               ((p.ldc | p.ldr | p.ldaux) & 3) == 0);
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!p.slab && p.bias) {
        if (gn < p.N) b4.x = p.bias[gn];
        if (gn + 1 < p.N) b4.y = p.bias[gn + 1];
        if (gn + 2 < p.N) b4.z = p.bias[gn + 2];
        if (gn + 3 < p.N) b4.w = p.bias[gn + 3];
    }
    float* slab = p.slab ? p.slab + (size_t)kz * p.M * p.N : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(ct + l31 * CT_LD + j * 32 + 8 * g + 4 * hi) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + r4;
            const int gm = m0 + wm * 64 + i * 32 + row;
            if (gm >= p.M || gn >= p.N) continue;
            const float4 x = *reinterpret_cast<const float4*>(ct + row * CT_LD + c4);
            if (v16 && gn + 3 < p.N) {
                if (slab) st4(slab + (size_t)gm * p.N + gn, x);
                else f32_epilogue_store4(p, gm, gn, x, b4);
            } else {
                const float xs[4] = {x.x, x.y, x.z, x.w}, bs[4] = {b4.x, b4.y, b4.z, b4.w};
                for (int e = 0; e < 4; ++e) {
                    if (gn + e >= p.N) break;
                    if (slab) slab[(size_t)gm * p.N + gn + e] = xs[e];
                    else f32_epilogue_store(p, gm, gn + e, xs[e], bs[e]);
                }
            }
        }
    }
}

template <int LD>
__global__ __launch_bounds__(256, LD == LD_ANY ? 2 : 3) void gemm_x3_kernel(F32Params p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * XPLANE];
    gemm_x3_body<LD, LD>(p, blockIdx.x, blockIdx.y, smem);
}

// several small products in one launch: blockIdx.y = problem, blockIdx.x = tile (grid sized for the largest problem)
constexpr int X3_MAX_BATCH = 24;
struct X3Batch { F32Params p[X3_MAX_BATCH]; };
template <int LDA, int LDB>
__global__ __launch_bounds__(256, (LDA == LD_ANY || LDB == LD_ANY) ? 2 : 3) void gemm_x3_batch_kernel(X3Batch b) {
    __shared__ __attribute__((aligned(16))) char smem[4 * XPLANE];
    const F32Params& p = b.p[blockIdx.y];
    if ((int)blockIdx.x >= cdiv_dev(p.N, XT) * cdiv_dev(p.M, XT)) return;
    gemm_x3_body<LDA, LDB>(p, blockIdx.x, 0, smem);
}

}  // namespace

int tcow_launch_slab_reduce(hipStream_t stream, const float* slab, int nz, long slab_stride, long rows, long cols, float* out, long ldo, int accumulate,
                            const float* bias_part, int bias_nparts, int bias_n, float* bias_out);

int tcow_gemm_nt_x3(hipStream_t stream, const tcow_gemm_args* a) {
    F32Params p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = (const float*)a->A; p.sai = a->lda; p.sak = 1;
    p.B = (const float*)a->W; p.sbj = a->ldw; p.sbk = 1;
    p.C = a->C; p.ldc = a->ldc; p.bias = a->bias; p.row_scale = a->row_scale; p.resid = a->resid; p.ldr = a->ldr;
    p.act = a->act; p.aux = (float*)a->aux; p.ldaux = a->ldaux; p.bias2 = a->bias2; p.row_scale2 = a->row_scale2; p.kps = ((a->K + XK - 1) / XK) * XK; p.slab = nullptr;
    const bool vec = ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B)) & 15) == 0 && ((p.sai | p.sbj | p.K) & 3) == 0;
    const dim3 grid(cdiv(a->N, XT) * cdiv(a->M, XT), 1);
    if (vec) hipLaunchKernelGGL(gemm_x3_kernel<LD_KVEC>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gemm_x3_kernel<LD_ANY>, grid, dim3(256), 0, stream, p);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// token-dimension slices of the weight-gradient form: as many as keep tiles x slices within ONE round of 3 workgroups per CU
// (the vector loaders fit 168 registers and 40 KiB of LDS), at least 256 tokens each
int tcow_tn_splits_x3(int M, int N, int K) {
    const int tiles = cdiv(N, XT) * cdiv(K, XT);
    int s = (3 * 256) / tiles;
    const int max_s = M / 256;
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}

int tcow_gemm_tn_x3(hipStream_t stream, int M, int N, int K, const float* dY, long ldy, const float* X, long ldx, float* dW, long lddw,
                    int accumulate, float* slab, int splits, const float* bias_part, int bias_nparts, float* bias_out) {
    F32Params p;
    p.M = N; p.N = K; p.K = M;                       // output [N,K], contraction over tokens
    p.A = dY; p.sai = 1; p.sak = ldy;
    p.B = X; p.sbj = 1; p.sbk = ldx;
    p.C = nullptr; p.ldc = 0; p.bias = nullptr; p.row_scale = nullptr; p.resid = nullptr; p.ldr = 0; p.act = 0; p.aux = nullptr; p.ldaux = 0; p.bias2 = nullptr; p.row_scale2 = nullptr;
    int kps = cdiv(M, splits); kps = ((kps + XK - 1) / XK) * XK;
    const int nz = cdiv(M, kps);
    p.kps = kps; p.slab = slab;
    const bool vec = ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) & 15) == 0 && ((ldy | ldx | N | K) & 3) == 0;
    const dim3 grid(cdiv(K, XT) * cdiv(N, XT), nz);
    if (vec) hipLaunchKernelGGL(gemm_x3_kernel<LD_RVEC>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gemm_x3_kernel<LD_ANY>, grid, dim3(256), 0, stream, p);
    TCOW_CHECK_LAUNCH();
    return tcow_launch_slab_reduce(stream, slab, nz, (long)N * K, N, K, dW, lddw, accumulate, bias_part, bias_nparts, N, bias_out);       // (bias partials of tcow_launch_colsum_partials ride on the fold)
}

// ---- tcow_sgemm_x3_batched: C = A B for up to 24 small f32 problems in one launch; every operand is addressed by (row stride, k stride),
// one of them 1 -- which picks the vector loader per operand (all problems of a call share the two loaders)
static int x3_loader(const float* P, long s_row, long s_k, int nrows, int K) {
    const bool al = (reinterpret_cast<uintptr_t>(P) & 15) == 0;
    if (s_k == 1 && al && (s_row & 3) == 0 && (K & 3) == 0 && K >= 4) return LD_KVEC;
    if (s_row == 1 && al && (s_k & 3) == 0 && (nrows & 3) == 0 && nrows >= 4) return LD_RVEC;
    return LD_ANY;
}

extern "C" int tcow_sgemm_x3_batched(void* stream, int n, const tcow_sgemm* probs) {
    TCOW_CHECK_ARG(n >= 1 && n <= X3_MAX_BATCH && probs, "tcow_sgemm_x3_batched: 1 .. %d problems per call (got %d)", X3_MAX_BATCH, n);
    X3Batch b;
    int lda = -1, ldb = -1, maxtiles = 0;
    for (int i = 0; i < n; ++i) {
        const tcow_sgemm& s = probs[i];
        TCOW_CHECK_ARG(s.M > 0 && s.N > 0 && s.K > 0 && s.A && s.B && s.C && s.ldc >= s.N, "tcow_sgemm_x3_batched: bad problem %d", i);
        F32Params& p = b.p[i];
        p.M = s.M; p.N = s.N; p.K = s.K; p.A = s.A; p.sai = s.sai; p.sak = s.sak; p.B = s.B; p.sbj = s.sbj; p.sbk = s.sbk;
        p.C = s.C; p.ldc = s.ldc; p.bias = nullptr; p.row_scale = nullptr; p.resid = nullptr; p.ldr = 0; p.act = 0; p.aux = nullptr; p.ldaux = 0; p.bias2 = nullptr; p.row_scale2 = nullptr;
        p.kps = ((s.K + XK - 1) / XK) * XK; p.slab = nullptr;
        if (s.accumulate) { p.resid = s.C; p.ldr = s.ldc; }           // (the epilogue reads the residual element it then overwrites)
        const int la = x3_loader(s.A, s.sai, s.sak, s.M, s.K), lb = x3_loader(s.B, s.sbj, s.sbk, s.N, s.K);
        lda = (lda < 0 || lda == la) ? la : LD_ANY; ldb = (ldb < 0 || ldb == lb) ? lb : LD_ANY;
        const int t = cdiv(s.N, XT) * cdiv(s.M, XT);
        if (t > maxtiles) maxtiles = t;
    }
    const dim3 grid(maxtiles, n);
    hipStream_t st = (hipStream_t)stream;
    if (lda == LD_KVEC && ldb == LD_KVEC) hipLaunchKernelGGL((gemm_x3_batch_kernel<LD_KVEC, LD_KVEC>), grid, dim3(256), 0, st, b);
    else if (lda == LD_KVEC && ldb == LD_RVEC) hipLaunchKernelGGL((gemm_x3_batch_kernel<LD_KVEC, LD_RVEC>), grid, dim3(256), 0, st, b);
    else if (lda == LD_RVEC && ldb == LD_RVEC) hipLaunchKernelGGL((gemm_x3_batch_kernel<LD_RVEC, LD_RVEC>), grid, dim3(256), 0, st, b);
    else if (lda == LD_RVEC && ldb == LD_ANY) hipLaunchKernelGGL((gemm_x3_batch_kernel<LD_RVEC, LD_ANY>), grid, dim3(256), 0, st, b);
    else hipLaunchKernelGGL((gemm_x3_batch_kernel<LD_ANY, LD_ANY>), grid, dim3(256), 0, st, b);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
