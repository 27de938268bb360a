// Exact-f32 GEMMs (parity mode) on the f32-input MFMA (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain).
// One stride-generic kernel serves the NT form (forward / input-gradient) and the TN form (weight gradient):
//   C[i,j] = sum_k A(i,k) * B(j,k),  A(i,k) = A[i*sai + k*sak],  B(j,k) = B[j*sbj + k*sbk].
// 64x64 tile per 256-thread workgroup (4 waves, one 32x32 MFMA tile each), k walked 16 at a time through LDS
// stored k-major so that fragment reads are conflict-free ds_read_b32.  Speed is secondary here: this path
// exists so that the HIP pipeline can be compared with the fp32 reference below 1e-3.
#include "gemm_f32.h"

namespace {

constexpr int FT = 64;   // tile edge
constexpr int FK = 16;   // k-slice
constexpr int FLD = FT + 4;

__device__ __forceinline__ void load_tile(const float* __restrict__ P, long s_row, long s_k, int row0, int nrows, int k0, int kend,
                                          float (*dst)[FLD], int tid) {
    // 64 rows x 16 k = 1024 elements, 4 per thread, vectorised along whichever index is contiguous.
    if (s_k == 1) {
        const int r = tid >> 2, kk = (tid & 3) * 4;
        const int gr = row0 + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (gr < nrows) {
            const float* src = P + (size_t)gr * s_row + k0 + kk;
            if (k0 + kk + 3 < kend && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
                float4 t = *reinterpret_cast<const float4*>(src); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                for (int e = 0; e < 4; ++e) if (k0 + kk + e < kend) v[e] = src[e];
            }
        }
        for (int e = 0; e < 4; ++e) dst[kk + e][r] = v[e];
    } else {
        const int kk = tid >> 4, r = (tid & 15) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k0 + kk < kend) {
            const float* src = P + (size_t)(k0 + kk) * s_k;
            if (s_row == 1 && row0 + r + 3 < nrows && ((reinterpret_cast<uintptr_t>(src + row0 + r) & 15) == 0)) {
                float4 t = *reinterpret_cast<const float4*>(src + row0 + r); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                for (int e = 0; e < 4; ++e) if (row0 + r + e < nrows) v[e] = src[(size_t)(row0 + r + e) * s_row];
            }
        }
        *reinterpret_cast<float4*>(&dst[kk][r]) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(F32Params p) {
    __shared__ __attribute__((aligned(16))) float As[FK][FLD];
    __shared__ __attribute__((aligned(16))) float Bs[FK][FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.y * FT, n0 = blockIdx.x * FT;
    const int kbeg = blockIdx.z * p.kps;
    const int kend = (kbeg + p.kps < p.K) ? kbeg + p.kps : p.K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += FK) {
        load_tile(p.A, p.sai, p.sak, m0, p.M, k0, kend, As, tid);
        load_tile(p.B, p.sbj, p.sbk, n0, p.N, k0, kend, Bs, tid);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < FK / 2; ++s) {
            const float a = As[2 * s + hi][wm * 32 + l31];
            const float b = Bs[2 * s + hi][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int gn = n0 + wn * 32 + l31;
    if (gn >= p.N) return;
    if (p.slab) {
        float* out = p.slab + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * 32 + crow32(r, hi);
            if (gm < p.M) out[(size_t)gm * p.N + gn] = acc[r];
        }
        return;
    }
    const float bv = p.bias ? p.bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 32 + crow32(r, hi);
        if (gm < p.M) f32_epilogue_store(p, gm, gn, acc[r], bv);
    }
}

// out[i] (+)= sum_z slab[z][i]  (float4 lanes; cols, ldo and slab_stride multiples of 4 on the vector path)
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int nz, long slab_stride, long n, float* __restrict__ out, long rows, long cols, long ldo, int accumulate) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += slab[(size_t)z * slab_stride + i];
        const long r = i / cols, c = i - r * cols;
        float* o = out + r * ldo + c;
        *o = accumulate ? (*o + s) : s;
    }
}
// Workgroups [0, slab_blocks) fold the split-K slabs; the optional tail workgroups fold the bias-gradient partial table of the same
// weight-gradient GEMM (4 columns x 16 row groups per 64-thread workgroup), so one launch finishes both.
__device__ __forceinline__ void slab_reduce4_body(const int bx, const float* __restrict__ slab, int nz, long slab_stride, long n4, float* __restrict__ out, long cols4, long ldo, int accumulate,
                                                          int slab_blocks, const float* __restrict__ part, int nparts, int N, float* __restrict__ bias_out) {
    if (bx >= slab_blocks) {
        __shared__ float red[16][4];
        const int cq = threadIdx.x & 3, g = threadIdx.x >> 2;
        const int c = (bx - slab_blocks) * 4 + cq;
        float a = 0.f;
        if (c < N)
            for (int r = g; r < nparts; r += 16) a += part[(size_t)r * N + c];
        red[g][cq] = a;
        __syncthreads();
        if (g == 0 && c < N) {
            for (int k = 1; k < 16; ++k) a += red[k][cq];
            bias_out[c] = accumulate ? bias_out[c] + a : a;
        }
        return;
    }
    long i = bx * (long)blockDim.x + threadIdx.x;
    const long stride = (long)slab_blocks * blockDim.x;
    for (; i < n4; i += stride) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // 8 independent loads in flight per thread: the slabs are streamed once, latency not bandwidth is the enemy
        for (int z0 = 0; z0 < nz; z0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (z0 + u < nz) ? ld4(slab + (size_t)(z0 + u) * slab_stride + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        const long r = i / cols4, c = (i - r * cols4) * 4;
        float* o = out + r * ldo + c;
        if (accumulate) { const float4 p = ld4(o); s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
        st4(o, s);
    }
}

__global__ __launch_bounds__(64) void slab_reduce4_kernel(const float* __restrict__ slab, int nz, long slab_stride, long n4, float* __restrict__ out, long cols4, long ldo, int accumulate,
                                                          int slab_blocks, const float* __restrict__ part, int nparts, int N, float* __restrict__ bias_out) {
    slab_reduce4_body(blockIdx.x, slab, nz, slab_stride, n4, out, cols4, ldo, accumulate, slab_blocks, part, nparts, N, bias_out);
}
// The folds of a grouped weight-gradient launch (tcow_gemm_tn_grouped) as ONE flat grid of 256-thread workgroups: job k owns workgroups
// [first[k], first[k + 1]) -- its slab workgroups (512 float4 columns each: two per thread, 2 x nz independent 16-byte loads in flight, the
// slabs read once and not kept in cache) followed by its bias-table workgroups (16 columns x 16 row groups each).
// (The first version was a (max blocks, jobs) grid of one-wave workgroups with one float4 per thread: 52 us for the 170 MB of a ViT-B block's
// seven weights = 3.3 TB/s, a third of its workgroups empty.)
// (measured equal: non-temporal vs plain loads, one vs two float4 columns per thread, 64-bit vs 32-bit row / column split: the launch moves its
// 198-226 MB at 4.1-4.4 TB/s either way)
#define FOLD_LD ld4
#define FOLD_BLK 512
struct FoldJob { const float* slab; long slab_stride, n4, cols4, ldo; float* out; const float* part; float* bias_out; int nz, accumulate, slab_blocks, nparts, N, blocks; };
struct FoldGroup { int n; int first[41]; FoldJob j[40]; };       // (as many jobs as a grouped weight-gradient launch has problems: TN_GROUP_MAX)
__global__ __launch_bounds__(256) void slab_reduce4_group_kernel(FoldGroup g) {
    int k = 0;
    while (k + 1 < g.n && (int)blockIdx.x >= g.first[k + 1]) ++k;          // workgroup-uniform
    const FoldJob& j = g.j[k];
    const int bx = (int)blockIdx.x - g.first[k];
    const int tid = threadIdx.x;
    if (bx >= j.slab_blocks) {
        __shared__ float red[16][17];
        const int cq = tid & 15, rg = tid >> 4;
        const int c = (bx - j.slab_blocks) * 16 + cq;
        float a = 0.f;
        if (c < j.N)
            for (int r = rg; r < j.nparts; r += 16) a += j.part[(size_t)r * j.N + c];
        red[rg][cq] = a;
        __syncthreads();
        if (rg == 0 && c < j.N) {
            for (int q = 1; q < 16; ++q) a += red[q][cq];
            j.bias_out[c] = j.accumulate ? j.bias_out[c] + a : a;
        }
        return;
    }
    const long i0 = (long)bx * FOLD_BLK + tid, i1 = i0 + 256;
    const bool ok0 = i0 < j.n4, ok1 = FOLD_BLK > 256 && i1 < j.n4;
    const int nz = j.nz;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    for (int z0 = 0; z0 < nz; z0 += 8) {
        float4 v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (z0 + u < nz) {                  // (uniform: no load is issued for an absent slice)
                const float* b = j.slab + (size_t)(z0 + u) * j.slab_stride;
                v0[u] = ok0 ? FOLD_LD(b + i0 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                v1[u] = ok1 ? FOLD_LD(b + i1 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                v0[u] = make_float4(0.f, 0.f, 0.f, 0.f); v1[u] = v0[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {          // (slice order 0, 1, 2, ...: the same sum as the one-weight kernel)
            s0.x += v0[u].x; s0.y += v0[u].y; s0.z += v0[u].z; s0.w += v0[u].w;
            s1.x += v1[u].x; s1.y += v1[u].y; s1.z += v1[u].z; s1.w += v1[u].w;
        }
    }
    // (dense outputs need no row / column split; otherwise 32-bit division: the 64-bit one is ~100 instructions, twice per thread = 9 us of a 49 us launch)
    const bool dense = j.ldo == j.cols4 * 4;
    if (ok0) {
        const unsigned r = dense ? 0u : (unsigned)i0 / (unsigned)j.cols4;
        float* o = dense ? j.out + i0 * 4 : j.out + (long)r * j.ldo + ((unsigned)i0 - r * (unsigned)j.cols4) * 4;
        if (j.accumulate) { const float4 p = ld4(o); s0.x += p.x; s0.y += p.y; s0.z += p.z; s0.w += p.w; }
        st4(o, s0);
    }
    if (ok1) {
        const unsigned r = dense ? 0u : (unsigned)i1 / (unsigned)j.cols4;
        float* o = dense ? j.out + i1 * 4 : j.out + (long)r * j.ldo + ((unsigned)i1 - r * (unsigned)j.cols4) * 4;
        if (j.accumulate) { const float4 p = ld4(o); s1.x += p.x; s1.y += p.y; s1.z += p.z; s1.w += p.w; }
        st4(o, s1);
    }
}

// out[c] (+)= sum_r part[r][c] for a tall-skinny partial table (many rows, few columns): 16 columns x 16 row groups per block
// (columns >= N1 go to out2[c - N1]: LayerNorm's dgamma | dbeta table is folded by one launch)
__device__ __forceinline__ void row_reduce_body(int bx, const float* __restrict__ part, int nrows, long ld, int N, float* __restrict__ out, int accumulate,
                                                int N1, float* __restrict__ out2, int N12, float* __restrict__ out3);
__global__ __launch_bounds__(256) void row_reduce_kernel(const float* __restrict__ part, int nrows, long ld, int N, float* __restrict__ out, int accumulate,
                                                         int N1, float* __restrict__ out2, int N12 = 1 << 30, float* __restrict__ out3 = nullptr) {
    row_reduce_body(blockIdx.x, part, nrows, ld, N, out, accumulate, N1, out2, N12, out3);
}
// several such folds in one launch (blockIdx.y = job): the dgamma | dbeta (| bias) tables of all LayerNorm backward calls of a group of blocks
struct RowReduceJob { const float* part; float* out; float* out2; float* out3; long ld; int nrows, N, N1, N12, accumulate; };
struct RowReduceGroup { RowReduceJob j[16]; };
__global__ __launch_bounds__(256) void row_reduce_group_kernel(RowReduceGroup g) {
    const RowReduceJob& j = g.j[blockIdx.y];
    if ((int)blockIdx.x * 16 >= j.N) return;
    row_reduce_body(blockIdx.x, j.part, j.nrows, j.ld, j.N, j.out, j.accumulate, j.N1, j.out2, j.N12, j.out3);
}
__device__ __forceinline__ void row_reduce_body(int bx, const float* __restrict__ part, int nrows, long ld, int N, float* __restrict__ out, int accumulate,
                                                int N1, float* __restrict__ out2, int N12, float* __restrict__ out3) {
    __shared__ float red[16][17];
    const int c = bx * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    float s = 0.f;
    if (c < N) {
        // 8 independent loads in flight per thread: the table is small (a few MB), the kernel is pure load latency
        int r = g;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (; r + 7 * 16 < nrows; r += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += part[(size_t)(r + 16 * u) * ld + c];
        }
        for (; r < nrows; r += 16) a[0] += part[(size_t)r * ld + c];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[g][threadIdx.x & 15] = s;
    __syncthreads();
    if (g == 0 && c < N) {
        for (int k = 1; k < 16; ++k) s += red[k][threadIdx.x & 15];
        float* o = c < N1 ? out + c : (c < N12 ? out2 + (c - N1) : out3 + (c - N12));      // (columns >= N12: a third table, LayerNorm's fused bias gradient)
        *o = accumulate ? *o + s : s;
    }
}

// column sums of Y[M,N] (bias gradient): thread = 4 consecutive columns, a block covers 128 columns x a row slice with
// 8 row groups folded through LDS; partial rows go to `part` and tcow_launch_slab_reduce finishes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ Y, long ldy, int M, int N, int rows_per_blk, float* __restrict__ part) {
    __shared__ float4 red[8][32];
    const int cq = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 128 + cq * 4;
    const int r0 = blockIdx.y * rows_per_blk;
    const int r1 = (r0 + rows_per_blk < M) ? r0 + rows_per_blk : M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < N) {
        // four row groups in flight per trip (one load per trip ran at the memory latency: 2.7 TB/s at N = 768 in the f32-storage modes' step)
        int r = r0 + rg;
        for (; r + 24 < r1; r += 32) {
            const float4 v0 = ld4(Y + (size_t)r * ldy + c), v1 = ld4(Y + (size_t)(r + 8) * ldy + c), v2 = ld4(Y + (size_t)(r + 16) * ldy + c), v3 = ld4(Y + (size_t)(r + 24) * ldy + c);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y); s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; r < r1; r += 8) { const float4 v = ld4(Y + (size_t)r * ldy + c); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    } else if (c < N) {
        for (int r = r0 + rg; r < r1; r += 8) {
            const T* p = Y + (size_t)r * ldy + c;
            s.x += Elem<T>::ld(p); if (c + 1 < N) s.y += Elem<T>::ld(p + 1); if (c + 2 < N) s.z += Elem<T>::ld(p + 2);
        }
    }
    red[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && c < N) {
        for (int g = 1; g < 8; ++g) { const float4 v = red[g][cq]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        float* o = part + (size_t)blockIdx.y * N + c;
        o[0] = s.x; if (c + 1 < N) o[1] = s.y; if (c + 2 < N) o[2] = s.z; if (c + 3 < N) o[3] = s.w;
    }
}

}  // namespace

int tcow_gemm_nt_f32(hipStream_t stream, const tcow_gemm_args* a) {
    F32Params p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = (const float*)a->A; p.sai = a->lda; p.sak = 1;
    p.B = (const float*)a->W; p.sbj = a->ldw; p.sbk = 1;
    p.C = a->C; p.ldc = a->ldc; p.bias = a->bias; p.row_scale = a->row_scale; p.resid = a->resid; p.ldr = a->ldr;
    p.act = a->act; p.aux = (float*)a->aux; p.ldaux = a->ldaux; p.bias2 = a->bias2; p.row_scale2 = a->row_scale2; p.kps = ((a->K + FK - 1) / FK) * FK; p.slab = nullptr;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(a->N, FT), cdiv(a->M, FT), 1), dim3(256), 0, stream, p);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// number of token-dimension slices used by the weight-gradient GEMMs (both dtypes): ~2 workgroups per CU, a multiple of
// 8 so that every XCD owns the same number of slices (gemm_tn_bf16_kernel pins slice z to XCD z % 8), >= 256 tokens each.
int tcow_tn_splits(int M, int N, int K, int tile_outputs) {
    const int tiles = cdiv(N, tile_outputs) * cdiv(K, tile_outputs);
    int s = ((cdiv(512, tiles) + 7) / 8) * 8;
    const int max_s = M / 256;
    if (s > max_s) s = max_s >= 8 ? (max_s / 8) * 8 : max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

int tcow_launch_row_reduce(hipStream_t stream, const float* part, int nrows, long ld, int N, float* out, int accumulate);

int tcow_launch_slab_reduce(hipStream_t stream, const float* slab, int nz, long slab_stride, long rows, long cols, float* out, long ldo, int accumulate,
                            const float* bias_part, int bias_nparts, int bias_n, float* bias_out) {
    const long n = rows * cols;
    const bool vec = (cols % 4 == 0) && (ldo % 4 == 0) && (slab_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    if (vec) {
        int blocks = cdiv(n / 4, 64); if (blocks > 8192) blocks = 8192;
        const int tail = bias_part ? cdiv(bias_n, 4) : 0;
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3(blocks + tail), dim3(64), 0, stream, slab, nz, slab_stride, n / 4, out, cols / 4, ldo, accumulate,
                           blocks, bias_part, bias_nparts, bias_n, bias_out);
    } else {
        if (bias_part) { const int rc = tcow_launch_row_reduce(stream, bias_part, bias_nparts, bias_n, bias_n, bias_out, accumulate); if (rc) return rc; }
        int blocks = cdiv(n, 256); if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slab, nz, slab_stride, n, out, rows, cols, ldo, accumulate);
    }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// n <= 8 folds in one launch; every job must satisfy the vector path's alignment rules (the caller checks with tcow_fold_vec_ok)
bool tcow_fold_vec_ok(const float* slab, long slab_stride, long cols, float* out, long ldo) {
    return (cols % 4 == 0) && (ldo % 4 == 0) && (slab_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
}
int tcow_launch_slab_reduce_group(hipStream_t stream, int n, const float* const* slab, int nz, const long* rows, const long* cols, float* const* out, const long* ldo,
                                  const int* accumulate, const float* const* bias_part, const int* bias_nparts, float* const* bias_out) {
    FoldGroup g;
    g.n = n;
    int first = 0;
    for (int i = 0; i < n; ++i) {
        FoldJob& j = g.j[i];
        const long nel = rows[i] * cols[i];
        j.slab = slab[i]; j.slab_stride = nel; j.n4 = nel / 4; j.cols4 = cols[i] / 4; j.ldo = ldo[i]; j.out = out[i];
        j.part = bias_part[i]; j.bias_out = bias_out[i]; j.nz = nz; j.accumulate = accumulate[i];
        j.slab_blocks = slab[i] == out[i] ? 0 : (int)cdiv(nel / 4, FOLD_BLK);       // (slab == destination: the GEMM wrote its single slice in place)
        j.nparts = bias_nparts[i]; j.N = (int)rows[i];
        j.blocks = j.slab_blocks + (bias_part[i] ? cdiv(rows[i], 16) : 0);
        g.first[i] = first;
        first += j.blocks;
    }
    for (int i = n; i < 40; ++i) { g.j[i] = g.j[0]; }
    for (int i = n; i <= 40; ++i) g.first[i] = first;
    if (first == 0) return TCOW_OK;
    hipLaunchKernelGGL(slab_reduce4_group_kernel, dim3(first), dim3(256), 0, stream, g);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// n <= 16 three-way folds (out | out2 | out3 = columns [0, N1) | [N1, N12) | [N12, N)) in one launch
int tcow_launch_row_reduce_group(hipStream_t stream, int n, const float* const* part, const int* nrows, const long* ld, const int* N1, float* const* out1, const int* N2,
                                 float* const* out2, const int* N3, float* const* out3, const int* accumulate) {
    RowReduceGroup g;
    int gx = 0;
    for (int i = 0; i < n; ++i) {
        RowReduceJob& j = g.j[i];
        j.part = part[i]; j.out = out1[i]; j.out2 = out2[i]; j.out3 = out3[i]; j.ld = ld[i]; j.nrows = nrows[i];
        j.N = N1[i] + N2[i] + (out3[i] ? N3[i] : 0); j.N1 = N1[i]; j.N12 = out3[i] ? N1[i] + N2[i] : (1 << 30); j.accumulate = accumulate[i];
        const int b = cdiv(j.N, 16); if (b > gx) gx = b;
    }
    for (int i = n; i < 16; ++i) g.j[i] = g.j[0];
    hipLaunchKernelGGL(row_reduce_group_kernel, dim3(gx, n), dim3(256), 0, stream, g);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_launch_row_reduce(hipStream_t stream, const float* part, int nrows, long ld, int N, float* out, int accumulate) {
    hipLaunchKernelGGL(row_reduce_kernel, dim3(cdiv(N, 16)), dim3(256), 0, stream, part, nrows, ld, N, out, accumulate, N, (float*)nullptr);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// out1[c] (+)= column sums of part[:, c] for c < N1, out2[c - N1] for N1 <= c < N1 + N2
int tcow_launch_row_reduce2(hipStream_t stream, const float* part, int nrows, long ld, int N1, float* out1, int N2, float* out2, int accumulate) {
    hipLaunchKernelGGL(row_reduce_kernel, dim3(cdiv(N1 + N2, 16)), dim3(256), 0, stream, part, nrows, ld, N1 + N2, out1, accumulate, N1, out2);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_launch_row_reduce3(hipStream_t stream, const float* part, int nrows, long ld, int N1, float* out1, int N2, float* out2, int N3, float* out3, int accumulate) {
    hipLaunchKernelGGL(row_reduce_kernel, dim3(cdiv(N1 + N2 + N3, 16)), dim3(256), 0, stream, part, nrows, ld, N1 + N2 + N3, out1, accumulate, N1, out2, N1 + N2, out3);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// the partial sums only: `part` [parts][N], returns the number of parts (the caller folds them -- with the weight-gradient slabs, in one launch)
int tcow_launch_colsum_partials(hipStream_t stream, int dtype, const void* Y, long ldy, int M, int N, float* part, int max_parts, int* nparts) {
    int parts = cdiv(M, 256); if (parts > max_parts) parts = max_parts; if (parts < 1) parts = 1;
    const int rpb = cdiv(M, parts);
    parts = cdiv(M, rpb);
    if (dtype == TCOW_BF16)
        hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(cdiv(N, 128), parts), dim3(256), 0, stream, (const bf16_t*)Y, ldy, M, N, rpb, part);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(cdiv(N, 128), parts), dim3(256), 0, stream, (const float*)Y, ldy, M, N, rpb, part);
    TCOW_CHECK_LAUNCH();
    *nparts = parts;
    return TCOW_OK;
}

int tcow_launch_colsum(hipStream_t stream, int dtype, const void* Y, long ldy, int M, int N, float* out, int accumulate, float* part, int max_parts) {
    int parts = cdiv(M, 256); if (parts > max_parts) parts = max_parts; if (parts < 1) parts = 1;
    const int rpb = cdiv(M, parts);
    parts = cdiv(M, rpb);
    if (dtype == TCOW_BF16)
        hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(cdiv(N, 128), parts), dim3(256), 0, stream, (const bf16_t*)Y, ldy, M, N, rpb, part);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(cdiv(N, 128), parts), dim3(256), 0, stream, (const float*)Y, ldy, M, N, rpb, part);
    TCOW_CHECK_LAUNCH();
    return tcow_launch_slab_reduce(stream, part, parts, N, 1, N, out, N, accumulate, nullptr, 0, 0, nullptr);
}

int tcow_gemm_tn_f32(hipStream_t stream, int M, int N, int K, const float* dY, long ldy, const float* X, long ldx, float* dW, long lddw,
                     int accumulate, float* slab, int splits, const float* bias_part, int bias_nparts, float* bias_out) {
    F32Params p;
    p.M = N; p.N = K; p.K = M;                       // output [N,K], contraction over tokens
    p.A = dY; p.sai = 1; p.sak = ldy;
    p.B = X; p.sbj = 1; p.sbk = ldx;
    p.C = nullptr; p.ldc = 0; p.bias = nullptr; p.row_scale = nullptr; p.resid = nullptr; p.ldr = 0; p.act = 0; p.aux = nullptr; p.ldaux = 0; p.bias2 = nullptr; p.row_scale2 = nullptr;
    int kps = cdiv(M, splits); kps = ((kps + FK - 1) / FK) * FK;
    const int nz = cdiv(M, kps);
    p.kps = kps; p.slab = slab;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(K, FT), cdiv(N, FT), nz), dim3(256), 0, stream, p);
    TCOW_CHECK_LAUNCH();
    return tcow_launch_slab_reduce(stream, slab, nz, (long)N * K, N, K, dW, lddw, accumulate, bias_part, bias_nparts, N, bias_out);
}
