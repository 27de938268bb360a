// LayerNorm over the channel dimension of the f32 residual stream (nn.LayerNorm(D, eps=1e-6):
// vit.py:135 norm1, :142 temporal_norm1, :150 norm2, :283 norm; eps from vit.py:428).
// One wave per token row, the row lives in registers (D <= 2048), statistics by wavefront reduction.
// Memory-bound: reads 4*D bytes and writes sizeof(T)*D bytes per row.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int LN_MAXV = 8;  // float4 per lane -> D <= 64*4*8 = 2048; kernels are instantiated for NV = 1, 2, 3, 4, 8 (NV = 3 is D = 768)

// FULL: D == 256 * NV, i.e. every lane owns a float4 in every one of the NV passes -- the per-pass lane guards (divergent-branch code and a basic
// block per pass, which keep hipcc from issuing a row's loads together) are compiled out.  D = 768 / 512 / 1024 take this form.
template <typename T, int NV, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int rows, int D, const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ y, long ldy,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    // Grid-stride over rows with the NEXT row's loads issued before this row's reductions and stores (one row per wave and launch-sized
    // grids left every wave a serial load -> reduce -> reduce -> store chain: 5.0 TB/s where a plain copy of the same bytes reaches 7).
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;  // float4 count
    const int rstep = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 g[NV], b[NV], nx[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (FULL || c < nv) { g[i] = ld4(gamma + c * 4); b[i] = ld4(beta + c * 4); nx[i] = ld4(x + (size_t)row * ldx + c * 4); }
    }
    for (; row < rows; row += rstep) {
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < nv) { v[i] = nx[i]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
        }
        if (row + rstep < rows) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                if (FULL || c < nv) nx[i] = ld4(x + (size_t)(row + rstep) * ldx + c * 4);
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < nv) {
                const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + bb * bb) + (cc * cc + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        T* yr = y + (size_t)row * ldy;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < nv) {
                float4 o;
                o.x = (v[i].x - mean) * rstd * g[i].x + b[i].x; o.y = (v[i].y - mean) * rstd * g[i].y + b[i].y;
                o.z = (v[i].z - mean) * rstd * g[i].z + b[i].z; o.w = (v[i].w - mean) * rstd * g[i].w + b[i].w;
                st4(yr + c * 4, o);
            }
        }
    }
}

// Backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma, xhat = (x - mean) * rstd;
// dx_out = dres + dx (the gradient arriving through the residual connection is added here).
// dgamma / dbeta partial sums: each wave keeps per-column partials over the rows it visits, the block folds its
// 4 waves through LDS and writes one partial row per block; tcow_launch_slab_reduce finishes the sum.
template <typename T, int NV, bool CSUM, bool FULL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int rows, int D, const T* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres, long lddres,
                                                     float* __restrict__ dx, long lddx, float* __restrict__ part /* [grid][NP][D] or NULL */,
                                                     T* __restrict__ dxc, long lddxc, const float* __restrict__ cscale, const float* __restrict__ sumscale) {
    // CSUM: a third column sum rides along, sum_rows sumscale[row] * dx[row] -- the bias gradient of the Linear layer BELOW this norm when
    // that layer's output was row-masked (temporal_fc: vit.py:174-176 with the cls rows excluded), in f32 instead of from the bf16 operand
    constexpr int NP = CSUM ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    float4 gsum[NV], bsum[NV], gam[NV], csum[CSUM ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        gsum[i] = make_float4(0.f, 0.f, 0.f, 0.f); bsum[i] = gsum[i];
        if (CSUM) csum[i] = gsum[i];
        const int c = lane + i * 64;
        gam[i] = (FULL || c < nv) ? ld4(gamma + c * 4) : gsum[i];
    }
    // Software-pipelined over the rows a wave visits: the loads of the NEXT row are issued before this row's results are stored.
    // (vmcnt counts stores too: a "load, compute, store" loop body makes every row wait for the previous row's stores.)
    const int rstep = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    float4 xn[NV], dn[NV], rn[NV];
    float mu_n = 0.f, rs_n = 0.f;
    auto fetch = [&](int r) {
        mu_n = mean[r]; rs_n = rstd[r];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < nv) {
                {   // (the layer's saved input: written a forward pass ago, read once -- non-temporal, see attention_tiles.h)
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(x + (size_t)r * ldx + c * 4));
                    xn[i] = make_float4(t[0], t[1], t[2], t[3]);
                }
                dn[i] = ld4(dy + (size_t)r * lddy + c * 4);
            }
        }
        if (dres) {                                              // (one uniform branch per row, not one per pass: the passes stay one basic block)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                if (FULL || c < nv) rn[i] = ld4(dres + (size_t)r * lddres + c * 4);
            }
        }
    };
    if (row < rows) fetch(row);
    for (; row < rows; row += rstep) {
        const float mu = mu_n, rs = rs_n;
        float4 xh[NV], g[NV], rr[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < nv) {
                const float4 xv = xn[i], d = dn[i];
                rr[i] = rn[i];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(d.x * gam[i].x, d.y * gam[i].y, d.z * gam[i].z, d.w * gam[i].w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                gsum[i].x += d.x * xh[i].x; gsum[i].y += d.y * xh[i].y; gsum[i].z += d.z * xh[i].z; gsum[i].w += d.w * xh[i].w;
                bsum[i].x += d.x; bsum[i].y += d.y; bsum[i].z += d.z; bsum[i].w += d.w;
            }
        }
        if (row + rstep < rows) fetch(row + rstep);
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
        float* dxr = dx + (size_t)row * lddx;
        const float cs_row = (dxc && cscale) ? cscale[row] : 1.0f;
        const float ss_row = (CSUM && sumscale) ? sumscale[row] : 1.0f;
        float4 o[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i)
            o[i] = make_float4(rs * (g[i].x - m1 - xh[i].x * m2), rs * (g[i].y - m1 - xh[i].y * m2), rs * (g[i].z - m1 - xh[i].z * m2), rs * (g[i].w - m1 - xh[i].w * m2));
        if (dres) {
#pragma unroll
            for (int i = 0; i < NV; ++i) { o[i].x += rr[i].x; o[i].y += rr[i].y; o[i].z += rr[i].z; o[i].w += rr[i].w; }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < nv) {
                st4(dxr + c * 4, o[i]);
                if (CSUM) { csum[i].x = fmaf(ss_row, o[i].x, csum[i].x); csum[i].y = fmaf(ss_row, o[i].y, csum[i].y); csum[i].z = fmaf(ss_row, o[i].z, csum[i].z); csum[i].w = fmaf(ss_row, o[i].w, csum[i].w); }
            }
        }
        if (dxc) {           // the same gradient as the next GEMM's operand: dtype(dx * row scale), saves a separate cast pass
            const float cs = cs_row;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                if (FULL || c < nv) st4(dxc + (size_t)row * lddxc + c * 4, make_float4(o[i].x * cs, o[i].y * cs, o[i].z * cs, o[i].w * cs));
            }
        }
    }
    if (!part) return;
    // fold the four waves' column partials through ONE [4][D] LDS image, one quantity after the other (12 KiB at D = 768 instead of 24 / 36:
    // the image is held for the whole kernel, and this streaming kernel wants its 6 workgroups per CU)
#pragma unroll
    for (int qn = 0; qn < NP; ++qn) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < nv) *reinterpret_cast<float4*>(red + (size_t)wave * D + c * 4) = qn == 0 ? gsum[i] : (qn == 1 ? bsum[i] : csum[CSUM ? i : 0]);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < D; i += 256)
            part[((size_t)blockIdx.x * NP + qn) * D + i] = (red[i] + red[D + i]) + (red[2 * D + i] + red[3 * D + i]);
        __syncthreads();
    }
}

}  // namespace

int tcow_launch_row_reduce2(hipStream_t stream, const float* part, int nrows, long ld, int N1, float* out1, int N2, float* out2, int accumulate);
int tcow_launch_row_reduce_group(hipStream_t stream, int n, const float* const* part, const int* nrows, const long* ld, const int* N1, float* const* out1, const int* N2,
                                 float* const* out2, const int* N3, float* const* out3, const int* accumulate);
int tcow_launch_row_reduce3(hipStream_t stream, const float* part, int nrows, long ld, int N1, float* out1, int N2, float* out2, int N3, float* out3, int accumulate);

static const int kLnBwdBlocks = 768;
// (grid-stride blocks of the backward: 768 = three 4-wave workgroups per CU in ONE round -- the kernel needs 154 VGPRs at D = 768, i.e. three
// waves per SIMD; 512 -> 768: 78 -> 73.5 us, 1024 (1.33 rounds) 84 us.  The colsum variant needs 180 VGPRs = two waves per SIMD: 512 blocks.)
static const int kLnBwdBlocksCsum = 512;
static const int kLnFwdBlocks = 4096;     // grid-stride blocks of the forward (512 ... 8192 measured within 6 %: tools/dev_ln_time.py)

extern "C" {

int tcow_layernorm_fwd(void* stream, int dtype, int rows, int D, const float* x, long ldx, const float* gamma, const float* beta, float eps, void* y,
                       long ldy, float* mean, float* rstd) {
    TCOW_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "tcow_layernorm_fwd: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * LN_MAXV);
    TCOW_CHECK_ARG(x && gamma && beta && y && ldx % 4 == 0 && ldy % 4 == 0, "tcow_layernorm_fwd: bad pointers / strides");
    int fblocks = cdiv(rows, 4); if (fblocks > kLnFwdBlocks) fblocks = kLnFwdBlocks;
    const dim3 grid(fblocks), block(256);
    if (dtype != TCOW_BF16 && dtype != TCOW_F32) { tcow_set_error("tcow_layernorm_fwd: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    const int nvl = (D / 4 + 63) / 64;
#define LN_FWD(NVV, FULLV)                                                                                                                               \
    do {                                                                                                                                          \
        if (dtype == TCOW_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NVV, FULLV>), grid, block, 0, (hipStream_t)stream, rows, D, x, ldx, gamma, beta, eps, (bf16_t*)y, ldy, mean, rstd); \
        else hipLaunchKernelGGL((ln_fwd_kernel<float, NVV, FULLV>), grid, block, 0, (hipStream_t)stream, rows, D, x, ldx, gamma, beta, eps, (float*)y, ldy, mean, rstd); \
    } while (0)
    const bool full = D == 256 * nvl && nvl <= 4;
    if (full) { if (nvl == 1) LN_FWD(1, true); else if (nvl == 2) LN_FWD(2, true); else if (nvl == 3) LN_FWD(3, true); else LN_FWD(4, true); }
    else if (nvl <= 1) LN_FWD(1, false); else if (nvl == 2) LN_FWD(2, false); else if (nvl == 3) LN_FWD(3, false); else if (nvl == 4) LN_FWD(4, false); else LN_FWD(8, false);
#undef LN_FWD
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

long tcow_layernorm_bwd_workspace_bytes(int D) { return (long)kLnBwdBlocks * 3 * D * 4; }

int tcow_layernorm_bwd(void* stream, int dtype, int rows, int D, const void* dy, long lddy, const float* x, long ldx, const float* mean,
                       const float* rstd, const float* gamma, const float* dres, long lddres, float* dx, long lddx, float* dgamma, float* dbeta,
                       int accumulate, void* workspace, long workspace_bytes, void* dx_cast, long lddx_cast, const float* cast_row_scale,
                       const float* colsum_row_scale, float* colsum_out) {
    TCOW_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "tcow_layernorm_bwd: bad D=%d", D);
    TCOW_CHECK_ARG(dy && x && mean && rstd && gamma && dx, "tcow_layernorm_bwd: null pointer");
    TCOW_CHECK_ARG(!dx_cast || lddx_cast % 4 == 0, "tcow_layernorm_bwd: dx_cast stride must be a multiple of 4");
    const bool want_param_grads = dgamma != nullptr || dbeta != nullptr;
    TCOW_CHECK_ARG(!want_param_grads || (dgamma && dbeta && workspace && workspace_bytes >= tcow_layernorm_bwd_workspace_bytes(D)),
                   "tcow_layernorm_bwd: parameter gradients need dgamma, dbeta and a workspace of %ld bytes", tcow_layernorm_bwd_workspace_bytes(D));
    TCOW_CHECK_ARG(!colsum_out || want_param_grads, "tcow_layernorm_bwd: colsum_out rides on the parameter-gradient pass (dgamma / dbeta needed)");
    const bool csum = colsum_out != nullptr;
    const int max_blocks = csum ? kLnBwdBlocksCsum : kLnBwdBlocks;
    int blocks = cdiv(rows, 4); if (blocks > max_blocks) blocks = max_blocks;
    float* part = want_param_grads ? (float*)workspace : nullptr;
    const size_t lds = want_param_grads ? (size_t)4 * D * 4 : 0;
    if (dtype != TCOW_BF16 && dtype != TCOW_F32) { tcow_set_error("tcow_layernorm_bwd: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    const int nvl = (D / 4 + 63) / 64;
#define LN_BWD(NVV, FULLV)                                                                                                                               \
    do {                                                                                                                                          \
        if (dtype == TCOW_BF16 && csum) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, NVV, true, FULLV>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, rows, D, (const bf16_t*)dy, lddy, x, ldx, mean, rstd, gamma, dres, lddres, dx, lddx, part, (bf16_t*)dx_cast, lddx_cast, cast_row_scale, colsum_row_scale); \
        else if (dtype == TCOW_BF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, NVV, false, FULLV>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, rows, D, (const bf16_t*)dy, lddy, x, ldx, mean, rstd, gamma, dres, lddres, dx, lddx, part, (bf16_t*)dx_cast, lddx_cast, cast_row_scale, colsum_row_scale); \
        else if (csum) hipLaunchKernelGGL((ln_bwd_kernel<float, NVV, true, FULLV>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, rows, D, (const float*)dy, lddy, x, ldx, mean, rstd, gamma, dres, lddres, dx, lddx, part, (float*)dx_cast, lddx_cast, cast_row_scale, colsum_row_scale); \
        else hipLaunchKernelGGL((ln_bwd_kernel<float, NVV, false, FULLV>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, rows, D, (const float*)dy, lddy, x, ldx, mean, rstd, gamma, dres, lddres, dx, lddx, part, (float*)dx_cast, lddx_cast, cast_row_scale, colsum_row_scale); \
    } while (0)
    const bool full = D == 256 * nvl && nvl <= 4;
    if (full) { if (nvl == 1) LN_BWD(1, true); else if (nvl == 2) LN_BWD(2, true); else if (nvl == 3) LN_BWD(3, true); else LN_BWD(4, true); }
    else if (nvl <= 1) LN_BWD(1, false); else if (nvl == 2) LN_BWD(2, false); else if (nvl == 3) LN_BWD(3, false); else if (nvl == 4) LN_BWD(4, false); else LN_BWD(8, false);
#undef LN_BWD
    TCOW_CHECK_LAUNCH();
    if (want_param_grads && !(accumulate & 2)) {
        // part is [blocks][2 or 3][D]: dgamma partials first, dbeta partials at +D, the fused bias gradient at +2D
        if (csum) return tcow_launch_row_reduce3((hipStream_t)stream, part, blocks, 3L * D, D, dgamma, D, dbeta, D, colsum_out, accumulate & 1);
        return tcow_launch_row_reduce2((hipStream_t)stream, part, blocks, 2L * D, D, dgamma, D, dbeta, accumulate & 1);
    }
    return TCOW_OK;            // (accumulate & 2: the partial table stays in `workspace` for tcow_layernorm_fold)
}

int tcow_layernorm_bwd_parts(int rows, int with_colsum) {
    const int max_blocks = with_colsum ? kLnBwdBlocksCsum : kLnBwdBlocks;
    int blocks = cdiv(rows, 4); if (blocks > max_blocks) blocks = max_blocks;
    return blocks;
}

int tcow_layernorm_fold(void* stream, int n, const tcow_ln_fold_job* jobs) {
    TCOW_CHECK_ARG(n > 0 && n <= 16 && jobs, "tcow_layernorm_fold: 1..16 jobs");
    const float* part[16]; int nrows[16]; long ld[16]; int N1[16], N2[16], N3[16], acc[16]; float* o1[16]; float* o2[16]; float* o3[16];
    for (int i = 0; i < n; ++i) {
        const tcow_ln_fold_job& j = jobs[i];
        TCOW_CHECK_ARG(j.part && j.dgamma && j.dbeta && j.parts > 0 && j.D > 0, "tcow_layernorm_fold: bad job %d", i);
        part[i] = j.part; nrows[i] = j.parts; ld[i] = (j.colsum_out ? 3L : 2L) * j.D; N1[i] = N2[i] = N3[i] = j.D; acc[i] = j.accumulate;
        o1[i] = j.dgamma; o2[i] = j.dbeta; o3[i] = j.colsum_out;
    }
    return tcow_launch_row_reduce_group((hipStream_t)stream, n, part, nrows, ld, N1, o1, N2, o2, N3, o3, acc);
}

}  // extern "C"
