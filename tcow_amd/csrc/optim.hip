// Fused gradient clipping + AdamW over all parameters of the Seeker in three launches (train.py:99-102:
// clip_grad_norm_(seeker.parameters(), 0.3) then AdamW.step(); torch defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0.01 on
// every parameter).  The reference's foreach implementation walks ~250 tensors from Python / ATen per step;
// here a device-resident chunk table (<= 65536 elements per chunk) lets one grid cover every tensor.
//   1. sumsq_kernel   : per-chunk sum of squares of the gradient
//   2. clipcoef_kernel: total norm -> clip coefficient min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_);
//                       a NaN / infinite norm -> -1 = "skip this step" (the update kernel returns at once)
//   3. adamw_kernel   : p, m, v update with the clipped gradient (decoupled weight decay, bias correction as torch.optim.AdamW)
#include "common.h"

namespace {

struct Chunk { float* p; const float* g; float* m; float* v; long n; };

__global__ __launch_bounds__(256) void sumsq_kernel(const Chunk* __restrict__ chunks, float* __restrict__ partial) {
    __shared__ float red[4];
    const Chunk c = chunks[blockIdx.x];
    float s = 0.f;
    const long n4 = c.n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) { const float4 g = ld4(c.g + i * 4); s += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w; }
    for (long i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) s += c.g[i] * c.g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// gscale (may be NULL): the gradients in memory are g_true / *gscale' -- i.e. still multiplied by a loss scale whose INVERSE is *gscale (a power of two).
// The norm is taken of the true gradients and the coefficient handed to the update kernel carries the inverse scale, so the update kernel's
// g * coef is clip(g_true) exactly (powers of two commute with every rounding here): the binary16 mode's unscaling pass over all gradients
// (488 MB read + written per step) disappears into this one-workgroup kernel.
__global__ __launch_bounds__(256) void clipcoef_kernel(const float* __restrict__ partial, int n, float max_norm, float* __restrict__ out /* [0]=coef, [1]=norm, [2]+=skipped */,
                                                       const float* __restrict__ gscale) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float gs = gscale ? gscale[0] : 1.0f;
        const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3])) * gs;
        float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.0f;
        // a non-finite gradient norm (an overflow in a 16-bit backward) marks the step as skipped: coefficient -1
        const bool skip = !(norm <= 3.0e38f);
        out[0] = skip ? -1.0f : (coef < 1.0f ? coef : 1.0f) * gs;
        out[1] = norm;
        if (skip) out[2] += 1.0f;                  // steps skipped so far (every precision): the update kernel counts them out of its bias correction
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(const Chunk* __restrict__ chunks, const float* __restrict__ coef_ptr, float lr, float beta1, float beta2, float eps,
                                                    float weight_decay, int step) {
    const Chunk c = chunks[blockIdx.x];
    const float coef = coef_ptr[0];
    if (coef < 0.f) return;                        // non-finite gradients: parameters and moments stay as they are (like a GradScaler skip)
    // bias correction with the number of updates actually APPLIED: the caller counts calls, skipped ones are subtracted here, so the moments
    // and their correction stay in step after an overflow (torch's GradScaler does not advance the optimizer on a skipped step either)
    const float eff = fmaxf((float)step - coef_ptr[2], 1.0f);
    const float bc1 = 1.0f - powf(beta1, eff), bc2_sqrt = sqrtf(1.0f - powf(beta2, eff));
    const float decay = 1.0f - lr * weight_decay, step_size = lr / bc1;
    const long n4 = c.n >> 2;
    auto upd = [&](float& p, float g, float& m, float& v) {
        g *= coef;
        p *= decay;
        m = beta1 * m + (1.0f - beta1) * g;
        v = beta2 * v + (1.0f - beta2) * g * g;
        p -= step_size * m / (sqrtf(v) / bc2_sqrt + eps);
    };
    for (long i = threadIdx.x; i < n4; i += 256) {
        float4 p = ld4(c.p + i * 4), m = ld4(c.m + i * 4), v = ld4(c.v + i * 4);
        const float4 g = ld4(c.g + i * 4);
        upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
        st4(c.p + i * 4, p); st4(c.m + i * 4, m); st4(c.v + i * 4, v);
    }
    for (long i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) upd(c.p[i], c.g[i], c.m[i], c.v[i]);
}

// The same update for one 64 x 64 tile of a GEMM weight [N, K] -- and, while the new values are in registers, the two 16-bit operand copies the GEMMs read
// (W [N, K] and W^T [K, N] in the library's 16-bit format): the separate re-cast pass (tcow_cast_transpose_batched: 488 MB of f32 read again + the same
// 488 MB of copies written) shrinks to the weights this kernel does not own (the folded products).  Same arithmetic, same operation order as adamw_kernel:
// parameters and moments are bit-identical; the copies equal a cast of the stored f32 value.  Tile pointers are the tile's ORIGIN in each array.
struct CastTile { float* p; const float* g; float* m; float* v; bf16_t* wc; bf16_t* wt; int K, N; };

__global__ __launch_bounds__(256) void adamw_cast_kernel(const CastTile* __restrict__ tiles, const float* __restrict__ coef_ptr, float lr, float beta1, float beta2, float eps,
                                                         float weight_decay, int step) {
    __shared__ float tile[64][65];
    const float coef = coef_ptr[0];
    if (coef < 0.f) return;                        // skipped step: parameters unchanged, the copies stay valid
    const CastTile d = tiles[blockIdx.x];
    const float eff = fmaxf((float)step - coef_ptr[2], 1.0f);
    const float bc1 = 1.0f - powf(beta1, eff), bc2_sqrt = sqrtf(1.0f - powf(beta2, eff));
    const float decay = 1.0f - lr * weight_decay, step_size = lr / bc1;
    auto upd = [&](float& p, float g, float& m, float& v) {
        g *= coef;
        p *= decay;
        m = beta1 * m + (1.0f - beta1) * g;
        v = beta2 * v + (1.0f - beta2) * g * g;
        p -= step_size * m / (sqrtf(v) / bc2_sqrt + eps);
    };
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i;
        const size_t o = (size_t)r * d.K + c4;
        float4 p = ld4(d.p + o), m = ld4(d.m + o), v = ld4(d.v + o);
        const float4 g = ld4(d.g + o);
        upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
        st4(d.p + o, p); st4(d.m + o, m); st4(d.v + o, v);
        st4(d.wc + o, p);
        tile[r][c4] = p.x; tile[r][c4 + 1] = p.y; tile[r][c4 + 2] = p.z; tile[r][c4 + 3] = p.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = r0 + 16 * i;
        st4(d.wt + (size_t)k * d.N + c4, make_float4(tile[c4][k], tile[c4 + 1][k], tile[c4 + 2][k], tile[c4 + 3][k]));
    }
}

}  // namespace

extern "C" {

long tcow_adamw_chunk_bytes(void) { return (long)sizeof(Chunk); }
long tcow_adamw_tile_bytes(void) { return (long)sizeof(CastTile); }

// tcow_adamw_clip_step_scaled with the operand copies of the GEMM weights written by the update itself (ABI 10).  chunks[0 .. n_chunks): every parameter
// (the gradient norm is taken over all of them, partial sums in this order); flat_chunks[0 .. n_flat): the parameters the flat kernel updates; the others tile by tile:
// tiles[0 .. n_tiles) = {p, g, m, v, wc, wt, K, N} records of 64 x 64 tiles (pointers at the tile's origin; N, K multiples of 64; wc / wt in this
// library's 16-bit format).  scratch as in tcow_adamw_clip_step: f32 [n_chunks + 3].
int tcow_adamw_clip_step_cast(void* stream, const void* chunks, int n_chunks, const void* flat_chunks, int n_flat, const void* tiles, int n_tiles, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float max_norm, float* scratch, const float* grad_inv_scale) {
    TCOW_CHECK_ARG(chunks && scratch && n_chunks > 0 && step >= 1 && n_flat >= 0 && (flat_chunks || n_flat == 0) && n_tiles >= 0 && (tiles || n_tiles == 0),
                   "tcow_adamw_clip_step_cast: bad arguments");
    const Chunk* c = (const Chunk*)chunks;
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, c, scratch);
    TCOW_CHECK_LAUNCH();
    hipLaunchKernelGGL(clipcoef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, n_chunks, max_norm, scratch + n_chunks, grad_inv_scale);
    TCOW_CHECK_LAUNCH();
    if (n_flat > 0) {
        hipLaunchKernelGGL(adamw_kernel, dim3(n_flat), dim3(256), 0, (hipStream_t)stream, (const Chunk*)flat_chunks, scratch + n_chunks, lr, beta1, beta2, eps, weight_decay, step);
        TCOW_CHECK_LAUNCH();
    }
    if (n_tiles > 0) {
        hipLaunchKernelGGL(adamw_cast_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, (const CastTile*)tiles, scratch + n_chunks, lr, beta1, beta2, eps, weight_decay, step);
        TCOW_CHECK_LAUNCH();
    }
    return TCOW_OK;
}

// chunks: device array of n_chunks {p, g, m, v, n} records (all f32, 16-byte aligned starts); scratch: f32 [n_chunks + 3];
// on return scratch[n_chunks] = clip coefficient (-1 = step skipped), scratch[n_chunks + 1] = total gradient norm,
// scratch[n_chunks + 2] += 1 if the step was skipped (the caller zeroes it once; device side, no sync).  `step` = number of calls so far
// (>= 1); the bias correction uses step - scratch[n_chunks + 2].
// tcow_adamw_clip_step_scaled: the gradients are still multiplied by a loss scale; grad_inv_scale (device scalar, a power of two; NULL = 1) is its inverse.
int tcow_adamw_clip_step_scaled(void* stream, const void* chunks, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                float max_norm, float* scratch, const float* grad_inv_scale) {
    TCOW_CHECK_ARG(chunks && scratch && n_chunks > 0 && step >= 1, "tcow_adamw_clip_step: bad arguments");
    const Chunk* c = (const Chunk*)chunks;
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, c, scratch);
    TCOW_CHECK_LAUNCH();
    hipLaunchKernelGGL(clipcoef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, n_chunks, max_norm, scratch + n_chunks, grad_inv_scale);
    TCOW_CHECK_LAUNCH();
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, c, scratch + n_chunks, lr, beta1, beta2, eps, weight_decay, step);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_adamw_clip_step(void* stream, const void* chunks, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         float max_norm, float* scratch) {
    return tcow_adamw_clip_step_scaled(stream, chunks, n_chunks, lr, beta1, beta2, eps, weight_decay, step, max_norm, scratch, nullptr);
}

}  // extern "C"
