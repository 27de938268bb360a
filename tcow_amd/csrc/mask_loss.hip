// Caller row L: the per-channel mask objective of loss.py:164-225 as one forward+backward pass family.
//
//   loss = [ aot * (boot + jac) / 2 + (1 - aot) * mean(w * bce) ] * sqrt(n_sel / n)           loss.py:196-222
//   boot = mean(top-k of (w *) bce), k = int(topk_frac * n_sel)                               loss.py:13-17
//   jac  = 1 - sum(p g) / (sum(p g) + sum(p (1-g)) + sum((1-p) g) + 0.1)   (or = boot)        loss.py:20-32
//
// Everything the reference decides on the host (which frames carry weight, mean weight >= 1e-4, empty target, k == n)
// is decided on the device from a control block, so the step has no host synchronisation.  The k-th largest loss value
// is found exactly by an 11/11/9-bit radix select over the float bit pattern (loss values are >= 0, so the unsigned
// pattern orders like the value); elements equal to the k-th value share the remaining weight equally (a valid
// sub-gradient where torch.topk picks an arbitrary subset of the ties).
// Pass A and the gradient pass stream x, t, w once (12 B/pixel; + 4 B/pixel written: the loss values' bit patterns / the gradient); the two
// radix passes in between read only those bit patterns (4 B/pixel, no arithmetic -- round 5: they recomputed the loss from x, t, w before).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kVecPerBlock = 1024;          // float4 groups per workgroup (4 per thread)
constexpr int kBins12 = 2048, kBins3 = 512;
constexpr int kNPart = 6;

struct MlView {
    const float* x; long xs;
    const float* t; long ts;
    const float* pw;                        // dense [n_frames * frame_len] or null
    const float* fw;                        // [n_frames] or null
    float* dx; long dxs;
    int T; int L4; int n_frames; int weighted;
    int focal;                              // loss.py:49-51: sigmoid focal loss (alpha 0.25, gamma 2) in place of the plain BCE term
};

struct MlCtl {
    long n_sel, n_all, k, above, cnt_eq, r;
    int valid, mode_all, jac_valid, pad;
    unsigned b1, b2, tau, pad2;
    double S[kNPart];
    double topk_sum;
};

struct MlWs {
    int* fsel; double* fwsum; double* part; unsigned* hist; MlCtl* ctl; int nblk;
    unsigned* vb;                           // [n_frames * frame_len] bit patterns of the (weighted) loss values: written by pass A, read by the radix passes
};

// Up to kMaxJobs channels of the objective as ONE set of launches (round 5: the three channels of a step were 30 launches of 2-40 us each; the
// workgroups of a pass now carry their job in blockIdx.z, the one-workgroup kernels run one workgroup per job): job = one tcow_mask_loss call.
constexpr int kMaxJobs = 4;
struct MlJob { MlView v; MlWs ws; float aot, lw; double topk_frac; float* loss; int radix; };
struct MlBatch { int n; float* total; MlJob j[kMaxJobs]; };

// bce-with-logits and sigmoid of one pixel; identical instruction sequence in every pass (the radix passes rely on it)
__device__ __forceinline__ void bce_sig(float x, float t, float& bce, float& p) {
    const float e = expf(-fabsf(x));
    const float inv = 1.0f / (1.0f + e);
    p = x >= 0.0f ? inv : e * inv;
    bce = fmaxf(x, 0.0f) - x * t + log1pf(e);
}

// The per-pixel loss value and its derivative d(value)/d(logit).  focal = 0: BCE, derivative p - t.  focal = 1 (train_args.focal_loss, loss.py:49-51 =
// torchvision.ops.sigmoid_focal_loss with its defaults alpha = 0.25, gamma = 2, reduction 'none'): value = a_t (1 - p_t)^2 bce with
// p_t = p t + (1 - p)(1 - t), a_t = 0.25 t + 0.75 (1 - t); derivative a_t (1 - p_t) [(1 - p_t)(p - t) - 2 p (1 - p)(2 t - 1) bce].  One instruction
// sequence for every pass (the radix select compares bit patterns of the value across passes).
__device__ __forceinline__ void pix_loss(float x, float t, int focal, float& val, float& dval, float& p) {
    float bce; bce_sig(x, t, bce, p);
    if (!focal) { val = bce; dval = p - t; return; }
    const float pt = p * t + (1.0f - p) * (1.0f - t), at = 0.25f * t + 0.75f * (1.0f - t), q = 1.0f - pt;
    val = at * bce * (q * q);
    dval = at * q * (q * (p - t) - 2.0f * p * (1.0f - p) * (2.0f * t - 1.0f) * bce);
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < kThreads / 64; ++i) s += red[i];
    return s;
}

__device__ __forceinline__ long elem_off(const MlView& v, int f, long stride) {
    const int s = f / v.T;
    return (long)s * stride + (long)(f - s * v.T) * v.L4 * 4;
}

// ---- pass 0: which frames carry weight (loss.py:176-181) and the sum of the weights (loss.py:184)
__device__ __forceinline__ void ml_frames_body(const MlView& v, const MlWs& ws) {
    __shared__ double red[kThreads / 64];
    const int f = blockIdx.x;
    const float fwv = v.fw ? v.fw[f] : 1.0f;
    double sum = 0.0; int any = 0;
    if (v.pw) {
        const float4* pw = reinterpret_cast<const float4*>(v.pw + (long)f * v.L4 * 4);
        float s = 0.f;
        for (int i = threadIdx.x; i < v.L4; i += kThreads) {
            float4 w = pw[i];
            w.x *= fwv; w.y *= fwv; w.z *= fwv; w.w *= fwv;
            any |= (w.x != 0.f) | (w.y != 0.f) | (w.z != 0.f) | (w.w != 0.f);
            s += (w.x + w.y) + (w.z + w.w);
        }
        sum = s;
    } else if (threadIdx.x == 0) {
        any = fwv != 0.f; sum = (double)fwv * (v.L4 * 4.0);
    }
    sum = block_sum(sum, red);
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) { ws.fsel[f] = any; ws.fwsum[f] = sum; }
}

// ---- control: n_sel, validity, k; clears the histograms
__device__ __forceinline__ void ml_ctl_body(const MlView& v, const MlWs& ws, double topk_frac) {
    __shared__ double red[kThreads / 64];
    double nf = 0.0, wsum = 0.0;
    for (int f = threadIdx.x; f < v.n_frames; f += kThreads) { nf += ws.fsel[f] ? 1.0 : 0.0; wsum += ws.fwsum[f]; }
    nf = block_sum(nf, red); wsum = block_sum(wsum, red);
    for (int i = threadIdx.x; i < 2 * kBins12 + kBins3; i += kThreads) ws.hist[i] = 0u;
    if (threadIdx.x == 0) {
        MlCtl c = {};
        c.n_all = (long)v.n_frames * v.L4 * 4;
        c.n_sel = (long)nf * v.L4 * 4;
        const float wmean = (float)(wsum / (double)c.n_all);
        c.valid = c.n_sel > 0 && wmean >= 1e-4f;                           // loss.py:184
        c.k = (long)(topk_frac * (double)c.n_sel);                         // loss.py:14
        c.mode_all = c.k >= c.n_sel;
        *ws.ctl = c;
    }
}

// Iterates this workgroup's float4 groups of frame blockIdx.y; F(x4, t4, w4, offset into the frame in elements)
template <typename F> __device__ __forceinline__ void for_each_vec(const MlView& v, int f, F&& fn) {
    const float4* x = reinterpret_cast<const float4*>(v.x + elem_off(v, f, v.xs));
    const float4* t = reinterpret_cast<const float4*>(v.t + elem_off(v, f, v.ts));
    const float4* pw = v.pw ? reinterpret_cast<const float4*>(v.pw + (long)f * v.L4 * 4) : nullptr;
    const float fwv = v.fw ? v.fw[f] : 1.0f;
    const int base = blockIdx.x * kVecPerBlock;
#pragma unroll
    for (int j = 0; j < kVecPerBlock / kThreads; ++j) {
        const int i = base + j * kThreads + threadIdx.x;
        if (i < v.L4) {
            float4 w = pw ? pw[i] : make_float4(1.f, 1.f, 1.f, 1.f);
            w.x *= fwv; w.y *= fwv; w.z *= fwv; w.w *= fwv;
            fn(x[i], t[i], w, i);
        }
    }
}

__device__ __forceinline__ unsigned vbits(float bce, float w, int weighted) {
    const unsigned u = __builtin_bit_cast(unsigned, weighted ? bce * w : bce);
    return (int)u < 0 ? 0u : u;                 // (-0 / rounding below zero with soft targets) orders as zero
}

// ---- pass A: sums for the three terms + level-1 histogram of the loss values
__device__ __forceinline__ void ml_stats_body(const MlView& v, const MlWs& ws, int radix) {
    __shared__ unsigned hist[kBins12];
    __shared__ double red[kThreads / 64];
    const int f = blockIdx.y;
    const MlCtl* c = ws.ctl;
    const bool live = c->valid && ws.fsel[f];
    const bool need_hist = live && !c->mode_all && radix;      // (no radix select -- topk_frac = 1 or aot_loss = 0 --: no histogram, and the workspace has no bit-pattern image)
    if (need_hist) { for (int i = threadIdx.x; i < kBins12; i += kThreads) hist[i] = 0u; }
    __syncthreads();
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        uint4* vbo = need_hist ? reinterpret_cast<uint4*>(ws.vb + (long)f * v.L4 * 4) : nullptr;
        for_each_vec(v, f, [&](float4 x, float4 t, float4 w, int i) {
            const float xs[4] = {x.x, x.y, x.z, x.w}, ts[4] = {t.x, t.y, t.z, t.w}, wv[4] = {w.x, w.y, w.z, w.w};
            unsigned vbs[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bce, dval, p; pix_loss(xs[e], ts[e], v.focal, bce, dval, p);
                const unsigned vb = vbits(bce, wv[e], v.weighted);
                vbs[e] = vb;
                s[0] += bce * wv[e]; s[1] += __builtin_bit_cast(float, vb);
                s[2] += p * ts[e]; s[3] += p; s[4] += ts[e];
                if (need_hist) atomicAdd(&hist[vb >> 20], 1u);
            }
            // the radix passes that follow only need the bit patterns: 4 B/pixel instead of x, t, w again (12 B/pixel) and no arithmetic
            if (vbo) vbo[i] = make_uint4(vbs[0], vbs[1], vbs[2], vbs[3]);
        });
    }
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    for (int q = 0; q < 5; ++q) {
        const double tot = block_sum((double)s[q], red);
        if (threadIdx.x == 0) ws.part[(long)blk * kNPart + q] = tot;
    }
    if (need_hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < kBins12; i += kThreads)
            if (hist[i]) atomicAdd(&ws.hist[i], hist[i]);
    }
}

// ---- levels 2 and 3 of the radix select: histogram the next digit of the values that share the known prefix
template <int LEVEL> __device__ __forceinline__ void ml_hist_body(const MlView& v, const MlWs& ws) {
    __shared__ unsigned hist[kBins12];
    const int f = blockIdx.y;
    const MlCtl* c = ws.ctl;
    if (!(c->valid && ws.fsel[f]) || c->mode_all) return;
    constexpr int NB = LEVEL == 2 ? kBins12 : kBins3;
    for (int i = threadIdx.x; i < NB; i += kThreads) hist[i] = 0u;
    __syncthreads();
    const unsigned prefix = LEVEL == 2 ? c->b1 : ((c->b1 << 11) | c->b2);
    const uint4* vbi = reinterpret_cast<const uint4*>(ws.vb + (long)f * v.L4 * 4);          // pass A's bit patterns of this frame
    const int base = blockIdx.x * kVecPerBlock;
#pragma unroll
    for (int j = 0; j < kVecPerBlock / kThreads; ++j) {
        const int i = base + j * kThreads + threadIdx.x;
        if (i >= v.L4) continue;
        const uint4 q = vbi[i];
        const unsigned vbs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned vb = vbs[e];
            if (LEVEL == 2) { if ((vb >> 20) == prefix) atomicAdd(&hist[(vb >> 9) & 2047u], 1u); }
            else            { if ((vb >> 9) == prefix) atomicAdd(&hist[vb & 511u], 1u); }
        }
    }
    __syncthreads();
    unsigned* gh = ws.hist + (LEVEL == 2 ? kBins12 : 2 * kBins12);
    for (int i = threadIdx.x; i < NB; i += kThreads)
        if (hist[i]) atomicAdd(&gh[i], hist[i]);
}

// ---- one wave walks a histogram from the top for the bin holding the (k - above)-th largest value;
//      level 1 also folds the pass-A partial sums (fixed order: deterministic)
template <int LEVEL> __device__ __forceinline__ void ml_scan_body(const MlView& v, const MlWs& ws) {
    __shared__ double red[kThreads / 64];
    MlCtl* c = ws.ctl;
    if (LEVEL == 1) {
        for (int q = 0; q < 5; ++q) {
            double s = 0.0;
            for (int b = threadIdx.x; b < ws.nblk; b += kThreads) s += ws.part[(long)b * kNPart + q];
            s = block_sum(s, red);
            if (threadIdx.x == 0) c->S[q] = s;
        }
        if (threadIdx.x == 0) {
            const float tmean = c->n_sel > 0 ? (float)(c->S[4] / (double)c->n_sel) : 0.f;
            c->jac_valid = !v.weighted && tmean >= 1e-6f;                  // loss.py:21
        }
    }
    if (!c->valid || c->mode_all || threadIdx.x >= 64) return;
    constexpr int NB = LEVEL == 3 ? kBins3 : kBins12;
    constexpr int PER = NB / 64;
    const unsigned* h = ws.hist + (LEVEL == 1 ? 0 : LEVEL == 2 ? kBins12 : 2 * kBins12);
    const long krem = c->k - (LEVEL == 1 ? 0 : c->above);
    const int lane = threadIdx.x;
    long mine = 0;
    for (int i = 0; i < PER; ++i) mine += h[lane * PER + i];
    long incl = mine;                                                     // inclusive suffix sum over lanes >= lane
    for (int o = 1; o < 64; o <<= 1) {
        const long up = __shfl_down(incl, o, 64);
        if (lane + o < 64) incl += up;
    }
    const long above_lane = incl - mine;
    if (above_lane < krem && krem <= incl) {                               // exactly one lane
        long acc = above_lane;
        for (int b = lane * PER + PER - 1; b >= lane * PER; --b) {
            const long hb = h[b];
            if (acc + hb >= krem) {
                const long above = (LEVEL == 1 ? 0 : c->above) + acc;
                c->above = above;
                if (LEVEL == 1) c->b1 = (unsigned)b;
                else if (LEVEL == 2) c->b2 = (unsigned)b;
                else { c->tau = (c->b1 << 20) | (c->b2 << 9) | (unsigned)b; c->cnt_eq = hb; c->r = c->k - above; }
                break;
            }
            acc += hb;
        }
    }
}

// ---- pass E: gradient of the channel loss w.r.t. the logits + the top-k sum
__device__ __forceinline__ void ml_grad_body(const MlView& v, const MlWs& ws, float aot, float lw) {
    __shared__ double red[kThreads / 64];
    const int f = blockIdx.y;
    const MlCtl* c = ws.ctl;
    const bool live = c->valid && ws.fsel[f];
    float4* dx = v.dx ? reinterpret_cast<float4*>(v.dx + elem_off(v, f, v.dxs)) : nullptr;
    float tk = 0.f;
    if (!live) {
        if (dx) {
            const int base = blockIdx.x * kVecPerBlock;
            for (int j = 0; j < kVecPerBlock / kThreads; ++j) {
                const int i = base + j * kThreads + threadIdx.x;
                if (i < v.L4) dx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    } else {
        const float sf = lw * sqrtf((float)((double)c->n_sel / (double)c->n_all));     // loss.py:222
        const float inv_n = (float)(1.0 / (double)c->n_sel);
        const float inv_k = c->k > 0 ? (float)(1.0 / (double)c->k) : 0.f;
        const float c_custom = aot > 0.f ? sf * (1.f - aot) * inv_n : sf * inv_n;
        const float c_boot = aot > 0.f ? sf * aot * (v.weighted ? 1.f : 0.5f) * inv_k : 0.f;
        const float c_jac = (aot > 0.f && c->jac_valid) ? sf * aot * 0.5f : 0.f;
        const float num = (float)c->S[2];
        const float D = (float)(c->S[3] + c->S[4] - c->S[2]) + 0.1f;                  // num + sum p(1-g) + sum (1-p)g + eps
        const float invD2 = 1.f / (D * D);
        const int mode_all = c->mode_all || aot <= 0.f;
        const unsigned tau = c->tau;
        const float tie = (mode_all || c->cnt_eq <= 0) ? 1.f : (float)((double)c->r / (double)c->cnt_eq);
        for_each_vec(v, f, [&](float4 x, float4 t, float4 w, int i) {
            const float xs[4] = {x.x, x.y, x.z, x.w}, ts[4] = {t.x, t.y, t.z, t.w}, wv[4] = {w.x, w.y, w.z, w.w};
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bce, d, p; pix_loss(xs[e], ts[e], v.focal, bce, d, p);
                const unsigned vb = vbits(bce, wv[e], v.weighted);
                const float sel = mode_all ? 1.f : (vb > tau ? 1.f : (vb == tau ? tie : 0.f));
                tk += sel * __builtin_bit_cast(float, vb);
                float gr = c_custom * wv[e] * d + c_boot * sel * (v.weighted ? wv[e] : 1.f) * d;
                const float dJ = -(ts[e] * D - num * (1.f - ts[e])) * invD2;
                gr += c_jac * dJ * p * (1.f - p);
                g[e] = gr;
            }
            if (dx) dx[i] = make_float4(g[0], g[1], g[2], g[3]);
        });
    }
    const double tot = block_sum((double)tk, red);
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) ws.part[(long)blk * kNPart + 5] = tot;
}

// ---- final: the scalar
__device__ __forceinline__ void ml_final_body(const MlView& v, const MlWs& ws, float aot, float loss_weight, float* loss, float* total) {
    __shared__ double red[kThreads / 64];
    MlCtl* c = ws.ctl;
    double s = 0.0;
    for (int b = threadIdx.x; b < ws.nblk; b += kThreads) s += ws.part[(long)b * kNPart + 5];
    s = block_sum(s, red);
    if (threadIdx.x != 0) return;
    c->topk_sum = s;
    float L = 0.f;
    if (c->valid) {
        const double n = (double)c->n_sel;
        const float custom = (float)(c->S[0] / n);
        if (aot > 0.f) {
            const float boot = c->mode_all ? (float)(c->S[1] / n) : (float)(s / (double)c->k);
            float jac = boot;
            if (!v.weighted) {
                const float num = (float)c->S[2];
                const float den = (float)(c->S[3] + c->S[4] - c->S[2]);
                jac = c->jac_valid ? 1.f - num / (den + 0.1f) : 0.f;
            }
            L = (boot + jac) / 2.f * aot + custom * (1.f - aot);
        } else {
            L = custom;
        }
        L *= sqrtf((float)(n / (double)c->n_all));
    }
    *loss = L;
    if (total) *total += loss_weight * L;
}

// ---- the launches: one per pass for all jobs of a batch
__global__ void __launch_bounds__(kThreads) ml_frames_kernel(MlBatch b) { const MlJob& J = b.j[blockIdx.y]; ml_frames_body(J.v, J.ws); }
__global__ void __launch_bounds__(kThreads) ml_ctl_kernel(MlBatch b) { const MlJob& J = b.j[blockIdx.x]; ml_ctl_body(J.v, J.ws, J.topk_frac); }
__global__ void __launch_bounds__(kThreads) ml_stats_kernel(MlBatch b) { const MlJob& J = b.j[blockIdx.z]; ml_stats_body(J.v, J.ws, J.radix); }
template <int LEVEL> __global__ void __launch_bounds__(kThreads) ml_hist_kernel(MlBatch b) { const MlJob& J = b.j[blockIdx.z]; if (J.radix) ml_hist_body<LEVEL>(J.v, J.ws); }
template <int LEVEL> __global__ void __launch_bounds__(kThreads) ml_scan_kernel(MlBatch b) { const MlJob& J = b.j[blockIdx.x]; if (LEVEL == 1 || J.radix) ml_scan_body<LEVEL>(J.v, J.ws); }
__global__ void __launch_bounds__(kThreads) ml_grad_kernel(MlBatch b) { const MlJob& J = b.j[blockIdx.z]; ml_grad_body(J.v, J.ws, J.aot, J.lw); }
// (one workgroup walks the jobs in order: `total` is accumulated in job order, as the one-launch-per-channel form did)
__global__ void __launch_bounds__(kThreads) ml_final_kernel(MlBatch b) {
    for (int i = 0; i < b.n; ++i) { const MlJob& J = b.j[i]; ml_final_body(J.v, J.ws, J.aot, J.lw, J.loss, b.total); __syncthreads(); }
}

static int ml_nblk(long n_frames, long frame_len) { return (int)(n_frames * ((frame_len / 4 + kVecPerBlock - 1) / kVecPerBlock)); }

}  // namespace

extern "C" {

static size_t ml_ws_bytes(long n_frames, long frame_len, bool radix);
size_t tcow_mask_loss_workspace_bytes(long n_frames, long frame_len) { return ml_ws_bytes(n_frames, frame_len, true); }
size_t tcow_mask_loss_workspace_bytes_for(long n_frames, long frame_len, double topk_frac, float aot_loss) {
    return ml_ws_bytes(n_frames, frame_len, topk_frac < 1.0 && aot_loss > 0.f);
}
static size_t ml_ws_bytes(long n_frames, long frame_len, bool radix) {
    if (n_frames <= 0 || frame_len <= 0) return 0;
    size_t b = 0;
    b += ((size_t)n_frames * sizeof(int) + 255) & ~(size_t)255;
    b += ((size_t)n_frames * sizeof(double) + 255) & ~(size_t)255;
    b += ((size_t)ml_nblk(n_frames, frame_len) * kNPart * sizeof(double) + 255) & ~(size_t)255;
    b += (2 * kBins12 + kBins3) * sizeof(unsigned);
    b += 512;
    if (radix) b += (size_t)n_frames * (size_t)frame_len * sizeof(unsigned);           // the loss values' bit patterns between pass A and the radix passes
    return b;
}

static int ml_make_job(const tcow_mask_loss_args* a, MlJob& J) {
    TCOW_CHECK_ARG(a && a->logits && a->target && a->loss && a->ws, "tcow_mask_loss: null argument");
    TCOW_CHECK_ARG(a->n_frames > 0 && a->frame_len > 0 && a->frames_per_seq > 0 && a->n_frames % a->frames_per_seq == 0,
                   "tcow_mask_loss: bad frame counts");
    TCOW_CHECK_ARG(a->frame_len % 4 == 0, "tcow_mask_loss: frame_len %ld must be a multiple of 4", a->frame_len);
    TCOW_CHECK_ARG(a->logits_seq_stride % 4 == 0 && a->target_seq_stride % 4 == 0 && a->dlogits_seq_stride % 4 == 0,
                   "tcow_mask_loss: sequence strides must be multiples of 4 elements");
    TCOW_CHECK_ARG(a->n_frames * (a->frame_len / 4) < (1L << 31) && a->n_frames < 65536, "tcow_mask_loss: too many pixels for one call");
    TCOW_CHECK_ARG(a->topk_frac > 0.0 && a->topk_frac <= 1.0, "tcow_mask_loss: topk_frac %g outside (0, 1]", a->topk_frac);
    TCOW_CHECK_ARG(a->ws_bytes >= tcow_mask_loss_workspace_bytes_for(a->n_frames, a->frame_len, a->topk_frac, a->aot_loss), "tcow_mask_loss: workspace too small");
    MlView& v = J.v;
    v.x = a->logits; v.xs = a->logits_seq_stride; v.t = a->target; v.ts = a->target_seq_stride;
    v.pw = a->pixel_w; v.fw = a->frame_w; v.dx = a->dlogits; v.dxs = a->dlogits_seq_stride;
    v.T = (int)a->frames_per_seq; v.L4 = (int)(a->frame_len / 4); v.n_frames = (int)a->n_frames; v.weighted = a->weighted_aot ? 1 : 0; v.focal = a->focal ? 1 : 0;
    MlWs& ws = J.ws; char* p = (char*)a->ws;
    ws.fsel = (int*)p; p += ((size_t)v.n_frames * sizeof(int) + 255) & ~(size_t)255;
    ws.fwsum = (double*)p; p += ((size_t)v.n_frames * sizeof(double) + 255) & ~(size_t)255;
    ws.nblk = ml_nblk(a->n_frames, a->frame_len);
    ws.part = (double*)p; p += ((size_t)ws.nblk * kNPart * sizeof(double) + 255) & ~(size_t)255;
    ws.hist = (unsigned*)p; p += (2 * kBins12 + kBins3) * sizeof(unsigned);
    ws.ctl = (MlCtl*)p; p += 512;
    ws.vb = (unsigned*)p;
    static_assert(sizeof(MlCtl) <= 512, "control block");
    J.aot = a->aot_loss; J.lw = a->loss_weight; J.topk_frac = a->topk_frac; J.loss = a->loss;
    J.radix = (a->topk_frac < 1.0 && a->aot_loss > 0.f) ? 1 : 0;
    return TCOW_OK;
}

// n <= 4 channels of one objective in one set of launches: the same frame geometry, the same `total` (may be NULL) -- accumulated in argument
// order --, separate workspaces.
int tcow_mask_loss_batch(void* stream, const tcow_mask_loss_args* a, int n) {
    TCOW_CHECK_ARG(a && n >= 1 && n <= kMaxJobs, "tcow_mask_loss_batch: 1 .. %d jobs (got %d)", kMaxJobs, n);
    MlBatch b; b.n = n; b.total = a[0].total;
    int any_radix = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = ml_make_job(a + i, b.j[i]);
        if (rc != TCOW_OK) return rc;
        TCOW_CHECK_ARG(a[i].n_frames == a[0].n_frames && a[i].frame_len == a[0].frame_len && a[i].total == a[0].total,
                       "tcow_mask_loss_batch: the jobs of a batch share n_frames, frame_len and total");
        for (int k = 0; k < i; ++k) TCOW_CHECK_ARG(a[k].ws != a[i].ws, "tcow_mask_loss_batch: every job needs its own workspace");
        any_radix |= b.j[i].radix;
    }
    for (int i = n; i < kMaxJobs; ++i) b.j[i] = b.j[0];
    hipStream_t st = (hipStream_t)stream;
    const MlView& v = b.j[0].v;
    const dim3 grid((v.L4 + kVecPerBlock - 1) / kVecPerBlock, v.n_frames, n);
    ml_frames_kernel<<<dim3(v.n_frames, n), kThreads, 0, st>>>(b);
    ml_ctl_kernel<<<n, kThreads, 0, st>>>(b);
    ml_stats_kernel<<<grid, kThreads, 0, st>>>(b);
    ml_scan_kernel<1><<<n, kThreads, 0, st>>>(b);
    if (any_radix) {
        ml_hist_kernel<2><<<grid, kThreads, 0, st>>>(b);
        ml_scan_kernel<2><<<n, kThreads, 0, st>>>(b);
        ml_hist_kernel<3><<<grid, kThreads, 0, st>>>(b);
        ml_scan_kernel<3><<<n, kThreads, 0, st>>>(b);
    }
    ml_grad_kernel<<<grid, kThreads, 0, st>>>(b);
    ml_final_kernel<<<1, kThreads, 0, st>>>(b);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_mask_loss(void* stream, const tcow_mask_loss_args* a) { return tcow_mask_loss_batch(stream, a, 1); }

}  // extern "C"

// ------------------------------------------------------------------------------------------ IoU area counts (caller row M)
// eval/metrics.py:19-20,55-66: per frame, |target|, |output & target|, |output | target| with output = logit > 0 and
// target = value > 0.5.  One pass over both tensors; integer counts, so the result is exact and order-independent.
namespace {
__global__ void __launch_bounds__(256) iou_counts_kernel(const float* __restrict__ logits, const float* __restrict__ target, int L4, int* __restrict__ counts) {
    __shared__ int red[3][4];
    const int f = blockIdx.x;
    const float4* x = reinterpret_cast<const float4*>(logits) + (size_t)f * L4;
    const float4* t = reinterpret_cast<const float4*>(target) + (size_t)f * L4;
    int ta = 0, in = 0, un = 0;
    for (int i = threadIdx.x; i < L4; i += 256) {
        const float4 xv = x[i], tv = t[i];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ts[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = xs[e] > 0.0f, g = ts[e] > 0.5f;
            ta += g; in += o & g; un += o | g;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { ta += __shfl_xor(ta, o, 64); in += __shfl_xor(in, o, 64); un += __shfl_xor(un, o, 64); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = ta; red[1][w] = in; red[2][w] = un; }
    __syncthreads();
    if (threadIdx.x < 3) counts[f * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
}  // namespace

extern "C" int tcow_iou_counts(void* stream, const float* logits, const float* target, long n_frames, long frame_len, int* counts) {
    TCOW_CHECK_ARG(logits && target && counts && n_frames > 0 && frame_len > 0, "tcow_iou_counts: bad arguments");
    TCOW_CHECK_ARG(frame_len % 4 == 0 && frame_len / 4 < (1L << 31) && n_frames < (1L << 31), "tcow_iou_counts: frame_len %ld must be a multiple of 4", frame_len);
    hipLaunchKernelGGL(iou_counts_kernel, dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, logits, target, (int)(frame_len / 4), counts);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// ------------------------------------------------------------------------------------------ query / target masks (caller row P)
// data/data_utils.py:414-510 for all (b, q, t) in one pass over the segmentation maps: query mask = visible pixels of the queried
// instance at the query frame (:431), target channel 0 = its amodal mask (:441), channel 1 = amodal mask of the frontmost occluder
// when the frame has one (:455-463), channel 2 = outermost container (:468-492), snitch_occl_by_ptr = occluded snitch pixels tagged
// with the occluder's id (:435-437).  Which instance is the occluder / container of a frame is decided beforehand on (B,Q,T)-sized
// tensors; this kernel does the per-pixel work: 4 B read, 17 B written per pixel.
namespace {
struct BuildMasksArgs {
    const uint8_t* segm; const uint8_t* div;          // (B,1,T,H,W), (B,M,T,H,W)
    const int* qidx;                                   // [B*Q] queried instance
    const int* front; const int* cont;                 // [B*Q*T] instance index of the frontmost occluder / outermost container, -1 = none
    float* qmask; float* target; uint8_t* ptr;         // (B,Q,1,T,H,W), (B,Q,3,T,H,W), (B,Q,1,T,H,W)
    int* counts;                                       // [1 + 2*Q]: amodal pixels of channel 0; per query: any(query mask), any(target)
    float* fw_occl; float* fw_cont; float has_weight;  // optional [B*Q*T] frame weights of channels 1 / 2 (preset to the empty-frame weight): has_weight where the frame has a mask
    int B, Q, M, T, HW16, qt;
};
__global__ void __launch_bounds__(256) build_masks_kernel(BuildMasksArgs a) {
    const int f = blockIdx.y;                          // (b*Q + q)*T + t
    const int t = f % a.T, bq = f / a.T, q = bq % a.Q, b = bq / a.Q;
    const int qi = a.qidx[bq], fi = a.front[f], ci = a.cont[f];
    const size_t plane = (size_t)a.HW16 * 16;
    const uint4* seg = reinterpret_cast<const uint4*>(a.segm + ((size_t)b * a.T + t) * plane);
    const uint4* dq = reinterpret_cast<const uint4*>(a.div + (((size_t)b * a.M + qi) * a.T + t) * plane);
    const uint4* df = fi >= 0 ? reinterpret_cast<const uint4*>(a.div + (((size_t)b * a.M + fi) * a.T + t) * plane) : nullptr;
    const uint4* dc = ci >= 0 ? reinterpret_cast<const uint4*>(a.div + (((size_t)b * a.M + ci) * a.T + t) * plane) : nullptr;
    float4* qm = reinterpret_cast<float4*>(a.qmask + (size_t)f * plane);
    float4* t0 = reinterpret_cast<float4*>(a.target + (((size_t)bq * 3 + 0) * a.T + t) * plane);
    float4* t1 = reinterpret_cast<float4*>(a.target + (((size_t)bq * 3 + 1) * a.T + t) * plane);
    float4* t2 = reinterpret_cast<float4*>(a.target + (((size_t)bq * 3 + 2) * a.T + t) * plane);
    uint4* pt = reinterpret_cast<uint4*>(a.ptr + (size_t)f * plane);
    int n_amodal = 0, any_q = 0, any_t = 0, any_f = 0, any_c = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.HW16; i += gridDim.x * 256) {
        const uint4 sv = seg[i], qv = dq[i];
        const uint4 fv = df ? df[i] : make_uint4(0, 0, 0, 0), cv = dc ? dc[i] : make_uint4(0, 0, 0, 0);
        const uint32_t sw[4] = {sv.x, sv.y, sv.z, sv.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w}, fw[4] = {fv.x, fv.y, fv.z, fv.w}, cw[4] = {cv.x, cv.y, cv.z, cv.w};
        uint32_t pw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float qo[4], o0[4], o1[4], o2[4];
            uint32_t pword = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t s = (sw[w] >> (8 * e)) & 255u;
                const bool isq = s == (uint32_t)(qi + 1);
                const bool amodal = ((qw[w] >> (8 * e)) & 255u) == 1u;
                const bool fr = ((fw[w] >> (8 * e)) & 255u) == 1u, co = ((cw[w] >> (8 * e)) & 255u) == 1u;
                qo[e] = (t == a.qt && isq) ? 1.f : 0.f;
                o0[e] = amodal ? 1.f : 0.f; o1[e] = fr ? 1.f : 0.f; o2[e] = co ? 1.f : 0.f;
                if (amodal && !isq) pword |= s << (8 * e);
                n_amodal += amodal; any_q |= (t == a.qt && isq); any_t |= amodal | fr | co; any_f |= fr; any_c |= co;
            }
            pw[w] = pword;
            qm[i * 4 + w] = make_float4(qo[0], qo[1], qo[2], qo[3]);
            t0[i * 4 + w] = make_float4(o0[0], o0[1], o0[2], o0[3]);
            t1[i * 4 + w] = make_float4(o1[0], o1[1], o1[2], o1[3]);
            t2[i * 4 + w] = make_float4(o2[0], o2[1], o2[2], o2[3]);
        }
        pt[i] = make_uint4(pw[0], pw[1], pw[2], pw[3]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        n_amodal += __shfl_xor(n_amodal, o, 64); any_q |= __shfl_xor(any_q, o, 64); any_t |= __shfl_xor(any_t, o, 64);
        any_f |= __shfl_xor(any_f, o, 64); any_c |= __shfl_xor(any_c, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_amodal) atomicAdd(&a.counts[0], n_amodal);
        if (any_q) atomicOr(&a.counts[1 + 2 * q], 1);
        if (any_t) atomicOr(&a.counts[2 + 2 * q], 1);
        // loss.py:285-292, 302-308: the frame weight of the occluder / container channel where the frame has a mask (every writer stores the same value)
        if (any_f && a.fw_occl) a.fw_occl[f] = a.has_weight;
        if (any_c && a.fw_cont) a.fw_cont[f] = a.has_weight;
    }
}

// The per-frame decisions of data_utils.py:455-492 and the frame weights of loss.py:55-83 for all (b, q, t): which instance is the frontmost
// occluder / outermost container of the queried instance in each frame, the ids and flags that go with it, the queried instances' occlusion
// fractions, the snitch frame weights -- (B, Q, T)-sized tables that were ~40 tensor operations per step.  One thread per (b, q, t).
struct QueryTablesArgs {
    const float* occl_fracs; const float* dag; const long long* sel;     // (B,K,T,3), (B,T,M,M,3), (B,Q)
    int B, Q, K, T, M, qt;
    float front_thres, front_half_thres, outer_thres, occluded_weight, zero_weight;
    int* query_idx; int* front; int* cont;                                // [B*Q], [B*Q*T] x 2
    uint8_t* ids; float* flags; float* sel_occl; float* frame_w;          // (B,Q,T,2) u8, (B,Q,T,3), (B,Q,T,3), [3][B*Q*T] (snitch | occluder | container)
    int* counts; int n_counts;
};
__global__ void __launch_bounds__(256) query_tables_kernel(QueryTablesArgs a) {
    const int n = a.B * a.Q * a.T;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < a.n_counts) a.counts[f] = 0;                                  // (n_counts <= n: 1 + 2 Q <= B Q T checked by the caller)
    if (f >= n) return;
    const int t = f % a.T, bq = f / a.T, b = bq / a.Q;
    int s = (int)a.sel[bq];
    const int smax = (a.K < a.M ? a.K : a.M) - 1;                         // (callers validate 0 <= sel < min(K, M) on the host -- pipeline.forward_kubric; a bad index
    s = s < 0 ? 0 : (s > smax ? smax : s);                                //  handed to the C ABI directly is clamped rather than read through)
    if (t == 0) a.query_idx[bq] = s;
    const float* of = a.occl_fracs + (((size_t)b * a.K + s) * a.T + t) * 3;
    const float* frame = a.dag + ((size_t)b * a.T + t) * a.M * a.M * 3;
    const float* row = frame + (size_t)s * a.M * 3;                       // the queried instance's row: [m][0] = contained by m, [m][2] = occluded by m
    const float of0 = of[0];
    float fmax = row[2]; int farg = 0;                                    // frontmost occluder: the first maximum (data_utils.py:455-458)
    for (int m = 1; m < a.M; ++m) { const float v = row[m * 3 + 2]; if (v > fmax) { fmax = v; farg = m; } }
    const bool has_front = of0 >= a.front_thres && fmax >= a.front_half_thres;
    // outermost container (:468-492): among the instances that contain the query, the one least contained itself; a single candidate
    // (or none) falls back to the strongest container
    int n_cand = 0, outer = 0, amax_i = 0;
    float best = __builtin_inff(), cmax = row[0];
    for (int m = 0; m < a.M; ++m) {
        const float c = row[m * 3];
        if (m > 0 && c > cmax) { cmax = c; amax_i = m; }
        if (c >= a.outer_thres) {
            ++n_cand;
            const float* rm = frame + (size_t)m * a.M * 3;
            float sc = rm[0];
            for (int j = 1; j < a.M; ++j) sc = fmaxf(sc, rm[j * 3]);
            if (sc < best) { best = sc; outer = m; }
        }
    }
    if (n_cand <= 1) outer = amax_i;
    const bool has_cont = n_cand > 0;
    a.front[f] = has_front ? farg : -1;
    a.cont[f] = has_cont ? outer : -1;
    a.ids[f * 2] = (uint8_t)(has_front ? farg + 1 : 0); a.ids[f * 2 + 1] = (uint8_t)(has_cont ? outer + 1 : 0);
    a.flags[f * 3] = has_front ? 1.f : 0.f; a.flags[f * 3 + 1] = has_cont ? 1.f : 0.f; a.flags[f * 3 + 2] = of0;
    a.sel_occl[f * 3] = of0; a.sel_occl[f * 3 + 1] = of[1]; a.sel_occl[f * 3 + 2] = of[2];
    float fw = fmaxf(of0 * a.occluded_weight, 1.0f);                      // loss.py:55-83
    if (b == a.B - 1 && t == a.qt) fw *= 0.2f;                            // (:79 indexes with the leaked loop variable b == B - 1: only the last clip)
    a.frame_w[f] = fw; a.frame_w[n + f] = a.zero_weight; a.frame_w[2 * n + f] = a.zero_weight;
}

// eval/metrics.py:55-113 from the per-frame area tables of tcow_iou_counts: the six masked IoU means and their frame counts
__global__ void __launch_bounds__(256) iou_means_kernel(const int* __restrict__ counts, int n_seq, int C, int T, float* __restrict__ mean, int* __restrict__ count) {
    __shared__ double ssum[6][4];
    __shared__ int scnt[6][4];
    double sum[6] = {0., 0., 0., 0., 0., 0.};
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n_seq * T; i += 256) {
        const int sq = i / T, t = i - sq * T;
        double iou[3] = {0., 0., 0.}; bool has[3] = {false, false, false};
        for (int c = 0; c < C && c < 3; ++c) {
            const int* k = counts + (((size_t)sq * C + c) * T + t) * 3;
            iou[c] = (double)k[1] / ((double)k[2] + 1e-7);                 // metrics.py:55-66
            has[c] = k[0] > 0;
        }
        const bool sel[6] = {has[0], has[1], has[2], has[0] && !has[1] && C >= 2, has[0] && has[1], has[0] && has[2]};
        const double val[6] = {iou[0], iou[1], iou[2], iou[0], iou[0], iou[0]};
#pragma unroll
        for (int k = 0; k < 6; ++k) if (sel[k]) { sum[k] += val[k]; ++cnt[k]; }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double s = sum[k]; int c = cnt[k];
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
        if ((threadIdx.x & 63) == 0) { ssum[k][threadIdx.x >> 6] = s; scnt[k][threadIdx.x >> 6] = c; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        const double s = (ssum[k][0] + ssum[k][1]) + (ssum[k][2] + ssum[k][3]);
        const int c = scnt[k][0] + scnt[k][1] + scnt[k][2] + scnt[k][3];
        mean[k] = c > 0 ? (float)(s / (double)c) : -1.0f;
        count[k] = c;
    }
}

// ------------------------------------------------------------------------------------------ snitch pixel weights (caller row L)
// loss.py:85-148 times the frame weights of loss.py:55-83: class balancing (powers 0.7 / -0.3 of the rarer / commoner class
// fraction, 5 % floor), x2 on occluded snitch pixels, x hard_negative_factor on the band = (k x k box dilation of the target) minus
// the target, k = odd(int(sqrt(HW) / 12)) -- the reference's gaussian_blur(target, k, sigma=k) > 0.  Separable dilation: rows here,
// columns inside the weight kernel.
__global__ void __launch_bounds__(256) dilate_rows_kernel(const float* __restrict__ target, long frame_stride_t, int frames_per_seq, long seq_stride_t, int H, int W, int r,
                                                          uint8_t* __restrict__ tmp) {
    const int f = blockIdx.y;
    const int s = f / frames_per_seq, ft = f - s * frames_per_seq;
    const float* src = target + (size_t)s * seq_stride_t + (size_t)ft * frame_stride_t;
    uint8_t* dst = tmp + (size_t)f * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        const int x0 = x - r > 0 ? x - r : 0, x1 = x + r < W - 1 ? x + r : W - 1;
        int any = 0;
        for (int xx = x0; xx <= x1; ++xx) any |= src[y * W + xx] > 0.f;
        dst[i] = (uint8_t)any;
    }
}
// The same row dilation as 64-bit masks (W a multiple of 64, r <= 63): one wave per row; ballots give the row's set pixels, every lane tests
// its window [x - r, x + r] against the three words around it, a second ballot is the dilated word.  4 B read + 1 bit written per pixel
// (the byte version walks 2r + 1 = 23 floats per pixel at 240 x 320: 68 us per step); the weight kernel ORs 2r + 1 of these words per 64
// pixels instead of reading 2r + 1 bytes per pixel.
__device__ __forceinline__ unsigned long long bit_range(int a, int b) {            // bits a..b (0 <= a <= b <= 63)
    return ((~0ull) >> (63 - (b - a))) << a;
}
__global__ void __launch_bounds__(256) dilate_rows_bits_kernel(const float* __restrict__ target, long frame_stride_t, int frames_per_seq, long seq_stride_t, int H, int W, int r,
                                                               unsigned long long* __restrict__ bits) {
    const int f = blockIdx.y;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (y >= H) return;
    const int s = f / frames_per_seq, ft = f - s * frames_per_seq;
    const float* src = target + (size_t)s * seq_stride_t + (size_t)ft * frame_stride_t + (size_t)y * W;
    const int nw = W >> 6;
    unsigned long long* dst = bits + ((size_t)f * H + y) * nw;
    unsigned long long prev = 0ull, cur = __ballot(src[lane] > 0.f);
    for (int c = 0; c < nw; ++c) {
        const unsigned long long next = c + 1 < nw ? __ballot(src[(c + 1) * 64 + lane] > 0.f) : 0ull;
        const int lo = lane - r, hi = lane + r;
        bool any = (cur & bit_range(lo > 0 ? lo : 0, hi < 63 ? hi : 63)) != 0ull;
        if (lo < 0) any = any || (prev & bit_range(64 + lo, 63)) != 0ull;
        if (hi > 63) any = any || (next & bit_range(0, hi - 64)) != 0ull;
        const unsigned long long d = __ballot(any);
        if (lane == 0) dst[c] = d;
        prev = cur; cur = next;
    }
}
struct WeightsArgs {
    const float* target; long frame_stride_t; int frames_per_seq; long seq_stride_t;   // channel 0 of (BQ,3,T,H,W)
    const uint8_t* ptr; const uint8_t* rowdil; const unsigned long long* rowbits; const float* frame_w; const int* pos_count;   // rowbits != NULL: bit-mask rows (W % 64 == 0)
    float* out; int H, W, r; long n_pixels; int class_balancing; float hard_negative_factor;
};
__global__ void __launch_bounds__(256) snitch_weights_kernel(WeightsArgs a) {
    __shared__ float corr[2];
    if (threadIdx.x == 0) {
        float pc = 1.f, nc = 1.f;
        if (a.class_balancing) {                                            // loss.py:100-119 (host float64 arithmetic there, double here)
            const long pos = *a.pos_count, neg = a.n_pixels - pos;
            float pf = (float)pos / (float)a.n_pixels, nf = (float)neg / (float)a.n_pixels;
            pf = pf < 0.05f ? 0.05f : pf; nf = nf < 0.05f ? 0.05f : nf;
            const double p = pf, n = nf;
            if (p > n) { pc = (float)pow(n / p, 0.7); nc = (float)pow(n / p, -0.3); }
            else { pc = (float)pow(p / n, -0.3); nc = (float)pow(p / n, 0.7); }
        }
        corr[0] = nc; corr[1] = pc;
    }
    __syncthreads();
    const int f = blockIdx.y;
    const int s = f / a.frames_per_seq, ft = f - s * a.frames_per_seq;
    const float* tg = a.target + (size_t)s * a.seq_stride_t + (size_t)ft * a.frame_stride_t;
    const size_t fo = (size_t)f * a.H * a.W;
    const float fw = a.frame_w[f];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.H * a.W; i += gridDim.x * 256) {
        const int y = i / a.W, x = i - y * a.W;
        const float t = tg[i];
        float w = 1.f;
        if (a.class_balancing) { if (t == 0.f) w *= corr[0]; if (t == 1.f) w *= corr[1]; }
        if (a.ptr[fo + i] != 0) w *= 2.f;
        if (a.hard_negative_factor > 1.f && a.rowbits) {
            // the wave's 64 pixels share one word per row (W % 64 == 0, i % 64 == lane): lane j fetches row y - r + j's word, an OR butterfly
            // over the wave gives the column-dilated word, every lane takes its bit
            const int y0 = y - a.r, nw = a.W >> 6, lane = threadIdx.x & 63;
            const int yy = y0 + lane;
            unsigned long long m = (lane <= 2 * a.r && yy >= 0 && yy < a.H) ? a.rowbits[((size_t)f * a.H + yy) * nw + (x >> 6)] : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
            if (!(t >= 0.5f) && ((m >> (x & 63)) & 1ull)) w *= a.hard_negative_factor;
        } else if (a.hard_negative_factor > 1.f && !(t >= 0.5f)) {
            const int y0 = y - a.r > 0 ? y - a.r : 0, y1 = y + a.r < a.H - 1 ? y + a.r : a.H - 1;
            int any = 0;
            for (int yy = y0; yy <= y1; ++yy) any |= a.rowdil[fo + (size_t)yy * a.W + x];
            if (any) w *= a.hard_negative_factor;
        }
        a.out[fo + i] = fw * w;
    }
}
}  // namespace

extern "C" int tcow_build_masks(void* stream, int B, int Q, int M, int T, long HW, int query_time, const uint8_t* segm, const uint8_t* div_segm,
                                const int* query_idx, const int* front_idx, const int* cont_idx, float* query_mask, float* target_mask,
                                uint8_t* snitch_occl_by_ptr, int* counts) {
    TCOW_CHECK_ARG(B > 0 && Q > 0 && M > 0 && T > 0 && HW > 0 && HW % 16 == 0, "tcow_build_masks: bad shape (H*W must be a multiple of 16)");
    TCOW_CHECK_ARG(segm && div_segm && query_idx && front_idx && cont_idx && query_mask && target_mask && snitch_occl_by_ptr && counts, "tcow_build_masks: null pointer");
    TCOW_CHECK_ARG((long)B * Q * T < 65536, "tcow_build_masks: too many frames for one call");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (1 + 2 * Q), st);
    if (e != hipSuccess) { tcow_set_error("tcow_build_masks: memset failed: %s", hipGetErrorString(e)); return TCOW_ERR_LAUNCH; }
    BuildMasksArgs a;
    a.segm = segm; a.div = div_segm; a.qidx = query_idx; a.front = front_idx; a.cont = cont_idx; a.qmask = query_mask; a.target = target_mask;
    a.ptr = snitch_occl_by_ptr; a.counts = counts; a.B = B; a.Q = Q; a.M = M; a.T = T; a.HW16 = (int)(HW / 16); a.qt = query_time;
    a.fw_occl = nullptr; a.fw_cont = nullptr; a.has_weight = 1.0f;
    const int gx = cdiv(a.HW16, 256) < 8 ? cdiv(a.HW16, 256) : 8;
    hipLaunchKernelGGL(build_masks_kernel, dim3(gx, B * Q * T), dim3(256), 0, st, a);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

extern "C" int tcow_build_query_masks(void* stream, int B, int Q, int K, int M, int T, long HW, int query_time, const uint8_t* segm, const uint8_t* div_segm,
                                      const float* occl_fracs, const float* dag, const long long* sel, float front_thres, float front_half_thres,
                                      float outer_thres, float occluded_weight, float zero_weight, float has_weight, int* idx_ws, uint8_t* ids, float* flags,
                                      float* sel_occl_fracs, float* frame_w, float* query_mask, float* target_mask, uint8_t* snitch_occl_by_ptr, int* counts) {
    TCOW_CHECK_ARG(B > 0 && Q > 0 && K > 0 && M > 0 && T > 0 && HW > 0 && HW % 16 == 0, "tcow_build_query_masks: bad shape (H*W must be a multiple of 16)");
    TCOW_CHECK_ARG(segm && div_segm && occl_fracs && dag && sel && idx_ws && ids && flags && sel_occl_fracs && frame_w && query_mask && target_mask &&
                   snitch_occl_by_ptr && counts, "tcow_build_query_masks: null pointer");
    TCOW_CHECK_ARG((long)B * Q * T < 65536 && 1 + 2 * Q <= B * Q * T && M <= 255, "tcow_build_query_masks: B*Q*T must be in [1 + 2Q, 65536), M <= 255");
    TCOW_CHECK_ARG(query_time >= 0 && query_time < T, "tcow_build_query_masks: query_time %d outside [0, %d)", query_time, T);
    hipStream_t st = (hipStream_t)stream;
    const int n = B * Q * T;
    QueryTablesArgs qa;
    qa.occl_fracs = occl_fracs; qa.dag = dag; qa.sel = sel; qa.B = B; qa.Q = Q; qa.K = K; qa.T = T; qa.M = M; qa.qt = query_time;
    qa.front_thres = front_thres; qa.front_half_thres = front_half_thres; qa.outer_thres = outer_thres; qa.occluded_weight = occluded_weight; qa.zero_weight = zero_weight;
    qa.query_idx = idx_ws; qa.front = idx_ws + B * Q; qa.cont = idx_ws + B * Q + n;
    qa.ids = ids; qa.flags = flags; qa.sel_occl = sel_occl_fracs; qa.frame_w = frame_w; qa.counts = counts; qa.n_counts = 1 + 2 * Q;
    hipLaunchKernelGGL(query_tables_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, qa);
    TCOW_CHECK_LAUNCH();
    BuildMasksArgs a;
    a.segm = segm; a.div = div_segm; a.qidx = qa.query_idx; a.front = qa.front; a.cont = qa.cont; a.qmask = query_mask; a.target = target_mask;
    a.ptr = snitch_occl_by_ptr; a.counts = counts; a.B = B; a.Q = Q; a.M = M; a.T = T; a.HW16 = (int)(HW / 16); a.qt = query_time;
    a.fw_occl = frame_w + n; a.fw_cont = frame_w + 2 * n; a.has_weight = has_weight;
    const int gx = cdiv(a.HW16, 256) < 8 ? cdiv(a.HW16, 256) : 8;
    hipLaunchKernelGGL(build_masks_kernel, dim3(gx, n), dim3(256), 0, st, a);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

extern "C" int tcow_iou_means(void* stream, const int* counts, int n_seq, int C, int T, float* mean, int* count) {
    TCOW_CHECK_ARG(counts && mean && count && n_seq > 0 && C >= 1 && T > 0, "tcow_iou_means: bad arguments");
    hipLaunchKernelGGL(iou_means_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, n_seq, C, T, mean, count);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

extern "C" size_t tcow_snitch_weights_workspace_bytes(long n_frames, int H, int W) { return (size_t)n_frames * H * W + 256; }

extern "C" int tcow_snitch_weights(void* stream, long n_seq, int T, int H, int W, const float* target_ch0, long target_seq_stride, const uint8_t* snitch_occl_by_ptr,
                                   const float* frame_w, const int* pos_count, int class_balancing, float hard_negative_factor, float* weights, void* ws,
                                   size_t ws_bytes) {
    TCOW_CHECK_ARG(n_seq > 0 && T > 0 && H > 0 && W > 0 && target_ch0 && snitch_occl_by_ptr && frame_w && weights, "tcow_snitch_weights: bad arguments");
    TCOW_CHECK_ARG(!class_balancing || pos_count, "tcow_snitch_weights: class balancing needs the positive-pixel count");
    const long frames = n_seq * T;
    TCOW_CHECK_ARG(frames < 65536, "tcow_snitch_weights: too many frames for one call");
    hipStream_t st = (hipStream_t)stream;
    int k = (int)(sqrt((double)H * (double)W) / 12.0);                       // loss.py:138-140
    if (k % 2 == 0) k += 1;
    const int r = k / 2;
    uint8_t* tmp = (uint8_t*)ws;
    const int gx = cdiv((long)H * W, 256 * 8) < 1 ? 1 : cdiv((long)H * W, 256 * 8);
    const bool band = hard_negative_factor > 1.f;
    const bool bitrows = band && W % 64 == 0 && r <= 31 && ((long)H * W) % 256 == 0;     // (2r + 1 <= 64 rows in one wave; whole waves inside the frame)
    if (band) {
        TCOW_CHECK_ARG(ws && ws_bytes >= tcow_snitch_weights_workspace_bytes(frames, H, W), "tcow_snitch_weights: workspace too small");
        if (bitrows) hipLaunchKernelGGL(dilate_rows_bits_kernel, dim3(cdiv(H, 4), (unsigned)frames), dim3(256), 0, st, target_ch0, (long)H * W, T, target_seq_stride, H, W, r,
                                        (unsigned long long*)ws);
        else hipLaunchKernelGGL(dilate_rows_kernel, dim3(gx, (unsigned)frames), dim3(256), 0, st, target_ch0, (long)H * W, T, target_seq_stride, H, W, r, tmp);
        TCOW_CHECK_LAUNCH();
    }
    WeightsArgs a;
    a.target = target_ch0; a.frame_stride_t = (long)H * W; a.frames_per_seq = T; a.seq_stride_t = target_seq_stride; a.ptr = snitch_occl_by_ptr; a.rowdil = tmp; a.rowbits = bitrows ? (const unsigned long long*)ws : nullptr;
    a.frame_w = frame_w; a.pos_count = pos_count; a.out = weights; a.H = H; a.W = W; a.r = r; a.n_pixels = frames * H * W; a.class_balancing = class_balancing;
    a.hard_negative_factor = band ? hard_negative_factor : 1.f;
    hipLaunchKernelGGL(snitch_weights_kernel, dim3(gx, (unsigned)frames), dim3(256), 0, st, a);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
