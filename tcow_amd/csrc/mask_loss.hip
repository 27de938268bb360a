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
// All passes stream x, t, w once (12 B/pixel; the gradient pass also writes 4 B/pixel): HBM-bound.
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
};

// bce-with-logits and sigmoid of one pixel; identical instruction sequence in every pass (the radix passes rely on it)
__device__ __forceinline__ void bce_sig(float x, float t, float& bce, float& p) {
    const float e = expf(-fabsf(x));
    const float inv = 1.0f / (1.0f + e);
    p = x >= 0.0f ? inv : e * inv;
    bce = fmaxf(x, 0.0f) - x * t + log1pf(e);
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
__global__ void __launch_bounds__(kThreads) ml_frames_kernel(MlView v, MlWs ws) {
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
__global__ void __launch_bounds__(kThreads) ml_ctl_kernel(MlView v, MlWs ws, double topk_frac) {
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
__global__ void __launch_bounds__(kThreads) ml_stats_kernel(MlView v, MlWs ws) {
    __shared__ unsigned hist[kBins12];
    __shared__ double red[kThreads / 64];
    const int f = blockIdx.y;
    const MlCtl* c = ws.ctl;
    const bool live = c->valid && ws.fsel[f];
    const bool need_hist = live && !c->mode_all;
    if (need_hist) { for (int i = threadIdx.x; i < kBins12; i += kThreads) hist[i] = 0u; }
    __syncthreads();
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        for_each_vec(v, f, [&](float4 x, float4 t, float4 w, int) {
            const float xs[4] = {x.x, x.y, x.z, x.w}, ts[4] = {t.x, t.y, t.z, t.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bce, p; bce_sig(xs[e], ts[e], bce, p);
                const unsigned vb = vbits(bce, wv[e], v.weighted);
                s[0] += bce * wv[e]; s[1] += __builtin_bit_cast(float, vb);
                s[2] += p * ts[e]; s[3] += p; s[4] += ts[e];
                if (need_hist) atomicAdd(&hist[vb >> 20], 1u);
            }
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
template <int LEVEL> __global__ void __launch_bounds__(kThreads) ml_hist_kernel(MlView v, MlWs ws) {
    __shared__ unsigned hist[kBins12];
    const int f = blockIdx.y;
    const MlCtl* c = ws.ctl;
    if (!(c->valid && ws.fsel[f]) || c->mode_all) return;
    constexpr int NB = LEVEL == 2 ? kBins12 : kBins3;
    for (int i = threadIdx.x; i < NB; i += kThreads) hist[i] = 0u;
    __syncthreads();
    const unsigned prefix = LEVEL == 2 ? c->b1 : ((c->b1 << 11) | c->b2);
    for_each_vec(v, f, [&](float4 x, float4 t, float4 w, int) {
        const float xs[4] = {x.x, x.y, x.z, x.w}, ts[4] = {t.x, t.y, t.z, t.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float bce, p; bce_sig(xs[e], ts[e], bce, p);
            const unsigned vb = vbits(bce, wv[e], v.weighted);
            if (LEVEL == 2) { if ((vb >> 20) == prefix) atomicAdd(&hist[(vb >> 9) & 2047u], 1u); }
            else            { if ((vb >> 9) == prefix) atomicAdd(&hist[vb & 511u], 1u); }
        }
    });
    __syncthreads();
    unsigned* gh = ws.hist + (LEVEL == 2 ? kBins12 : 2 * kBins12);
    for (int i = threadIdx.x; i < NB; i += kThreads)
        if (hist[i]) atomicAdd(&gh[i], hist[i]);
}

// ---- one wave walks a histogram from the top for the bin holding the (k - above)-th largest value;
//      level 1 also folds the pass-A partial sums (fixed order: deterministic)
template <int LEVEL> __global__ void __launch_bounds__(kThreads) ml_scan_kernel(MlView v, MlWs ws) {
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
__global__ void __launch_bounds__(kThreads) ml_grad_kernel(MlView v, MlWs ws, float aot, float lw) {
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
                float bce, p; bce_sig(xs[e], ts[e], bce, p);
                const unsigned vb = vbits(bce, wv[e], v.weighted);
                const float sel = mode_all ? 1.f : (vb > tau ? 1.f : (vb == tau ? tie : 0.f));
                tk += sel * __builtin_bit_cast(float, vb);
                const float d = p - ts[e];
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
__global__ void __launch_bounds__(kThreads) ml_final_kernel(MlView v, MlWs ws, float aot, float loss_weight, float* loss, float* total) {
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

static int ml_nblk(long n_frames, long frame_len) { return (int)(n_frames * ((frame_len / 4 + kVecPerBlock - 1) / kVecPerBlock)); }

}  // namespace

extern "C" {

size_t tcow_mask_loss_workspace_bytes(long n_frames, long frame_len) {
    if (n_frames <= 0 || frame_len <= 0) return 0;
    size_t b = 0;
    b += ((size_t)n_frames * sizeof(int) + 255) & ~(size_t)255;
    b += ((size_t)n_frames * sizeof(double) + 255) & ~(size_t)255;
    b += ((size_t)ml_nblk(n_frames, frame_len) * kNPart * sizeof(double) + 255) & ~(size_t)255;
    b += (2 * kBins12 + kBins3) * sizeof(unsigned);
    b += 512;
    return b;
}

int tcow_mask_loss(void* stream, const tcow_mask_loss_args* a) {
    TCOW_CHECK_ARG(a && a->logits && a->target && a->loss && a->ws, "tcow_mask_loss: null argument");
    TCOW_CHECK_ARG(a->n_frames > 0 && a->frame_len > 0 && a->frames_per_seq > 0 && a->n_frames % a->frames_per_seq == 0,
                   "tcow_mask_loss: bad frame counts");
    TCOW_CHECK_ARG(a->frame_len % 4 == 0, "tcow_mask_loss: frame_len %ld must be a multiple of 4", a->frame_len);
    TCOW_CHECK_ARG(a->logits_seq_stride % 4 == 0 && a->target_seq_stride % 4 == 0 && a->dlogits_seq_stride % 4 == 0,
                   "tcow_mask_loss: sequence strides must be multiples of 4 elements");
    TCOW_CHECK_ARG(a->n_frames * (a->frame_len / 4) < (1L << 31) && a->n_frames < 65536, "tcow_mask_loss: too many pixels for one call");
    TCOW_CHECK_ARG(a->ws_bytes >= tcow_mask_loss_workspace_bytes(a->n_frames, a->frame_len), "tcow_mask_loss: workspace too small");
    TCOW_CHECK_ARG(a->topk_frac > 0.0 && a->topk_frac <= 1.0, "tcow_mask_loss: topk_frac %g outside (0, 1]", a->topk_frac);
    hipStream_t st = (hipStream_t)stream;
    MlView v;
    v.x = a->logits; v.xs = a->logits_seq_stride; v.t = a->target; v.ts = a->target_seq_stride;
    v.pw = a->pixel_w; v.fw = a->frame_w; v.dx = a->dlogits; v.dxs = a->dlogits_seq_stride;
    v.T = (int)a->frames_per_seq; v.L4 = (int)(a->frame_len / 4); v.n_frames = (int)a->n_frames; v.weighted = a->weighted_aot ? 1 : 0;
    MlWs ws; char* p = (char*)a->ws;
    ws.fsel = (int*)p; p += ((size_t)v.n_frames * sizeof(int) + 255) & ~(size_t)255;
    ws.fwsum = (double*)p; p += ((size_t)v.n_frames * sizeof(double) + 255) & ~(size_t)255;
    ws.nblk = ml_nblk(a->n_frames, a->frame_len);
    ws.part = (double*)p; p += ((size_t)ws.nblk * kNPart * sizeof(double) + 255) & ~(size_t)255;
    ws.hist = (unsigned*)p; p += (2 * kBins12 + kBins3) * sizeof(unsigned);
    ws.ctl = (MlCtl*)p;
    static_assert(sizeof(MlCtl) <= 512, "control block");
    const dim3 grid((v.L4 + kVecPerBlock - 1) / kVecPerBlock, v.n_frames);
    ml_frames_kernel<<<v.n_frames, kThreads, 0, st>>>(v, ws);
    ml_ctl_kernel<<<1, kThreads, 0, st>>>(v, ws, a->topk_frac);
    ml_stats_kernel<<<grid, kThreads, 0, st>>>(v, ws);
    ml_scan_kernel<1><<<1, kThreads, 0, st>>>(v, ws);
    if (a->topk_frac < 1.0 && a->aot_loss > 0.f) {
        ml_hist_kernel<2><<<grid, kThreads, 0, st>>>(v, ws);
        ml_scan_kernel<2><<<1, kThreads, 0, st>>>(v, ws);
        ml_hist_kernel<3><<<grid, kThreads, 0, st>>>(v, ws);
        ml_scan_kernel<3><<<1, kThreads, 0, st>>>(v, ws);
    }
    ml_grad_kernel<<<grid, kThreads, 0, st>>>(v, ws, a->aot_loss, a->loss_weight);
    ml_final_kernel<<<1, kThreads, 0, st>>>(v, ws, a->aot_loss, a->loss_weight, a->loss, a->total);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

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
