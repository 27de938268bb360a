// Photometric augmentation of the rgb modality on the device: torchvision ColorJitter (brightness / contrast / saturation / hue in the drawn
// order) + GaussianBlur(5) + Grayscale(3) of the reference's input pipeline (data/augs.py:33-35,175-181), fused:
//   pass 1 (only when contrast is drawn)  per-frame mean of the grayscale image AS IT IS when the contrast adjustment runs, i.e. after the
//                                         adjustments drawn in front of it -- elementwise, so recomputed per pixel -- as fixed-order partial sums;
//   pass 2                                one 32 x 32 output tile per workgroup: the 36 x 36 halo (reflect-padded at the frame border, like
//                                         torchvision's blur) is read once from the SOURCE clip -- frame selection and centre crop are index
//                                         arithmetic on the way in --, jittered per pixel into LDS, then the separable 5-tap blur (rows, then
//                                         columns) and the grayscale fold run out of LDS and the tile is written in the (3, T, h, w) layout
//                                         the crop / flip / resize kernels read.
// The reference runs these as 6-10 full-tensor passes of torchvision ops on the CPU loader workers; round 3 ran them as ~40 torch elementwise
// launches on the device.  Arithmetic follows torchvision's tensor definitions operation by operation in f32 (functional_tensor: _blend with
// clamp, rgb_to_grayscale 0.2989 / 0.587 / 0.114, _rgb2hsv / _hsv2rgb, the Gaussian taps exp(-0.5 (x / sigma)^2) normalised on the host in f32).
#include "common.h"

namespace {

struct PhotoParams {
    const float* src; long s_c, s_t, s_y;         // element strides of the (3, Tv, H, W) source: channel, frame, row
    const int* frame_idx; int T;                  // source frame of each output frame
    int y0, x0, h, w;                             // centre-crop rectangle inside a source frame = the output frame size
    int n_ops, ops[4];                            // jitter adjustments in application order: 0 brightness, 1 contrast, 2 saturation, 3 hue
    float fb, fc, fs, fh;
    int blur; float taps[5];
    int gray;
    float* mean_part; int nblk;                   // [T][nblk] partial sums of pass 1
    float* out;                                   // (3, T, h, w)
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
__device__ __forceinline__ float blend(float a, float b, float ratio) { return clamp01(ratio * a + (1.0f - ratio) * b); }

// torchvision adjust_hue on one pixel: _rgb2hsv, h <- (h + shift) mod 1, _hsv2rgb
__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float shift) {
    const float maxc = fmaxf(fmaxf(r, g), b), minc = fminf(fminf(r, g), b);
    const bool eq = maxc == minc;
    const float cr = maxc - minc;
    const float sat = cr / (eq ? 1.0f : maxc);
    const float div = eq ? 1.0f : cr;
    const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
    const bool is_r = maxc == r, is_g = maxc == g;
    float hh = is_r ? (bc - gc) : (is_g ? (2.0f + rc - bc) : (4.0f + gc - rc));
    hh = fmodf(hh / 6.0f + 1.0f, 1.0f);
    hh = hh + shift;
    hh = hh - floorf(hh);                                              // python's % 1.0 on a value in (-1, 2)
    const float i = floorf(hh * 6.0f);
    const float f = hh * 6.0f - i;
    const int k = ((int)i) % 6;
    const float v = maxc;
    const float p = clamp01(v * (1.0f - sat)), q = clamp01(v * (1.0f - sat * f)), t = clamp01(v * (1.0f - sat * (1.0f - f)));
    r = k == 0 ? v : k == 1 ? q : k == 2 ? p : k == 3 ? p : k == 4 ? t : v;
    g = k == 0 ? t : k == 1 ? v : k == 2 ? v : k == 3 ? q : k == 4 ? p : p;
    b = k == 0 ? p : k == 1 ? p : k == 2 ? t : k == 3 ? v : k == 4 ? v : q;
}

// the first `upto` adjustments of the drawn chain on one pixel (`mean` = the frame's grayscale mean for the contrast adjustment)
__device__ __forceinline__ void jitter(const PhotoParams& p, int upto, float mean, float& r, float& g, float& b) {
    for (int i = 0; i < upto; ++i) {
        const int op = p.ops[i];
        if (op == 0) { r = blend(r, 0.0f, p.fb); g = blend(g, 0.0f, p.fb); b = blend(b, 0.0f, p.fb); }
        else if (op == 1) { r = blend(r, mean, p.fc); g = blend(g, mean, p.fc); b = blend(b, mean, p.fc); }
        else if (op == 2) { const float y = gray_of(r, g, b); r = blend(r, y, p.fs); g = blend(g, y, p.fs); b = blend(b, y, p.fs); }
        else hue_shift(r, g, b, p.fh);
    }
}

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// pass 1: mean_part[t][blk] = sum over the pixels blk, blk + nblk, ... (in chunks of 256) of gray(jitter prefix)
__global__ __launch_bounds__(256) void photo_mean_kernel(PhotoParams p, int upto) {
    __shared__ float red[256];
    const int t = blockIdx.y, tid = threadIdx.x;
    const float* f = p.src + (long)p.frame_idx[t] * p.s_t + (long)p.y0 * p.s_y + p.x0;
    const long npix = (long)p.h * p.w;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + tid; i < npix; i += (long)p.nblk * 256) {
        const int y = (int)(i / p.w), x = (int)(i - (long)y * p.w);
        float r = f[(long)y * p.s_y + x], g = f[p.s_c + (long)y * p.s_y + x], b = f[2 * p.s_c + (long)y * p.s_y + x];
        jitter(p, upto, 0.f, r, g, b);
        acc += gray_of(r, g, b);
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) p.mean_part[(long)t * p.nblk + blockIdx.x] = red[0];
}

constexpr int PT = 32, PH = PT + 4, PLD = PH + 1;

__global__ __launch_bounds__(256) void photo_apply_kernel(PhotoParams p, int has_contrast) {
    __shared__ float tile[3][PH][PLD];       // jittered halo tile
    __shared__ float hor[3][PH][PT + 1];     // after the horizontal taps
    __shared__ float s_mean;
    const int t = blockIdx.z, tid = threadIdx.x;
    const int ty0 = blockIdx.y * PT, tx0 = blockIdx.x * PT;
    if (tid == 0) {
        float m = 0.f;
        if (has_contrast) { for (int i = 0; i < p.nblk; ++i) m += p.mean_part[(long)t * p.nblk + i]; m /= (float)((long)p.h * p.w); }
        s_mean = m;
    }
    __syncthreads();
    const float mean = s_mean;
    const float* f = p.src + (long)p.frame_idx[t] * p.s_t + (long)p.y0 * p.s_y + p.x0;
    for (int i = tid; i < PH * PH; i += 256) {
        const int ly = i / PH, lx = i - ly * PH;
        const int y = reflect(ty0 + ly - 2, p.h), x = reflect(tx0 + lx - 2, p.w);
        float r = 0.f, g = 0.f, b = 0.f;
        if (y >= 0 && y < p.h && x >= 0 && x < p.w) {                       // (a reflected index of a tile far outside a tiny frame can still miss)
            r = f[(long)y * p.s_y + x]; g = f[p.s_c + (long)y * p.s_y + x]; b = f[2 * p.s_c + (long)y * p.s_y + x];
            jitter(p, p.n_ops, mean, r, g, b);
        }
        tile[0][ly][lx] = r; tile[1][ly][lx] = g; tile[2][ly][lx] = b;
    }
    __syncthreads();
    if (p.blur) {
        for (int i = tid; i < 3 * PH * PT; i += 256) {
            const int c = i / (PH * PT), rem = i - c * (PH * PT), ly = rem / PT, lx = rem - ly * PT;
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) a += p.taps[j] * tile[c][ly][lx + j];
            hor[c][ly][lx] = a;
        }
        __syncthreads();
    }
    const long plane = (long)p.h * p.w;
    for (int i = tid; i < PT * PT; i += 256) {
        const int ly = i / PT, lx = i - ly * PT;
        const int y = ty0 + ly, x = tx0 + lx;
        if (y >= p.h || x >= p.w) continue;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (p.blur) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) a += p.taps[j] * hor[c][ly + j][lx];
                v[c] = a;
            } else v[c] = tile[c][ly + 2][lx + 2];
        }
        if (p.gray) { const float y3 = gray_of(v[0], v[1], v[2]); v[0] = v[1] = v[2] = y3; }
        float* o = p.out + (long)t * plane + (long)y * p.w + x;
        o[0] = v[0]; o[(long)p.T * plane] = v[1]; o[2L * p.T * plane] = v[2];
    }
}

}  // namespace

extern "C" {

long tcow_photometric_workspace_bytes(int Tc) { return (long)(Tc > 0 ? Tc : 0) * 64 * 4; }

int tcow_photometric(void* stream, int Tv, int H, int W, int Tc, int y0, int x0, int h, int w, const float* src, const int* frame_idx, int n_ops,
                     const int* ops, float brightness, float contrast, float saturation, float hue, int blur, const float* taps, int gray,
                     float* workspace, long workspace_bytes, float* out) {
    TCOW_CHECK_ARG(Tv > 0 && H > 0 && W > 0 && Tc > 0 && h > 0 && w > 0, "tcow_photometric: bad shape Tv=%d H=%d W=%d Tc=%d h=%d w=%d", Tv, H, W, Tc, h, w);
    TCOW_CHECK_ARG(y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 + w <= W, "tcow_photometric: crop rectangle (%d, %d, %d, %d) outside the %d x %d frame", y0, x0, h, w, H, W);
    TCOW_CHECK_ARG(src && frame_idx && out, "tcow_photometric: null pointer");
    TCOW_CHECK_ARG(n_ops >= 0 && n_ops <= 4 && (n_ops == 0 || ops), "tcow_photometric: 0 .. 4 jitter adjustments expected (got %d)", n_ops);
    TCOW_CHECK_ARG(!blur || (taps && h >= 3 && w >= 3), "tcow_photometric: the blur needs five taps and frames of at least 3 x 3 (reflect padding)");
    PhotoParams p;
    p.src = src; p.s_y = W; p.s_t = (long)H * W; p.s_c = (long)Tv * H * W;
    p.frame_idx = frame_idx; p.T = Tc; p.y0 = y0; p.x0 = x0; p.h = h; p.w = w;
    p.n_ops = n_ops; p.fb = brightness; p.fc = contrast; p.fs = saturation; p.fh = hue;
    int contrast_at = -1;
    for (int i = 0; i < 4; ++i) {
        p.ops[i] = i < n_ops ? ops[i] : 0;
        TCOW_CHECK_ARG(i >= n_ops || (ops[i] >= 0 && ops[i] <= 3), "tcow_photometric: adjustment %d is not 0..3", i < n_ops ? ops[i] : 0);
        if (i < n_ops && ops[i] == 1 && contrast_at < 0) contrast_at = i;
    }
    p.blur = blur ? 1 : 0; p.gray = gray ? 1 : 0;
    for (int j = 0; j < 5; ++j) p.taps[j] = blur ? taps[j] : 0.f;
    p.nblk = 64; p.mean_part = workspace; p.out = out;
    hipStream_t st = (hipStream_t)stream;
    if (contrast_at >= 0) {
        TCOW_CHECK_ARG(workspace && workspace_bytes >= tcow_photometric_workspace_bytes(Tc), "tcow_photometric: workspace too small");
        hipLaunchKernelGGL(photo_mean_kernel, dim3(p.nblk, Tc), dim3(256), 0, st, p, contrast_at);
        TCOW_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(photo_apply_kernel, dim3(cdiv(w, PT), cdiv(h, PT), Tc), dim3(256), 0, st, p, contrast_at >= 0 ? 1 : 0);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

}  // extern "C"
