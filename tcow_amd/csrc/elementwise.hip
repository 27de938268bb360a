// Bandwidth-bound glue kernels of the Seeker path: patch gather, embeddings, cls handling, mask head re-layout,
// coarsening (avg-pool + bilinear), flags, weight/gradient casts.  All are simple grid-stride kernels with
// 8-16 byte vector accesses along the contiguous dimension.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------ im2col
// rows (b,t,s): s = 0 -> zeros, s = 1+n -> flattened patch n of frame (b,t) of cat([rgb, query]) in the order
// c*P*P + py*P + px (Conv2d weight flattening, vit.py:233-240; cat at mask_tracker.py:107-108).  Optional
// (x-0.45)/0.225 on the rgb channels only (vision_tf.py:81-89).
template <typename T>
__global__ void im2col_kernel(int B, int T_, int H, int W, int P, int Crgb, int Cq, const float* __restrict__ rgb, const float* __restrict__ qm, int norm,
                              T* __restrict__ out) {
    const int Hp = H / P, Wp = W / P, N = Hp * Wp, S = N + 1, C = Crgb + Cq;
    const int Kc = C * P * P, q4 = Kc / 4;
    const long total = (long)B * T_ * S * q4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / q4; const int k = (int)(i - row * q4) * 4;
        const int s = (int)(row % S); const long bt = row / S; const int t = (int)(bt % T_); const int b = (int)(bt / T_);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s > 0) {
            const int n = s - 1, hp = n / Wp, wp = n - hp * Wp;
            const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
            const int y = hp * P + py, x = wp * P + px;
            if (c < Crgb) {
                v = ld4(rgb + ((((size_t)b * Crgb + c) * T_ + t) * H + y) * W + x);
                if (norm) { v.x = (v.x - 0.45f) / 0.225f; v.y = (v.y - 0.45f) / 0.225f; v.z = (v.z - 0.45f) / 0.225f; v.w = (v.w - 0.45f) / 0.225f; }
            } else {
                v = ld4(qm + ((((size_t)b * Cq + (c - Crgb)) * T_ + t) * H + y) * W + x);
            }
        }
        st4(out + row * Kc + k, v);
    }
}

// ------------------------------------------------------------------------------------------ clip gather (input pipeline)
// out[c, t, y, x] = src[c, frame_idx[t], src_y[y], src_x[x]]: temporal sub-sampling, centre / random crop, flip and nearest resize of
// data/augs.py:150-203 composed into three index tables (tcow_amd/augs.py); byte or 4-byte elements.
template <typename T>
__global__ void gather_frames_kernel(int C, int Tv, int H, int W, int Tc, int h, int w, const T* __restrict__ src, const int* __restrict__ frame_idx,
                                     const int* __restrict__ src_y, const int* __restrict__ src_x, T* __restrict__ out) {
    const long total = (long)C * Tc * h * w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w); long r = i / w; const int y = (int)(r % h); r /= h; const int t = (int)(r % Tc); const int c = (int)(r / Tc);
        out[i] = src[(((size_t)c * Tv + frame_idx[t]) * H + src_y[y]) * W + src_x[x]];
    }
}

// Antialiased bilinear resize of a clip with the crop / flip / frame selection folded into the source addressing (data/augs.py:150-201 for
// the float modalities): out[c,t,Y,X] = sum_j wy[Y][j] * (sum_i wx[X][i] * src[c, frame_idx[t], ys[ymin[Y] + j], xs[xmin[X] + i]]) -- the
// separable triangle filter of F.interpolate(mode='bilinear', antialias=True, align_corners=False), horizontal pass inside, vertical
// outside (ATen's order), tables from tcow_amd/augs.py::aa_tables.  One thread per output pixel: 2 x 2 taps when enlarging, ~(2 s + 1)^2
// when shrinking by s; the source rows of a pixel's taps are shared by the neighbouring threads of its row (L1 / L2 hits).
__global__ void resize_aa_kernel(int C, int Tv, int H, int W, int Tc, int oh, int ow, const float* __restrict__ src, const int* __restrict__ frame_idx,
                                 const int* __restrict__ ys, const int* __restrict__ xs, const int* __restrict__ ymin, const int* __restrict__ ysize,
                                 const float* __restrict__ wy, int ky, const int* __restrict__ xmin, const int* __restrict__ xsize, const float* __restrict__ wx, int kx,
                                 float* __restrict__ out) {
    const long total = (long)C * Tc * oh * ow;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % ow); long r = i / ow; const int Y = (int)(r % oh); r /= oh; const int t = (int)(r % Tc); const int c = (int)(r / Tc);
        const float* plane = src + ((size_t)c * Tv + frame_idx[t]) * H * W;
        const int x0 = xmin[X], nx = xsize[X], y0 = ymin[Y], ny = ysize[Y];
        float acc = 0.f;
        for (int j = 0; j < ny; ++j) {
            const float* row = plane + (size_t)ys[y0 + j] * W;
            float h = 0.f;
            for (int k = 0; k < nx; ++k) h += wx[(size_t)X * kx + k] * row[xs[x0 + k]];
            acc += wy[(size_t)Y * ky + j] * h;
        }
        out[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------ embeddings
// x[b,t,s,:] = (s == 0) ? cls + pos[0] : x + pos[s] + time[t]      (vision_tf.py:99-138; f32 residual stream)
__global__ void embed_fwd_kernel(int B, int T_, int S, int D, float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                 const float* __restrict__ time) {
    const int d4 = D / 4;
    const long total = (long)B * T_ * S * d4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / d4; const int c = (int)(i - row * d4) * 4;
        const int s = (int)(row % S); const int t = (int)((row / S) % T_);
        const float4 p = ld4(pos + (size_t)s * D + c);
        float4 o;
        if (s == 0) { const float4 cv = ld4(cls + c); o = make_float4(cv.x + p.x, cv.y + p.y, cv.z + p.z, cv.w + p.w); }
        else {
            const float4 v = ld4(x + row * D + c), te = ld4(time + (size_t)t * D + c);
            o = make_float4(v.x + p.x + te.x, v.y + p.y + te.y, v.z + p.z + te.z, v.w + p.w + te.w);
        }
        st4(x + row * D + c, o);
    }
}
// dpos[s] = sum_{b,t} g[b,t,s];  one workgroup per (s, 64 channel quads = 1 KiB of the row): four thread rows split the B T frames (frame bt goes to row
// bt & 3), six loads in flight per thread, partial sums folded in LDS in a fixed order.  (One thread per (s, quad) walking all B T frames alone kept 58 000
// x 16 bytes in flight: 43 us for the 83 MB of configs[1].)
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(int B, int T_, int S, int D, const float* __restrict__ g, float* __restrict__ dpos, int accumulate) {
    __shared__ float4 red[256];
    const int d4 = D / 4, s = blockIdx.x, q = blockIdx.y * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6, BT = B * T_;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < d4) {
        const float* src = g + (size_t)s * D + q * 4;
        const size_t fs = (size_t)S * D;
        int bt = part;
        for (; bt + 20 < BT; bt += 24) {
            float4 v[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) v[u] = ld4(src + (size_t)(bt + 4 * u) * fs);
#pragma unroll
            for (int u = 0; u < 6; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; bt < BT; bt += 4) { const float4 v = ld4(src + (size_t)bt * fs); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (part == 0 && q < d4) {
#pragma unroll
        for (int p = 1; p < 4; ++p) { const float4 v = red[p * 64 + threadIdx.x]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        float* o = dpos + (size_t)s * D + q * 4;
        if (accumulate) { const float4 p = ld4(o); a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
        st4(o, a);
    }
}
// dtime[t] = sum_{b, s>=1} g[b,t,s];  one workgroup per (t, 64-channel-quad group): threads split s, reduce in LDS
__global__ __launch_bounds__(256) void embed_bwd_time_kernel(int B, int T_, int S, int D, const float* __restrict__ g, float* __restrict__ dtime, int accumulate) {
    __shared__ float4 red[256];
    const int t = blockIdx.x, cq = blockIdx.y * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;  // 16 channel quads x 16 row slices
    const int d4 = D / 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cq < d4) {
        for (int b = 0; b < B; ++b)
            for (int s = 1 + part; s < S; s += 16) {
                const float4 v = ld4(g + (((size_t)b * T_ + t) * S + s) * D + cq * 4); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (part == 0 && cq < d4) {
        for (int p = 1; p < 16; ++p) { const float4 v = red[p * 16 + (threadIdx.x & 15)]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        float* o = dtime + (size_t)t * D + cq * 4;
        if (accumulate) { const float4 p = ld4(o); a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
        st4(o, a);
    }
}

// ------------------------------------------------------------------------------------------ cls merge
// After the spatial projection every frame's slot 0 holds cls + attn_out(frame).  The reference keeps ONE cls per
// clip: frame 0's row for causal_attention == 1, the mean over frames for 0 (vit.py:189-198,215).  Write that row
// back to every frame's slot 0.  mode: 1 -> frame 0, 0 -> mean.  Backward: sum of the replicas' gradients goes to
// frame 0 (mode 1, others get zero) or is spread as mean (mode 0).
// CT* cast_out (backward only, may be NULL): the 16-bit operand copy of the gradient (cast_scale[row] * x[row], what the LayerNorm backward
// in front of this call wrote for every row) is refreshed for the slot-0 rows this kernel changes, so no separate cast pass is needed.
// sum over the T frames' slot-0 rows in frame order, eight independent loads in flight (one thread per 4 columns of one clip: the loop was
// T dependent-latency loads, 20 us for 276 KB)
__device__ __forceinline__ float4 sum_frames(const float* base, size_t fs, int T_) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < T_; t0 += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = t0 + u < T_ ? ld4(base + (t0 + u) * fs) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    return a;
}
template <typename CT>
__global__ void cls_merge_kernel(int B, int T_, int S, int D, float* __restrict__ x, int mode, int backward, CT* __restrict__ cast_out, long ldc,
                                 const float* __restrict__ cast_scale) {
    const int d4 = D / 4;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * d4) return;
    const int b = (int)(i / d4), c = (int)(i - (long)b * d4) * 4;
    float* base = x + (size_t)b * T_ * S * D + c;
    const size_t fs = (size_t)S * D;
    if (!backward) {
        float4 a;
        if (mode == 1) a = ld4(base);
        else {
            a = sum_frames(base, fs, T_);
            const float inv = 1.0f / (float)T_; a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        }
        for (int t = 0; t < T_; ++t) st4(base + t * fs, a);
    } else {
        float4 a = sum_frames(base, fs, T_);
        if (mode == 1) {
            st4(base, a);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 1; t < T_; ++t) st4(base + t * fs, z);
        } else {
            const float inv = 1.0f / (float)T_; a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
            for (int t = 0; t < T_; ++t) st4(base + t * fs, a);
        }
        if (cast_out) {
            for (int t0 = 0; t0 < T_; t0 += 8) {                 // (the row scales eight at a time: the loads used to wait for each other)
                float cs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) cs[u] = (cast_scale && t0 + u < T_ && !(mode == 1 && t0 + u > 0)) ? cast_scale[((long)b * T_ + t0 + u) * S] : 1.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + u;
                    if (t < T_) {
                        const float4 v = (mode == 1 && t > 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(a.x * cs[u], a.y * cs[u], a.z * cs[u], a.w * cs[u]);
                        st4(cast_out + ((long)b * T_ + t) * S * ldc + c, v);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ mask head re-layout + avg-pool
// pm rows (b,t,1+n) hold C*P*P outputs in (c, py, px) order (mask_tracker.py:114-115); pooled[bt][c][Y][X] is the
// mean over the st x st pixel block (mask_tracker.py:121-122), Y = hp*(P/st) + py/st.
template <typename T>
__global__ void unpatchify_pool_fwd_kernel(int BT, int Hp, int Wp, int P, int C, int st, const T* __restrict__ pm, float* __restrict__ pooled) {
    const int S = Hp * Wp + 1, Ps = P / st, Ho = Hp * Ps, Wo = Wp * Ps, K = C * P * P;
    const long total = (long)BT * C * Ho * Wo;
    const float inv = 1.0f / (float)(st * st);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % Wo); const int Y = (int)((i / Wo) % Ho); const int c = (int)((i / ((long)Wo * Ho)) % C); const long bt = i / ((long)Wo * Ho * C);
        const int hp = Y / Ps, py0 = (Y - hp * Ps) * st, wp = X / Ps, px0 = (X - wp * Ps) * st;
        const T* src = pm + ((size_t)bt * S + 1 + hp * Wp + wp) * K + (size_t)c * P * P;
        float a = 0.f;
        for (int dy = 0; dy < st; ++dy)
            for (int dx = 0; dx < st; ++dx) a += Elem<T>::ld(src + (py0 + dy) * P + px0 + dx);
        pooled[i] = a * inv;
    }
}
// The same for stride 4 and P % 8 == 0 (the path's head: P = 16) with 16-byte loads: one thread takes 8 consecutive patch pixels of FOUR consecutive
// patch rows (a 4 x 8 block of one channel of one token = two pooled outputs), i.e. four loads 2 P elements apart and one 8-byte store -- the
// one-output version reads its 4 x 4 window as sixteen 2-byte loads (48 us for 41 MB at configs[1]).
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_pool4_fwd_kernel(int BT, int Hp, int Wp, int P, int C, const T* __restrict__ pm, float* __restrict__ pooled) {
    const int S = Hp * Wp + 1, Ps = P / 4, Ho = Hp * Ps, Wo = Wp * Ps, K = C * P * P, P8 = P / 8;
    const int per_tok = C * Ps * P8;                         // threads per token: channel x pooled row x 8-pixel piece
    const long total = (long)BT * (S - 1) * per_tok;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tok = (int)(i / per_tok), r = (int)(i - (long)tok * per_tok);
        const int piece = r % P8, yb = (r / P8) % Ps, c = r / (P8 * Ps);
        const int bt = tok / (S - 1), n = tok - bt * (S - 1), hp = n / Wp, wp = n - hp * Wp;
        const T* src = pm + ((size_t)bt * S + 1 + n) * K + (size_t)c * P * P + (4 * yb) * P + piece * 8;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const float4 u = ld4(src + dy * P), v = ld4(src + dy * P + 4);
            a0 += (u.x + u.y) + (u.z + u.w); a1 += (v.x + v.y) + (v.z + v.w);
        }
        float* o = pooled + (((size_t)bt * C + c) * Ho + hp * Ps + yb) * Wo + wp * Ps + piece * 2;
        *reinterpret_cast<float2*>(o) = make_float2(a0 * 0.0625f, a1 * 0.0625f);
    }
}

template <typename T>
__global__ void unpatchify_pool_bwd_kernel(int BT, int Hp, int Wp, int P, int C, int st, const float* __restrict__ dpooled, T* __restrict__ dpm) {
    const int S = Hp * Wp + 1, Ps = P / st, Ho = Hp * Ps, Wo = Wp * Ps, K = C * P * P;
    const long total = (long)BT * S * K;
    const float inv = 1.0f / (float)(st * st);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K); const long row = i / K; const int s = (int)(row % S); const long bt = row / S;
        float v = 0.f;
        if (s > 0) {
            const int n = s - 1, hp = n / Wp, wp = n - hp * Wp;
            const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
            const int Y = hp * Ps + py / st, X = wp * Ps + px / st;
            v = dpooled[(((size_t)bt * C + c) * Ho + Y) * Wo + X] * inv;
        }
        Elem<T>::st(dpm + i, v);
    }
}

// The same with 8 consecutive patch pixels per thread (P % 8 == 0): 32-bit index arithmetic once per 8 outputs and one 16-byte (16-bit) /
// two 16-byte (f32) stores -- the one-element version spends its time in 64-bit divisions (98 us for 20.8 M elements at configs[1]).
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_pool_bwd8_kernel(int rows, int Hp, int Wp, int P, int C, int st, const float* __restrict__ dpooled, T* __restrict__ dpm) {
    const int S = Hp * Wp + 1, Ps = P / st, Ho = Hp * Ps, Wo = Wp * Ps, K = C * P * P, K8 = K >> 3;
    const float inv = 1.0f / (float)(st * st);
    const long total = (long)rows * K8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / K8), k = (int)(i - (long)row * K8) * 8;
        const int bt = row / S, s = row - bt * S;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const int n = s - 1, hp = n / Wp, wp = n - hp * Wp;
            const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
            const float* src = dpooled + (((size_t)bt * C + c) * Ho + hp * Ps + py / st) * Wo + wp * Ps;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[(px + e) / st] * inv;
        }
        T* dst = dpm + (size_t)row * K + k;
        st4(dst, make_float4(v[0], v[1], v[2], v[3]));
        st4(dst + 4, make_float4(v[4], v[5], v[6], v[7]));
    }
}

// ------------------------------------------------------------------------------------------ upsample
// F.interpolate(scale_factor=st, mode='bilinear', align_corners=True) or 'nearest' (mask_tracker.py:124-130),
// fused with the '(B T) C H W -> B C T H W' re-layout (mask_tracker.py:132).  Source index rule of ATen:
// align_corners: src = dst * (in-1)/(out-1); i0 = (int)src; i1 = i0 + (i0 < in-1); w1 = src - i0.
__device__ __forceinline__ void bil_src(int dst, int in, int out, int& i0, int& i1, float& w0, float& w1) {
    const float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float src = scale * (float)dst;
    i0 = (int)src; if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    w1 = src - (float)i0; w0 = 1.0f - w1;
}
// one workgroup column per (b, c, t) frame (blockIdx.y), 4 consecutive output pixels per thread (W is a multiple of st; a 16-byte
// store when W % 4 == 0), 32-bit index arithmetic only
__global__ void upsample_fwd_kernel(int B, int T_, int C, int h, int w, int st, int bilinear, const float* __restrict__ pooled, float* __restrict__ out) {
    const int H = h * st, W = w * st;
    const int f = blockIdx.y;                                   // (b*C + c)*T_ + t  (output layout B,C,T,H,W)
    const int t = f % T_, bc = f / T_, c = bc % C, b = bc / C;
    const float* src = pooled + (((size_t)(b * T_ + t)) * C + c) * h * w;
    float* dst = out + (size_t)f * H * W;
    const int Wq = (W + 3) / 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * Wq; i += gridDim.x * blockDim.x) {
        const int y = i / Wq, x0q = (i - y * Wq) * 4;
        float v[4];
        int y0, y1; float wy0, wy1;
        if (bilinear) bil_src(y, h, H, y0, y1, wy0, wy1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = x0q + e < W ? x0q + e : W - 1;
            if (bilinear) {
                int x0, x1; float wx0, wx1;
                bil_src(x, w, W, x0, x1, wx0, wx1);
                v[e] = wy0 * (wx0 * src[y0 * w + x0] + wx1 * src[y0 * w + x1]) + wy1 * (wx0 * src[y1 * w + x0] + wx1 * src[y1 * w + x1]);
            } else {
                v[e] = src[(y / st) * w + (x / st)];
            }
        }
        float* o = dst + (size_t)y * W + x0q;
        if ((W & 3) == 0) st4(o, make_float4(v[0], v[1], v[2], v[3]));
        else for (int e = 0; e < 4 && x0q + e < W; ++e) o[e] = v[e];
    }
}
// gather-form backward: one thread per pooled pixel sums the output pixels that read it
// amax_bits (may be NULL): atomic maximum of the bit patterns of |dout| over everything the fast path reads = max |dout| (every pixel is read by some
// window; the maximum is idempotent) -- the statistic the binary16 mode's loss scale needs, without two more passes over dout.
__global__ void upsample_bwd_kernel(int B, int T_, int C, int h, int w, int st, int bilinear, const float* __restrict__ dout, float* __restrict__ dpooled,
                                    unsigned* __restrict__ amax_bits) {
    const int H = h * st, W = w * st;
    const long total = (long)B * T_ * C * h * w;
    float amax = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % w); const int Y = (int)((i / w) % h); const int c = (int)((i / ((long)w * h)) % C); const long bt = i / ((long)w * h * C);
        const int b = (int)(bt / T_), t = (int)(bt - (long)b * T_);
        const float* g = dout + (((size_t)b * C + c) * T_ + t) * H * W;
        float a = 0.f;
        if (bilinear && st == 4 && h > 4 && w > 4) {
            // stride 4 (the path's head): output column x reads pooled columns floor(s x), floor(s x) + 1 with 1 / s = 4 + 3 / (w - 1), so X is
            // read by columns inside [4X - 4, 4X + 8) only (same for rows): twelve column weights once per thread, then per candidate row one
            // row weight and three aligned float4 loads -- instead of ~100 per-pixel weight evaluations and scalar loads (122 us -> see HISTORY)
            float wx[12];
            const int xa = 4 * X - 4, ya = 4 * Y - 4;
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int x = xa + e;
                wx[e] = 0.f;
                if (x >= 0 && x < W) { int x0, x1; float wx0, wx1; bil_src(x, w, W, x0, x1, wx0, wx1); if (x0 == X) wx[e] += wx0; if (x1 == X) wx[e] += wx1; }
            }
            for (int e = 0; e < 12; ++e) {
                const int y = ya + e;
                if (y < 0 || y >= H) continue;
                int y0, y1; float wy0, wy1; bil_src(y, h, H, y0, y1, wy0, wy1);
                float wy = 0.f; if (y0 == Y) wy += wy0; if (y1 == Y) wy += wy1;
                if (wy == 0.f) continue;
                const float* row = g + (size_t)y * W + xa;
                float r = 0.f;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (xa + 4 * q >= 0 && xa + 4 * q + 3 < W) {
                        const float4 v = ld4(row + 4 * q);
                        r += wx[4 * q] * v.x + wx[4 * q + 1] * v.y + wx[4 * q + 2] * v.z + wx[4 * q + 3] * v.w;
                        if (amax_bits && q == 1) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));   // (columns 4X .. 4X+3: each pixel once per row window)
                    }
                }
                a += wy * r;
            }
        } else if (bilinear) {
            // candidate output rows: those whose i0 or i1 can equal Y
            // an output row y reads pooled rows floor(s*y) and floor(s*y)+1, s = (h-1)/(H-1): Y is among them only for
            // (Y-1)/s < y < (Y+1)/s -- about 2/s + 1 rows (9 at stride 4) instead of the 4*st + 1 of the safe bound
            const float isy = (h > 1) ? (float)(H - 1) / (float)(h - 1) : 0.f, isx = (w > 1) ? (float)(W - 1) / (float)(w - 1) : 0.f;
            int ylo = (int)((float)(Y - 1) * isy) - 1, yhi = (int)((float)(Y + 1) * isy) + 2; if (ylo < 0) ylo = 0; if (yhi > H - 1) yhi = H - 1;
            int xlo = (int)((float)(X - 1) * isx) - 1, xhi = (int)((float)(X + 1) * isx) + 2; if (xlo < 0) xlo = 0; if (xhi > W - 1) xhi = W - 1;
            if (h <= 1) { ylo = 0; yhi = H - 1; }
            if (w <= 1) { xlo = 0; xhi = W - 1; }
            for (int y = ylo; y <= yhi; ++y) {
                int y0, y1; float wy0, wy1; bil_src(y, h, H, y0, y1, wy0, wy1);
                float wy = 0.f; if (y0 == Y) wy += wy0; if (y1 == Y) wy += wy1;
                if (wy == 0.f) continue;
                for (int x = xlo; x <= xhi; ++x) {
                    int x0, x1; float wx0, wx1; bil_src(x, w, W, x0, x1, wx0, wx1);
                    float wx = 0.f; if (x0 == X) wx += wx0; if (x1 == X) wx += wx1;
                    if (wx != 0.f) a += wy * wx * g[(size_t)y * W + x];
                }
            }
        } else {
            for (int dy = 0; dy < st; ++dy)
                for (int dx = 0; dx < st; ++dx) a += g[(size_t)(Y * st + dy) * W + X * st + dx];
        }
        dpooled[i] = a;
    }
    if (amax_bits) {
        // one atomic per WORKGROUP, spread over TCOW_AMAX_SLOTS words (the caller takes the maximum of the slots): device-scope atomics on ONE word are
        // served one after the other (~10 ns each) -- with one per wave on one word the 16 000 of them were 160 us of a 200 us kernel
        __shared__ float wmax[4];
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (m > 0.f) atomicMax(amax_bits + (blockIdx.x & (TCOW_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, m));      // (non-negative floats order as their bit patterns)
        }
    }
}

// ------------------------------------------------------------------------------------------ flags
// flags[b,t,f] = mean_n (Wf . x[b,t,1+n] + bf) = Wf . mean_n x + bf   (mask_tracker.py:135-137)
// One workgroup of 1024 threads per (b,t): thread (g, c) sums rows 1+g, 1+g+G, ... of float4 column c (G = 1024 / (D/4) row
// groups run in parallel; a single thread per column would walk all S rows serially and leave the load pipe empty).
__global__ __launch_bounds__(1024) void flags_fwd_kernel(int S, int D, int F, const float* __restrict__ x, const float* __restrict__ Wf, const float* __restrict__ bf,
                                                         float* __restrict__ flags) {
    extern __shared__ float4 part[];  // [G][D/4]; row 0 ends up holding the mean
    const int bt = blockIdx.x;
    const int D4 = D >> 2;
    const int G = 1024 / D4 > 0 ? 1024 / D4 : 1;
    const float4* xb = reinterpret_cast<const float4*>(x + (size_t)bt * S * D);
    for (int idx = threadIdx.x; idx < G * D4; idx += 1024) {
        const int g = idx / D4, c = idx - g * D4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int s = 1 + g; s < S; s += G) {
            const float4 v = xb[(size_t)s * D4 + c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        part[idx] = a;
    }
    __syncthreads();
    const float inv = 1.0f / (float)(S - 1);
    for (int c = threadIdx.x; c < D4; c += 1024) {
        float4 a = part[c];
        for (int g = 1; g < G; ++g) { const float4 v = part[g * D4 + c]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        part[c] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
    __syncthreads();
    const float* mean = reinterpret_cast<const float*>(part);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int f = wave; f < F; f += 16) {
        float a = 0.f;
        for (int d = lane; d < D; d += 64) a += mean[d] * Wf[(size_t)f * D + d];
        a = wave_sum(a);
        if (lane == 0) flags[(size_t)bt * F + f] = a + bf[f];
    }
}

// All operand copies of a step in ONE launch: a device-resident table of (W, Wc, Wt, N, K, first tile) records, one tile per workgroup; the
// workgroup finds its record by bisection on the first-tile column (86 weights -> 7 steps).  Tile edge = record.tile (32 or 64): the caller
// counts 64 x 64 tiles when every N and K is a multiple of 64 (all GEMM weights of the reference geometries) -- 16-byte loads and 8-byte
// (16-bit) / 16-byte (f32) stores instead of one element per thread (389 -> ~200 us for the 120 M weights of ViT-B).
struct CastDesc { const float* W; void* Wc; void* Wt; int N, K, tile_begin, tile; };
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const CastDesc* __restrict__ tab, int n) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].tile_begin <= b) lo = mid; else hi = mid - 1; }
    const CastDesc d = tab[lo];
    T* Wc = reinterpret_cast<T*>(d.Wc); T* Wt = reinterpret_cast<T*>(d.Wt);
    if (d.tile == 64) {                                   // N % 64 == 0 and K % 64 == 0: no edge handling
        const int t = b - d.tile_begin, tiles_k = d.K >> 6;
        const int n0 = (t / tiles_k) * 64, k0 = (t - (t / tiles_k) * tiles_k) * 64;
        const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 16 * i;
            const float4 v = ld4(d.W + (size_t)(n0 + r) * d.K + k0 + c4);
            if (Wc) st4(Wc + (size_t)(n0 + r) * d.K + k0 + c4, v);
            tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
        }
        __syncthreads();
        if (Wt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = r0 + 16 * i;
                st4(Wt + (size_t)(k0 + k) * d.N + n0 + c4, make_float4(tile[c4][k], tile[c4 + 1][k], tile[c4 + 2][k], tile[c4 + 3][k]));
            }
        }
        return;
    }
    const int t = b - d.tile_begin, tiles_k = (d.K + 31) / 32;
    const int n0 = (t / tiles_k) * 32, k0 = (t - (t / tiles_k) * tiles_k) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int nn = n0 + r, k = k0 + tx;
        float v = 0.f;
        if (nn < d.N && k < d.K) { v = d.W[(size_t)nn * d.K + k]; if (Wc) Elem<T>::st(Wc + (size_t)nn * d.K + k, v); }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (Wt)
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r, nn = n0 + tx;
            if (k < d.K && nn < d.N) Elem<T>::st(Wt + (size_t)k * d.N + nn, tile[tx][r]);
        }
}

// ------------------------------------------------------------------------------------------ casts
// dst = T(src * row_scale[row])  (f32 -> T), used to turn residual-stream gradients into GEMM operands
template <typename T>
__global__ void scale_cast_kernel(long rows, int D, const float* __restrict__ src, long lds_, const float* __restrict__ row_scale, T* __restrict__ dst, long ldd) {
    const int d4 = D / 4;
    const long total = rows * d4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / d4; const int c = (int)(i - row * d4) * 4;
        float4 v = ld4(src + row * lds_ + c);
        if (row_scale) { const float s = row_scale[row]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
        st4(dst + row * ldd + c, v);
    }
}
// W [N,K] f32 -> Wc [N,K] (T, optional) and Wt [K,N] (T, optional): 32x32 tile transpose through LDS
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(int N, int K, const float* __restrict__ Wsrc, T* __restrict__ Wc, T* __restrict__ Wt) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        float v = 0.f;
        if (n < N && k < K) { v = Wsrc[(size_t)n * K + k]; if (Wc) Elem<T>::st(Wc + (size_t)n * K + k, v); }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (Wt)
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r, n = n0 + tx;
            if (n < N && k < K) Elem<T>::st(Wt + (size_t)k * N + n, tile[tx][r]);
        }
}

}  // namespace

static inline int gs_blocks(long total, int per_block = 256) { long b = (total + per_block - 1) / per_block; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

// DropPath row scales of every block of a divided space-time step (vit_utils.py:139-154 applied at vit.py:172-186): keep = floor(u + keep_p),
// scale = keep / keep_p, one draw per (clip, spatial position) for the temporal branch, per (clip, frame) for the spatial one, per clip for the
// MLP.  out [4][depth][B*T*S]: temporal | spatial | MLP | temporal x mask0 (the row scale of the folded temporal projection).
__global__ void __launch_bounds__(256) droppath_rows_kernel(int depth, int B, int T, int S, const float* __restrict__ u, const float* __restrict__ keep_p,
                                                            const float* __restrict__ mask0, float* __restrict__ out) {
    const long M = (long)B * T * S, n = (long)depth * M;
    const int N = S - 1, per = B * N + B * T + B;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int l = (int)(i / M); const long r = i - (long)l * M;
        const int b = (int)(r / ((long)T * S)), ts = (int)(r - (long)b * T * S), t = ts / S, sp = ts - t * S;
        const float kp = keep_p[l];
        const float* ul = u + (long)l * per;
        const float st = sp == 0 ? 1.0f : floorf(ul[b * N + sp - 1] + kp) / kp;
        const float ss = floorf(ul[B * N + b * T + t] + kp) / kp;
        const float sm = floorf(ul[B * N + B * T + b] + kp) / kp;
        out[i] = st; out[n + i] = ss; out[2 * n + i] = sm; out[3 * n + i] = st * mask0[r];
    }
}

extern "C" {

int tcow_im2col(void* stream, int dtype, int B, int T_, int H, int W, int P, const float* rgb, const float* query, int pretrained_norm, void* out) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && P > 0 && P % 4 == 0 && H % P == 0 && W % P == 0, "tcow_im2col: bad geometry B=%d T=%d H=%d W=%d P=%d", B, T_, H, W, P);
    TCOW_CHECK_ARG(rgb && query && out, "tcow_im2col: null pointer");
    const long total = (long)B * T_ * ((H / P) * (W / P) + 1) * (4 * P * P / 4);
    if (dtype == TCOW_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, B, T_, H, W, P, 3, 1, rgb, query, pretrained_norm, (bf16_t*)out);
    else if (dtype == TCOW_F32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, B, T_, H, W, P, 3, 1, rgb, query, pretrained_norm, (float*)out);
    else { tcow_set_error("tcow_im2col: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_gather_frames(void* stream, int elem_bytes, int C, int Tv, int H, int W, int Tc, int h, int w, const void* src, const int* frame_idx,
                       const int* src_y, const int* src_x, void* out) {
    TCOW_CHECK_ARG(C > 0 && Tv > 0 && H > 0 && W > 0 && Tc > 0 && h > 0 && w > 0 && src && frame_idx && src_y && src_x && out, "tcow_gather_frames: bad arguments");
    TCOW_CHECK_ARG(elem_bytes == 1 || elem_bytes == 4, "tcow_gather_frames: elem_bytes must be 1 or 4 (got %d)", elem_bytes);
    const long total = (long)C * Tc * h * w;
    if (elem_bytes == 1) hipLaunchKernelGGL(gather_frames_kernel<uint8_t>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, C, Tv, H, W, Tc, h, w, (const uint8_t*)src, frame_idx, src_y, src_x, (uint8_t*)out);
    else hipLaunchKernelGGL(gather_frames_kernel<uint32_t>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, C, Tv, H, W, Tc, h, w, (const uint32_t*)src, frame_idx, src_y, src_x, (uint32_t*)out);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_resize_aa(void* stream, int C, int Tv, int H, int W, int Tc, int hc, int wc, int oh, int ow, const float* src, const int* frame_idx, const int* ys, const int* xs,
                   const int* ymin, const int* ysize, const float* wy, int ky, const int* xmin, const int* xsize, const float* wx, int kx, float* out) {
    TCOW_CHECK_ARG(C > 0 && Tv > 0 && H > 0 && W > 0 && Tc > 0 && hc > 0 && wc > 0 && oh > 0 && ow > 0 && ky > 0 && kx > 0, "tcow_resize_aa: bad geometry");
    TCOW_CHECK_ARG(src && frame_idx && ys && xs && ymin && ysize && wy && xmin && xsize && wx && out, "tcow_resize_aa: null pointer");
    const long total = (long)C * Tc * oh * ow;
    hipLaunchKernelGGL(resize_aa_kernel, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, C, Tv, H, W, Tc, oh, ow, src, frame_idx, ys, xs, ymin, ysize, wy, ky, xmin, xsize,
                       wx, kx, out);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_im2col_channels(void* stream, int dtype, int B, int T_, int H, int W, int P, int C, const float* src, int normalise, void* out) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && C > 0 && P > 0 && P % 4 == 0 && H % P == 0 && W % P == 0, "tcow_im2col_channels: bad geometry B=%d T=%d H=%d W=%d P=%d C=%d", B, T_, H, W, P, C);
    TCOW_CHECK_ARG(src && out, "tcow_im2col_channels: null pointer");
    const long total = (long)B * T_ * ((H / P) * (W / P) + 1) * (C * P * P / 4);
    if (dtype == TCOW_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, B, T_, H, W, P, C, 0, src, src, normalise, (bf16_t*)out);
    else if (dtype == TCOW_F32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, B, T_, H, W, P, C, 0, src, src, normalise, (float*)out);
    else { tcow_set_error("tcow_im2col_channels: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_embed_fwd(void* stream, int B, int T_, int S, int D, float* x, const float* cls, const float* pos, const float* time_embed) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && S > 1 && D % 4 == 0 && x && cls && pos && time_embed, "tcow_embed_fwd: bad arguments");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(gs_blocks((long)B * T_ * S * D / 4)), dim3(256), 0, (hipStream_t)stream, B, T_, S, D, x, cls, pos, time_embed);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_embed_bwd(void* stream, int B, int T_, int S, int D, const float* g, float* dpos, float* dtime, int accumulate) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && S > 1 && D % 4 == 0 && g && dpos && dtime, "tcow_embed_bwd: bad arguments");
    hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3(S, cdiv(D / 4, 64)), dim3(256), 0, (hipStream_t)stream, B, T_, S, D, g, dpos, accumulate);
    TCOW_CHECK_LAUNCH();
    hipLaunchKernelGGL(embed_bwd_time_kernel, dim3(T_, cdiv(D / 4, 16)), dim3(256), 0, (hipStream_t)stream, B, T_, S, D, g, dtime, accumulate);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_cls_merge(void* stream, int B, int T_, int S, int D, float* x, int mode, int backward) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && S > 1 && D % 4 == 0 && x && (mode == 0 || mode == 1), "tcow_cls_merge: bad arguments");
    hipLaunchKernelGGL(cls_merge_kernel<float>, dim3(cdiv((long)B * D / 4, 64)), dim3(64), 0, (hipStream_t)stream, B, T_, S, D, x, mode, backward, (float*)nullptr, 0L, (const float*)nullptr);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_cls_merge_bwd_cast(void* stream, int dtype, int B, int T_, int S, int D, float* x, int mode, void* cast_out, long ldc, const float* cast_scale) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && S > 1 && D % 4 == 0 && x && (mode == 0 || mode == 1) && cast_out && ldc % 4 == 0, "tcow_cls_merge_bwd_cast: bad arguments");
    const dim3 grid(cdiv((long)B * D / 4, 64));      // (one wave per workgroup: 576 threads of latency-bound work over 9 CUs rather than 3)
    if (dtype == TCOW_BF16) hipLaunchKernelGGL(cls_merge_kernel<bf16_t>, grid, dim3(64), 0, (hipStream_t)stream, B, T_, S, D, x, mode, 1, (bf16_t*)cast_out, ldc, cast_scale);
    else if (dtype == TCOW_F32) hipLaunchKernelGGL(cls_merge_kernel<float>, grid, dim3(64), 0, (hipStream_t)stream, B, T_, S, D, x, mode, 1, (float*)cast_out, ldc, cast_scale);
    else { tcow_set_error("tcow_cls_merge_bwd_cast: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_unpatchify_pool_fwd(void* stream, int dtype, int BT, int Hp, int Wp, int P, int C, int st, const void* pm, float* pooled) {
    TCOW_CHECK_ARG(BT > 0 && Hp > 0 && Wp > 0 && P > 0 && C > 0 && st > 0 && P % st == 0 && pm && pooled, "tcow_unpatchify_pool_fwd: bad arguments");
    const long total = (long)BT * C * Hp * Wp * (P / st) * (P / st);
    if (st == 4 && P % 8 == 0 && ((P / 4) * Wp) % 2 == 0 && (dtype == TCOW_BF16 || dtype == TCOW_F32)) {      // the path's head: 16-byte loads (see the kernel)
        const long nthr = (long)BT * Hp * Wp * C * (P / 4) * (P / 8);
        if (dtype == TCOW_BF16) hipLaunchKernelGGL(unpatchify_pool4_fwd_kernel<bf16_t>, dim3(gs_blocks(nthr)), dim3(256), 0, (hipStream_t)stream, BT, Hp, Wp, P, C, (const bf16_t*)pm, pooled);
        else hipLaunchKernelGGL(unpatchify_pool4_fwd_kernel<float>, dim3(gs_blocks(nthr)), dim3(256), 0, (hipStream_t)stream, BT, Hp, Wp, P, C, (const float*)pm, pooled);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    if (dtype == TCOW_BF16) hipLaunchKernelGGL(unpatchify_pool_fwd_kernel<bf16_t>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, BT, Hp, Wp, P, C, st, (const bf16_t*)pm, pooled);
    else if (dtype == TCOW_F32) hipLaunchKernelGGL(unpatchify_pool_fwd_kernel<float>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, BT, Hp, Wp, P, C, st, (const float*)pm, pooled);
    else { tcow_set_error("tcow_unpatchify_pool_fwd: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_unpatchify_pool_bwd(void* stream, int dtype, int BT, int Hp, int Wp, int P, int C, int st, const float* dpooled, void* dpm) {
    TCOW_CHECK_ARG(BT > 0 && Hp > 0 && Wp > 0 && P > 0 && C > 0 && st > 0 && P % st == 0 && dpooled && dpm, "tcow_unpatchify_pool_bwd: bad arguments");
    const long total = (long)BT * (Hp * Wp + 1) * C * P * P;
    if (P % 8 == 0 && (dtype == TCOW_BF16 || dtype == TCOW_F32) && (reinterpret_cast<uintptr_t>(dpm) & 15) == 0) {
        const int rows = BT * (Hp * Wp + 1);
        if (dtype == TCOW_BF16) hipLaunchKernelGGL(unpatchify_pool_bwd8_kernel<bf16_t>, dim3(gs_blocks(total / 8)), dim3(256), 0, (hipStream_t)stream, rows, Hp, Wp, P, C, st, dpooled, (bf16_t*)dpm);
        else hipLaunchKernelGGL(unpatchify_pool_bwd8_kernel<float>, dim3(gs_blocks(total / 8)), dim3(256), 0, (hipStream_t)stream, rows, Hp, Wp, P, C, st, dpooled, (float*)dpm);
        TCOW_CHECK_LAUNCH();
        return TCOW_OK;
    }
    if (dtype == TCOW_BF16) hipLaunchKernelGGL(unpatchify_pool_bwd_kernel<bf16_t>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, BT, Hp, Wp, P, C, st, dpooled, (bf16_t*)dpm);
    else if (dtype == TCOW_F32) hipLaunchKernelGGL(unpatchify_pool_bwd_kernel<float>, dim3(gs_blocks(total)), dim3(256), 0, (hipStream_t)stream, BT, Hp, Wp, P, C, st, dpooled, (float*)dpm);
    else { tcow_set_error("tcow_unpatchify_pool_bwd: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_upsample_fwd(void* stream, int B, int T_, int C, int h, int w, int st, int bilinear, const float* pooled, float* out) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && C > 0 && h > 0 && w > 0 && st > 0 && pooled && out, "tcow_upsample_fwd: bad arguments");
    TCOW_CHECK_ARG((long)B * C * T_ < 65536, "tcow_upsample_fwd: too many frames for one call");
    { const int per = h * st * ((w * st + 3) / 4); int gx = cdiv(per, 256); if (gx > 64) gx = 64;
      hipLaunchKernelGGL(upsample_fwd_kernel, dim3(gx, B * C * T_), dim3(256), 0, (hipStream_t)stream, B, T_, C, h, w, st, bilinear, pooled, out); }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_upsample_bwd(void* stream, int B, int T_, int C, int h, int w, int st, int bilinear, const float* dout, float* dpooled) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && C > 0 && h > 0 && w > 0 && st > 0 && dout && dpooled, "tcow_upsample_bwd: bad arguments");
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(gs_blocks((long)B * C * T_ * h * w)), dim3(256), 0, (hipStream_t)stream, B, T_, C, h, w, st, bilinear, dout, dpooled, (unsigned*)nullptr);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_upsample_bwd_amax(void* stream, int B, int T_, int C, int h, int w, int st, const float* dout, float* dpooled, unsigned* amax_bits, int n_slots) {
    TCOW_CHECK_ARG(B > 0 && T_ > 0 && C > 0 && h > 4 && w > 4 && st == 4 && dout && dpooled && amax_bits, "tcow_upsample_bwd_amax: bilinear stride 4 with h, w > 4 only");
    TCOW_CHECK_ARG(n_slots == TCOW_AMAX_SLOTS, "tcow_upsample_bwd_amax: amax_bits must hold TCOW_AMAX_SLOTS = %d words (got %d)", TCOW_AMAX_SLOTS, n_slots);
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(gs_blocks((long)B * C * T_ * h * w)), dim3(256), 0, (hipStream_t)stream, B, T_, C, h, w, st, 1, dout, dpooled, amax_bits);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_flags_fwd(void* stream, int BT, int S, int D, int F, const float* x, const float* Wf, const float* bf, float* flags) {
    TCOW_CHECK_ARG(BT > 0 && S > 1 && D > 0 && F > 0 && x && Wf && bf && flags, "tcow_flags_fwd: bad arguments");
    TCOW_CHECK_ARG(D % 4 == 0, "tcow_flags_fwd: D %d must be a multiple of 4", D);
    const int D4 = D / 4, G = 1024 / D4 > 0 ? 1024 / D4 : 1;
    hipLaunchKernelGGL(flags_fwd_kernel, dim3(BT), dim3(1024), (size_t)G * D4 * 16, (hipStream_t)stream, S, D, F, x, Wf, bf, flags);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

// x *= *scale unless *scale == 1 (then no memory is touched): the upstream gradient of a scalar loss is 1 in the plain training step, and a
// device-side branch is cheaper than a pass over an 83 MB gradient for it (the host cannot look at a device scalar without a synchronisation).
__global__ __launch_bounds__(256) void scale_unless_one_kernel(float* __restrict__ x, long n4, long n, const float* __restrict__ scale) {
    const float s = scale[0];
    if (s == 1.0f) return;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = ld4(x + i * 4); v.x *= s; v.y *= s; v.z *= s; v.w *= s; st4(x + i * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) x[n4 * 4 + threadIdx.x] *= s;
}

int tcow_scale_unless_one(void* stream, float* x, long n, const float* scale) {
    TCOW_CHECK_ARG(x && scale && n > 0 && ((uintptr_t)x & 15) == 0, "tcow_scale_unless_one: bad arguments");
    hipLaunchKernelGGL(scale_unless_one_kernel, dim3(gs_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, n / 4, n, scale);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_scale_cast(void* stream, int dtype, long rows, int D, const float* src, long ld_src, const float* row_scale, void* dst, long ld_dst) {
    TCOW_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && src && dst && ld_src % 4 == 0 && ld_dst % 4 == 0, "tcow_scale_cast: bad arguments");
    if (dtype == TCOW_BF16) hipLaunchKernelGGL(scale_cast_kernel<bf16_t>, dim3(gs_blocks(rows * D / 4)), dim3(256), 0, (hipStream_t)stream, rows, D, src, ld_src, row_scale, (bf16_t*)dst, ld_dst);
    else if (dtype == TCOW_F32) hipLaunchKernelGGL(scale_cast_kernel<float>, dim3(gs_blocks(rows * D / 4)), dim3(256), 0, (hipStream_t)stream, rows, D, src, ld_src, row_scale, (float*)dst, ld_dst);
    else { tcow_set_error("tcow_scale_cast: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_cast_transpose(void* stream, int dtype, int N, int K, const float* W, void* Wc, void* Wt) {
    TCOW_CHECK_ARG(N > 0 && K > 0 && W && (Wc || Wt), "tcow_cast_transpose: bad arguments");
    const dim3 grid(cdiv(K, 32), cdiv(N, 32));
    if (dtype == TCOW_BF16) hipLaunchKernelGGL(cast_transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, N, K, W, (bf16_t*)Wc, (bf16_t*)Wt);
    else if (dtype == TCOW_F32) hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, N, K, W, (float*)Wc, (float*)Wt);
    else { tcow_set_error("tcow_cast_transpose: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

long tcow_cast_desc_bytes(void) { return (long)sizeof(CastDesc); }

int tcow_cast_transpose_batched(void* stream, int dtype, const void* table, int n, int total_tiles) {
    TCOW_CHECK_ARG(table && n > 0 && total_tiles > 0, "tcow_cast_transpose_batched: bad arguments");
    if (dtype == TCOW_BF16) hipLaunchKernelGGL(cast_transpose_batched_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const CastDesc*)table, n);
    else if (dtype == TCOW_F32) hipLaunchKernelGGL(cast_transpose_batched_kernel<float>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const CastDesc*)table, n);
    else { tcow_set_error("tcow_cast_transpose_batched: unknown dtype %d", dtype); return TCOW_ERR_INVALID_ARG; }
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

int tcow_droppath_rows(void* stream, int depth, int B, int T, int S, const float* u, const float* keep_p, const float* mask0, float* out) {
    TCOW_CHECK_ARG(depth > 0 && B > 0 && T > 0 && S > 1 && u && keep_p && mask0 && out, "tcow_droppath_rows: bad arguments");
    const long n = (long)depth * B * T * S;
    hipLaunchKernelGGL(droppath_rows_kernel, dim3(gs_blocks(n)), dim3(256), 0, (hipStream_t)stream, depth, B, T, S, u, keep_p, mask0, out);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

}  // extern "C"
