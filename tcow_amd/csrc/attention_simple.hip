// Exact-f32-arithmetic attention over strided token sequences (parity mode for both dtypes' storage).
// Replaces the core of Attention.forward, softmax(q k^T * d^-0.5 [causal mask]) v (vit.py:88-109), for
//   temporal attention: one sequence of T frames per (clip, patch slot s>=1, head)      (vit.py:169-172)
//   spatial attention : one sequence of S (or S-1) tokens per (clip, frame, head)        (vit.py:184-186,206-208)
// directly on the [rows, 3D] output of the qkv GEMM (row order which*D + head*64 + j, vit.py:81-83), writing
// [rows, D] in the head*64 + j order the reference gets from transpose(1,2).reshape (vit.py:109).
// No score matrix is materialised: one thread per query walks the keys 32 at a time through LDS with an online
// softmax.  The backward recomputes probabilities from the saved log-sum-exp.
// These kernels do all arithmetic in f32 on the VALU; the bf16 MFMA kernels live in attention_bf16.hip.
#include <stdlib.h>

#include "attention_common.h"

namespace {

constexpr int HD = ATT_HD;
constexpr int KC = 32;    // keys (or queries) per LDS chunk

template <typename T>
__device__ __forceinline__ void load_row64(const T* p, float* dst) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) { const float4 v = ld4(p + 4 * i); dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w; }
}

// cooperative load of up to KC rows x 64 elements (positions p0.. of the sequence, column offset col) into LDS as f32
template <typename T>
__device__ __forceinline__ void stage_rows(const T* base, long pos_stride_elems, int p0, int L, float (*dst)[HD], int tid) {
    for (int i = tid; i < KC * (HD / 4); i += 64) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + r < L) v = ld4(base + (size_t)(p0 + r) * pos_stride_elems + c4);
        *reinterpret_cast<float4*>(&dst[r][c4]) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_fwd_simple(SeqDesc sd, const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse) {
    __shared__ __attribute__((aligned(16))) float Ks[KC][HD];
    __shared__ __attribute__((aligned(16))) float Vs[KC][HD];
    const int tid = threadIdx.x;
    const int item = blockIdx.x / sd.heads, h = blockIdx.x - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D;
    const int qi = blockIdx.y * 64 + tid;
    const bool active = qi < sd.L;
    const T* qkv_h = qkv + base * ld3 + h * HD;
    const long pse = sd.pos_stride * ld3;
    float q[HD], o[HD];
    float m = -INFINITY, l = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = 0.f; o[d] = 0.f; }
    if (active) {
        load_row64(qkv_h + (size_t)qi * pse, q);
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] *= 0.125f;
    }
    // keys needed by this block: up to max query position + diag
    const int qmax = min(sd.L - 1, blockIdx.y * 64 + 63);
    const long kend_l = (long)qmax + sd.diag + 1;
    const int kend = kend_l < sd.L ? (int)kend_l : sd.L;
    for (int k0 = 0; k0 < kend; k0 += KC) {
        __syncthreads();
        stage_rows(qkv_h + sd.D, pse, k0, sd.L, Ks, tid);
        stage_rows(qkv_h + 2 * sd.D, pse, k0, sd.L, Vs, tid);
        __syncthreads();
        if (!active) continue;
        float s[KC];
        float cmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(&Ks[j][d]);
                a = fmaf(q[d], kv.x, a); a = fmaf(q[d + 1], kv.y, a); a = fmaf(q[d + 2], kv.z, a); a = fmaf(q[d + 3], kv.w, a);
            }
            const int kp = k0 + j;
            const bool ok = kp < sd.L && (long)kp <= (long)qi + sd.diag;
            s[j] = ok ? a : -INFINITY;
            cmax = fmaxf(cmax, s[j]);
        }
        if (cmax == -INFINITY) continue;
        const float mn = fmaxf(m, cmax);
        const float alpha = __expf(m - mn);
        l *= alpha;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const float p = __expf(s[j] - mn);
            l += p;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(&Vs[j][d]);
                o[d] = fmaf(p, vv.x, o[d]); o[d + 1] = fmaf(p, vv.y, o[d + 1]); o[d + 2] = fmaf(p, vv.z, o[d + 2]); o[d + 3] = fmaf(p, vv.w, o[d + 3]);
            }
        }
        m = mn;
    }
    if (!active) return;
    const float inv = 1.0f / l;
    const long row = base + (long)qi * sd.pos_stride;
    T* orow = out + row * sd.D + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) st4(orow + d, make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv));
    if (lse) lse[row * sd.heads + h] = m + __logf(l);
}

// delta[row][h] = sum_d dO * O
template <typename T>
__global__ void attn_delta_kernel(long rows, int heads, int D, const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= rows * heads) return;
    const long row = i / heads; const int h = (int)(i - row * heads);
    const T* a = o + row * D + h * HD; const T* b = dout + row * D + h * HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) { const float4 x = ld4(a + d), y = ld4(b + d); s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; }
    delta[i] = s;
}

// dQ: one thread per query.
template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_dq_simple(SeqDesc sd, const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                                                         const float* __restrict__ delta, T* __restrict__ dqkv) {
    __shared__ __attribute__((aligned(16))) float Ks[KC][HD];
    __shared__ __attribute__((aligned(16))) float Vs[KC][HD];
    const int tid = threadIdx.x;
    const int item = blockIdx.x / sd.heads, h = blockIdx.x - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D;
    const int qi = blockIdx.y * 64 + tid;
    const bool active = qi < sd.L;
    const T* qkv_h = qkv + base * ld3 + h * HD;
    const long pse = sd.pos_stride * ld3;
    const long row = base + (long)qi * sd.pos_stride;
    float q[HD], dq[HD], dO[HD];
    float ls = 0.f, dl = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = 0.f; dq[d] = 0.f; dO[d] = 0.f; }
    if (active) {
        load_row64(qkv_h + (size_t)qi * pse, q);
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] *= 0.125f;
        load_row64(dout + row * sd.D + h * HD, dO);
        ls = lse[row * sd.heads + h]; dl = delta[row * sd.heads + h];
    }
    const int qmax = min(sd.L - 1, blockIdx.y * 64 + 63);
    const long kend_l = (long)qmax + sd.diag + 1;
    const int kend = kend_l < sd.L ? (int)kend_l : sd.L;
    for (int k0 = 0; k0 < kend; k0 += KC) {
        __syncthreads();
        stage_rows(qkv_h + sd.D, pse, k0, sd.L, Ks, tid);
        stage_rows(qkv_h + 2 * sd.D, pse, k0, sd.L, Vs, tid);
        __syncthreads();
        if (!active) continue;
        for (int j = 0; j < KC; ++j) {
            const int kp = k0 + j;
            if (!(kp < sd.L && (long)kp <= (long)qi + sd.diag)) continue;
            float a = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(&Ks[j][d]);
                const float4 vv = *reinterpret_cast<const float4*>(&Vs[j][d]);
                a = fmaf(q[d], kv.x, a); a = fmaf(q[d + 1], kv.y, a); a = fmaf(q[d + 2], kv.z, a); a = fmaf(q[d + 3], kv.w, a);
                dp = fmaf(dO[d], vv.x, dp); dp = fmaf(dO[d + 1], vv.y, dp); dp = fmaf(dO[d + 2], vv.z, dp); dp = fmaf(dO[d + 3], vv.w, dp);
            }
            const float p = __expf(a - ls);
            const float ds = p * (dp - dl);
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(&Ks[j][d]);
                dq[d] = fmaf(ds, kv.x, dq[d]); dq[d + 1] = fmaf(ds, kv.y, dq[d + 1]); dq[d + 2] = fmaf(ds, kv.z, dq[d + 2]); dq[d + 3] = fmaf(ds, kv.w, dq[d + 3]);
            }
        }
    }
    if (!active) return;
    T* drow = dqkv + row * ld3 + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) st4(drow + d, make_float4(dq[d] * 0.125f, dq[d + 1] * 0.125f, dq[d + 2] * 0.125f, dq[d + 3] * 0.125f));
}

// dK, dV: two threads per key (each owns 32 of the 64 channels), 32 keys per block; queries staged through LDS.
template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_dkv_simple(SeqDesc sd, const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, T* __restrict__ dqkv) {
    __shared__ __attribute__((aligned(16))) float Qs[KC][HD];
    __shared__ __attribute__((aligned(16))) float Os[KC][HD];
    __shared__ float Ls[KC], Dl[KC];
    const int tid = threadIdx.x;
    const int item = blockIdx.x / sd.heads, h = blockIdx.x - item * sd.heads;
    const long base = seq_base(sd, item);
    const long ld3 = 3L * sd.D;
    const int kp = blockIdx.y * 32 + (tid >> 1);
    const int half = tid & 1;
    const bool active = kp < sd.L;
    const T* qkv_h = qkv + base * ld3 + h * HD;
    const long pse = sd.pos_stride * ld3;
    float k[32], v[32], dk[32], dv[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    if (active) {
        const T* kr = qkv_h + (size_t)kp * pse + sd.D + half * 32;
        const T* vr = qkv_h + (size_t)kp * pse + 2 * sd.D + half * 32;
#pragma unroll
        for (int d = 0; d < 32; d += 4) {
            const float4 a = ld4(kr + d), b = ld4(vr + d);
            k[d] = a.x; k[d + 1] = a.y; k[d + 2] = a.z; k[d + 3] = a.w; v[d] = b.x; v[d + 1] = b.y; v[d + 2] = b.z; v[d + 3] = b.w;
        }
    }
    // queries that can see any key of this block: q >= kmin - diag
    const long qstart_l = (long)blockIdx.y * 32 - sd.diag;
    const int qstart = qstart_l > 0 ? (int)((qstart_l / KC) * KC) : 0;
    for (int q0 = qstart; q0 < sd.L; q0 += KC) {
        __syncthreads();
        stage_rows(qkv_h, pse, q0, sd.L, Qs, tid);
        for (int i = tid; i < KC * (HD / 4); i += 64) {
            const int r = i >> 4, c4 = (i & 15) * 4;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + r < sd.L) x = ld4(dout + (base + (long)(q0 + r) * sd.pos_stride) * sd.D + h * HD + c4);
            *reinterpret_cast<float4*>(&Os[r][c4]) = x;
        }
        if (tid < KC) {
            const int qi = q0 + tid;
            const long row = base + (long)qi * sd.pos_stride;
            Ls[tid] = qi < sd.L ? lse[row * sd.heads + h] : 0.f;
            Dl[tid] = qi < sd.L ? delta[row * sd.heads + h] : 0.f;
        }
        __syncthreads();
        for (int j = 0; j < KC; ++j) {
            const int qi = q0 + j;
            if (qi >= sd.L) break;
            float a = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; d += 4) {
                const float4 qv = *reinterpret_cast<const float4*>(&Qs[j][half * 32 + d]);
                const float4 ov = *reinterpret_cast<const float4*>(&Os[j][half * 32 + d]);
                a = fmaf(qv.x, k[d], a); a = fmaf(qv.y, k[d + 1], a); a = fmaf(qv.z, k[d + 2], a); a = fmaf(qv.w, k[d + 3], a);
                dp = fmaf(ov.x, v[d], dp); dp = fmaf(ov.y, v[d + 1], dp); dp = fmaf(ov.z, v[d + 2], dp); dp = fmaf(ov.w, v[d + 3], dp);
            }
            a += __shfl_xor(a, 1, 64); dp += __shfl_xor(dp, 1, 64);
            const bool ok = active && (long)kp <= (long)qi + sd.diag;
            const float p = ok ? __expf(a * 0.125f - Ls[j]) : 0.f;
            const float ds = p * (dp - Dl[j]) * 0.125f;
#pragma unroll
            for (int d = 0; d < 32; d += 4) {
                const float4 qv = *reinterpret_cast<const float4*>(&Qs[j][half * 32 + d]);
                const float4 ov = *reinterpret_cast<const float4*>(&Os[j][half * 32 + d]);
                dk[d] = fmaf(ds, qv.x, dk[d]); dk[d + 1] = fmaf(ds, qv.y, dk[d + 1]); dk[d + 2] = fmaf(ds, qv.z, dk[d + 2]); dk[d + 3] = fmaf(ds, qv.w, dk[d + 3]);
                dv[d] = fmaf(p, ov.x, dv[d]); dv[d + 1] = fmaf(p, ov.y, dv[d + 1]); dv[d + 2] = fmaf(p, ov.z, dv[d + 2]); dv[d + 3] = fmaf(p, ov.w, dv[d + 3]);
            }
        }
    }
    if (!active) return;
    const long row = base + (long)kp * sd.pos_stride;
    T* dkr = dqkv + row * ld3 + sd.D + h * HD + half * 32;
    T* dvr = dqkv + row * ld3 + 2 * sd.D + h * HD + half * 32;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
        st4(dkr + d, make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]));
        st4(dvr + d, make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]));
    }
}

// zero `width` elements of every row r = outer*outer_stride + j, j in [0, count)
template <typename T>
__global__ void zero_rows_kernel(T* p, long ld, int n_outer, long outer_stride, int width) {
    const int o = blockIdx.x;
    if (o >= n_outer) return;
    T* r = p + (size_t)o * outer_stride * ld;
    for (int i = threadIdx.x * 4; i < width; i += blockDim.x * 4) st4(r + i, make_float4(0.f, 0.f, 0.f, 0.f));
}

}  // namespace

template <typename T>
static int launch_fwd(hipStream_t st, const SeqDesc& d, const void* qkv, void* out, float* lse) {
    const dim3 grid(d.n_outer * d.n_inner * d.heads, cdiv(d.L, 64));
    hipLaunchKernelGGL(attn_fwd_simple<T>, grid, dim3(64), 0, st, d, (const T*)qkv, (T*)out, lse);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
template <typename T>
static int launch_bwd(hipStream_t st, const SeqDesc& d, long rows, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv) {
    const long n = rows * d.heads;
    hipLaunchKernelGGL(attn_delta_kernel<T>, dim3(cdiv(n, 256)), dim3(256), 0, st, rows, d.heads, d.D, (const T*)out, (const T*)dout, delta);
    TCOW_CHECK_LAUNCH();
    const int items = d.n_outer * d.n_inner;
    hipLaunchKernelGGL(attn_bwd_dq_simple<T>, dim3(items * d.heads, cdiv(d.L, 64)), dim3(64), 0, st, d, (const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv);
    TCOW_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_simple<T>, dim3(items * d.heads, cdiv(d.L, 32)), dim3(64), 0, st, d, (const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}
template <typename T>
static int zero_slot0(hipStream_t st, void* p, long ld, const tcow_attn_shape* s, int width) {
    hipLaunchKernelGGL(zero_rows_kernel<T>, dim3(s->B * s->T), dim3(256), 0, st, (T*)p, ld, s->B * s->T, (long)s->S, width);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

static int check_shape(const tcow_attn_shape* s, const char* who) {
    TCOW_CHECK_ARG(s != nullptr, "%s: null shape", who);
    TCOW_CHECK_ARG(s->B > 0 && s->T > 0 && s->S > 1 && s->heads > 0, "%s: bad shape B=%d T=%d S=%d heads=%d", who, s->B, s->T, s->S, s->heads);
    TCOW_CHECK_ARG(s->D == s->heads * HD, "%s: head_dim must be 64 (D=%d heads=%d)", who, s->D, s->heads);
    TCOW_CHECK_ARG(s->dtype == TCOW_F32 || s->dtype == TCOW_BF16, "%s: unknown dtype %d", who, s->dtype);
    return TCOW_OK;
}

bool tcow_attn_mfma_supported(const SeqDesc& d, bool shared);
bool tcow_attn_mfma_zeroes_slot0(const SeqDesc& d, bool shared, bool backward);
int tcow_attn_mfma_fwd(hipStream_t st, const SeqDesc& d, bool shared, const void* qkv, void* out, float* lse);
long tcow_attn_mfma_bwd_workspace_bytes(const SeqDesc& d);
int tcow_attn_mfma_bwd(hipStream_t st, const SeqDesc& d, bool shared, const void* qkv, const void* out, const void* dout, const float* lse, void* ws,
                       void* dqkv);

int tcow_attn_f32_fwd(hipStream_t st, const SeqDesc& d, const void* qkv, void* out, float* lse);
int tcow_attn_f32_bwd(hipStream_t st, const SeqDesc& d, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv);

// TCOW_ATTN_SIMPLE=1 forces the one-thread-per-query VALU kernels for both storage types (A/B checks of the MFMA kernels)
static bool force_simple() {
    static const bool f = getenv("TCOW_ATTN_SIMPLE") != nullptr && getenv("TCOW_ATTN_SIMPLE")[0] == '1';
    return f;
}
static bool use_mfma(const tcow_attn_shape* s, const SeqDesc& d, int spatial) {
    return s->dtype == TCOW_BF16 && !force_simple() && tcow_attn_mfma_supported(d, spatial != 0);
}

int tcow_attn_fwd_dispatch(hipStream_t st, const tcow_attn_shape* s, int spatial, const void* qkv, void* out, float* lse) {
    const SeqDesc d = spatial ? spatial_desc(s) : temporal_desc(s);
    int rc;
    if ((!spatial || d.offset == 1) && !(use_mfma(s, d, spatial) && tcow_attn_mfma_zeroes_slot0(d, spatial != 0, false))) {   // slot-0 rows are not produced by the kernels: define them as zero
        rc = (s->dtype == TCOW_BF16) ? zero_slot0<bf16_t>(st, out, s->D, s, s->D) : zero_slot0<float>(st, out, s->D, s, s->D);
        if (rc) return rc;
    }
    if (use_mfma(s, d, spatial)) return tcow_attn_mfma_fwd(st, d, spatial != 0, qkv, out, lse);
    if (s->dtype == TCOW_F32 && !force_simple()) return tcow_attn_f32_fwd(st, d, qkv, out, lse);      // exact-f32 MFMA kernels (attention_f32.hip)
    return (s->dtype == TCOW_BF16) ? launch_fwd<bf16_t>(st, d, qkv, out, lse) : launch_fwd<float>(st, d, qkv, out, lse);
}

static long bwd_ws_bytes(const tcow_attn_shape* s) {
    const long simple = (long)s->B * s->T * s->S * s->heads * 4;
    const SeqDesc dt = temporal_desc(s), ds = spatial_desc(s);
    long m = simple;
    if (s->dtype == TCOW_BF16) {
        const long a = tcow_attn_mfma_bwd_workspace_bytes(dt), b = tcow_attn_mfma_bwd_workspace_bytes(ds);
        if (a > m) m = a;
        if (b > m) m = b;
    }
    return m + 256;
}

int tcow_attn_bwd_dispatch(hipStream_t st, const tcow_attn_shape* s, int spatial, const void* qkv, const void* out, const void* dout, const float* lse,
                           void* ws, void* dqkv) {
    const SeqDesc d = spatial ? spatial_desc(s) : temporal_desc(s);
    const long rows = (long)s->B * s->T * s->S;
    int rc;
    if ((!spatial || d.offset == 1) && !(use_mfma(s, d, spatial) && tcow_attn_mfma_zeroes_slot0(d, spatial != 0, true))) {
        rc = (s->dtype == TCOW_BF16) ? zero_slot0<bf16_t>(st, dqkv, 3L * s->D, s, 3 * s->D) : zero_slot0<float>(st, dqkv, 3L * s->D, s, 3 * s->D);
        if (rc) return rc;
    }
    if (use_mfma(s, d, spatial)) return tcow_attn_mfma_bwd(st, d, spatial != 0, qkv, out, dout, lse, ws, dqkv);
    if (s->dtype == TCOW_F32 && !force_simple()) return tcow_attn_f32_bwd(st, d, qkv, out, dout, lse, (float*)ws, dqkv);
    return (s->dtype == TCOW_BF16) ? launch_bwd<bf16_t>(st, d, rows, qkv, out, dout, lse, (float*)ws, dqkv)
                                   : launch_bwd<float>(st, d, rows, qkv, out, dout, lse, (float*)ws, dqkv);
}

extern "C" {

int tcow_attn_temporal_fwd(void* stream, const tcow_attn_shape* s, const void* qkv, void* out, float* lse) {
    int rc = check_shape(s, "tcow_attn_temporal_fwd"); if (rc) return rc;
    TCOW_CHECK_ARG(qkv && out, "tcow_attn_temporal_fwd: null pointer");
    return tcow_attn_fwd_dispatch((hipStream_t)stream, s, 0, qkv, out, lse);
}
int tcow_attn_spatial_fwd(void* stream, const tcow_attn_shape* s, const void* qkv, void* out, float* lse) {
    int rc = check_shape(s, "tcow_attn_spatial_fwd"); if (rc) return rc;
    TCOW_CHECK_ARG(qkv && out, "tcow_attn_spatial_fwd: null pointer");
    return tcow_attn_fwd_dispatch((hipStream_t)stream, s, 1, qkv, out, lse);
}
long tcow_attn_bwd_workspace_bytes(const tcow_attn_shape* s) { return s ? bwd_ws_bytes(s) : 0; }
int tcow_attn_temporal_bwd(void* stream, const tcow_attn_shape* s, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                           void* workspace, long workspace_bytes) {
    int rc = check_shape(s, "tcow_attn_temporal_bwd"); if (rc) return rc;
    TCOW_CHECK_ARG(qkv && out && dout && lse && dqkv && workspace && workspace_bytes >= tcow_attn_bwd_workspace_bytes(s), "tcow_attn_temporal_bwd: bad pointers / workspace");
    return tcow_attn_bwd_dispatch((hipStream_t)stream, s, 0, qkv, out, dout, lse, workspace, dqkv);
}
int tcow_attn_spatial_bwd(void* stream, const tcow_attn_shape* s, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                          void* workspace, long workspace_bytes) {
    int rc = check_shape(s, "tcow_attn_spatial_bwd"); if (rc) return rc;
    TCOW_CHECK_ARG(qkv && out && dout && lse && dqkv && workspace && workspace_bytes >= tcow_attn_bwd_workspace_bytes(s), "tcow_attn_spatial_bwd: bad pointers / workspace");
    return tcow_attn_bwd_dispatch((hipStream_t)stream, s, 1, qkv, out, dout, lse, workspace, dqkv);
}

}  // extern "C"
