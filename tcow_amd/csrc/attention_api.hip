// C ABI of the attention entry points (include/tcow_hip.h: tcow_attn_temporal_* / tcow_attn_spatial_*): argument checks, the sequence
// descriptors of the two attention kinds and the dispatch to the kernels -- attention_bf16.hip for 16-bit storage (MFMA flash attention),
// attention_f32.hip for f32 storage (exact-f32 MFMA).  Replaces the core of Attention.forward, softmax(q k^T * d^-0.5 [causal mask]) v
// (vit.py:88-109), for
//   temporal attention: one sequence of T frames per (clip, patch slot s>=1, head)      (vit.py:169-172)
//   spatial attention : one sequence of S (or S-1) tokens per (clip, frame, head)        (vit.py:184-186,206-208)
// directly on the [rows, 3D] output of the qkv GEMM (row order which*D + head*64 + j, vit.py:81-83), writing [rows, D] in the head*64 + j order
// the reference gets from transpose(1,2).reshape (vit.py:109).  (The one-thread-per-query VALU kernels of rounds 1-3 -- TCOW_ATTN_SIMPLE -- are
// gone: both storage types have MFMA kernels for every shape.)
#include <stdlib.h>

#include "attention_common.h"

namespace {

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
static int zero_slot0(hipStream_t st, void* p, long ld, const tcow_attn_shape* s, int width) {
    hipLaunchKernelGGL(zero_rows_kernel<T>, dim3(s->B * s->T), dim3(256), 0, st, (T*)p, ld, s->B * s->T, (long)s->S, width);
    TCOW_CHECK_LAUNCH();
    return TCOW_OK;
}

static int check_shape(const tcow_attn_shape* s, const char* who) {
    TCOW_CHECK_ARG(s != nullptr, "%s: null shape", who);
    TCOW_CHECK_ARG(s->B > 0 && s->T > 0 && s->S > 1 && s->heads > 0, "%s: bad shape B=%d T=%d S=%d heads=%d", who, s->B, s->T, s->S, s->heads);
    TCOW_CHECK_ARG(s->D == s->heads * ATT_HD, "%s: head_dim must be 64 (D=%d heads=%d)", who, s->D, s->heads);
    TCOW_CHECK_ARG(s->dtype == TCOW_F32 || s->dtype == TCOW_BF16 || s->dtype == TCOW_F32X3, "%s: unknown dtype %d", who, s->dtype);
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

// f32 storage, split-bf16 MFMA arithmetic (attention_x3.hip; dtype TCOW_F32X3): sequences of two or more tiles
bool tcow_attn_x3_supported(const SeqDesc& d);
int tcow_attn_x3_fwd(hipStream_t st, const SeqDesc& d, const void* qkv, void* out, float* lse);
int tcow_attn_x3_bwd(hipStream_t st, const SeqDesc& d, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv);

static bool use_mfma(const tcow_attn_shape* s, const SeqDesc& d, int spatial) {
    return s->dtype == TCOW_BF16 && tcow_attn_mfma_supported(d, spatial != 0);
}

int tcow_attn_fwd_dispatch(hipStream_t st, const tcow_attn_shape* s, int spatial, const void* qkv, void* out, float* lse) {
    const SeqDesc d = spatial ? spatial_desc(s) : temporal_desc(s);
    int rc;
    if ((!spatial || d.offset == 1) && !(use_mfma(s, d, spatial) && tcow_attn_mfma_zeroes_slot0(d, spatial != 0, false))) {   // slot-0 rows are not produced by the kernels: define them as zero
        rc = (s->dtype == TCOW_BF16) ? zero_slot0<bf16_t>(st, out, s->D, s, s->D) : zero_slot0<float>(st, out, s->D, s, s->D);
        if (rc) return rc;
    }
    if (use_mfma(s, d, spatial)) return tcow_attn_mfma_fwd(st, d, spatial != 0, qkv, out, lse);      // 16-bit storage: attention_bf16.hip
    if (s->dtype == TCOW_F32X3 && tcow_attn_x3_supported(d)) return tcow_attn_x3_fwd(st, d, qkv, out, lse);   // f32 storage, bf16 x 3 split products (attention_x3.hip)
    return tcow_attn_f32_fwd(st, d, qkv, out, lse);                                                   // f32 storage: exact-f32 MFMA kernels (attention_f32.hip)
}

static long bwd_ws_bytes(const tcow_attn_shape* s) {
    const long delta = (long)s->B * s->T * s->S * s->heads * 4;       // f32 modes: the delta table
    const SeqDesc dt = temporal_desc(s), ds = spatial_desc(s);
    long m = delta;
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
    int rc;
    if ((!spatial || d.offset == 1) && !(use_mfma(s, d, spatial) && tcow_attn_mfma_zeroes_slot0(d, spatial != 0, true))) {
        rc = (s->dtype == TCOW_BF16) ? zero_slot0<bf16_t>(st, dqkv, 3L * s->D, s, 3 * s->D) : zero_slot0<float>(st, dqkv, 3L * s->D, s, 3 * s->D);
        if (rc) return rc;
    }
    if (use_mfma(s, d, spatial)) return tcow_attn_mfma_bwd(st, d, spatial != 0, qkv, out, dout, lse, ws, dqkv);
    if (s->dtype == TCOW_F32X3 && tcow_attn_x3_supported(d)) return tcow_attn_x3_bwd(st, d, qkv, out, dout, lse, (float*)ws, dqkv);
    return tcow_attn_f32_bwd(st, d, qkv, out, dout, lse, (float*)ws, dqkv);
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
