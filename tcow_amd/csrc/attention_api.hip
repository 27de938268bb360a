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

#include <mutex>
#include <vector>

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

// ---- optional HIP-event timing of the four attention entry points on the launch stream (bench.py's `roofline_attention` block: the north-star asks for
// the achieved fraction of the attention roofline next to clips/s).  Classes: 0 spatial forward, 1 spatial backward, 2 temporal forward, 3 temporal backward.
// Algorithmic work of a call: FLOPs = 4 L^2 d per (sequence, head) forward (Q K^T and P V, dense -- a causal mask does not discount it), 2.5 x that
// backward (five products); bytes = q, k, v in + o out forward (4 R D e), q, k, v, dO (+ O for the spatial kernel, which forms delta itself) in + dq, dk, dv
// out backward (8 / 7 R D e), R = rows the sequences cover, e = element size.
namespace {
struct AttnProf {
    std::mutex mu;
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;
    std::vector<double> flops, bytes;
    size_t used = 0, cap = 0;
    bool on = false;
};
AttnProf g_ap;

template <typename F>
int attn_profiled(void* stream, const tcow_attn_shape* s, int spatial, int backward, F&& launch) {
    if (!g_ap.on) return launch();
    std::lock_guard<std::mutex> lock(g_ap.mu);
    if (!g_ap.on || g_ap.used >= g_ap.cap) return launch();
    const size_t i = g_ap.used++;
    const SeqDesc d = spatial ? spatial_desc(s) : temporal_desc(s);
    const double seqs = (double)d.n_outer * d.n_inner, rows = seqs * d.L, es = s->dtype == TCOW_BF16 ? 2.0 : 4.0;
    const double fwd_flops = 4.0 * d.L * (double)d.L * ATT_HD * seqs * d.heads;
    g_ap.cls[i] = (spatial ? 0 : 2) + (backward ? 1 : 0);
    g_ap.flops[i] = backward ? 2.5 * fwd_flops : fwd_flops;
    g_ap.bytes[i] = (backward ? (spatial ? 8.0 : 7.0) : 4.0) * rows * d.D * es;
    (void)hipEventRecord(g_ap.ev[2 * i], (hipStream_t)stream);
    const int rc = launch();
    (void)hipEventRecord(g_ap.ev[2 * i + 1], (hipStream_t)stream);
    return rc;
}
}  // namespace

extern "C" {

int tcow_prof_attn_begin(int max_launches) {
    TCOW_CHECK_ARG(max_launches > 0, "tcow_prof_attn_begin: max_launches must be positive");
    std::lock_guard<std::mutex> lock(g_ap.mu);
    while (g_ap.ev.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { tcow_set_error("tcow_prof_attn_begin: hipEventCreate failed"); return TCOW_ERR_LAUNCH; }
        g_ap.ev.push_back(e);
    }
    g_ap.cls.assign(max_launches, 0); g_ap.flops.assign(max_launches, 0.0); g_ap.bytes.assign(max_launches, 0.0);
    g_ap.used = 0; g_ap.cap = (size_t)max_launches; g_ap.on = true;
    return TCOW_OK;
}

int tcow_prof_attn_end(double* ms4, double* flops4, double* bytes4, long* launches4) {
    TCOW_CHECK_ARG(ms4 && flops4 && bytes4 && launches4, "tcow_prof_attn_end: null output");
    std::lock_guard<std::mutex> lock(g_ap.mu);
    g_ap.on = false;
    for (int c = 0; c < 4; ++c) { ms4[c] = 0.0; flops4[c] = 0.0; bytes4[c] = 0.0; launches4[c] = 0; }
    for (size_t i = 0; i < g_ap.used; ++i) {
        if (hipEventSynchronize(g_ap.ev[2 * i + 1]) != hipSuccess) { tcow_set_error("tcow_prof_attn_end: event sync failed"); return TCOW_ERR_LAUNCH; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_ap.ev[2 * i], g_ap.ev[2 * i + 1]);
        const int c = g_ap.cls[i];
        ms4[c] += t; flops4[c] += g_ap.flops[i]; bytes4[c] += g_ap.bytes[i]; launches4[c] += 1;
    }
    return TCOW_OK;
}

int tcow_attn_temporal_fwd(void* stream, const tcow_attn_shape* s, const void* qkv, void* out, float* lse) {
    int rc = check_shape(s, "tcow_attn_temporal_fwd"); if (rc) return rc;
    TCOW_CHECK_ARG(qkv && out, "tcow_attn_temporal_fwd: null pointer");
    return attn_profiled(stream, s, 0, 0, [&] { return tcow_attn_fwd_dispatch((hipStream_t)stream, s, 0, qkv, out, lse); });
}
int tcow_attn_spatial_fwd(void* stream, const tcow_attn_shape* s, const void* qkv, void* out, float* lse) {
    int rc = check_shape(s, "tcow_attn_spatial_fwd"); if (rc) return rc;
    TCOW_CHECK_ARG(qkv && out, "tcow_attn_spatial_fwd: null pointer");
    return attn_profiled(stream, s, 1, 0, [&] { return tcow_attn_fwd_dispatch((hipStream_t)stream, s, 1, qkv, out, lse); });
}
long tcow_attn_bwd_workspace_bytes(const tcow_attn_shape* s) { return s ? bwd_ws_bytes(s) : 0; }
int tcow_attn_temporal_bwd(void* stream, const tcow_attn_shape* s, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                           void* workspace, long workspace_bytes) {
    int rc = check_shape(s, "tcow_attn_temporal_bwd"); if (rc) return rc;
    TCOW_CHECK_ARG(qkv && out && dout && lse && dqkv && workspace && workspace_bytes >= tcow_attn_bwd_workspace_bytes(s), "tcow_attn_temporal_bwd: bad pointers / workspace");
    return attn_profiled(stream, s, 0, 1, [&] { return tcow_attn_bwd_dispatch((hipStream_t)stream, s, 0, qkv, out, dout, lse, workspace, dqkv); });
}
int tcow_attn_spatial_bwd(void* stream, const tcow_attn_shape* s, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                          void* workspace, long workspace_bytes) {
    int rc = check_shape(s, "tcow_attn_spatial_bwd"); if (rc) return rc;
    TCOW_CHECK_ARG(qkv && out && dout && lse && dqkv && workspace && workspace_bytes >= tcow_attn_bwd_workspace_bytes(s), "tcow_attn_spatial_bwd: bad pointers / workspace");
    return attn_profiled(stream, s, 1, 1, [&] { return tcow_attn_bwd_dispatch((hipStream_t)stream, s, 1, qkv, out, dout, lse, workspace, dqkv); });
}

}  // extern "C"
