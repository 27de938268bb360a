// C ABI glue of libtcow_hip.so: argument validation, dtype dispatch, thread-local error text.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void tcow_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: remember (device, kernel) pairs so that a
// process driving several GPUs (torch.nn.DataParallel replicas are threads of one process, train.py:222-223) sets it on each.
void tcow_ensure_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.insert(std::make_pair(dev, kernel)).second)
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int tcow_gemm_nt_bf16(hipStream_t stream, const tcow_gemm_args* a);
int tcow_gemm_nt_f32(hipStream_t stream, const tcow_gemm_args* a);
int tcow_gemm_tn_bf16(hipStream_t stream, int M, int N, int K, const bf16_t* dY, long ldy, const bf16_t* X, long ldx, float* slab, int splits, int* nz_out,
                      float* bias_part, int* bias_parts_out);
int tcow_gemm_tn_f32(hipStream_t stream, int M, int N, int K, const float* dY, long ldy, const float* X, long ldx, float* dW, long lddw, int accumulate,
                     float* slab, int splits, const float* bias_part, int bias_nparts, float* bias_out);
int tcow_gemm_nt_x3(hipStream_t stream, const tcow_gemm_args* a);
int tcow_gemm_tn_x3(hipStream_t stream, int M, int N, int K, const float* dY, long ldy, const float* X, long ldx, float* dW, long lddw, int accumulate,
                    float* slab, int splits, const float* bias_part, int bias_nparts, float* bias_out);
int tcow_tn_group_max(void);
bool tcow_tn_group_ok(int n, const tcow_tn_problem* pr);
int tcow_tn_group_slices(int n, const tcow_tn_problem* pr);
int tcow_gemm_tn_bf16_group(hipStream_t stream, int n, const tcow_tn_problem* pr, int nz_req, float* const* slabs, float* const* bias_parts, int* nz_out,
                            int* bias_nparts);
int tcow_tn_splits(int M, int N, int K, int tile_outputs);
int tcow_tn_splits_x3(int M, int N, int K);
int tcow_tn_splits_256(int M, int N, int K);
bool tcow_tn_use_256(int M, int N, int K);
int tcow_launch_slab_reduce(hipStream_t stream, const float* slab, int nz, long slab_stride, long rows, long cols, float* out, long ldo, int accumulate,
                            const float* bias_part, int bias_nparts, int bias_n, float* bias_out);
int tcow_launch_row_reduce(hipStream_t stream, const float* part, int nrows, long ld, int N, float* out, int accumulate);
bool tcow_fold_vec_ok(const float* slab, long slab_stride, long cols, float* out, long ldo);
int tcow_launch_slab_reduce_group(hipStream_t stream, int n, const float* const* slab, int nz, const long* rows, const long* cols, float* const* out, const long* ldo,
                                  const int* accumulate, const float* const* bias_part, const int* bias_nparts, float* const* bias_out);
int tcow_launch_colsum(hipStream_t stream, int dtype, const void* Y, long ldy, int M, int N, float* out, int accumulate, float* part, int max_parts);
int tcow_launch_colsum_partials(hipStream_t stream, int dtype, const void* Y, long ldy, int M, int N, float* part, int max_parts, int* nparts);

extern "C" {

int tcow_version(void) { return TCOW_ABI_VERSION; }
const char* tcow_last_error(void) { return g_err; }

// ---- optional low-overhead HIP-event timing of the dominant kernel (the NT GEMM), on the launch stream
static std::vector<hipEvent_t> g_prof_events;
static std::vector<double> g_prof_flops;
static size_t g_prof_used = 0;
static bool g_prof_on = false;
static std::mutex g_prof_mu;   // the profiling aid is process-wide: one measuring thread at a time, launches from others are serialised here

int tcow_prof_gemm_begin(int max_launches) {
    TCOW_CHECK_ARG(max_launches > 0, "tcow_prof_gemm_begin: max_launches must be positive");
    std::lock_guard<std::mutex> lock(g_prof_mu);
    while (g_prof_events.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { tcow_set_error("tcow_prof_gemm_begin: hipEventCreate failed"); return TCOW_ERR_LAUNCH; }
        g_prof_events.push_back(e);
    }
    g_prof_flops.assign(max_launches, 0.0);
    g_prof_used = 0;
    g_prof_on = true;
    return TCOW_OK;
}

int tcow_prof_gemm_end(double* total_ms, double* total_flops, long* launches) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof_on = false;
    double ms = 0.0, fl = 0.0;
    for (size_t i = 0; i < g_prof_used; ++i) {
        if (hipEventSynchronize(g_prof_events[2 * i + 1]) != hipSuccess) { tcow_set_error("tcow_prof_gemm_end: event sync failed"); return TCOW_ERR_LAUNCH; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_prof_events[2 * i], g_prof_events[2 * i + 1]);
        ms += t; fl += g_prof_flops[i];
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (launches) *launches = (long)g_prof_used;
    return TCOW_OK;
}

static int gemm_nt_dispatch(void* stream, const tcow_gemm_args* a);

int tcow_gemm_nt(void* stream, const tcow_gemm_args* a) {
    if (!g_prof_on) return gemm_nt_dispatch(stream, a);
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof_on || a == nullptr || g_prof_used >= g_prof_flops.size()) return gemm_nt_dispatch(stream, a);
    const size_t i = g_prof_used++;
    g_prof_flops[i] = 2.0 * a->M * (double)a->N * a->K;
    (void)hipEventRecord(g_prof_events[2 * i], (hipStream_t)stream);
    const int rc = gemm_nt_dispatch(stream, a);
    (void)hipEventRecord(g_prof_events[2 * i + 1], (hipStream_t)stream);
    return rc;
}

static int gemm_nt_dispatch(void* stream, const tcow_gemm_args* a) {
    TCOW_CHECK_ARG(a != nullptr, "tcow_gemm_nt: null args");
    TCOW_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "tcow_gemm_nt: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    TCOW_CHECK_ARG(a->A && a->W && a->C, "tcow_gemm_nt: null operand");
    TCOW_CHECK_ARG((a->act != TCOW_ACT_DGELU && a->act != TCOW_ACT_GELU_DSAVE && a->act != TCOW_ACT_MUL_AUX) || a->aux, "tcow_gemm_nt: this activation needs aux");
    TCOW_CHECK_ARG(a->act >= TCOW_ACT_NONE && a->act <= TCOW_ACT_MUL_AUX, "tcow_gemm_nt: unknown activation %d", a->act);
    TCOW_CHECK_ARG(a->bias2 || !a->row_scale2, "tcow_gemm_nt: row_scale2 without bias2");
    if (a->dtype == TCOW_BF16) {
        return tcow_gemm_nt_bf16((hipStream_t)stream, a);
    }
    if (a->dtype == TCOW_F32) return tcow_gemm_nt_f32((hipStream_t)stream, a);
    if (a->dtype == TCOW_F32X3) return tcow_gemm_nt_x3((hipStream_t)stream, a);
    tcow_set_error("tcow_gemm_nt: unknown dtype %d", a->dtype);
    return TCOW_ERR_INVALID_ARG;
}

static const int kColsumParts = 64 * 24 * 2;   // >= nz * tiles_k * 2 partial rows of the fused bias gradient (nz <= 64, K <= 3072)

long tcow_gemm_tn_workspace_bytes(int M, int N, int K) {
    const int s_bf = tcow_tn_splits(M, N, K, 128), s_f = tcow_tn_splits(M, N, K, 64), s_big = tcow_tn_splits_256(M, N, K);
    int s = s_bf > s_f ? s_bf : s_f;
    if (s_big > s) s = s_big;
    if (tcow_tn_splits_x3(M, N, K) > s) s = tcow_tn_splits_x3(M, N, K);
    return ((long)(s + 1) * N * K + (long)kColsumParts * N) * 4 + 256;
}

int tcow_gemm_tn(void* stream, int dtype, int M, int N, int K, const void* dY, long ldy, const void* X, long ldx, float* dW, long lddw,
                 float* bias_grad, int accumulate, void* workspace, long workspace_bytes) {
    TCOW_CHECK_ARG(M > 0 && N > 0 && K > 0 && dY && X && dW && workspace, "tcow_gemm_tn: bad arguments");
    TCOW_CHECK_ARG(workspace_bytes >= tcow_gemm_tn_workspace_bytes(M, N, K), "tcow_gemm_tn: workspace too small (%ld < %ld)", workspace_bytes,
                   tcow_gemm_tn_workspace_bytes(M, N, K));
    float* slab = (float*)workspace;
    float* part = slab + (size_t)(workspace_bytes / 4 - (long)kColsumParts * N - 8);
    int rc;
    if (dtype == TCOW_BF16) {
        const int splits = tcow_tn_use_256(M, N, K) ? tcow_tn_splits_256(M, N, K) : tcow_tn_splits(M, N, K, 128);
        int nz = 0, nparts = 0;
        const bool fuse_bias = bias_grad != nullptr && (long)splits * ((K + 127) / 128) * 2 <= kColsumParts;
        rc = tcow_gemm_tn_bf16((hipStream_t)stream, M, N, K, (const bf16_t*)dY, ldy, (const bf16_t*)X, ldx, slab, splits, &nz, fuse_bias ? part : nullptr, &nparts);
        if (rc) return rc;
        rc = tcow_launch_slab_reduce((hipStream_t)stream, slab, nz, (long)N * K, N, K, dW, lddw, accumulate, fuse_bias ? part : nullptr, nparts, N, bias_grad);
        if (rc) return rc;
        if (fuse_bias) return TCOW_OK;
    } else if (dtype == TCOW_F32 || dtype == TCOW_F32X3) {
        // f32 storage: the bias gradient's column-sum partials first, their fold rides on the weight gradient's slab fold (one launch instead of two per Linear)
        int nparts = 0;
        if (bias_grad) {
            rc = tcow_launch_colsum_partials((hipStream_t)stream, TCOW_F32, dY, ldy, M, N, part, 64, &nparts);      // (5.4 TB/s averaged over the step's shapes; 256 slices measured the same and a dearer fold)
            if (rc) return rc;
        }
        if (dtype == TCOW_F32) rc = tcow_gemm_tn_f32((hipStream_t)stream, M, N, K, (const float*)dY, ldy, (const float*)X, ldx, dW, lddw, accumulate, slab, tcow_tn_splits(M, N, K, 64),
                                                     bias_grad ? part : nullptr, nparts, bias_grad);
        else rc = tcow_gemm_tn_x3((hipStream_t)stream, M, N, K, (const float*)dY, ldy, (const float*)X, ldx, dW, lddw, accumulate, slab, tcow_tn_splits_x3(M, N, K),
                                  bias_grad ? part : nullptr, nparts, bias_grad);
        return rc;
    } else {
        tcow_set_error("tcow_gemm_tn: unknown dtype %d", dtype);
        return TCOW_ERR_INVALID_ARG;
    }
    if (bias_grad) {
        rc = tcow_launch_colsum((hipStream_t)stream, dtype == TCOW_BF16 ? TCOW_BF16 : TCOW_F32, dY, ldy, M, N, bias_grad, accumulate, part, 64);
        if (rc) return rc;
    }
    return TCOW_OK;
}

// ---- grouped weight gradients
static long tn_group_bytes(int n, const tcow_tn_problem* pr, int nz) {
    long b = 256;
    for (int i = 0; i < n; ++i) b += ((long)nz * pr[i].N * pr[i].K + (long)nz * ((pr[i].K + 255) / 256) * 2 * pr[i].N) * 4 + 64;
    return b;
}

int tcow_gemm_tn_group_max(void) { return tcow_tn_group_max(); }

long tcow_gemm_tn_grouped_workspace_bytes(int dtype, int n, const tcow_tn_problem* pr) {
    if (n <= 0 || pr == nullptr) return 0;
    long single = 0;
    for (int i = 0; i < n; ++i) { const long b = tcow_gemm_tn_workspace_bytes(pr[i].M, pr[i].N, pr[i].K); if (b > single) single = b; }
    if (dtype == TCOW_BF16 && tcow_tn_group_ok(n, pr)) { const long g = tn_group_bytes(n, pr, tcow_tn_group_slices(n, pr) + 1); if (g > single) single = g; }
    return single;
}

int tcow_gemm_tn_grouped(void* stream, int dtype, int n, const tcow_tn_problem* pr, void* workspace, long workspace_bytes) {
    TCOW_CHECK_ARG(n > 0 && pr && workspace, "tcow_gemm_tn_grouped: bad arguments");
    TCOW_CHECK_ARG(workspace_bytes >= tcow_gemm_tn_grouped_workspace_bytes(dtype, n, pr), "tcow_gemm_tn_grouped: workspace too small (%ld < %ld)", workspace_bytes,
                   tcow_gemm_tn_grouped_workspace_bytes(dtype, n, pr));
    for (int i = 0; i < n; ++i)
        TCOW_CHECK_ARG(pr[i].M > 0 && pr[i].N > 0 && pr[i].K > 0 && pr[i].dY && pr[i].X && pr[i].dW, "tcow_gemm_tn_grouped: bad problem %d", i);
    if (!(dtype == TCOW_BF16 && tcow_tn_group_ok(n, pr))) {
        for (int i = 0; i < n; ++i) {
            const int rc = tcow_gemm_tn(stream, dtype, pr[i].M, pr[i].N, pr[i].K, pr[i].dY, pr[i].ldy, pr[i].X, pr[i].ldx, pr[i].dW, pr[i].lddw, pr[i].bias_grad,
                                        pr[i].accumulate, workspace, workspace_bytes);
            if (rc) return rc;
        }
        return TCOW_OK;
    }
    const int nz_req = tcow_tn_group_slices(n, pr);
    float* slabs[40]; float* parts[40]; int nparts[40];          // (tcow_tn_group_ok: n <= tcow_tn_group_max() = 40)
    char* w = (char*)workspace;
    for (int i = 0; i < n; ++i) {
        slabs[i] = (float*)w; w += (long)(nz_req + 1) * pr[i].N * pr[i].K * 4;
        parts[i] = pr[i].bias_grad ? (float*)w : nullptr; w += (long)(nz_req + 1) * ((pr[i].K + 255) / 256) * 2 * pr[i].N * 4 + 64;
        // ONE token slice (a group whose tiles fill whole rounds of the chip by themselves: five ViT-B blocks = 765 tiles): nothing to fold, so the
        // kernel writes a dense, non-accumulating weight gradient straight into its destination -- no slab image written, re-read and copied
        // (2 x 177 MB per five-block group); only the bias-gradient partials still go through the fold launch
        if (nz_req == 1 && !pr[i].accumulate && pr[i].lddw == pr[i].K && tcow_fold_vec_ok(pr[i].dW, (long)pr[i].N * pr[i].K, pr[i].K, pr[i].dW, pr[i].lddw))
            slabs[i] = pr[i].dW;
    }
    int nz = 0;
    int rc = tcow_gemm_tn_bf16_group((hipStream_t)stream, n, pr, nz_req, slabs, parts, &nz, nparts);
    if (rc) return rc;
    bool vec = true;
    for (int i = 0; i < n; ++i) vec = vec && tcow_fold_vec_ok(slabs[i], (long)pr[i].N * pr[i].K, pr[i].K, pr[i].dW, pr[i].lddw);
    if (vec) {       // one fold launch for the whole group
        long rows[40], cols[40], ldo[40]; int acc[40]; float* outs[40]; float* bouts[40]; const float* cparts[40]; const float* cslabs[40];
        for (int i = 0; i < n; ++i) {
            rows[i] = pr[i].N; cols[i] = pr[i].K; ldo[i] = pr[i].lddw; acc[i] = pr[i].accumulate; outs[i] = pr[i].dW; bouts[i] = pr[i].bias_grad;
            cparts[i] = parts[i]; cslabs[i] = slabs[i]; if (!parts[i]) nparts[i] = 0;
        }
        return tcow_launch_slab_reduce_group((hipStream_t)stream, n, cslabs, nz, rows, cols, outs, ldo, acc, cparts, nparts, bouts);
    }
    for (int i = 0; i < n; ++i) {
        rc = tcow_launch_slab_reduce((hipStream_t)stream, slabs[i], nz, (long)pr[i].N * pr[i].K, pr[i].N, pr[i].K, pr[i].dW, pr[i].lddw, pr[i].accumulate,
                                     parts[i], parts[i] ? nparts[i] : 0, pr[i].N, pr[i].bias_grad);
        if (rc) return rc;
    }
    return TCOW_OK;
}

}  // extern "C"
