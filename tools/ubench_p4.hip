// Phase stamps of the persistent spatial-attention forward (tools/attn_fwd_p4.hip: measured, not in the library -- profiles/r05_attn_fwd_p4.txt) at BASELINE configs[1]: B*Qs = 3, T = 30, S = 301,
// 12 heads.  Per wave index (mean over the workgroups): shader cycles in the whole kernel, waiting for its own LDS-DMA pieces (vmcnt), at the
// key-tile barriers, in the item prologues, in the rounds (barrier waits included), in the stores.   build: make build/ubench_p4   run: build/ubench_p4 [S] [frames]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define P4_STAMPS 1
#include "attn_fwd_p4.hip"

void tcow_set_error(const char*, ...) {}
void tcow_ensure_lds(const void* k, int bytes) { (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f2h(float f) {
#ifdef TCOW_FP16
    _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u;
#else
    uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16);
#endif
}

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 301, frames = argc > 2 ? atoi(argv[2]) : 90, heads = 12, D = heads * 64;
    const long M = (long)frames * S;
    std::vector<uint16_t> h((size_t)M * 3 * D);
    uint32_t st = 12345u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = f2h(((st >> 8) & 0xffff) / 32768.0f - 1.0f); }
    uint16_t *qkv, *out; float* lse; long long* dbg;
    CK(hipMalloc(&qkv, h.size() * 2)); CK(hipMalloc(&out, (size_t)M * D * 2)); CK(hipMalloc(&lse, (size_t)M * heads * 4)); CK(hipMalloc(&dbg, 256 * 8 * 8 * 8));
    CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_p4_dbg), &dbg, sizeof(dbg)));
    SeqDesc d; d.n_outer = frames; d.n_inner = 1; d.outer_stride = S; d.inner_stride = 0; d.offset = 0; d.pos_stride = 1; d.L = S; d.diag = 1 << 28; d.heads = heads; d.D = D;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int waves = 4; waves <= 8; waves += 4) {
        CK(hipMemset(dbg, 0, 256 * 8 * 8 * 8));
        for (int i = 0; i < 3; ++i) tcow_attn_fwd_p4(0, d, qkv, out, lse, waves, 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) tcow_attn_fwd_p4(0, d, qkv, out, lse, waves, 0);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> t(256 * 8 * 8);
        CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
        printf("attn_fwd_p4<%d> S=%d frames=%d: %.1f us per launch\n", waves, S, frames, ms * 1000.f / 20);
        printf("  wave  items  tiles |   total    vmcnt  barrier  prologue   rounds   stores   (mean shader cycles per workgroup; rounds include vmcnt + barrier)\n");
        for (int w = 0; w < waves; ++w) {
            double a[8] = {0}; int n = 0;
            for (int b = 0; b < 256; ++b) { const long long* r = &t[((size_t)b * waves + w) * 8]; if (r[0]) { for (int k = 0; k < 8; ++k) a[k] += (double)r[k]; ++n; } }
            if (!n) continue;
            printf("  %4d  %5.2f  %5.1f | %7.0f  %7.0f  %7.0f  %8.0f  %7.0f  %7.0f\n", w, a[6] / n, a[7] / n, a[0] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n);
        }
    }
    return 0;
}
