// Phase stamps of the persistent spatial-attention forward (tools/attn_fwd_p4.hip: measured, not in the library -- profiles/r05_attn_fwd_p4.txt) at BASELINE configs[1]: B*Qs = 3, T = 30, S = 301,
// 12 heads.  Per wave index (mean over the workgroups): shader cycles in the whole kernel, waiting for its own LDS-DMA pieces (vmcnt), at the
// key-tile barriers, in the item prologues, in the rounds (barrier waits included), in the stores.   build: make build/ubench_p4   run: build/ubench_p4 [S] [frames]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
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

static float h2f(uint16_t u) {
#ifdef TCOW_FP16
    _Float16 h; memcpy(&h, &u, 2); return (float)h;
#else
    uint32_t w = (uint32_t)u << 16; float f; memcpy(&f, &w, 4); return f;
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
        CK(hipMemset(dbg, 0, 256 * 8 * 8 * 8)); CK(hipMemset(out, 0xff, (size_t)M * D * 2)); CK(hipMemset(lse, 0xff, (size_t)M * heads * 4));
        for (int i = 0; i < 3; ++i) tcow_attn_fwd_p4(0, d, qkv, out, lse, waves, 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) tcow_attn_fwd_p4(0, d, qkv, out, lse, waves, 0);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> t(256 * 8 * 8);
        CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
        printf("attn_fwd_p4<%d> S=%d frames=%d: %.1f us per launch\n", waves, S, frames, ms * 1000.f / 20);
        {   // a few (frame, head) pairs against a host double-precision softmax(q k^T / 8) v
            std::vector<uint16_t> ho((size_t)M * D); std::vector<float> hl((size_t)M * heads);
            CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl.data(), lse, hl.size() * 4, hipMemcpyDeviceToHost));
            double eo = 0, el = 0; long bad = 0;
            const int pf[5] = {0, 1, frames / 2, frames - 2, frames - 1}, ph[5] = {0, 7, 3, 11, 5};
            for (int t = 0; t < 5; ++t) {
                const int f = pf[t], hd = ph[t];
                std::vector<double> sc(S);
                for (int q = 0; q < S; ++q) {
                    const uint16_t* qr = &h[((size_t)(f * (long)S + q)) * 3 * D + hd * 64];
                    double mx = -1e300;
                    for (int k = 0; k < S; ++k) {
                        const uint16_t* kr = &h[((size_t)(f * (long)S + k)) * 3 * D + D + hd * 64];
                        double a = 0; for (int d2 = 0; d2 < 64; ++d2) a += (double)h2f(qr[d2]) * (double)h2f(kr[d2]);
                        sc[k] = a * 0.125; if (sc[k] > mx) mx = sc[k];
                    }
                    double l = 0; for (int k = 0; k < S; ++k) { sc[k] = exp(sc[k] - mx); l += sc[k]; }
                    for (int d2 = 0; d2 < 64; ++d2) {
                        double o = 0;
                        for (int k = 0; k < S; ++k) o += sc[k] * (double)h2f(h[((size_t)(f * (long)S + k)) * 3 * D + 2 * D + hd * 64 + d2]);
                        o /= l;
                        const double got = (double)h2f(ho[((size_t)(f * (long)S + q)) * D + hd * 64 + d2]);
                        if (!(got == got) || fabs(got) > 1e30) ++bad; else if (fabs(got - o) > eo) eo = fabs(got - o);
                    }
                    const double lr = mx + log(l), gl = hl[((size_t)(f * (long)S + q)) * heads + hd];
                    if (fabs(gl - lr) > el) el = fabs(gl - lr);
                }
            }
            printf("  check (5 pairs vs host f64): max|d| out %.2e  lse %.2e  non-finite %ld\n", eo, el, bad);
        }
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
