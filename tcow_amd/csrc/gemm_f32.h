// Parameter block and element epilogue shared by the two f32-storage GEMM kernels (gemm_f32.hip: exact f32 MFMA, gemm_x3.hip: bf16 x 3).
#pragma once
#include "common.h"

//   C[i,j] = sum_k A(i,k) * B(j,k),  A(i,k) = A[i*sai + k*sak],  B(j,k) = B[j*sbj + k*sbk]   (NT: sak = sbk = 1, TN: sai = sbj = 1)
struct F32Params {
    int M, N, K;
    const float* A; long sai, sak;
    const float* B; long sbj, sbk;
    void* C; long ldc;
    const float* bias; const float* row_scale; const float* resid; long ldr;
    int act; float* aux; long ldaux;
    const float* bias2; const float* row_scale2;      // second bias with its own row scale (folded temporal projection), or NULL
    int kps;          // contraction elements per z-slice (multiple of the kernel's k-slice)
    float* slab;      // if non-NULL: raw partial sums to slab[z][M][N], no epilogue
};

// bias, DropPath row scale, activation (with its side input / output), residual -- the semantics of tcow_gemm_args, one element
__device__ __forceinline__ void f32_epilogue_store(const F32Params& p, int gm, int gn, float x, float bv) {
    x += bv;
    if (p.row_scale) x *= p.row_scale[gm];
    if (p.act == TCOW_ACT_GELU) {
        if (p.aux) p.aux[(size_t)gm * p.ldaux + gn] = x;
        x = gelu_erf(x);
    } else if (p.act == TCOW_ACT_DGELU) {
        x *= gelu_erf_grad(p.aux[(size_t)gm * p.ldaux + gn]);
    } else if (p.act == TCOW_ACT_GELU_DSAVE) {
        p.aux[(size_t)gm * p.ldaux + gn] = gelu_erf_grad(x);
        x = gelu_erf(x);
    } else if (p.act == TCOW_ACT_MUL_AUX) {
        x *= p.aux[(size_t)gm * p.ldaux + gn];
    }
    if (p.bias2) x += (p.row_scale2 ? p.row_scale2[gm] : 1.0f) * p.bias2[gn];
    if (p.resid) x += p.resid[(size_t)gm * p.ldr + gn];
    reinterpret_cast<float*>(p.C)[(size_t)gm * p.ldc + gn] = x;
}

// the same on four consecutive columns gn .. gn+3 of row gm (all pointers / pitches 16-byte compatible: the caller checks)
__device__ __forceinline__ void f32_epilogue_store4(const F32Params& p, int gm, int gn, float4 x, float4 bv) {
    x.x += bv.x; x.y += bv.y; x.z += bv.z; x.w += bv.w;
    if (p.row_scale) { const float rs = p.row_scale[gm]; x.x *= rs; x.y *= rs; x.z *= rs; x.w *= rs; }
    float* ax = p.aux ? p.aux + (size_t)gm * p.ldaux + gn : nullptr;
    if (p.act == TCOW_ACT_GELU) {
        if (ax) st4(ax, x);
        x = make_float4(gelu_erf(x.x), gelu_erf(x.y), gelu_erf(x.z), gelu_erf(x.w));
    } else if (p.act == TCOW_ACT_DGELU) {
        const float4 a = ld4(ax);
        x.x *= gelu_erf_grad(a.x); x.y *= gelu_erf_grad(a.y); x.z *= gelu_erf_grad(a.z); x.w *= gelu_erf_grad(a.w);
    } else if (p.act == TCOW_ACT_GELU_DSAVE) {
        st4(ax, make_float4(gelu_erf_grad(x.x), gelu_erf_grad(x.y), gelu_erf_grad(x.z), gelu_erf_grad(x.w)));
        x = make_float4(gelu_erf(x.x), gelu_erf(x.y), gelu_erf(x.z), gelu_erf(x.w));
    } else if (p.act == TCOW_ACT_MUL_AUX) {
        const float4 a = ld4(ax);
        x.x *= a.x; x.y *= a.y; x.z *= a.z; x.w *= a.w;
    }
    if (p.bias2) { const float r2 = p.row_scale2 ? p.row_scale2[gm] : 1.0f; const float4 b2 = ld4(p.bias2 + gn); x.x += r2 * b2.x; x.y += r2 * b2.y; x.z += r2 * b2.z; x.w += r2 * b2.w; }
    if (p.resid) { const float4 r = ld4(p.resid + (size_t)gm * p.ldr + gn); x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w; }
    st4(reinterpret_cast<float*>(p.C) + (size_t)gm * p.ldc + gn, x);
}
