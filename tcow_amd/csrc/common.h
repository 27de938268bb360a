// Shared device/host helpers for the tcow_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tcow_hip.h"

// The 16-bit storage format of the library's 16-bit mode (dtype TCOW_BF16 of the C ABI) is a BUILD parameter: the same sources give
//   libtcow_hip.so       bfloat16  (8 significand bits;  precision='bf16', the benchmarked mode of BASELINE configs[1])
//   libtcow_hip_fp16.so  binary16  (11 significand bits; precision='fp16': -DTCOW_FP16) -- same kernels, same speed, 8x smaller rounding
//                        error (mask logits within 1e-3 of the reference); gradients need the static loss scale of engine.py.
// Everything format-specific is in this block: the element type, the MFMA instruction and the conversion helpers below.  (The names
// bf16_t / bf16x8 / pack_bf2 ... are kept for both builds: read them as "the 16-bit type".)
#ifdef TCOW_FP16
typedef _Float16 tcow_h16;
#define TCOW_MFMA_32x32x16_H16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define TCOW_MFMA_32x32x16_H16_ASM "v_mfma_f32_32x32x16_f16"
#else
typedef __bf16 tcow_h16;
#define TCOW_MFMA_32x32x16_H16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define TCOW_MFMA_32x32x16_H16_ASM "v_mfma_f32_32x32x16_bf16"
#endif
typedef __attribute__((ext_vector_type(8))) tcow_h16 bf16x8;
typedef __attribute__((ext_vector_type(4))) tcow_h16 bf16x4;
typedef __attribute__((ext_vector_type(2))) tcow_h16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef uint16_t bf16_t;  // raw storage type of a bf16 element in global memory

#define LDS_PTR(T) __attribute__((address_space(3))) T*
#define GLB_PTR(T) __attribute__((address_space(1))) T*

// ---- error plumbing (thread-local message, negative status codes; never aborts the process)
void tcow_set_error(const char* fmt, ...);
#define TCOW_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            tcow_set_error(__VA_ARGS__);          \
            return TCOW_ERR_INVALID_ARG;          \
        }                                         \
    } while (0)
#define TCOW_CHECK_LAUNCH()                                                        \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            tcow_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return TCOW_ERR_LAUNCH;                                                \
        }                                                                          \
    } while (0)

// ---- 16-bit <-> f32 (round to nearest even)
__device__ __forceinline__ float bf2f(bf16_t h) {
#ifdef TCOW_FP16
    return (float)__builtin_bit_cast(_Float16, h);
#else
    return __builtin_bit_cast(float, ((uint32_t)h) << 16);
#endif
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (tcow_h16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, b);
}
// the two elements of a packed pair
#ifdef TCOW_FP16
__device__ __forceinline__ float bflo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float bfhi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
#else
__device__ __forceinline__ float bflo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
#endif

// element load/store generic over storage type
template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// load / store 4 consecutive elements as float4 (16 B for f32, 8 B for bf16); pointers must be aligned
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 u; u.x = pack_bf2(v.x, v.y); u.y = pack_bf2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}

// 16-byte store / load with a cache policy chosen per call site: NT = non-temporal (`nt` on the instruction) -- for a stream far larger than the
// L2s (4 MB per XCD) that nobody re-reads soon.  A normal store allocates its line in L2 and pushes out lines other workgroups still share (GEMM
// operand tiles); a non-temporal one does not stay.  Measured per stream: profiles/r05_nontemporal.txt (one stream of the path qualifies).
typedef uint32_t tcow_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st16c(void* p, uint4 w) {
    if constexpr (NT) __builtin_nontemporal_store((tcow_u32x4){w.x, w.y, w.z, w.w}, reinterpret_cast<tcow_u32x4*>(p)); else *reinterpret_cast<uint4*>(p) = w;
}
template <bool NT> __device__ __forceinline__ uint4 ld16c(const void* p) {
    if constexpr (NT) { const tcow_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const tcow_u32x4*>(p)); return make_uint4(v[0], v[1], v[2], v[3]); } else return *reinterpret_cast<const uint4*>(p);
}

// ---- math
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---- wave reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// MFMA C/D layout of the 32x32 tiles: register r of lane (l&31, hi=l>>5) holds row crow(r,hi), col l&31.
__host__ __device__ __forceinline__ constexpr int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// raises the dynamic-LDS limit of a kernel once per (device, kernel); thread-safe (api.cpp)
void tcow_ensure_lds(const void* kernel, int bytes);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
