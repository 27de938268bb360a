/*
 * tcow_hip.h -- C ABI of libtcow_hip.so: the MI355X (gfx950) kernels behind the TCOW Seeker hot path.
 *
 * The reference (basilevh/tcow) is pure Python on stock PyTorch ops and has no FFI / operator registry;
 * its boundary for this path is the nn.Module surface of `Seeker` (model/seeker.py:17-25,
 * model/mask_tracker.py:92-142).  This header therefore defines the C entry points that a maintainer would
 * bind from Python (ctypes stub in INTEGRATION.md) to replace, op for op, the stock torch calls of
 *   third_party/TimeSformer/timesformer/models/vit.py   (PatchEmbed, Attention, Mlp, Block)
 *   model/vision_tf.py:68-169                            (embeddings, token re-layout)
 *   model/mask_tracker.py:102-137                        (input concat, mask head, coarsen, flags)
 * Each entry cites the reference lines it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, int status return: 0 = ok, <0 = error (see TCOW_ERR_*);
 *    tcow_last_error() returns a thread-local message.  Nothing aborts the process: the reference's training
 *    loop catches per-step exceptions (train.py:77-84), so failures must surface as Python exceptions.
 *  - All pointers are DEVICE pointers borrowed for the duration of the call; the library never allocates,
 *    frees or synchronises.  `stream` is a hipStream_t passed as void*.  Calls are re-entrant per stream.
 *  - Token rows.  The residual stream is a row-major [rows, D] matrix with
 *        row(b, t, s) = (b*T + t)*S + s,   S = N + 1,  N = (H/P)*(W/P) patches, n = h'*W' + w',
 *    slot s = 0 of every frame holds a replica of the clip's cls token and s = 1 + n holds patch n.  The
 *    reference keeps one cls token per clip and replicates it per frame for spatial attention
 *    (vit.py:180-185); replicating it in storage is mathematically identical (cls gradients of the
 *    replicas sum, see tcow_cls_merge_bwd) and keeps every GEMM a plain [rows, D] x [D, *] product.
 *    Reference token index of (b, t, n): 1 + n*T + t (vision_tf.py:137).
 *  - dtype: TCOW_F32 = everything in float (parity mode, exact-f32 MFMA / FMA), TCOW_BF16 = activations and
 *    GEMM operands in bfloat16 with f32 accumulation; the residual stream, LayerNorm statistics, softmax
 *    statistics, biases and all gradients of parameters stay f32 in both modes.
 */
#ifndef TCOW_HIP_H
#define TCOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define TCOW_OK 0
#define TCOW_ERR_INVALID_ARG (-1)
#define TCOW_ERR_LAUNCH (-2)
#define TCOW_ERR_UNSUPPORTED (-3)

#define TCOW_F32 0
#define TCOW_BF16 1

#define TCOW_ACT_NONE 0
#define TCOW_ACT_GELU 1      /* C = GELU_erf(v); optionally also stores v (pre-activation) to aux         */
#define TCOW_ACT_DGELU 2     /* C = v * GELU_erf'(aux)   (backward of the fc1 activation)                 */

/* ABI version (bumped on any signature change) and last error text of the calling thread. */
int tcow_version(void);
const char* tcow_last_error(void);

/* ------------------------------------------------------------------------------------------------ GEMM
 * C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  v = acc + bias[n];  v *= row_scale[m];  act;  v += resid[m,n].
 * Replaces every nn.Linear on the path: Attention.qkv / .proj (vit.py:74-76,81,111), Block.temporal_fc
 * (vit.py:146,174), Mlp.fc1 / GELU / fc2 (vit.py:50-61), PatchEmbed.proj as a GEMM over flattened patches
 * (vit.py:233-240), tracker_post_linear (mask_tracker.py:113) and their input-gradient GEMMs
 * (dA = dC . W with W passed pre-transposed).  Residual adds x + f(x) (vit.py:176,215-216) and DropPath row
 * scaling (vit_utils.py:139-154) are fused through `resid` / `row_scale`.
 * A, W: `dtype` elements, K-contiguous rows (lda, ldw in elements, multiples of 8; K % 8 == 0).
 * C: `dtype` elements, or f32 when out_f32 != 0.  resid: f32 [M, ldr] or NULL (may alias C when out_f32).
 * aux: `dtype` [M, ldaux]; written for TCOW_ACT_GELU when non-NULL, read for TCOW_ACT_DGELU.
 */
typedef struct {
    int M, N, K;
    int dtype;
    const void* A; long lda;
    const void* W; long ldw;
    void* C; long ldc; int out_f32;
    const float* bias;
    const float* row_scale;
    const float* resid; long ldr;
    int act;
    void* aux; long ldaux;
} tcow_gemm_args;
int tcow_gemm_nt(void* stream, const tcow_gemm_args* args);

/* Weight-gradient GEMM  dW[N,K] (f32) (+)= dY[M,N]^T . X[M,K]   (autograd of nn.Linear.weight).
 * dY, X: `dtype` row-major (ldy, ldx in elements).  The M (token) dimension is split across workgroups;
 * partial tiles go to `workspace` (f32, tcow_gemm_tn_workspace_bytes) and are reduced deterministically.
 * If bias_grad != NULL it also receives (+)= column sums of dY (autograd of nn.Linear.bias).
 * accumulate != 0 adds into dW / bias_grad instead of overwriting. */
long tcow_gemm_tn_workspace_bytes(int M, int N, int K);
int tcow_gemm_tn(void* stream, int dtype, int M, int N, int K, const void* dY, long ldy, const void* X,
                 long ldx, float* dW, long lddw, float* bias_grad, int accumulate, void* workspace,
                 long workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* TCOW_HIP_H */
