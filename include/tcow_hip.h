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
 *    statistics, biases and all gradients of parameters stay f32 in both modes.  TCOW_F32X3 is accepted by the two GEMM
 *    entry points only (tcow_gemm_nt, tcow_gemm_tn): storage, arguments and epilogues of TCOW_F32, products computed as
 *    three bf16 MFMAs on hi / lo splits of the f32 operands (~1e-5 relative per product; gemm_x3.hip) -- every other
 *    entry point of the module's precision='bf16x3' mode is called with TCOW_F32.
 *  - Two builds of this ABI exist: libtcow_hip.so stores the 16-bit mode (TCOW_BF16) as bfloat16; libtcow_hip_fp16.so is the same
 *    source compiled with -DTCOW_FP16 and stores it as IEEE binary16 (csrc/common.h) -- identical entry points, argument meaning and
 *    speed, 8x smaller rounding error, binary16's range (the host scales gradients by a power of two, see tcow_amd/engine.py).
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
#define TCOW_F32X3 2   /* GEMM entry points only: f32 storage, bf16 x 3 split products */

#define TCOW_ACT_NONE 0
#define TCOW_ACT_GELU 1      /* C = GELU_erf(v); optionally also stores v (pre-activation) to aux         */
#define TCOW_ACT_DGELU 2     /* C = v * GELU_erf'(aux)   (backward of the fc1 activation)                 */
#define TCOW_ACT_GELU_DSAVE 3 /* C = GELU_erf(v) and aux = GELU_erf'(v): the erf / exp are shared, so the      */
#define TCOW_ACT_MUL_AUX 4   /* backward is C = v * aux -- one multiply instead of a second erf + exp per element */

/* ABI version (bumped on any signature change) and last error text of the calling thread.  A host compares tcow_version() with the TCOW_ABI_VERSION it
 * was built against before the first call (tcow_amd/_lib.py does, for both builds of the library). */
#define TCOW_ABI_VERSION 10
int tcow_version(void);
const char* tcow_last_error(void);

/* ------------------------------------------------------------------------------------------------ GEMM
 * C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  v = acc + bias[n];  v *= row_scale[m];  act;  v += row_scale2[m] * bias2[n];  v += resid[m,n].
 * Replaces every nn.Linear on the path: Attention.qkv / .proj (vit.py:74-76,81,111), Block.temporal_fc
 * (vit.py:146,174), Mlp.fc1 / GELU / fc2 (vit.py:50-61), PatchEmbed.proj as a GEMM over flattened patches
 * (vit.py:233-240), tracker_post_linear (mask_tracker.py:113) and their input-gradient GEMMs
 * (dA = dC . W with W passed pre-transposed).  Residual adds x + f(x) (vit.py:176,215-216) and DropPath row
 * scaling (vit_utils.py:139-154) are fused through `resid` / `row_scale`.  bias2 / row_scale2 (both or neither; NULL row_scale2 = 1)
 * serve the FOLDED temporal projection: temporal_attn.proj -> DropPath -> temporal_fc (vit.py:111,172-176) are two Linear layers with
 * only a per-row scale s between them, so R1 = R0 + mask0 * (s * (O W'^T + b') + b_fc) with W' = W_fc W_proj, b' = W_fc b_proj:
 * one GEMM with bias = b', row_scale = mask0 * s, bias2 = b_fc, row_scale2 = mask0 (engine.py).
 * A, W: `dtype` elements, K-contiguous rows (lda, ldw in elements, multiples of 8; K % 8 == 0).
 * C: `dtype` elements, or f32 when out_f32 != 0.  resid: f32 [M, ldr] or NULL (may alias C when out_f32).
 * aux: `dtype` [M, ldaux]; written for TCOW_ACT_GELU (when non-NULL) and TCOW_ACT_GELU_DSAVE, read for TCOW_ACT_DGELU
 * and TCOW_ACT_MUL_AUX.  In bf16 mode erf is evaluated with Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below
 * the bf16 rounding of the result); the f32 mode uses erff.
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
    int tile;              /* 0 = choose by shape; 128 / 256 / 320 = force that bf16 NT tile kernel (tests and A/B runs); 160 = the 160 x 256 kernel
                              that runs two workgroups per CU (gemm_nt_c2.hip; K % 128 == 0) */
    const float* bias2;        /* second bias [N] with its own row scale (may be NULL) */
    const float* row_scale2;   /* [M] or NULL */
} tcow_gemm_args;
int tcow_gemm_nt(void* stream, const tcow_gemm_args* args);

/* Measurement aid for bench.py: between _begin and _end every tcow_gemm_nt launch is bracketed by HIP events recorded
 * on its own stream; _end synchronises on them and returns the summed event time (ms), the summed 2*M*N*K and the
 * number of launches (at most max_launches are recorded). */
int tcow_prof_gemm_begin(int max_launches);
int tcow_prof_gemm_end(double* total_ms, double* total_flops, long* launches);
/* The same for the four attention entry points (ABI 10): per class c = 0 spatial forward, 1 spatial backward, 2 temporal forward, 3 temporal backward the summed
 * HIP-event time (ms), algorithmic FLOPs (4 L^2 d per sequence and head forward, 2.5 x backward) and bytes (4 / 8 / 7 x rows x D x element size) and the
 * launch count of the calls between begin and end: bench.py's `roofline_attention`. */
int tcow_prof_attn_begin(int max_launches);
int tcow_prof_attn_end(double* ms4, double* flops4, double* bytes4, long* launches4);

/* Weight-gradient GEMM  dW[N,K] (f32) (+)= dY[M,N]^T . X[M,K]   (autograd of nn.Linear.weight).
 * dY, X: `dtype` row-major (ldy, ldx in elements).  The M (token) dimension is split across workgroups;
 * partial tiles go to `workspace` (f32, tcow_gemm_tn_workspace_bytes) and are reduced deterministically.
 * If bias_grad != NULL it also receives (+)= column sums of dY (autograd of nn.Linear.bias).
 * accumulate != 0 adds into dW / bias_grad instead of overwriting. */
long tcow_gemm_tn_workspace_bytes(int M, int N, int K);
int tcow_gemm_tn(void* stream, int dtype, int M, int N, int K, const void* dY, long ldy, const void* X,
                 long ldx, float* dW, long lddw, float* bias_grad, int accumulate, void* workspace,
                 long workspace_bytes);

/* The same for several Linear layers at once (the weight gradients of one transformer block, train.py:98's backward through
 * vit.py:50-61,74-76,146): in bf16 mode, problems that share M run as ONE grid with a common, much smaller number of token slices
 * (see gemm_bf16.hip: gemm_tn_bf16_256_group_kernel; the more tiles a group has, the fewer slices fill the chip: five for one ViT-B block,
 * two for four blocks -- tcow_tn_group_slices); otherwise the call is a loop over tcow_gemm_tn.  Results are identical in
 * meaning to n calls of tcow_gemm_tn (f32 summation order over the token slices differs). */
typedef struct {
    int M, N, K;
    const void* dY; long ldy;
    const void* X; long ldx;
    float* dW; long lddw;
    float* bias_grad;      /* may be NULL */
    int accumulate;
} tcow_tn_problem;
long tcow_gemm_tn_grouped_workspace_bytes(int dtype, int n, const tcow_tn_problem* problems);
int tcow_gemm_tn_grouped(void* stream, int dtype, int n, const tcow_tn_problem* problems, void* workspace, long workspace_bytes);
/* most problems that run as one grid (more are accepted and run one by one): 40 = the weight gradients of five divided space-time blocks */
int tcow_gemm_tn_group_max(void);

/* Small f32 products C[M,N] = A B for the folded temporal projection (W' = W_fc W_proj after every optimizer step;
 * dW_fc = dW' W_proj^T + db' b_proj^T, dW_proj = W_fc^T dW', db_proj = W_fc^T db' in the backward): up to 24 problems per launch on the split-bf16 arithmetic of TCOW_F32X3
 * (three bf16 MFMAs per product, ~5e-6 relative).  A(i,k) = A[i*sai + k*sak], B(j,k) = B[j*sbj + k*sbk] -- one stride of each operand
 * should be 1 (16-byte aligned, extents multiples of 4) for the vector loaders; all problems of a call should share the operand forms. */
typedef struct {
    int M, N, K;
    const float* A; long sai, sak;
    const float* B; long sbj, sbk;
    float* C; long ldc;
    int accumulate;        /* != 0: C += A B */
} tcow_sgemm;
int tcow_sgemm_x3_batched(void* stream, int n, const tcow_sgemm* problems);

/* ------------------------------------------------------------------------------------------- LayerNorm
 * y = (x - mean) / sqrt(var + eps) * gamma + beta over the last dimension of the f32 residual stream
 * (nn.LayerNorm(D, eps=1e-6): vit.py:135 norm1, :142 temporal_norm1, :150 norm2, :283 norm; eps vit.py:428).
 * y has `dtype` elements; mean / rstd (f32 [rows], may be NULL in inference) are saved for the backward.
 * Backward: dx = dres + dLN(dy) (dres = gradient arriving through the residual connection, may be NULL),
 * dgamma / dbeta (+)= column sums (both NULL to skip; otherwise workspace of tcow_layernorm_bwd_workspace_bytes).
 * dx_cast (may be NULL) additionally receives dtype(dx * cast_row_scale[row]) (scale NULL = 1): the operand of the
 * input-gradient GEMM that consumes this gradient next, without a separate tcow_scale_cast pass.
 * colsum_out (may be NULL; needs dgamma / dbeta) (+)= sum_rows colsum_row_scale[row] * dx[row] (scale NULL = 1) in f32: the bias gradient
 * of the Linear layer whose output fed this norm's residual stream under a row mask (temporal_fc.bias, vit.py:146,174-176). */
int tcow_layernorm_fwd(void* stream, int dtype, int rows, int D, const float* x, long ldx, const float* gamma,
                       const float* beta, float eps, void* y, long ldy, float* mean, float* rstd);
long tcow_layernorm_bwd_workspace_bytes(int D);
int tcow_layernorm_bwd(void* stream, int dtype, int rows, int D, const void* dy, long lddy, const float* x, long ldx,
                       const float* mean, const float* rstd, const float* gamma, const float* dres, long lddres,
                       float* dx, long lddx, float* dgamma, float* dbeta, int accumulate, void* workspace,
                       long workspace_bytes, void* dx_cast, long lddx_cast, const float* cast_row_scale,
                       const float* colsum_row_scale, float* colsum_out);
/* Deferred parameter-gradient fold: with bit 1 of `accumulate` set, tcow_layernorm_bwd leaves its partial table
 * [tcow_layernorm_bwd_parts(rows, colsum_out != NULL)][2 or 3][D] f32 in `workspace` (the caller keeps one workspace per pending call) and
 * tcow_layernorm_fold later folds up to 16 such tables in ONE launch (the LayerNorm calls of a group of transformer blocks: three per
 * block, each fold a 5 us latency-bound launch on its own). */
typedef struct {
    const float* part; int parts, D;
    float* dgamma; float* dbeta; float* colsum_out;      /* colsum_out NULL: the table has two sections */
    int accumulate;
} tcow_ln_fold_job;
int tcow_layernorm_bwd_parts(int rows, int with_colsum);
int tcow_layernorm_fold(void* stream, int n, const tcow_ln_fold_job* jobs);

/* ------------------------------------------------------------------------------------------- attention
 * softmax(q k^T / 8 [mask]) v per head (head_dim 64) straight on the qkv GEMM output [rows, 3D]
 * (column order which*D + head*64 + j, vit.py:81-83) -> out [rows, D] (head*64 + j, vit.py:109).
 *   temporal: one sequence of T frames per (clip b, slot s >= 1, head) -- Block.forward vit.py:169-172 with the
 *             causal mask of Attention.forward vit.py:93-99: causal in {1,2}: key_t <= query_t; causal >= 3:
 *             key_t <= query_t + causal - 2; causal <= 0: none.  Rows of slot 0 are written as zeros.
 *   spatial : one sequence per (clip, frame, head) over slots 0..S-1 when causal in {0,1} (cls takes part,
 *             vit.py:180-186) or 1..S-1 otherwise (vit.py:202-208; slot-0 rows written as zeros).  No mask.
 * lse: f32 [rows, heads] log-sum-exp of the scaled scores (NULL in inference), consumed by the backward, which
 * recomputes the probabilities instead of storing the (T,T) / (S,S) score matrices the reference materialises.
 * Backward writes dqkv [rows, 3D] (rows of unused slots zeroed). workspace: tcow_attn_bwd_workspace_bytes. */
typedef struct {
    int B, T, S;        /* clips, frames, slots per frame (patches + 1) */
    int D, heads;       /* D == heads * 64 */
    int causal;         /* the reference's causal_attention value */
    int dtype;
} tcow_attn_shape;
int tcow_attn_temporal_fwd(void* stream, const tcow_attn_shape* shape, const void* qkv, void* out, float* lse);
int tcow_attn_spatial_fwd(void* stream, const tcow_attn_shape* shape, const void* qkv, void* out, float* lse);
long tcow_attn_bwd_workspace_bytes(const tcow_attn_shape* shape);
int tcow_attn_temporal_bwd(void* stream, const tcow_attn_shape* shape, const void* qkv, const void* out,
                           const void* dout, const float* lse, void* dqkv, void* workspace, long workspace_bytes);
int tcow_attn_spatial_bwd(void* stream, const tcow_attn_shape* shape, const void* qkv, const void* out,
                          const void* dout, const float* lse, void* dqkv, void* workspace, long workspace_bytes);

/* ------------------------------------------------------------------------------------------- token glue
 * tcow_im2col: cat([rgb (B,3,T,H,W), query (B,1,T,H,W)]) (mask_tracker.py:107-108), optional (rgb-0.45)/0.225
 *   (vision_tf.py:81-89), flattened per patch in Conv2d weight order c*P*P + py*P + px (vit.py:233-240) into
 *   rows (b,t,1+n); slot-0 rows are zeros.  out: `dtype` [B*T*S, 4*P*P].
 * tcow_embed_fwd: in place on the f32 patch-embed output: slot 0 <- cls_token + pos_embed[0]; slot s >= 1 <-
 *   x + pos_embed[s] + time_embed[t] (vision_tf.py:99-138; dropouts are p = 0).  pos [S,D], time [T,D].
 * tcow_embed_bwd: dpos[s] (+)= sum_{b,t} g[b,t,s];  dtime[t] (+)= sum_{b,s>=1} g[b,t,s]
 *   (d cls_token = dpos[0]; d patch_embed.bias = sum_t dtime[t]).
 * tcow_cls_merge: after the spatial projection, make every frame's slot 0 equal to the clip's single cls token:
 *   mode 1 = frame 0's row (causal_attention == 1), mode 0 = mean over frames (== 0) (vit.py:189-198,215).
 *   backward != 0 applies the adjoint to a gradient buffer in place. */
int tcow_im2col(void* stream, int dtype, int B, int T, int H, int W, int P, const float* rgb, const float* query,
                int pretrained_norm, void* out);
/* One channel group of the same gather: src (B,C,T,H,W) f32 -> out `dtype` [B*T*S, C*P*P] (normalise != 0: (x-0.45)/0.225).
 * The Qs queries of a clip share its rgb frames (pipeline.py:134-158 re-feeds the same seeker_input Qs times): the patch
 * embedding then splits into ONE rgb GEMM per clip (C = 3) plus a K = P*P mask-channel GEMM per query (C = 1). */
int tcow_im2col_channels(void* stream, int dtype, int B, int T, int H, int W, int P, int C, const float* src,
                         int normalise, void* out);
/* Input-pipeline gather (data/augs.py:150-203: frame sub-sampling, centre / random crop, horizontal flip, NEAREST resize composed into
 * index tables by tcow_amd/augs.py): out[c,t,y,x] = src[c, frame_idx[t], src_y[y], src_x[x]]; src (C,Tv,H,W), out (C,Tc,h,w), elements of
 * elem_bytes = 1 (uint8 segmentation / masks) or 4 (f32 frames); index tables are device int32 arrays with in-range entries. */
int tcow_gather_frames(void* stream, int elem_bytes, int C, int Tv, int H, int W, int Tc, int h, int w, const void* src,
                       const int* frame_idx, const int* src_y, const int* src_x, void* out);
/* tcow_resize_aa: the float modalities of the same pipeline (rgb / depth / coordinates: augs.py:40-43,199-201 -- torchvision Resize(BILINEAR,
 *   antialias=True) == torch.nn.functional.interpolate(mode='bilinear', antialias=True, align_corners=False) on tensors):
 *   out[c,t,Y,X] = sum_j wy[Y][j] * sum_i wx[X][i] * src[c, frame_idx[t], ys[ymin[Y]+j], xs[xmin[X]+i]].  ys [hc] / xs [wc] = source row /
 *   column of each row / column of the cropped (and flipped) image; (ymin, ysize, wy [oh, ky]) and (xmin, xsize, wx [ow, kx]) = first tap,
 *   tap count and normalised triangle-filter weights per output row / column (tcow_amd/augs.py::aa_tables restates ATen's tables). */
int tcow_resize_aa(void* stream, int C, int Tv, int H, int W, int Tc, int hc, int wc, int oh, int ow, const float* src, const int* frame_idx,
                   const int* ys, const int* xs, const int* ymin, const int* ysize, const float* wy, int ky, const int* xmin, const int* xsize,
                   const float* wx, int kx, float* out);

/* Photometric augmentation of the rgb modality (data/augs.py:33-35,175-181: torchvision ColorJitter + GaussianBlur(5) + Grayscale(3)), fused
 * (photometric.hip): src (3, Tv, H, W) f32 in [0, 1]; the Tc frames frame_idx[t] (device int32), centre-crop rectangle (y0, x0, h, w), are
 * jittered with the n_ops adjustments ops[] (host array, application order: 0 brightness, 1 contrast, 2 saturation, 3 hue; one factor each),
 * blurred with the five normalised taps (host array; blur = 0: skipped) with reflect padding, optionally folded to 3 equal grey channels, and
 * written to out (3, Tc, h, w).  workspace: tcow_photometric_workspace_bytes(Tc) bytes (device), needed when contrast is among the ops. */
long tcow_photometric_workspace_bytes(int Tc);
int tcow_photometric(void* stream, int Tv, int H, int W, int Tc, int y0, int x0, int h, int w, const float* src, const int* frame_idx, int n_ops,
                     const int* ops, float brightness, float contrast, float saturation, float hue, int blur, const float* taps, int gray,
                     float* workspace, long workspace_bytes, float* out);

int tcow_embed_fwd(void* stream, int B, int T, int S, int D, float* x, const float* cls, const float* pos,
                   const float* time_embed);
int tcow_embed_bwd(void* stream, int B, int T, int S, int D, const float* g, float* dpos, float* dtime,
                   int accumulate);
int tcow_cls_merge(void* stream, int B, int T, int S, int D, float* x, int mode, int backward);
/* the adjoint (backward != 0) that also refreshes the `dtype` operand copy of the gradient for the slot-0 rows it rewrites:
 * cast_out[row] = dtype(cast_scale[row] * x[row]) (scale NULL = 1) -- the copy the LayerNorm backward wrote for all rows (dx_cast). */
int tcow_cls_merge_bwd_cast(void* stream, int dtype, int B, int T, int S, int D, float* x, int mode, void* cast_out, long ldc,
                            const float* cast_scale);

/* ------------------------------------------------------------------------------------------- mask head
 * tcow_unpatchify_pool_fwd: head output pm [B*T*S, C*P*P] (`dtype`, (c,py,px) order, mask_tracker.py:113-115)
 *   -> pooled f32 [B*T, C, H/st, W/st] = avg_pool2d(st) of the un-patchified frame (mask_tracker.py:118-122).
 * tcow_upsample_fwd: pooled -> out f32 (B, C, T, H, W): bilinear (align_corners=True) or nearest x st
 *   (mask_tracker.py:124-132).  h, w are the pooled sizes.
 * tcow_flags_fwd: flags[bt, f] = Wf . mean_{s>=1} x[bt, s] + bf  (mask_tracker.py:135-137), x f32 [B*T*S, D]. */
int tcow_unpatchify_pool_fwd(void* stream, int dtype, int BT, int Hp, int Wp, int P, int C, int st, const void* pm,
                             float* pooled);
int tcow_unpatchify_pool_bwd(void* stream, int dtype, int BT, int Hp, int Wp, int P, int C, int st,
                             const float* dpooled, void* dpm);
int tcow_upsample_fwd(void* stream, int B, int T, int C, int h, int w, int st, int bilinear, const float* pooled,
                      float* out);
int tcow_upsample_bwd(void* stream, int B, int T, int C, int h, int w, int st, int bilinear, const float* dout,
                      float* dpooled);
/* tcow_upsample_bwd for the bilinear stride-4 head (h, w > 4) that also leaves max |dout| in amax_bits[0 .. TCOW_AMAX_SLOTS) as float bit patterns:
 * max |dout| = the maximum of the slots (atomic maxima, one per workgroup, spread over the slots; the caller zeroes all of them) -- the statistic the
 * binary16 mode's power-of-two loss scale is chosen from, taken in the pass that reads dout anyway.  n_slots = the words the caller's buffer holds and must
 * equal TCOW_AMAX_SLOTS (anything else is TCOW_ERR_INVALID_ARG: ABI 8 wrote ONE word, ABI 9 sixty-four behind an unchanged signature -- a host built
 * against the older header would have been overrun silently; ABI 10 makes the size part of the call). */
#define TCOW_AMAX_SLOTS 64
int tcow_upsample_bwd_amax(void* stream, int B, int T, int C, int h, int w, int st, const float* dout, float* dpooled,
                           unsigned* amax_bits, int n_slots);
int tcow_flags_fwd(void* stream, int BT, int S, int D, int F, const float* x, const float* Wf, const float* bf,
                   float* flags);

/* ------------------------------------------------------------------------------------------- casts
 * tcow_scale_cast: dst[r,:] = dtype(src[r,:] * row_scale[r]) (row_scale may be NULL): f32 residual-stream
 *   values / gradients -> GEMM operands (fuses the DropPath scale, vit_utils.py:139-154, in the backward).
 * tcow_cast_transpose: W f32 [N,K] -> Wc `dtype` [N,K] and/or Wt `dtype` [K,N] (either may be NULL): the
 *   per-step operand copies of the f32 master weights (Wt feeds the input-gradient GEMMs). */
/* x[0 .. n) *= *scale, decided on the device: nothing is read or written when *scale == 1 (the upstream gradient of a scalar loss).  (ABI 9) */
int tcow_scale_unless_one(void* stream, float* x, long n, const float* scale);
int tcow_scale_cast(void* stream, int dtype, long rows, int D, const float* src, long ld_src, const float* row_scale,
                    void* dst, long ld_dst);
/* DropPath row scales of all `depth` blocks of a divided space-time step (vit_utils.py:139-154 at vit.py:172-186) from uniform draws
 * u [depth][B*(S-1) + B*T + B] (temporal: per clip and spatial position; spatial: per clip and frame; MLP: per clip) and keep_p [depth]:
 * scale = floor(u + keep_p) / keep_p.  out f32 [4][depth][B*T*S]: temporal (1 on slot 0) | spatial | MLP | temporal x mask0. */
int tcow_droppath_rows(void* stream, int depth, int B, int T, int S, const float* u, const float* keep_p, const float* mask0, float* out);
int tcow_cast_transpose(void* stream, int dtype, int N, int K, const float* W, void* Wc, void* Wt);
/* The same for many weights in one launch: `table` is a device array of n records {const float* W; void* Wc; void* Wt; int N;
 * int K; int tile_begin; int tile} (tcow_cast_desc_bytes() bytes each, 8-byte aligned), tile = tile edge E of the record: 64 (only
 * if N % 64 == 0 and K % 64 == 0: the vectorised path) or anything else = 32; tile_begin = running sum of ceil(N/E) * ceil(K/E)
 * over the preceding records, total_tiles = that sum over all records.  Wc / Wt may be NULL per record. */
long tcow_cast_desc_bytes(void);
int tcow_cast_transpose_batched(void* stream, int dtype, const void* table, int n, int total_tiles);

/* ------------------------------------------------------------------------------------------- optimizer step
 * clip_grad_norm_(params, max_norm) followed by AdamW.step() (train.py:99-102; torch defaults) in three
 * launches over a device-resident chunk table: records {float* p; const float* g; float* m; float* v; long n} of
 * tcow_adamw_chunk_bytes() bytes each, n <= 65536.  scratch: f32 [n_chunks + 2]; afterwards scratch[n_chunks] is the clip
 * coefficient and scratch[n_chunks + 1] the total gradient norm.  max_norm <= 0 disables clipping.  step counts from 1. */
long tcow_adamw_chunk_bytes(void);
int tcow_adamw_clip_step(void* stream, const void* chunks, int n_chunks, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float max_norm, float* scratch);
/* The same on gradients that are still multiplied by a loss scale (binary16 training): *grad_inv_scale (device scalar, a power of two; NULL = 1) is
 * the inverse scale.  The norm and the clip coefficient are those of the true gradients, the update multiplies the stored gradients by
 * coefficient x inverse scale -- exact, and the separate unscaling pass over all gradients is gone.  (ABI 9) */
int tcow_adamw_clip_step_scaled(void* stream, const void* chunks, int n_chunks, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, float max_norm, float* scratch, const float* grad_inv_scale);
/* The same step with the 16-bit operand copies of the GEMM weights written by the update itself (ABI 10): chunks[0 .. n_chunks) cover every parameter (gradient
 * norm); flat_chunks[0 .. n_flat) the ones updated flat, the others as 64 x 64 tiles -- tiles[0 .. n_tiles) = records of tcow_adamw_tile_bytes() bytes
 * {float* p; const float* g; float* m; float* v; void* wc; void* wt; int K, N;} with the pointers at the tile's origin in p / g / m / v / wc ([N, K], row
 * stride K) and wt ([K, N], row stride N); wc / wt receive the updated values in this library's 16-bit format.  Replaces tcow_cast_transpose_batched for
 * those weights (one read of 4 bytes per parameter less per step). */
long tcow_adamw_tile_bytes(void);
int tcow_adamw_clip_step_cast(void* stream, const void* chunks, int n_chunks, const void* flat_chunks, int n_flat, const void* tiles, int n_tiles, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float max_norm, float* scratch, const float* grad_inv_scale);

/* ------------------------------------------------------------------------------------------- mask objective (caller row L)
 * One channel of the TCOW mask loss, forward value and d(loss)/d(logits) together (loss.py:164-225, with
 * bootstrap_warmup_loss loss.py:13-17 and tversky_loss loss.py:20-32):
 *   loss = [aot_loss * (boot + jac) / 2 + (1 - aot_loss) * mean(w * bce)] * sqrt(n_selected / n)
 * over the frames that carry any non-zero weight; 0 when none does or mean(w) < 1e-4.  weighted_aot is the reference's
 * apply_weights_for_aot (true for the occlusion / containment channels: boot over w * bce and jac := boot).
 * Pixel (f, i) of frame f (0 <= f < n_frames, 0 <= i < frame_len) lives at
 *   base + (f / frames_per_seq) * seq_stride + (f % frames_per_seq) * frame_len + i      (elements),
 * which addresses one channel of a (B*Q, 3, T, H, W) tensor in place.  weight = pixel_w[f * frame_len + i] * frame_w[f]
 * (either pointer may be NULL = 1).  dlogits (may be NULL) receives loss_weight * d(loss)/d(logits), same addressing;
 * *loss receives the channel loss, *total (may be NULL) is incremented by loss_weight * loss.  Every decision the
 * reference takes on the host is taken on the device: the call never synchronises. */
typedef struct {
    long n_frames, frame_len, frames_per_seq;
    const float* logits;  long logits_seq_stride;
    const float* target;  long target_seq_stride;
    const float* pixel_w;
    const float* frame_w;
    int weighted_aot;
    float aot_loss;                 /* args.aot_loss (args.py:196) */
    double topk_frac;               /* loss.py:200: min(max(1 - progress * 8.5, 0.15), 1) */
    float loss_weight;
    float* loss;
    float* total;
    float* dlogits;  long dlogits_seq_stride;
    void* ws;  size_t ws_bytes;
    int focal;                      /* train_args.focal_loss (loss.py:49-51): sigmoid focal loss (alpha 0.25, gamma 2) in place of the BCE term  (ABI 10) */
} tcow_mask_loss_args;
/* Workspace of one channel: tcow_mask_loss_workspace_bytes = the most any job of this geometry needs; tcow_mask_loss_workspace_bytes_for = what THIS job
 * needs -- the 4-byte-per-pixel image of loss bit patterns (n_frames * frame_len * 4 bytes, the bulk) only exists when the radix select runs, i.e. when
 * topk_frac < 1 and aot_loss > 0.  The call checks ws_bytes against the latter.  (ABI 10) */
size_t tcow_mask_loss_workspace_bytes(long n_frames, long frame_len);
size_t tcow_mask_loss_workspace_bytes_for(long n_frames, long frame_len, double topk_frac, float aot_loss);
int tcow_mask_loss(void* stream, const tcow_mask_loss_args* args);
/* n <= 4 channels of the objective (args[0 .. n)) as ONE set of launches: same n_frames / frame_len, the same `total` (accumulated in argument
 * order), a separate workspace each.  (ABI 9) */
int tcow_mask_loss_batch(void* stream, const tcow_mask_loss_args* args, int n);

/* ------------------------------------------------------------------------------------------- IoU areas (caller row M)
 * eval/metrics.py:19-20,55-66: for each of n_frames contiguous frames of frame_len pixels, counts[f] = {|target|,
 * |output & target|, |output | target|} with output = logit > 0, target = value > 0.5 (int32 [n_frames][3]). */
int tcow_iou_counts(void* stream, const float* logits, const float* target, long n_frames, long frame_len, int* counts);

/* ------------------------------------------------------------------------------------------- query / target masks (caller row P)
 * data/data_utils.py:414-510 for all (b, q, t) at once.  segm (B,1,T,H,W) u8 = visible instance id + 1 (0 = background),
 * div_segm (B,M,T,H,W) u8 = amodal mask per instance; query_idx [B*Q] = queried instance; front_idx / cont_idx [B*Q*T] = instance
 * of the frontmost occluder / outermost container of that frame or -1 (decided beforehand from the occlusion fractions and the
 * containment DAG, data_utils.py:455-492).  Writes query_mask (B,Q,1,T,H,W) f32 (visible pixels at frame query_time only, :431),
 * target_mask (B,Q,3,T,H,W) f32 (:441-492), snitch_occl_by_ptr (B,Q,1,T,H,W) u8 (:435-437) and counts [1 + 2*Q] int32:
 * counts[0] = number of amodal snitch pixels (the class-balancing statistic of loss.py:100), counts[1+2q] / counts[2+2q] != 0
 * iff query q has a non-empty query mask / target (the checks of pipeline.py:149-154).  H*W must be a multiple of 16. */
int tcow_build_masks(void* stream, int B, int Q, int M, int T, long HW, int query_time, const unsigned char* segm,
                     const unsigned char* div_segm, const int* query_idx, const int* front_idx, const int* cont_idx,
                     float* query_mask, float* target_mask, unsigned char* snitch_occl_by_ptr, int* counts);

/* tcow_build_masks with the per-frame decisions made on the device as well (data/data_utils.py:455-492 + loss.py:55-83, 285-308): one
 * launch over (b, q, t) picks the frontmost occluder (first maximum of the queried instance's occluded-by row, present when the query's
 * occlusion fraction >= front_thres and that maximum >= front_half_thres) and the outermost container (among the instances containing the
 * query at >= outer_thres the one least contained itself; one candidate or none: the strongest container) from occl_fracs (B,K,T,3) f32 and
 * the containment DAG dag (B,T,M,M,3) f32 for the queried instances sel (B,Q) int64, then the mask pass of tcow_build_masks runs on them.
 * Also written: idx_ws int32 [B*Q + 2*B*Q*T] (query | front | cont indices, -1 = none), ids (B,Q,T,2) u8 (index + 1, 0 = none: full_occl_cont_id),
 * flags (B,Q,T,3) f32 {has occluder, has container, occlusion fraction}, sel_occl_fracs (B,Q,T,3) f32, frame_w [3][B*Q*T] f32: the snitch frame
 * weights max(occl * occluded_weight, 1) with x0.2 at (B-1, :, query_time) (loss.py:55-83), and the occluder / container channel weights
 * (has_weight where the frame has such a mask, zero_weight elsewhere: loss.py:285-308 with the caller's f32 roundings of 1-z+z and z).
 * counts as in tcow_build_masks (zeroed here). */
int tcow_build_query_masks(void* stream, int B, int Q, int K, int M, int T, long HW, int query_time, const unsigned char* segm,
                           const unsigned char* div_segm, const float* occl_fracs, const float* dag, const long long* sel, float front_thres,
                           float front_half_thres, float outer_thres, float occluded_weight, float zero_weight, float has_weight, int* idx_ws,
                           unsigned char* ids, float* flags, float* sel_occl_fracs, float* frame_w, float* query_mask, float* target_mask,
                           unsigned char* snitch_occl_by_ptr, int* counts);

/* eval/metrics.py:55-113 from tcow_iou_counts' table counts (n_seq, C, T, 3) int32: mean[6] f32 / count[6] int32 in the order snitch_iou,
 * occl_mask_iou, cont_mask_iou, snitch_during_vis_iou, snitch_during_occl_iou, snitch_during_cont_iou (float64 sums; mean -1 where count 0;
 * channels beyond C count as absent). */
int tcow_iou_means(void* stream, const int* counts, int n_seq, int C, int T, float* mean, int* count);

/* Snitch pixel weights (loss.py:85-148) times the frame weights (loss.py:55-83): weights[f,i] = frame_w[f] * class-balance factor
 * (from *pos_count amodal pixels out of n_seq*T*H*W) * 2 on occluded snitch pixels * hard_negative_factor on the band between the
 * target and its k x k box dilation, k = odd(int(sqrt(H*W)/12)).  target_ch0 addresses channel 0 of a (n_seq,3,T,H,W) tensor
 * (sequence stride in elements); weights is dense (n_seq,T,H,W). */
size_t tcow_snitch_weights_workspace_bytes(long n_frames, int H, int W);
int tcow_snitch_weights(void* stream, long n_seq, int T, int H, int W, const float* target_ch0, long target_seq_stride,
                        const unsigned char* snitch_occl_by_ptr, const float* frame_w, const int* pos_count, int class_balancing,
                        float hard_negative_factor, float* weights, void* ws, size_t ws_bytes);

#ifdef __cplusplus
}
#endif
#endif /* TCOW_HIP_H */
