"""ctypes binding of libtcow_hip.so (the C ABI declared in include/tcow_hip.h).

The library is the product: if it is missing or fails to load, importing the kernels raises -- there is
no PyTorch/CPU fallback on this path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TCOW_LIB') or os.path.join(_HERE, 'libtcow_hip.so')     # TCOW_LIB: A/B builds of the library (dev aid)
LIB_PATH_FP16 = os.path.join(_HERE, 'libtcow_hip_fp16.so')                          # the binary16 build of the same sources (precision='fp16')

TCOW_F32, TCOW_BF16, TCOW_F32X3 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_DGELU, ACT_GELU_DSAVE, ACT_MUL_AUX = 0, 1, 2, 3, 4


class TcowError(RuntimeError):
    """Raised when a libtcow_hip entry point returns a non-zero status (message from tcow_last_error)."""


class AttnShape(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int), ('T', ctypes.c_int), ('S', ctypes.c_int), ('D', ctypes.c_int),
                ('heads', ctypes.c_int), ('causal', ctypes.c_int), ('dtype', ctypes.c_int)]


class GemmArgs(ctypes.Structure):
    _fields_ = [('M', ctypes.c_int), ('N', ctypes.c_int), ('K', ctypes.c_int), ('dtype', ctypes.c_int),
                ('A', ctypes.c_void_p), ('lda', ctypes.c_long), ('W', ctypes.c_void_p), ('ldw', ctypes.c_long),
                ('C', ctypes.c_void_p), ('ldc', ctypes.c_long), ('out_f32', ctypes.c_int),
                ('bias', ctypes.c_void_p), ('row_scale', ctypes.c_void_p), ('resid', ctypes.c_void_p),
                ('ldr', ctypes.c_long), ('act', ctypes.c_int), ('aux', ctypes.c_void_p), ('ldaux', ctypes.c_long), ('tile', ctypes.c_int),
                ('bias2', ctypes.c_void_p), ('row_scale2', ctypes.c_void_p)]


class SGemm(ctypes.Structure):
    _fields_ = [('M', ctypes.c_int), ('N', ctypes.c_int), ('K', ctypes.c_int),
                ('A', ctypes.c_void_p), ('sai', ctypes.c_long), ('sak', ctypes.c_long),
                ('B', ctypes.c_void_p), ('sbj', ctypes.c_long), ('sbk', ctypes.c_long),
                ('C', ctypes.c_void_p), ('ldc', ctypes.c_long), ('accumulate', ctypes.c_int)]


class TnProblem(ctypes.Structure):
    _fields_ = [('M', ctypes.c_int), ('N', ctypes.c_int), ('K', ctypes.c_int),
                ('dY', ctypes.c_void_p), ('ldy', ctypes.c_long), ('X', ctypes.c_void_p), ('ldx', ctypes.c_long),
                ('dW', ctypes.c_void_p), ('lddw', ctypes.c_long), ('bias_grad', ctypes.c_void_p), ('accumulate', ctypes.c_int)]


class LnFoldJob(ctypes.Structure):
    _fields_ = [('part', ctypes.c_void_p), ('parts', ctypes.c_int), ('D', ctypes.c_int), ('dgamma', ctypes.c_void_p), ('dbeta', ctypes.c_void_p),
                ('colsum_out', ctypes.c_void_p), ('accumulate', ctypes.c_int)]


class MaskLossArgs(ctypes.Structure):
    _fields_ = [('n_frames', ctypes.c_long), ('frame_len', ctypes.c_long), ('frames_per_seq', ctypes.c_long),
                ('logits', ctypes.c_void_p), ('logits_seq_stride', ctypes.c_long),
                ('target', ctypes.c_void_p), ('target_seq_stride', ctypes.c_long),
                ('pixel_w', ctypes.c_void_p), ('frame_w', ctypes.c_void_p),
                ('weighted_aot', ctypes.c_int), ('aot_loss', ctypes.c_float), ('topk_frac', ctypes.c_double),
                ('loss_weight', ctypes.c_float), ('loss', ctypes.c_void_p), ('total', ctypes.c_void_p),
                ('dlogits', ctypes.c_void_p), ('dlogits_seq_stride', ctypes.c_long),
                ('ws', ctypes.c_void_p), ('ws_bytes', ctypes.c_size_t), ('focal', ctypes.c_int)]


_libs = {}

_vp, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_ash = ctypes.POINTER(AttnShape)
# name -> (restype, argtypes): every entry point declared in include/tcow_hip.h.  Explicit argtypes matter:
# without them ctypes passes Python ints as 32-bit C ints and `long` strides arrive with garbage upper halves.
SIGNATURES = {
    'tcow_version': (_i, []),
    'tcow_last_error': (ctypes.c_char_p, []),
    'tcow_gemm_nt': (_i, [_vp, ctypes.POINTER(GemmArgs)]),
    'tcow_prof_gemm_begin': (_i, [_i]),
    'tcow_prof_gemm_end': (_i, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]),
    'tcow_prof_attn_begin': (_i, [_i]),
    'tcow_prof_attn_end': (_i, [_vp, _vp, _vp, _vp]),
    'tcow_gemm_tn_workspace_bytes': (_l, [_i, _i, _i]),
    'tcow_gemm_tn': (_i, [_vp, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _vp, _l]),
    'tcow_gemm_tn_grouped_workspace_bytes': (_l, [_i, _i, _vp]),
    'tcow_gemm_tn_grouped': (_i, [_vp, _i, _i, _vp, _vp, _l]),
    'tcow_gemm_tn_group_max': (_i, []),
    'tcow_sgemm_x3_batched': (_i, [_vp, _i, _vp]),
    'tcow_layernorm_fwd': (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _vp, _f, _vp, _l, _vp, _vp]),
    'tcow_layernorm_bwd_workspace_bytes': (_l, [_i]),
    'tcow_layernorm_bwd': (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _i, _vp, _l, _vp, _l, _vp, _vp, _vp]),
    'tcow_layernorm_bwd_parts': (_i, [_i, _i]),
    'tcow_layernorm_fold': (_i, [_vp, _i, _vp]),
    'tcow_attn_temporal_fwd': (_i, [_vp, _ash, _vp, _vp, _vp]),
    'tcow_attn_spatial_fwd': (_i, [_vp, _ash, _vp, _vp, _vp]),
    'tcow_attn_bwd_workspace_bytes': (_l, [_ash]),
    'tcow_attn_temporal_bwd': (_i, [_vp, _ash, _vp, _vp, _vp, _vp, _vp, _vp, _l]),
    'tcow_attn_spatial_bwd': (_i, [_vp, _ash, _vp, _vp, _vp, _vp, _vp, _vp, _l]),
    'tcow_im2col': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'tcow_gather_frames': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'tcow_resize_aa': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    'tcow_photometric_workspace_bytes': (_l, [_i]),
    'tcow_photometric': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, ctypes.POINTER(ctypes.c_int), _f, _f, _f, _f, _i, ctypes.POINTER(ctypes.c_float), _i,
                         _vp, _l, _vp]),
    'tcow_im2col_channels': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    'tcow_embed_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'tcow_embed_bwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i]),
    'tcow_cls_merge': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i]),
    'tcow_cls_merge_bwd_cast': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _l, _vp]),
    'tcow_unpatchify_pool_fwd': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'tcow_unpatchify_pool_bwd': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'tcow_upsample_fwd': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'tcow_upsample_bwd': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'tcow_upsample_bwd_amax': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i]),
    'tcow_flags_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'tcow_droppath_rows': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'tcow_scale_cast': (_i, [_vp, _i, _l, _i, _vp, _l, _vp, _vp, _l]),
    'tcow_scale_unless_one': (_i, [_vp, _vp, _l, _vp]),
    'tcow_cast_transpose': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'tcow_cast_desc_bytes': (_l, []),
    'tcow_cast_transpose_batched': (_i, [_vp, _i, _vp, _i, _i]),
    'tcow_adamw_chunk_bytes': (_l, []),
    'tcow_adamw_clip_step': (_i, [_vp, _vp, _i, _f, _f, _f, _f, _f, _i, _f, _vp]),
    'tcow_adamw_clip_step_scaled': (_i, [_vp, _vp, _i, _f, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    'tcow_adamw_tile_bytes': (_l, []),
    'tcow_adamw_clip_step_cast': (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _f, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    'tcow_mask_loss_workspace_bytes': (ctypes.c_size_t, [_l, _l]),
    'tcow_mask_loss_workspace_bytes_for': (ctypes.c_size_t, [_l, _l, ctypes.c_double, _f]),
    'tcow_mask_loss': (_i, [_vp, ctypes.POINTER(MaskLossArgs)]),
    'tcow_mask_loss_batch': (_i, [_vp, ctypes.POINTER(MaskLossArgs), _i]),
    'tcow_iou_counts': (_i, [_vp, _vp, _vp, _l, _l, _vp]),
    'tcow_build_masks': (_i, [_vp, _i, _i, _i, _i, _l, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'tcow_build_query_masks': (_i, [_vp, _i, _i, _i, _i, _i, _l, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'tcow_iou_means': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'tcow_snitch_weights_workspace_bytes': (ctypes.c_size_t, [_l, _i, _i]),
    'tcow_snitch_weights': (_i, [_vp, _l, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _i, _f, _vp, _vp, ctypes.c_size_t]),
}


ABI_VERSION = 10       # TCOW_ABI_VERSION of include/tcow_hip.h


def _declare(L, tolerant=False):
    for name, (res, args) in SIGNATURES.items():
        if tolerant and not hasattr(L, name):      # (TCOW_LIB: an older build on purpose -- entry points it lacks simply cannot be called)
            continue
        fn = getattr(L, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args



def lib(fmt='bf16'):
    """The library built for the given 16-bit storage format: 'bf16' -> libtcow_hip.so (also serves the f32-storage modes),
    'fp16' -> libtcow_hip_fp16.so (the same sources compiled with -DTCOW_FP16, csrc/common.h)."""
    L = _libs.get(fmt)
    if L is None:
        path = LIB_PATH if fmt == 'bf16' else LIB_PATH_FP16
        if not os.path.exists(path):
            raise TcowError(f'{path} not found: build it with `make` (or __graft_entry__.build()); '
                            'the Seeker HIP path has no fallback')
        # torch bundles its own libamdhip64; it must be in the process first so that this library binds to the
        # SAME HIP runtime (otherwise device pointers / streams of one runtime are foreign to the other).
        import torch  # noqa: F401
        L = ctypes.CDLL(path)
        L.tcow_last_error.restype = ctypes.c_char_p
        L.tcow_version.restype = ctypes.c_int
        if L.tcow_version() != ABI_VERSION and not (fmt == 'bf16' and os.environ.get('TCOW_LIB')):      # (TCOW_LIB: an older build on purpose, same-box A/B of whole libraries)
            # a stale build next to newer host code (or the reverse): signatures may differ -- refuse before the first call
            raise TcowError(f'{path} reports ABI version {L.tcow_version()}, this host code was written against {ABI_VERSION} (include/tcow_hip.h): rebuild with `make`')
        _declare(L, tolerant=(fmt == 'bf16' and bool(os.environ.get('TCOW_LIB'))))
        _libs[fmt] = L
    return L


def check(rc, what='', L=None):
    if rc != 0:
        raise TcowError(f'{what} failed (status {rc}): {(L or lib()).tcow_last_error().decode()}')
