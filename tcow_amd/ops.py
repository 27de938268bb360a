"""Thin tensor-level wrappers over the libtcow_hip C ABI (include/tcow_hip.h).

PyTorch is used here only for device memory and streams: every function takes CUDA(HIP) tensors, passes raw
device pointers plus the current stream to the library and returns.  There is no fallback path: a missing or
failing library raises TcowError.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib as L

F32, BF16 = L.TCOW_F32, L.TCOW_BF16
F32X3 = L.TCOW_F32X3      # GEMM wrappers only: f32 tensors, bf16 x 3 split products (csrc/gemm_x3.hip)
ACT_NONE, ACT_GELU, ACT_DGELU, ACT_GELU_DSAVE, ACT_MUL_AUX = L.ACT_NONE, L.ACT_GELU, L.ACT_DGELU, L.ACT_GELU_DSAVE, L.ACT_MUL_AUX


FP16 = 16                 # Python-side mode: the 16-bit mode (ABI dtype TCOW_BF16) of the binary16 build of the library, libtcow_hip_fp16.so


def tdtype(mode):
    return torch.bfloat16 if mode == BF16 else torch.float16 if mode == FP16 else torch.float32


def is16(mode):
    return mode in (BF16, FP16)


def _sel(mode):
    """(library, ABI dtype) of a mode: the binary16 build serves FP16 through its 16-bit dtype, everything else is the main library."""
    return (L.lib('fp16'), BF16) if mode == FP16 else (L.lib(), mode)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """Raw hipStream_t of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Python Stream object through
    four layers of device-index helpers: 8 us per call, ~430 calls per training step = 3.5 ms of the host's enqueue time.)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return t.data_ptr() if t is not None else None


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.TcowError('libtcow_hip kernels need CUDA/HIP tensors (no CPU fallback on this path)')


def gemm_nt(mode, A, W, out, bias=None, row_scale=None, resid=None, act=ACT_NONE, aux=None, tile=0, bias2=None, row_scale2=None):
    """out[M,N] = epilogue(A[M,K] @ W[N,K]^T); see tcow_gemm_nt. `out` dtype f32 or the mode's dtype."""
    _need_cuda(A, W, out)
    M, K = A.shape
    N = W.shape[0]
    lib, dm = _sel(mode)
    a = L.GemmArgs(M, N, K, dm, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), out.data_ptr(), out.stride(0),
                   1 if out.dtype == torch.float32 else 0, _p(bias), _p(row_scale), _p(resid),
                   resid.stride(0) if resid is not None else 0, act, _p(aux), aux.stride(0) if aux is not None else 0, int(tile), _p(bias2), _p(row_scale2))
    L.check(lib.tcow_gemm_nt(_stream(), ctypes.byref(a)), 'tcow_gemm_nt', lib)
    return out


_ws_cache = {}


def workspace(nbytes, device, tag='default'):
    """Grow-only scratch buffer per (device, STREAM, host thread, tag).  Kernels on one stream run in order, so reuse within a stream is safe; two streams of one
    device (two host threads driving replicas -- the torch.nn.DataParallel thread model of train.py:222-223 -- or a side stream) never share scratch:
    the key carries torch's current stream, the one the kernels that use the buffer are launched on (tests/test_gpu_seeker.py::test_two_threads_two_streams).
    A buffer that has to grow is replaced, not freed under a running kernel: the old tensor's storage returns to torch's caching allocator, which holds
    it for this stream until the work queued on it has drained."""
    # (_stream(): the current stream of the current device -- the same call the launch that follows makes; the thread id: two host threads that share
    # a stream -- torch.nn.DataParallel replicas placed on one device -- interleave their launches, and some scratch lives across two launches)
    key = (str(device), _stream(), tag, threading.get_ident())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is None and len(_ws_cache) >= 48:         # short-lived host threads (DataParallel starts new ones per forward): forget the scratch of threads that are gone
            alive = {t.ident for t in threading.enumerate()}
            for k in [k for k in _ws_cache if k[3] not in alive]:
                del _ws_cache[k]
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def gemm_tn(mode, dY, X, dW, bias_grad=None, accumulate=False):
    """dW[N,K] (+)= dY[M,N]^T @ X[M,K]; bias_grad[N] (+)= column sums of dY."""
    _need_cuda(dY, X, dW)
    M, N = dY.shape
    K = X.shape[1]
    lib, dm = _sel(mode)
    nbytes = lib.tcow_gemm_tn_workspace_bytes(M, N, K)
    ws = workspace(nbytes, dY.device, 'tn')
    L.check(lib.tcow_gemm_tn(_stream(), dm, M, N, K, dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0), dW.data_ptr(),
                             dW.stride(0), _p(bias_grad), int(accumulate), ws.data_ptr(), ws.numel()), 'tcow_gemm_tn', lib)
    return dW


def tn_group_max():
    """Most problems one grouped weight-gradient launch takes (tcow_gemm_tn_group_max)."""
    return int(L.lib().tcow_gemm_tn_group_max())


def gemm_tn_grouped(mode, problems):
    """Weight / bias gradients of several Linear layers in one call: problems = [(dY [M,N], X [M,K], dW [N,K] f32, bias_grad [N] or None), ...].
    In bf16 mode problems that share M run as one grid (tcow_gemm_tn_grouped); the other modes loop inside the library."""
    n = len(problems)
    arr = (L.TnProblem * n)()
    for i, (dY, X, dW, db) in enumerate(problems):
        _need_cuda(dY, X, dW)
        arr[i] = L.TnProblem(dY.shape[0], dY.shape[1], X.shape[1], dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0), dW.data_ptr(), dW.stride(0), _p(db), 0)
    lib, dm = _sel(mode)
    ws = workspace(lib.tcow_gemm_tn_grouped_workspace_bytes(dm, n, arr), problems[0][0].device, 'tn')
    L.check(lib.tcow_gemm_tn_grouped(_stream(), dm, n, arr, ws.data_ptr(), ws.numel()), 'tcow_gemm_tn_grouped', lib)


def sgemm_batched(problems, accumulate=False):
    """Small f32 products C (+)= A B, several per launch (tcow_sgemm_x3_batched).  problems = [(A, B, C), ...] with 2-D f32 tensors of ANY
    strides: A [M,K], B [K,N] (pass `.t()` views for transposed operands), C [M,N] row-major with unit column stride."""
    n = len(problems)
    arr = (L.SGemm * n)()
    for i, (A, B, C) in enumerate(problems):
        _need_cuda(A, B, C)
        M, K = A.shape
        N = B.shape[1]
        assert B.shape[0] == K and C.shape == (M, N) and C.stride(1) == 1 and A.dtype == B.dtype == C.dtype == torch.float32
        arr[i] = L.SGemm(M, N, K, A.data_ptr(), A.stride(0), A.stride(1), B.data_ptr(), B.stride(1), B.stride(0), C.data_ptr(), C.stride(0), 1 if accumulate else 0)
    L.check(L.lib().tcow_sgemm_x3_batched(_stream(), n, arr), 'tcow_sgemm_x3_batched')


def layernorm_fwd(mode, x, gamma, beta, out, mean=None, rstd=None, eps=1e-6):
    rows, D = x.shape
    lib, dm = _sel(mode)
    L.check(lib.tcow_layernorm_fwd(_stream(), dm, rows, D, x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), eps,
                                       out.data_ptr(), out.stride(0), _p(mean), _p(rstd)), 'tcow_layernorm_fwd', lib)
    return out


def layernorm_bwd(mode, dy, x, mean, rstd, gamma, dres, dx, dgamma=None, dbeta=None, accumulate=False, dx_cast=None, cast_scale=None,
                  colsum_out=None, colsum_scale=None, defer=None):
    """dx = dres + dLN(dy); dx_cast (mode dtype, optional) = dx * cast_scale[row]: the next input-gradient GEMM's operand;
    colsum_out [D] (optional) = sum over rows of colsum_scale[row] * dx[row].
    defer: a list -> the fold of the dgamma / dbeta (/ colsum) partial table is NOT launched; a job for layernorm_fold is appended instead (the
    table lives in its own workspace until then)."""
    rows, D = x.shape
    lib, dm = _sel(mode)
    if defer is not None and dgamma is not None:
        ws = workspace(lib.tcow_layernorm_bwd_workspace_bytes(D), x.device, 'ln_defer%d' % len(defer))
        defer.append((lib, ws, int(lib.tcow_layernorm_bwd_parts(rows, 1 if colsum_out is not None else 0)), D, dgamma, dbeta, colsum_out, bool(accumulate)))
        accumulate = int(accumulate) | 2
    else:
        ws = workspace(lib.tcow_layernorm_bwd_workspace_bytes(D), x.device, 'ln')
    L.check(lib.tcow_layernorm_bwd(_stream(), dm, rows, D, dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), mean.data_ptr(),
                                   rstd.data_ptr(), gamma.data_ptr(), _p(dres), dres.stride(0) if dres is not None else 0, dx.data_ptr(),
                                   dx.stride(0), _p(dgamma), _p(dbeta), int(accumulate), ws.data_ptr(), ws.numel(),
                                   _p(dx_cast), dx_cast.stride(0) if dx_cast is not None else 0, _p(cast_scale), _p(colsum_scale), _p(colsum_out)), 'tcow_layernorm_bwd', lib)
    return dx


def layernorm_fold(jobs):
    """The deferred dgamma / dbeta (/ colsum) folds of layernorm_bwd(..., defer=jobs): up to 16 per launch (tcow_layernorm_fold)."""
    for c0 in range(0, len(jobs), 16):
        chunk = jobs[c0:c0 + 16]
        lib = chunk[0][0]
        arr = (L.LnFoldJob * len(chunk))()
        for i, (_, ws, parts, D, dg, db, cs, acc) in enumerate(chunk):
            arr[i] = L.LnFoldJob(ws.data_ptr(), parts, D, dg.data_ptr(), db.data_ptr(), _p(cs), 1 if acc else 0)
        L.check(lib.tcow_layernorm_fold(_stream(), len(chunk), arr), 'tcow_layernorm_fold', lib)
    jobs.clear()


def attn_shape(mode, B, T, S, D, heads, causal):
    lib, dm = _sel(mode)
    sh = L.AttnShape(B, T, S, D, heads, int(causal), dm)
    sh.tcow_lib = lib              # which build of the library this shape's calls go to
    return sh


def attn_fwd(shape, spatial, qkv, out, lse=None):
    lib = shape.tcow_lib
    fn = lib.tcow_attn_spatial_fwd if spatial else lib.tcow_attn_temporal_fwd
    L.check(fn(_stream(), ctypes.byref(shape), qkv.data_ptr(), out.data_ptr(), _p(lse)), 'tcow_attn_fwd', lib)
    return out


def attn_bwd(shape, spatial, qkv, out, dout, lse, dqkv):
    lib = shape.tcow_lib
    ws = workspace(lib.tcow_attn_bwd_workspace_bytes(ctypes.byref(shape)), qkv.device, 'attn')
    fn = lib.tcow_attn_spatial_bwd if spatial else lib.tcow_attn_temporal_bwd
    L.check(fn(_stream(), ctypes.byref(shape), qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
               ws.data_ptr(), ws.numel()), 'tcow_attn_bwd', lib)
    return dqkv


def im2col(mode, rgb, query, P, pretrained_norm, out):
    B, _, T, H, W = rgb.shape
    L.check(_sel(mode)[0].tcow_im2col(_stream(), _sel(mode)[1], B, T, H, W, P, rgb.data_ptr(), query.data_ptr(), int(pretrained_norm), out.data_ptr()), 'tcow_im2col', _sel(mode)[0])
    return out


def gather_frames(frames, frame_idx, src_y, src_x):
    """frames (C,Tv,H,W) uint8 / 4-byte CUDA tensor, int32 CUDA index tables -> (C,Tc,h,w) (see tcow_gather_frames)."""
    _need_cuda(frames, frame_idx, src_y, src_x)
    if frames.element_size() not in (1, 4) or not frames.is_contiguous():
        raise L.TcowError('gather_frames: contiguous tensor of 1- or 4-byte elements expected')
    for t in (frame_idx, src_y, src_x):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise L.TcowError('gather_frames: index tables must be contiguous int32 tensors')
    C, Tv, H, W = frames.shape
    out = torch.empty(C, frame_idx.numel(), src_y.numel(), src_x.numel(), dtype=frames.dtype, device=frames.device)
    L.check(L.lib().tcow_gather_frames(_stream(), frames.element_size(), C, Tv, H, W, frame_idx.numel(), src_y.numel(), src_x.numel(), frames.data_ptr(),
                                       frame_idx.data_ptr(), src_y.data_ptr(), src_x.data_ptr(), out.data_ptr()), 'tcow_gather_frames')
    return out


def resize_aa(frames, frame_idx, ys, xs, ymin, ysize, wy, ky, xmin, xsize, wx, kx, out_h, out_w):
    """frames (C,Tv,H,W) f32 CUDA tensor + the index / weight tables of tcow_amd.augs (int32 / f32 CUDA tensors) -> (C,Tc,out_h,out_w) f32
    (see tcow_resize_aa)."""
    _need_cuda(frames, frame_idx, ys, xs, ymin, ysize, wy, xmin, xsize, wx)
    if frames.dtype != torch.float32 or not frames.is_contiguous():
        raise L.TcowError('resize_aa: contiguous f32 frames expected')
    for t, dt in ((frame_idx, torch.int32), (ys, torch.int32), (xs, torch.int32), (ymin, torch.int32), (ysize, torch.int32), (xmin, torch.int32), (xsize, torch.int32),
                  (wy, torch.float32), (wx, torch.float32)):
        if t.dtype != dt or not t.is_contiguous():
            raise L.TcowError('resize_aa: tables must be contiguous int32 / float32 tensors')
    C, Tv, H, W = frames.shape
    out = torch.empty(C, frame_idx.numel(), out_h, out_w, dtype=torch.float32, device=frames.device)
    L.check(L.lib().tcow_resize_aa(_stream(), C, Tv, H, W, frame_idx.numel(), ys.numel(), xs.numel(), out_h, out_w, frames.data_ptr(), frame_idx.data_ptr(), ys.data_ptr(),
                                   xs.data_ptr(), ymin.data_ptr(), ysize.data_ptr(), wy.data_ptr(), int(ky), xmin.data_ptr(), xsize.data_ptr(), wx.data_ptr(), int(kx),
                                   out.data_ptr()), 'tcow_resize_aa')
    return out


def photometric(frames, frame_idx, rect, order, factors, taps, gray):
    """frames (3,Tv,H,W) f32 CUDA tensor in [0, 1], frame_idx int32 CUDA table, rect = (y0, x0, h, w) -> (3,Tc,h,w) f32: ColorJitter adjustments
    `order` (list of 0 brightness / 1 contrast / 2 saturation / 3 hue, may be empty) with `factors` = (brightness, contrast, saturation, hue),
    5-tap blur with the normalised `taps` (None: no blur), grayscale fold (see tcow_photometric)."""
    import ctypes
    _need_cuda(frames, frame_idx)
    if frames.dtype != torch.float32 or not frames.is_contiguous() or frames.dim() != 4 or frames.shape[0] != 3:
        raise L.TcowError('photometric: contiguous (3, Tv, H, W) f32 frames expected')
    if frame_idx.dtype != torch.int32 or not frame_idx.is_contiguous():
        raise L.TcowError('photometric: frame_idx must be a contiguous int32 tensor')
    _, Tv, H, W = frames.shape
    y0, x0, h, w = (int(v) for v in rect)
    Tc = frame_idx.numel()
    lib = L.lib()
    nb = lib.tcow_photometric_workspace_bytes(Tc)
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=frames.device)
    out = torch.empty(3, Tc, h, w, dtype=torch.float32, device=frames.device)
    ops_arr = (ctypes.c_int * 4)(*([int(o) for o in order] + [0] * (4 - len(order))))
    taps_arr = (ctypes.c_float * 5)(*([float(t) for t in taps] if taps is not None else [0.0] * 5))
    fb, fc, fs, fh = (float(v) for v in factors)
    L.check(lib.tcow_photometric(_stream(), Tv, H, W, Tc, y0, x0, h, w, frames.data_ptr(), frame_idx.data_ptr(), len(order), ops_arr, fb, fc, fs, fh,
                                 1 if taps is not None else 0, taps_arr, 1 if gray else 0, ws.data_ptr(), nb, out.data_ptr()), 'tcow_photometric')
    return out


def im2col_channels(mode, src, P, normalise, out):
    """src (B,C,T,H,W) f32 -> out [B*T*S, C*P*P] (see tcow_im2col_channels)."""
    B, C, T, H, W = src.shape
    L.check(_sel(mode)[0].tcow_im2col_channels(_stream(), _sel(mode)[1], B, T, H, W, P, C, src.data_ptr(), int(normalise), out.data_ptr()), 'tcow_im2col_channels', _sel(mode)[0])
    return out


def embed_fwd(x, B, T, S, cls, pos, time_embed):
    L.check(L.lib().tcow_embed_fwd(_stream(), B, T, S, x.shape[1], x.data_ptr(), cls.data_ptr(), pos.data_ptr(), time_embed.data_ptr()), 'tcow_embed_fwd')
    return x


def embed_bwd(g, B, T, S, dpos, dtime, accumulate=False):
    L.check(L.lib().tcow_embed_bwd(_stream(), B, T, S, g.shape[1], g.data_ptr(), dpos.data_ptr(), dtime.data_ptr(), int(accumulate)), 'tcow_embed_bwd')


def cls_merge(x, B, T, S, mode, backward=False, cast_mode=None, cast_out=None, cast_scale=None):
    """cast_out (backward only): the 16-bit / f32 operand copy of x to refresh for the rewritten slot-0 rows (tcow_cls_merge_bwd_cast)."""
    if cast_out is not None:
        lib, dm = _sel(cast_mode)
        L.check(lib.tcow_cls_merge_bwd_cast(_stream(), dm, B, T, S, x.shape[1], x.data_ptr(), mode, cast_out.data_ptr(), cast_out.stride(0), _p(cast_scale)), 'tcow_cls_merge_bwd_cast', lib)
        return x
    L.check(L.lib().tcow_cls_merge(_stream(), B, T, S, x.shape[1], x.data_ptr(), mode, int(backward)), 'tcow_cls_merge')
    return x


def unpatchify_pool_fwd(mode, pm, BT, Hp, Wp, P, C, st, pooled):
    L.check(_sel(mode)[0].tcow_unpatchify_pool_fwd(_stream(), _sel(mode)[1], BT, Hp, Wp, P, C, st, pm.data_ptr(), pooled.data_ptr()), 'tcow_unpatchify_pool_fwd', _sel(mode)[0])
    return pooled


def unpatchify_pool_bwd(mode, dpooled, BT, Hp, Wp, P, C, st, dpm):
    L.check(_sel(mode)[0].tcow_unpatchify_pool_bwd(_stream(), _sel(mode)[1], BT, Hp, Wp, P, C, st, dpooled.data_ptr(), dpm.data_ptr()), 'tcow_unpatchify_pool_bwd', _sel(mode)[0])
    return dpm


def upsample_fwd(pooled, B, T, C, h, w, st, bilinear, out):
    L.check(L.lib().tcow_upsample_fwd(_stream(), B, T, C, h, w, st, int(bilinear), pooled.data_ptr(), out.data_ptr()), 'tcow_upsample_fwd')
    return out


def upsample_bwd(dout, B, T, C, h, w, st, bilinear, dpooled):
    L.check(L.lib().tcow_upsample_bwd(_stream(), B, T, C, h, w, st, int(bilinear), dout.data_ptr(), dpooled.data_ptr()), 'tcow_upsample_bwd')
    return dpooled


AMAX_SLOTS = 64          # TCOW_AMAX_SLOTS of include/tcow_hip.h


def upsample_bwd_amax(dout, B, T, C, h, w, st, dpooled):
    """upsample_bwd (bilinear, stride 4, h, w > 4) that also returns max |dout| as a 0-dim f32 device tensor (see tcow_upsample_bwd_amax)."""
    bits = torch.zeros(AMAX_SLOTS, dtype=torch.int32, device=dout.device)           # TCOW_AMAX_SLOTS partial maxima (one atomic per workgroup, spread over the slots)
    L.check(L.lib().tcow_upsample_bwd_amax(_stream(), B, T, C, h, w, st, dout.data_ptr(), dpooled.data_ptr(), bits.data_ptr(), AMAX_SLOTS), 'tcow_upsample_bwd_amax')
    return dpooled, bits.view(torch.float32).amax()


def flags_fwd(x, BT, S, Wf, bf, flags):
    L.check(L.lib().tcow_flags_fwd(_stream(), BT, S, x.shape[1], Wf.shape[0], x.data_ptr(), Wf.data_ptr(), bf.data_ptr(), flags.data_ptr()), 'tcow_flags_fwd')
    return flags


def scale_unless_one(x, scale):
    """x *= scale (a 0-dim / 1-element f32 device tensor) in place, decided on the device: no memory traffic when scale == 1 (tcow_scale_unless_one)."""
    _need_cuda(x, scale)
    if x.dtype != torch.float32 or not x.is_contiguous() or scale.dtype != torch.float32 or scale.numel() != 1:
        raise L.TcowError('scale_unless_one: a contiguous f32 tensor and a one-element f32 scale')
    L.check(L.lib().tcow_scale_unless_one(_stream(), x.data_ptr(), x.numel(), scale.data_ptr()), 'tcow_scale_unless_one')
    return x


def scale_cast(mode, src, row_scale, dst):
    rows, D = src.shape
    L.check(_sel(mode)[0].tcow_scale_cast(_stream(), _sel(mode)[1], rows, D, src.data_ptr(), src.stride(0), _p(row_scale), dst.data_ptr(), dst.stride(0)), 'tcow_scale_cast', _sel(mode)[0])
    return dst


def droppath_rows(u, keep_p, mask0, B, T, S):
    """u (depth, B*(S-1) + B*T + B) f32 uniform draws, keep_p (depth,) f32, mask0 [B*T*S] f32 -> (4, depth, B*T*S) f32 row scales
    (temporal | spatial | MLP | temporal x mask0), see tcow_droppath_rows."""
    _need_cuda(u, keep_p, mask0)
    depth = u.shape[0]
    if u.shape[1] != B * (S - 1) + B * T + B or u.dtype != torch.float32 or not u.is_contiguous() or mask0.numel() != B * T * S:
        raise L.TcowError('droppath_rows: u must be contiguous f32 (depth, B*(S-1) + B*T + B), mask0 [B*T*S]')
    out = torch.empty(4, depth, B * T * S, dtype=torch.float32, device=u.device)
    L.check(L.lib().tcow_droppath_rows(_stream(), depth, B, T, S, u.data_ptr(), keep_p.data_ptr(), mask0.data_ptr(), out.data_ptr()), 'tcow_droppath_rows')
    return out


def cast_transpose(mode, W, Wc=None, Wt=None):
    N, K = W.shape
    L.check(_sel(mode)[0].tcow_cast_transpose(_stream(), _sel(mode)[1], N, K, W.data_ptr(), _p(Wc), _p(Wt)), 'tcow_cast_transpose', _sel(mode)[0])


def cast_transpose_batched(mode, table, n, total_tiles):
    """One launch for all operand copies; `table` is the uint8 device tensor of tcow_cast_desc records (see tcow_cast_transpose_batched)."""
    L.check(_sel(mode)[0].tcow_cast_transpose_batched(_stream(), _sel(mode)[1], table.data_ptr(), int(n), int(total_tiles)), 'tcow_cast_transpose_batched', _sel(mode)[0])


def mask_loss(logits, target, channel, pixel_w=None, frame_w=None, weighted_aot=False, aot_loss=0.8, topk_frac=1.0,
              loss_weight=1.0, loss_out=None, total=None, dlogits=None, focal=False):
    """Channel `channel` of the TCOW mask objective on (BQ, C, T, H, W) f32 logits / targets (see tcow_mask_loss):
    writes loss_out[0], adds loss_weight * loss to total[0] and d(loss)/d(logits) into dlogits[:, channel]."""
    _need_cuda(logits, target, pixel_w, frame_w, loss_out, total, dlogits)
    BQ, C, T, H, W = logits.shape
    for t in (logits, target, dlogits):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.shape != logits.shape):
            raise L.TcowError('mask_loss: logits / target / dlogits must be contiguous f32 tensors of one shape')
    n_frames, frame_len = BQ * T, H * W
    if pixel_w is not None and (pixel_w.dtype != torch.float32 or not pixel_w.is_contiguous() or pixel_w.numel() != n_frames * frame_len):
        raise L.TcowError('mask_loss: pixel_w must be a contiguous f32 tensor with one weight per pixel')
    if frame_w is not None and (frame_w.dtype != torch.float32 or not frame_w.is_contiguous() or frame_w.numel() != n_frames):
        raise L.TcowError('mask_loss: frame_w must be a contiguous f32 tensor with one weight per frame')
    lib = L.lib()
    nb = lib.tcow_mask_loss_workspace_bytes_for(n_frames, frame_len, float(topk_frac), float(aot_loss))
    ws = workspace(nb, logits.device, 'mask_loss')
    off = channel * T * frame_len * 4
    a = L.MaskLossArgs(n_frames, frame_len, T, logits.data_ptr() + off, C * T * frame_len, target.data_ptr() + off, C * T * frame_len,
                       _p(pixel_w), _p(frame_w), 1 if weighted_aot else 0, float(aot_loss), float(topk_frac), float(loss_weight),
                       loss_out.data_ptr(), _p(total), (dlogits.data_ptr() + off) if dlogits is not None else None,
                       C * T * frame_len, ws.data_ptr(), ws.numel(), 1 if focal else 0)
    L.check(lib.tcow_mask_loss(_stream(), ctypes.byref(a)), 'tcow_mask_loss')
    return loss_out


def mask_loss_channels(logits, target, jobs, aot_loss=0.8, topk_frac=1.0, total=None, dlogits=None, focal=False):
    """Several channels of the TCOW mask objective in ONE set of launches (tcow_mask_loss_batch).  jobs: up to 4 tuples
    (channel, pixel_w | None, frame_w | None, weighted_aot, loss_weight, loss_out[1]); `total` receives sum_j loss_weight_j * loss_j in job order."""
    _need_cuda(logits, target, total, dlogits)
    BQ, C, T, H, W = logits.shape
    for t in (logits, target, dlogits):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.shape != logits.shape):
            raise L.TcowError('mask_loss: logits / target / dlogits must be contiguous f32 tensors of one shape')
    n_frames, frame_len = BQ * T, H * W
    lib = L.lib()
    nb = lib.tcow_mask_loss_workspace_bytes_for(n_frames, frame_len, float(topk_frac), float(aot_loss))      # (the 4 B / pixel bit-pattern image only when the radix select runs)
    arr = (L.MaskLossArgs * len(jobs))()
    for i, (channel, pixel_w, frame_w, weighted_aot, loss_weight, loss_out) in enumerate(jobs):
        _need_cuda(pixel_w, frame_w, loss_out)
        if pixel_w is not None and (pixel_w.dtype != torch.float32 or not pixel_w.is_contiguous() or pixel_w.numel() != n_frames * frame_len):
            raise L.TcowError('mask_loss: pixel_w must be a contiguous f32 tensor with one weight per pixel')
        if frame_w is not None and (frame_w.dtype != torch.float32 or not frame_w.is_contiguous() or frame_w.numel() != n_frames):
            raise L.TcowError('mask_loss: frame_w must be a contiguous f32 tensor with one weight per frame')
        ws = workspace(nb, logits.device, 'mask_loss%d' % i)             # every job of a batch needs its own workspace
        off = channel * T * frame_len * 4
        arr[i] = L.MaskLossArgs(n_frames, frame_len, T, logits.data_ptr() + off, C * T * frame_len, target.data_ptr() + off, C * T * frame_len,
                                _p(pixel_w), _p(frame_w), 1 if weighted_aot else 0, float(aot_loss), float(topk_frac), float(loss_weight),
                                loss_out.data_ptr(), _p(total), (dlogits.data_ptr() + off) if dlogits is not None else None,
                                C * T * frame_len, ws.data_ptr(), ws.numel(), 1 if focal else 0)
    L.check(lib.tcow_mask_loss_batch(_stream(), arr, len(jobs)), 'tcow_mask_loss_batch')


def iou_counts(logits, target):
    """(..., H, W) f32 logits / targets -> int32 (..., 3): target area, intersection, union per frame (see tcow_iou_counts)."""
    _need_cuda(logits, target)
    if logits.shape != target.shape or logits.dtype != torch.float32 or target.dtype != torch.float32:
        raise L.TcowError('iou_counts: logits and target must be f32 tensors of one shape')
    lo = logits.contiguous(); tg = target.contiguous()
    frame_len = lo.shape[-1] * lo.shape[-2]
    out = torch.empty(lo.shape[:-2] + (3,), dtype=torch.int32, device=lo.device)
    L.check(L.lib().tcow_iou_counts(_stream(), lo.data_ptr(), tg.data_ptr(), lo.numel() // frame_len, frame_len, out.data_ptr()), 'tcow_iou_counts')
    return out


def build_masks(segm, div_segm, query_idx, front_idx, cont_idx, query_time):
    """segm (B,1,T,H,W) u8, div_segm (B,M,T,H,W) u8, query_idx (B,Q) int32, front_idx / cont_idx (B,Q,T) int32 (-1 = none).
    Returns query_mask (B,Q,1,T,H,W) f32, target (B,Q,3,T,H,W) f32, snitch_occl_by_ptr (B,Q,1,T,H,W) u8, counts int32 [1 + 2Q]
    (see tcow_build_masks)."""
    _need_cuda(segm, div_segm, query_idx, front_idx, cont_idx)
    B, _, T, H, W = segm.shape
    M = div_segm.shape[1]; Q = query_idx.shape[1]
    if segm.dtype != torch.uint8 or div_segm.dtype != torch.uint8 or not segm.is_contiguous() or not div_segm.is_contiguous():
        raise L.TcowError('build_masks: segmentation maps must be contiguous uint8 tensors')
    dev = segm.device
    qm = torch.empty(B, Q, 1, T, H, W, dtype=torch.float32, device=dev); tg = torch.empty(B, Q, 3, T, H, W, dtype=torch.float32, device=dev)
    pt = torch.empty(B, Q, 1, T, H, W, dtype=torch.uint8, device=dev); counts = torch.empty(1 + 2 * Q, dtype=torch.int32, device=dev)
    qi = query_idx.to(torch.int32).contiguous(); fi = front_idx.to(torch.int32).contiguous(); ci = cont_idx.to(torch.int32).contiguous()
    L.check(L.lib().tcow_build_masks(_stream(), B, Q, M, T, H * W, int(query_time), segm.data_ptr(), div_segm.data_ptr(), qi.data_ptr(), fi.data_ptr(),
                                     ci.data_ptr(), qm.data_ptr(), tg.data_ptr(), pt.data_ptr(), counts.data_ptr()), 'tcow_build_masks')
    return qm, tg, pt, counts


def build_query_masks(segm, div_segm, occl_fracs, dag, sel, query_time, front_occl_thres, outer_cont_thres, occluded_weight, occl_cont_zero_weight):
    """data_utils.py:414-510 for every query of the batch, the per-frame occluder / container decisions included (tcow_build_query_masks):
    segm (B,1,T,H,W) u8, div_segm (B,M,T,H,W) u8, occl_fracs (B,K,T,3) f32, dag (B,T,M,M,3) f32, sel (B,Q) int64 queried instances.
    Returns a dict: query_mask (B,Q,1,T,H,W) f32, target (B,Q,3,T,H,W) f32, snitch_occl_by_ptr (B,Q,1,T,H,W) u8, counts int32 [1 + 2Q],
    ids (B,Q,T,2) u8, flags (B,Q,T,3) f32, sel_occl_fracs (B,Q,T,3) f32, frame_w (3,B,Q,T) f32 (snitch | occluder | container frame
    weights of loss.py:55-83, 285-308), front_idx / cont_idx (B,Q,T) int32."""
    _need_cuda(segm, div_segm, occl_fracs, dag, sel)
    B, _, T, H, W = segm.shape
    M = div_segm.shape[1]; K = occl_fracs.shape[1]; Q = sel.shape[1]
    if segm.dtype != torch.uint8 or div_segm.dtype != torch.uint8 or not segm.is_contiguous() or not div_segm.is_contiguous():
        raise L.TcowError('build_query_masks: segmentation maps must be contiguous uint8 tensors')
    if tuple(dag.shape) != (B, T, M, M, 3) or tuple(occl_fracs.shape[2:]) != (T, 3) or sel.dtype != torch.int64:
        raise L.TcowError('build_query_masks: occl_fracs (B,K,T,3), dag (B,T,M,M,3) and int64 sel (B,Q) expected')
    dev = segm.device
    of = occl_fracs.to(torch.float32).contiguous(); dg = dag.to(torch.float32).contiguous(); sl = sel.contiguous()
    n = B * Q * T
    qm = torch.empty(B, Q, 1, T, H, W, dtype=torch.float32, device=dev); tg = torch.empty(B, Q, 3, T, H, W, dtype=torch.float32, device=dev)
    pt = torch.empty(B, Q, 1, T, H, W, dtype=torch.uint8, device=dev); counts = torch.empty(1 + 2 * Q, dtype=torch.int32, device=dev)
    idx = torch.empty(B * Q + 2 * n, dtype=torch.int32, device=dev)
    ids = torch.empty(B, Q, T, 2, dtype=torch.uint8, device=dev)
    fl = torch.empty(2, B, Q, T, 3, dtype=torch.float32, device=dev)         # flags | sel_occl_fracs (one allocation)
    fw = torch.empty(3, B, Q, T, dtype=torch.float32, device=dev)
    f32 = np.float32
    z = f32(occl_cont_zero_weight)
    has_w = f32(f32(1.0 - occl_cont_zero_weight) + z)                        # has * (1 - z) + z in f32, as the tensor expression rounds it
    L.check(L.lib().tcow_build_query_masks(_stream(), B, Q, K, M, T, H * W, int(query_time), segm.data_ptr(), div_segm.data_ptr(), of.data_ptr(), dg.data_ptr(),
                                           sl.data_ptr(), float(f32(front_occl_thres)), float(f32(front_occl_thres / 2.0)), float(f32(outer_cont_thres)),
                                           float(f32(float(occluded_weight))), float(z), float(has_w), idx.data_ptr(), ids.data_ptr(), fl[0].data_ptr(),
                                           fl[1].data_ptr(), fw.data_ptr(), qm.data_ptr(), tg.data_ptr(), pt.data_ptr(), counts.data_ptr()),
            'tcow_build_query_masks')
    return {'query_mask': qm, 'target': tg, 'snitch_occl_by_ptr': pt, 'counts': counts, 'ids': ids, 'flags': fl[0], 'sel_occl_fracs': fl[1], 'frame_w': fw,
            'front_idx': idx[B * Q:B * Q + n].view(B, Q, T), 'cont_idx': idx[B * Q + n:].view(B, Q, T)}


def iou_means(counts):
    """counts (n_seq, C, T, 3) int32 from iou_counts -> (mean f32 [6], count int32 [6]): eval/metrics.py:55-113 (see tcow_iou_means)."""
    _need_cuda(counts)
    if counts.dtype != torch.int32 or counts.dim() != 4 or counts.shape[-1] != 3 or not counts.is_contiguous():
        raise L.TcowError('iou_means: contiguous int32 (n_seq, C, T, 3) counts expected')
    n_seq, C, T, _ = counts.shape
    out = torch.empty(12, dtype=torch.float32, device=counts.device)
    cnt = out[6:].view(torch.int32)
    L.check(L.lib().tcow_iou_means(_stream(), counts.data_ptr(), n_seq, C, T, out.data_ptr(), cnt.data_ptr()), 'tcow_iou_means')
    return out[:6], cnt


def snitch_weights(target, snitch_occl_by_ptr, frame_w, pos_count, class_balancing=True, hard_negative_factor=3.0):
    """target (B,Q,3,T,H,W) f32 (channel 0 is used), snitch_occl_by_ptr (B,Q,1,T,H,W) u8, frame_w (B,Q,T) f32, pos_count int32 [>=1]
    -> (B,Q,T,H,W) f32 pixel weights of the track loss (see tcow_snitch_weights)."""
    _need_cuda(target, snitch_occl_by_ptr, frame_w, pos_count)
    B, Q, C, T, H, W = target.shape
    if not target.is_contiguous() or not snitch_occl_by_ptr.is_contiguous() or target.dtype != torch.float32 or snitch_occl_by_ptr.dtype != torch.uint8:
        raise L.TcowError('snitch_weights: target must be contiguous f32 and snitch_occl_by_ptr contiguous u8')
    out = torch.empty(B, Q, T, H, W, dtype=torch.float32, device=target.device)
    fw = frame_w.to(torch.float32).contiguous()
    lib = L.lib()
    ws = workspace(lib.tcow_snitch_weights_workspace_bytes(B * Q * T, H, W), target.device, 'snitch_w')
    L.check(lib.tcow_snitch_weights(_stream(), B * Q, T, H, W, target.data_ptr(), C * T * H * W, snitch_occl_by_ptr.data_ptr(), fw.data_ptr(), pos_count.data_ptr(),
                                    1 if class_balancing else 0, float(hard_negative_factor), out.data_ptr(), ws.data_ptr(), ws.numel()), 'tcow_snitch_weights')
    return out
