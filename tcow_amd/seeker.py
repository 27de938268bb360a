"""Drop-in `Seeker` / `QueryMaskTracker` for the TCOW hot path, running on libtcow_hip (gfx950).

Mirrors the nn.Module surface of the reference (model/seeker.py:17-25, model/mask_tracker.py:24-142):
same constructor keywords, `forward(input_frames, query_mask) -> (output_mask, output_flags)`, and the same
251 state-dict keys / shapes (seeker.tracker_backbone.timesformer.model.*, seeker.tracker_post_linear.*,
seeker.flag_post_linear.*), so checkpoints (`net_seeker`, train.py:281 / eval/inference.py:52) load with
strict=True.  The stock nn.Linear / nn.Conv2d / nn.LayerNorm objects below are parameter containers only:
their forward is never called; all arithmetic runs in the HIP kernels through one autograd.Function whose
backward is hand-written (no autograd graph inside the model).

precision:
  'bf16' (default)  bf16 GEMM/attention operands, f32 accumulation, f32 residual stream / LayerNorm / softmax.
  'fp16'            the bf16 mode's kernels built for IEEE binary16 storage (libtcow_hip_fp16.so: 11 instead of 8 significand bits,
                    same speed): mask logits within 1e-3 of the fp32 reference (measured ~6e-4 at BASELINE configs[1]); the backward
                    runs on gradients scaled by a power of two (exact; `loss_scale`, chosen per backward on the device) to stay inside binary16's range.
  'fp32'            everything f32 on the exact-f32 MFMA/FMA kernels: the parity mode (mask logits within 1e-3
                    of the fp32 reference; measured ~1e-6).
  'bf16x3'          the fp32 mode with its GEMMs on the bf16 matrix cores: every f32 operand is split into two bf16
                    terms in registers and a product is three MFMAs (hi*hi + hi*lo + lo*hi, csrc/gemm_x3.hip) --
                    ~1e-5 relative per product; mask logits within ~1e-5 of the reference at twice the fp32 mode's rate.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from . import ops
from ._lib import TcowError

TIMESFORMER_MEAN = 0.45
TIMESFORMER_STD = 0.225


# ----------------------------------------------------------------------------------------------- containers

def _trunc_normal_(t, std=.02):
    with torch.no_grad():
        return nn.init.trunc_normal_(t, mean=0., std=std, a=-2., b=2.)   # vit_utils.py:25-76 defaults a=-2,b=2


class _Attention(nn.Module):                     # vit.py:64-76
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):                           # vit.py:45-53
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):                         # vit.py:126-153
    def __init__(self, dim, mlp_ratio, divided=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim)
        if divided:                              # temporal parameters exist only for divided_space_time (vit.py:140-146)
            self.temporal_norm1 = nn.LayerNorm(dim, eps=1e-6)
            self.temporal_attn = _Attention(dim)
            self.temporal_fc = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(nn.Module):                    # vit.py:220-233
    def __init__(self, patch, in_chans, dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)


class _VisionTransformer(nn.Module):             # vit.py:244-306
    def __init__(self, img_size, patch, in_chans, dim, depth, mlp_ratio, num_frames, divided=True):
        super().__init__()
        self.embed_dim = dim
        self.patch_embed = _PatchEmbed(patch, in_chans, dim)
        n_patches = (img_size[0] // patch) * (img_size[1] // patch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patches + 1, dim))
        self.time_embed = nn.Parameter(torch.zeros(1, num_frames, dim))      # stays zero at init (vit.py:268)
        self.blocks = nn.ModuleList([_Block(dim, mlp_ratio, divided) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        _trunc_normal_(self.pos_embed)
        _trunc_normal_(self.cls_token)
        for m in self.modules():                                             # vit.py:299-306
            if isinstance(m, nn.Linear):
                _trunc_normal_(m.weight)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        # vit.py:289-297: the loop counts the ModuleList itself as "Block 0", so every real block is zeroed.
        for blk in self.blocks if divided else []:
            nn.init.constant_(blk.temporal_fc.weight, 0)
            nn.init.constant_(blk.temporal_fc.bias, 0)


class _TimeSformer(nn.Module):                   # vit.py:416-459
    def __init__(self, **kw):
        super().__init__()
        self.model = _VisionTransformer(**kw)


class _Backbone(nn.Module):                      # vision_tf.py:27-66
    def __init__(self, **kw):
        super().__init__()
        self.timesformer = _TimeSformer(**kw)
        self.output_feature_dim = self.timesformer.model.embed_dim


_DEPTH_GEOMETRY = {12: (768, 12), 18: (896, 14), 24: (1024, 16)}      # vit.py:424-449


# ----------------------------------------------------------------------------------------------- the module

class QueryMaskTracker(nn.Module):
    """mask_tracker.py:24-142.  Extra keywords (not in the reference): `embed_dim`, `num_heads` to build
    geometries other than depth 12/18/24 (the reference raises for those, vit.py:449), `precision`."""

    def __init__(self, logger, num_total_frames=24, num_visible_frames=16, frame_height=224, frame_width=288,
                 tracker_pretrained=False, attention_type='divided_space_time', patch_size=16, causal_attention=False,
                 norm_embeddings=False, drop_path_rate=0.1, network_depth=12, track_map_stride=4,
                 track_map_resize='bilinear', query_channels=1, output_channels=3, flag_channels=3,
                 embed_dim=None, num_heads=None, precision='bf16'):
        super().__init__()
        self.logger = logger
        self.num_total_frames = num_total_frames
        self.num_visible_frames = num_visible_frames
        self.frame_height = frame_height
        self.frame_width = frame_width
        self.attention_type = attention_type
        self.patch_size = patch_size
        self.causal_attention = int(causal_attention)
        self.norm_embeddings = bool(norm_embeddings)
        self.drop_path_rate = drop_path_rate
        self.network_depth = network_depth
        self.track_map_stride = track_map_stride
        self.track_map_resize = track_map_resize
        self.query_channels = query_channels
        self.output_channels = output_channels
        self.flag_channels = flag_channels
        self.input_channels = 3 + query_channels
        self.set_precision(precision)

        from .checkpoint import parse_tracker_pretrained
        self.tracker_pretrained, self.pretrained_path = parse_tracker_pretrained(tracker_pretrained)   # mask_tracker.py:55-67
        if logger is not None:
            logger.info(f'(QueryMaskTracker) tracker_pretrained: {self.tracker_pretrained} '
                        f'pretrained_path: {self.pretrained_path}')

        assert attention_type in ['divided_space_time', 'space_only', 'joint_space_time']      # vit.py:133
        if attention_type == 'space_only':
            # the reference cannot run this setting either: DenseTimeSformer.forward reads model.time_embed (vision_tf.py:127-132), which a
            # space_only VisionTransformer does not create (vit.py:263-265) -> AttributeError on the first forward
            raise TcowError("attention_type='space_only' is not runnable through Seeker (vision_tf.py:127 needs time_embed, vit.py:263-265)")
        if query_channels != 1:
            raise TcowError('query_channels must be 1 (mask_tracker.py:105)')
        if embed_dim is None or num_heads is None:
            if network_depth not in _DEPTH_GEOMETRY:
                raise ValueError(f'Invalid network depth {network_depth}, must be one of 12, 18, 24.')  # vit.py:449
            embed_dim, num_heads = _DEPTH_GEOMETRY[network_depth]
        if embed_dim != num_heads * 64:
            raise TcowError(f'head_dim must be 64 (embed_dim={embed_dim}, num_heads={num_heads})')
        self.embed_dim, self.num_heads = embed_dim, num_heads
        assert frame_height % patch_size == 0                             # mask_tracker.py:89-90
        assert frame_width % patch_size == 0
        if track_map_stride > 1 and patch_size % track_map_stride != 0:
            raise TcowError('track_map_stride must divide patch_size')
        if track_map_resize not in ('bilinear', 'nearest'):
            raise TcowError(f'unsupported track_map_resize {track_map_resize}')

        self.tracker_backbone = _Backbone(img_size=(frame_height, frame_width), patch=patch_size,
                                          in_chans=self.input_channels, dim=embed_dim, depth=network_depth,
                                          mlp_ratio=4, num_frames=num_total_frames, divided=(attention_type == 'divided_space_time'))
        self.use_feature_dim = embed_dim
        self.tracker_post_linear = nn.Linear(embed_dim, output_channels * patch_size * patch_size)
        if flag_channels > 0:
            self.flag_post_linear = nn.Linear(embed_dim, flag_channels)
        if self.tracker_pretrained:
            from .checkpoint import load_pretrained_vit
            load_pretrained_vit(self, self.pretrained_path, logger)
        self._wcache = {}
        self._gbufs = {}
        # dynamic loss scale of precision='fp16' (engine.run_backward): log2 of the target magnitude of the largest gradient seed element.  A
        # NON-persistent buffer: it follows .to() / DataParallel replication like any buffer, stays out of the 251-key state dict, and
        # save_tcow_checkpoint / resume_tcow_checkpoint carry it beside the optimizer state.
        self.register_buffer('ls_log2', torch.full((), -2.0, dtype=torch.float32), persistent=False)
        # persistent_grads=True: the backward writes parameter gradients into flat per-bucket buffers that live across steps and
        # hands them to p.grad directly (same pointers every step: no autograd copies, the fused optimizer's pointer table stays
        # valid).  Gradients are then OVERWRITTEN, not accumulated, by each backward -- use only with one backward per step.
        self.persistent_grads = False
        self.forced_drop_masks = None     # tests can inject explicit DropPath keep masks
        self.grad_hook = None             # optional callable(bucket_name, tensors) fired during backward (DDP)

    # ---- configuration helpers
    def set_precision(self, precision):
        if precision not in ('bf16', 'fp16', 'fp32', 'bf16x3'):
            raise ValueError("precision must be 'bf16', 'fp16', 'fp32' or 'bf16x3'")
        self.precision = precision
        self.mode = ops.BF16 if precision == 'bf16' else ops.FP16 if precision == 'fp16' else ops.F32
        if 'ls_log2' in self._buffers:
            with torch.no_grad():
                self.ls_log2.fill_(-2.0)
        self.loss_scale = 'dynamic'                                             # fp16 only: power-of-two factor on the backward's gradients (engine.run_backward); a number = static
        self.gemm_mode = ops.F32X3 if precision == 'bf16x3' else self.mode     # GEMM arithmetic; storage / every other kernel follow `mode`
        self._wcache = {}
        self.__dict__.pop('_wreg', None); self.__dict__.pop('_wtab', None); self.__dict__.pop('_foldreg', None)
        self.__dict__['_wreg_gen'] = self.__dict__.get('_wreg_gen', 0) + 1
        return self

    def invalidate_weight_cache(self):
        """Call after updating parameters through raw pointers (tcow_amd.optim.FusedAdamWClip does): in-place torch ops bump the
        parameters' autograd version counters, which the operand cache checks on its own."""
        self._wepoch = getattr(self, '_wepoch', 0) + 1
        if self.__dict__.get('_wreg'):
            from . import engine
            engine.refresh_weights(self)       # all operand copies of the next step in one launch

    def _drop_operand_caches(self):
        """Everything derived from the parameters' storage: 16-bit W / W^T copies, their registries and pointer tables, the folded
        products and the persistent gradient buffers.  They are keyed by id(parameter), which survives a `.data` swap."""
        self._wcache = {}
        self.__dict__.pop('_wreg', None); self.__dict__.pop('_wtab', None); self.__dict__.pop('_foldreg', None)
        self.__dict__['_wreg_gen'] = self.__dict__.get('_wreg_gen', 0) + 1
        self._gbufs = {}

    def _apply(self, fn, *a, **kw):
        # .to() / .cuda() / .half() swap the parameters' .data (neither id() nor ._version changes): every cache that holds device
        # pointers or copies of the old storage must go, and so must the cached parameter list
        self._drop_operand_caches()
        self.invalidate_param_cache()
        return super()._apply(fn, *a, **kw)

    @property
    def vit(self):
        return self.tracker_backbone.timesformer.model

    def geometry(self, B):
        P = self.patch_size
        Hp, Wp = self.frame_height // P, self.frame_width // P
        N = Hp * Wp
        return dict(B=B, T=self.num_total_frames, Hp=Hp, Wp=Wp, N=N, S=N + 1, D=self.embed_dim, heads=self.num_heads,
                    P=P, M=B * self.num_total_frames * (N + 1))

    def param_list(self):
        """Fixed order of the parameters the autograd.Function sees.  Cached: walking ~250 module attributes costs 0.4 ms per call.  .to() /
        load_state_dict keep the Parameter objects; code that ASSIGNS a new Parameter inside the module tree must call invalidate_param_cache()
        (the first and last entries are re-checked on every call, which catches wholesale replacement)."""
        cached = self.__dict__.get('_param_list_cache')
        if cached is not None and cached[0] is self.vit.cls_token and cached[-1] is (self.flag_post_linear if self.flag_channels > 0 else self.tracker_post_linear).bias:
            return cached
        ps = self._param_list_uncached()
        self.__dict__['_param_list_cache'] = ps
        return ps

    def invalidate_param_cache(self):
        self.__dict__.pop('_param_list_cache', None)

    def _param_list_uncached(self):
        v = self.vit
        ps = [v.cls_token, v.pos_embed, v.time_embed, v.patch_embed.proj.weight, v.patch_embed.proj.bias]
        for b in v.blocks:                   # (order = engine._layout)
            if self.attention_type == 'divided_space_time':
                ps += [b.temporal_norm1.weight, b.temporal_norm1.bias, b.temporal_attn.qkv.weight, b.temporal_attn.qkv.bias,
                       b.temporal_attn.proj.weight, b.temporal_attn.proj.bias, b.temporal_fc.weight, b.temporal_fc.bias]
            ps += [b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.qkv.bias, b.attn.proj.weight, b.attn.proj.bias,
                   b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias]
        ps += [v.norm.weight, v.norm.bias, self.tracker_post_linear.weight, self.tracker_post_linear.bias]
        if self.flag_channels > 0:
            ps += [self.flag_post_linear.weight, self.flag_post_linear.bias]
        return ps

    # ---- forward
    def forward(self, input_frames, query_mask):
        """mask_tracker.py:92-142: (B,3,T,Hf,Wf), (B,1,T,Hf,Wf) -> (B,C,T,Hf,Wf) logits, (B,T,F) flags.
        Extension: query_mask may carry Qs masks per clip, (B*Qs,1,T,Hf,Wf) with clip b's queries at rows b*Qs..b*Qs+Qs-1;
        the outputs then have B*Qs rows and equal Qs separate calls with the same frames (pipeline.py:134-158), but the rgb part of the
        patch embedding is computed once per clip."""
        (B, _, T, Hf, Wf) = input_frames.shape
        assert query_mask.shape[1] == 1                                   # mask_tracker.py:105
        assert query_mask.shape[0] % B == 0 and query_mask.shape[2:] == input_frames.shape[2:]
        assert T == self.num_total_frames                                 # vision_tf.py:96
        assert Hf == self.frame_height and Wf == self.frame_width         # vision_tf.py:97 (W' check)
        assert self.attention_type == 'divided_space_time' or self.causal_attention == 0   # vit.py:160
        if not input_frames.is_cuda:
            raise TcowError('Seeker (tcow_amd) runs on the GPU only: move inputs and module to cuda')
        if query_mask.device != input_frames.device or self.vit.pos_embed.device != input_frames.device:
            raise TcowError(f'Seeker (tcow_amd): inputs ({input_frames.device}, {query_mask.device}) and parameters ({self.vit.pos_embed.device}) must share one device')
        if getattr(self, '_is_replica', False):
            # torch.nn.DataParallel (train.py:222-223) re-creates shallow replicas every forward: their dicts alias the original's, so give
            # each replica private operand / gradient caches for this call instead of growing the shared ones with dead entries
            self._wcache, self._gbufs = {}, {}
            self.__dict__.pop('_wreg', None); self.__dict__.pop('_wtab', None); self.__dict__.pop('_foldreg', None)
            self.__dict__['_wreg_gen'] = self.__dict__.get('_wreg_gen', 0) + 1
        rgb = input_frames.to(torch.float32).contiguous()                 # mask_tracker.py:103-104 (inputs not mutated)
        qm = query_mask.to(torch.float32).contiguous()
        from .engine import SeekerFunction
        with torch.cuda.device(input_frames.device):                      # kernels launch on the tensors' device and its current stream
            out_mask, out_flags = SeekerFunction.apply(self, rgb, qm, *self.param_list())
        if self.flag_channels <= 0:
            out_flags = None
        return (out_mask, out_flags)


class Seeker(nn.Module):
    """model/seeker.py:17-25."""

    shares_rgb = True          # forward accepts (B, ...) frames with (B * Qs, ...) query masks (see QueryMaskTracker.forward)

    def __init__(self, logger, **kwargs):
        super().__init__()
        self.logger = logger
        self.seeker_args = {k: v for k, v in kwargs.items() if k != 'precision'}     # what train.py:186-205 stores as checkpoint['seeker_args']
        self.seeker = QueryMaskTracker(logger, **kwargs)

    def forward(self, *args):
        return self.seeker(*args)
