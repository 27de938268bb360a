"""Deterministic synthetic weights and clips for the Seeker hot path.

Everything here is generated with numpy's counter-based Philox generator keyed by (seed, crc32(name)),
so the same tensors come out on every machine and independently of generation order or of torch's RNG.
The reference has no such generator: it trains from an ImageNet ViT checkpoint
(third_party/TimeSformer/timesformer/models/helpers.py:100-205) on Kubric clips (data/data_kubric.py);
neither is available offline, so benchmarks and parity fixtures use these tensors instead.

State-dict keys and shapes follow the reference's `Seeker.state_dict()` (model/seeker.py:17-25,
model/mask_tracker.py:70-87, third_party/TimeSformer/timesformer/models/vit.py:244-306).
"""
import zlib
from collections import OrderedDict

import numpy as np

PREFIX = 'seeker.tracker_backbone.timesformer.model.'


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(size=shape, dtype=np.float32) * std
    return np.clip(x, -2.0 * std, 2.0 * std).astype(np.float32)


def seeker_config(num_total_frames=30, frame_height=240, frame_width=320, patch_size=16, embed_dim=768,
                  depth=12, num_heads=12, causal_attention=1, norm_embeddings=False, query_channels=1,
                  output_channels=3, flag_channels=3, track_map_stride=4, track_map_resize='bilinear',
                  pretrained_norm=False, mlp_ratio=4):
    """Plain dict describing one Seeker geometry (defaults = BASELINE.json configs[1])."""
    return dict(num_total_frames=num_total_frames, frame_height=frame_height, frame_width=frame_width,
                patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                causal_attention=int(causal_attention), norm_embeddings=bool(norm_embeddings),
                query_channels=query_channels, output_channels=output_channels, flag_channels=flag_channels,
                track_map_stride=track_map_stride, track_map_resize=track_map_resize,
                pretrained_norm=bool(pretrained_norm), mlp_ratio=mlp_ratio)


def state_dict_shapes(cfg):
    """Ordered {key: shape} of the reference state dict for this geometry (251 entries at depth 12)."""
    D = cfg['embed_dim']; P = cfg['patch_size']; T = cfg['num_total_frames']
    N = (cfg['frame_height'] // P) * (cfg['frame_width'] // P)
    Ci = 3 + cfg['query_channels']; Hd = int(D * cfg['mlp_ratio'])
    s = OrderedDict()
    s[PREFIX + 'cls_token'] = (1, 1, D)
    s[PREFIX + 'pos_embed'] = (1, N + 1, D)
    s[PREFIX + 'time_embed'] = (1, T, D)
    s[PREFIX + 'patch_embed.proj.weight'] = (D, Ci, P, P)
    s[PREFIX + 'patch_embed.proj.bias'] = (D,)
    for i in range(cfg['depth']):
        b = PREFIX + f'blocks.{i}.'
        s[b + 'norm1.weight'] = (D,); s[b + 'norm1.bias'] = (D,)
        s[b + 'attn.qkv.weight'] = (3 * D, D); s[b + 'attn.qkv.bias'] = (3 * D,)
        s[b + 'attn.proj.weight'] = (D, D); s[b + 'attn.proj.bias'] = (D,)
        s[b + 'temporal_norm1.weight'] = (D,); s[b + 'temporal_norm1.bias'] = (D,)
        s[b + 'temporal_attn.qkv.weight'] = (3 * D, D); s[b + 'temporal_attn.qkv.bias'] = (3 * D,)
        s[b + 'temporal_attn.proj.weight'] = (D, D); s[b + 'temporal_attn.proj.bias'] = (D,)
        s[b + 'temporal_fc.weight'] = (D, D); s[b + 'temporal_fc.bias'] = (D,)
        s[b + 'norm2.weight'] = (D,); s[b + 'norm2.bias'] = (D,)
        s[b + 'mlp.fc1.weight'] = (Hd, D); s[b + 'mlp.fc1.bias'] = (Hd,)
        s[b + 'mlp.fc2.weight'] = (D, Hd); s[b + 'mlp.fc2.bias'] = (D,)
    s[PREFIX + 'norm.weight'] = (D,); s[PREFIX + 'norm.bias'] = (D,)
    s['seeker.tracker_post_linear.weight'] = (cfg['output_channels'] * P * P, D)
    s['seeker.tracker_post_linear.bias'] = (cfg['output_channels'] * P * P,)
    if cfg['flag_channels'] > 0:
        s['seeker.flag_post_linear.weight'] = (cfg['flag_channels'], D)
        s['seeker.flag_post_linear.bias'] = (cfg['flag_channels'],)
    return s


def make_state_dict(cfg, seed=900):
    """Deterministic fp32 numpy state dict. Unlike the reference's stock init (vit.py:284-297, where
    every temporal_fc and time_embed is zero) all tensors are non-zero so that temporal attention, the
    causal mask and every bias actually influence the output."""
    out = OrderedDict()
    for name, shape in state_dict_shapes(cfg).items():
        r = _rng(seed, name)
        leaf = name.split('.')[-1]
        if 'norm' in name.split('.')[-2] and leaf == 'weight':
            v = 1.0 + 0.1 * r.standard_normal(size=shape, dtype=np.float32)
        elif 'norm' in name.split('.')[-2] and leaf == 'bias':
            v = 0.05 * r.standard_normal(size=shape, dtype=np.float32)
        elif leaf == 'bias':
            v = 0.02 * r.standard_normal(size=shape, dtype=np.float32)
        elif name.endswith('patch_embed.proj.weight'):
            fan_in = shape[1] * shape[2] * shape[3]
            v = r.uniform(-1.0, 1.0, size=shape).astype(np.float32) / np.float32(np.sqrt(fan_in))
        elif leaf in ('cls_token', 'pos_embed', 'time_embed'):
            v = _trunc_normal(r, shape, 0.02)
        else:  # Linear weights: trunc_normal(std=.02) like vit.py:299-303, scaled so activations stay O(1)
            v = _trunc_normal(r, shape, 0.02)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def make_clip(B, T, H, W, seed=900, n_objects=4):
    """Synthetic Kubric-shaped clip: rgb ~ U[0,1) (B,3,T,H,W) f32 plus K moving rectangles drawn in
    painter's order. Returns dict with rgb, visible segmentation `segm` (B,1,T,H,W) uint8 (0 = bg,
    k+1 = visible instance k, cf. pipeline.py:94) and amodal masks `div_segm` (B,K,T,H,W) uint8."""
    r = _rng(seed, f'clip{B}x{T}x{H}x{W}')
    rgb = r.random(size=(B, 3, T, H, W), dtype=np.float32)
    segm = np.zeros((B, 1, T, H, W), np.uint8)
    div = np.zeros((B, n_objects, T, H, W), np.uint8)
    for b in range(B):
        for k in range(n_objects):
            h = int(r.integers(max(H // 8, 2), max(H // 3, 3))); w = int(r.integers(max(W // 8, 2), max(W // 3, 3)))
            y0 = r.uniform(0, H - h); x0 = r.uniform(0, W - w)
            vy = r.uniform(-1, 1) * H / (3.0 * T); vx = r.uniform(-1, 1) * W / (3.0 * T)
            col = r.random(size=3, dtype=np.float32)
            for t in range(T):
                y = int(np.clip(y0 + vy * t, 0, H - h)); x = int(np.clip(x0 + vx * t, 0, W - w))
                div[b, k, t, y:y + h, x:x + w] = 1
                segm[b, 0, t, y:y + h, x:x + w] = k + 1
                rgb[b, :, t, y:y + h, x:x + w] = 0.5 * rgb[b, :, t, y:y + h, x:x + w] + 0.5 * col[:, None, None]
    return dict(rgb=rgb, segm=segm, div_segm=div)


def make_query_mask(clip, instance=0, query_time=0):
    """Query mask = visible pixels of `instance` at `query_time` only, zero elsewhere
    (data/data_utils.py:431 builds the same thing from pv_segm)."""
    segm = clip['segm']
    q = np.zeros(segm.shape, np.float32)
    q[:, :, query_time] = (segm[:, :, query_time] == instance + 1).astype(np.float32)
    return q
