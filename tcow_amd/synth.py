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
                  pretrained_norm=False, mlp_ratio=4, attention_type='divided_space_time'):
    """Plain dict describing one Seeker geometry (defaults = BASELINE.json configs[1])."""
    return dict(num_total_frames=num_total_frames, frame_height=frame_height, frame_width=frame_width,
                patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                causal_attention=int(causal_attention), norm_embeddings=bool(norm_embeddings),
                query_channels=query_channels, output_channels=output_channels, flag_channels=flag_channels,
                track_map_stride=track_map_stride, track_map_resize=track_map_resize,
                pretrained_norm=bool(pretrained_norm), mlp_ratio=mlp_ratio, attention_type=attention_type)


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
        if cfg.get('attention_type', 'divided_space_time') == 'divided_space_time':      # vit.py:140-146
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


def trained_scale_state_dict(sd, qk_gain=3.0, w_gain=1.5, head_gain=1.0):
    """A state dict with the MAGNITUDES of a trained checkpoint instead of the init's (the synthetic weights are trunc-normal(0.02): attention
    scores of std ~0.3, i.e. near-uniform softmax rows, and mask logits of std ~0.15 -- a trained TCOW checkpoint has peaked attention and
    logits of |x| ~ 10).  The q and k rows of every qkv weight / bias are multiplied by `qk_gain` (scores x qk_gain^2), every other Linear
    weight of the blocks (v rows, proj, temporal_fc, fc1, fc2) by `w_gain`, and the mask head (tracker_post_linear weight AND bias, so the
    logits scale exactly linearly) by `head_gain`.  Used by the parity-at-scale fixture g16 and the bench's `trained_scale` case: 16-bit
    error is relative, so what must be shown is that max|d| / std(logits) and the binary masks hold when the logits are 30x larger."""
    out = OrderedDict()
    for k, v in sd.items():
        v = np.array(v, dtype=np.float32, copy=True)
        leaf = k.split('.')[-1]
        if '.qkv.' in k:
            D = v.shape[0] // 3
            v[:2 * D] *= np.float32(qk_gain)
            if leaf == 'weight':
                v[2 * D:] *= np.float32(w_gain)
        elif leaf == 'weight' and ('.blocks.' in k) and v.ndim == 2:
            v *= np.float32(w_gain)
        elif k.startswith('seeker.tracker_post_linear.'):
            v *= np.float32(head_gain)
        out[k] = np.ascontiguousarray(v)
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


def make_kubric_batch(B, T, H, W, seed=900, n_objects=5, M=36):
    """Synthetic Kubric-shaped `data_retval` (data/data_kubric.py:133-155,425-432,
    data/data_utils.py:95-241): K moving rectangles in painter's order (later = in front), amodal masks padded to
    M = 36 instances (data/data.py:101), occlusion fractions (occluded, visible-area, total-area fractions), an
    occlusion / containment DAG [containee/occludee, container/occluder] with channels (containment, direct
    occlusion, frontmost occlusion), and a desirability table whose column 0 ranks the queries."""
    K = n_objects
    r = _rng(seed, f'kubric{B}x{T}x{H}x{W}x{K}')
    rgb = r.random(size=(B, 3, T, H, W), dtype=np.float32)
    segm = np.zeros((B, 1, T, H, W), np.uint8)
    div = np.zeros((B, M, T, H, W), np.uint8)
    boxes = np.zeros((B, K, T, 4), np.int64)
    for b in range(B):
        # the last (frontmost) object is large: it acts as occluder / container; small objects start (mostly) outside it
        # so that they are valid queries at t = 0, and object 0 drifts behind it so that occlusion / containment occur.
        bh = int(r.integers(H // 2, 3 * H // 4)); bw = int(r.integers(W // 2, 3 * W // 4))
        by0 = r.uniform(0, H - bh); bx0 = r.uniform(0, W - bw)
        bvy = r.uniform(-1, 1) * H / (4.0 * T); bvx = r.uniform(-1, 1) * W / (4.0 * T)
        params = []
        for k in range(K - 1):
            h = int(r.integers(max(H // 8, 2), max(H // 4, 3))); w = int(r.integers(max(W // 8, 2), max(W // 4, 3)))
            for _ in range(200):
                y0 = r.uniform(0, H - h); x0 = r.uniform(0, W - w)
                oy = max(0.0, min(y0 + h, by0 + bh) - max(y0, by0)); ox = max(0.0, min(x0 + w, bx0 + bw) - max(x0, bx0))
                if oy * ox < 0.5 * h * w:
                    break
            if k == 0:      # ends inside the big object's final box
                ye = np.clip(by0 + bvy * (T - 1), 0, H - bh) + (bh - h) / 2.0; xe = np.clip(bx0 + bvx * (T - 1), 0, W - bw) + (bw - w) / 2.0
                vy = (ye - y0) / max(T - 1, 1); vx = (xe - x0) / max(T - 1, 1)
            else:
                vy = r.uniform(-1, 1) * H / (2.0 * T); vx = r.uniform(-1, 1) * W / (2.0 * T)
            params.append((h, w, y0, x0, vy, vx))
        params.append((bh, bw, by0, bx0, bvy, bvx))
        for k, (h, w, y0, x0, vy, vx) in enumerate(params):
            col = r.random(size=3, dtype=np.float32)
            for t in range(T):
                y = int(np.clip(y0 + vy * t, 0, H - h)); x = int(np.clip(x0 + vx * t, 0, W - w))
                boxes[b, k, t] = (y, x, h, w)
                div[b, k, t, y:y + h, x:x + w] = 1
                segm[b, 0, t, y:y + h, x:x + w] = k + 1
                rgb[b, :, t, y:y + h, x:x + w] = 0.5 * rgb[b, :, t, y:y + h, x:x + w] + 0.5 * col[:, None, None]
    occl_fracs = np.zeros((B, M, T, 3), np.float32)
    dag = np.zeros((B, T, M, M, 3), np.float32)
    for b in range(B):
        for t in range(T):
            for i in range(K):
                am = div[b, i, t] == 1
                area = float(am.sum())
                vis = float(((segm[b, 0, t] == i + 1) & am).sum())
                occl_fracs[b, i, t] = (1.0 - vis / max(area, 1.0), vis / (H * W), area / (H * W))
                for j in range(K):
                    if j == i:
                        continue
                    cover = float((am & (segm[b, 0, t] == j + 1)).sum()) / max(area, 1.0)     # j visibly in front of i
                    dag[b, t, i, j, 1] = cover
                    dag[b, t, i, j, 2] = cover
                    yi, xi, hi, wi = boxes[b, i, t]; yj, xj, hj, wj = boxes[b, j, t]
                    if j > i and yi >= yj and xi >= xj and yi + hi <= yj + hj and xi + wi <= xj + wj:
                        dag[b, t, i, j, 0] = 1.0                                               # box of i inside box of j
    des = -np.ones((B, M, 7), np.float64)
    for b in range(B):
        for k in range(K):
            vis0 = float(occl_fracs[b, k, 0, 1])
            des[b, k, 0] = (occl_fracs[b, k, :, 0].mean() + 0.1 * k) if vis0 > 0.002 else -1.0   # invisible at t=0 -> never sampled
            des[b, k, 1:] = r.random(size=6)
    return {
        'source_name': ['kubric'] * B,
        'within_batch_idx': np.arange(B, dtype=np.int64),
        'scene_dp': ['synthetic'] * B,
        'kubric_retval': {
            'pv_rgb_tf': rgb, 'pv_segm_tf': segm, 'pv_div_segm_tf': div,
            'pv_inst_count': np.full((B, 1), K, np.int32),
            'traject_retval_tf': {'query_time': np.zeros((B,), np.int64), 'occl_fracs_tf': occl_fracs,
                                  'occl_cont_dag_tf': dag, 'desirability_tf': des},
        },
    }


def make_plugin_video(Tv, H, W, seed=900, n_objects=5, annot_every=5, query_frames=(0,)):
    """Synthetic 'plugin' video (data/data_plugin.py:52-140): Tv rgb frames, the visible segmentation and amodal instance masks of a
    Kubric-shaped scene, query annotations at `query_frames` and sparse target annotations every `annot_every`-th frame -- what
    the reference reads from <video> + *_query.png / *_snitch.png files, as arrays."""
    kb = make_kubric_batch(1, Tv, H, W, seed=seed, n_objects=n_objects)['kubric_retval']
    return dict(rgb=kb['pv_rgb_tf'][0], segm=kb['pv_segm_tf'][0, 0], div=kb['pv_div_segm_tf'][0, :n_objects],
                annot_frames=list(range(0, Tv, annot_every)), query_frames=list(query_frames))


HOST_KEYS = ('query_time', 'pv_inst_count', 'desirability_tf')


def to_torch_tree(x, device=None, host_keys=()):
    """numpy leaves -> torch tensors (optionally on `device`), lists / strings untouched.  Leaves whose key is in
    `host_keys` stay on the host: the pipeline reads them to decide control flow (which instances to query, the query
    frame), exactly as the reference reads them from its CPU-side DataLoader batch (pipeline.py:120-140)."""
    import torch
    if isinstance(x, dict):
        return {k: to_torch_tree(v, None if k in host_keys else device, host_keys) for k, v in x.items()}
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(x)
        return t.to(device) if device is not None else t
    return x
