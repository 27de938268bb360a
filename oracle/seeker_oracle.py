"""ORACLE (test infrastructure, not product): CPU restatement of the TCOW Seeker forward.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
The product path (`tcow_amd/`) never does; it fails loudly when its HIP library is missing.

This is our own layout-explicit restatement, in plain torch fp32 (or fp64) CPU ops, of
  model/seeker.py:17-25, model/mask_tracker.py:92-142, model/vision_tf.py:68-169 and
  third_party/TimeSformer/timesformer/models/vit.py:45-241 (Mlp, Attention, Block, PatchEmbed)
of the reference (paths relative to /root/reference).  The residual stream is kept as X[b,t,n,:]
(frames outer, patches inner) plus a separate CLS[b,:], instead of the reference's
(B, 1+N*T, D) n-major/t-minor token list; `ref_token_index` gives the bit-exact map between them.

Parity pin: `oracle/make_golden.py` imports the real reference in the build container and checks this
restatement element-wise against it (fp32, <=1e-5 abs on mask logits), then writes the fixtures under
tests/golden/ that `tests/test_oracle_golden.py` re-checks everywhere (the reference itself never
travels).  The reference has no tests of its own for this path, so the pins are
those goldens plus the known-answer properties in tests/test_oracle_properties.py.
"""
import math

import torch
import torch.nn.functional as F

PREFIX = 'seeker.tracker_backbone.timesformer.model.'
TIMESFORMER_MEAN = 0.45   # model/vision_tf.py:23
TIMESFORMER_STD = 0.225   # model/vision_tf.py:24


# ----------------------------------------------------------------------------- index maps (bit-exact)

def ref_token_index(t, n, T):
    """Reference token position of (frame t, patch n) in its (B, 1+N*T, D) stream:
    1 + n*T + t with n = h'*W' + w' (model/vision_tf.py:122-138, rearrange '(b n) t m -> b (n t) m')."""
    return 1 + n * T + t


def patch_pixel_index(c, py, px, P):
    """Column of the patch-embed GEMM / row inside the un-patchify for (channel, row, col) of a patch:
    c*P*P + py*P + px (Conv2d weight flattening vit.py:233; 'B T H W (C h w)' mask_tracker.py:114)."""
    return (c * P + py) * P + px


# ----------------------------------------------------------------------------- building blocks

def layer_norm(x, w, b, eps=1e-6):
    """nn.LayerNorm(D, eps=1e-6) (vit.py:428 norm_layer; biased variance, eps inside the sqrt)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    """nn.GELU() exact erf form (vit.py:46 act_layer)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def causal_keep_mask(L, ca, device=None):
    """Boolean (L, L) keep-mask [query, key] of Attention.forward (vit.py:93-99):
    ca in {1,2}: key <= query; ca >= 3: key <= query + (ca-2); ca <= 0: everything."""
    if ca <= 0:
        return torch.ones(L, L, dtype=torch.bool, device=device)
    diag = 0 if ca <= 2 else ca - 2
    return torch.ones(L, L, dtype=torch.bool, device=device).tril(diagonal=diag)


def attention(x, wqkv, bqkv, wproj, bproj, heads, ca=0):
    """Attention.forward (vit.py:78-123) on x (Bq, L, D). Returns proj(softmax(q k^T * d^-0.5 [mask]) v).
    Masked scores are set to -1e10 (not -inf) before the softmax, as the reference does (vit.py:99)."""
    Bq, L, D = x.shape
    d = D // heads
    qkv = (x @ wqkv.t() + bqkv).reshape(Bq, L, 3, heads, d).permute(2, 0, 3, 1, 4)   # vit.py:81-83
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (d ** -0.5)                                      # vit.py:88
    if ca > 0:
        keep = causal_keep_mask(L, ca, x.device)
        attn = attn.masked_fill(~keep, -1e10)                                           # vit.py:93-99
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(Bq, L, D)                                    # vit.py:109
    return o @ wproj.t() + bproj


def _nearest_resize_pos(pos, Hp, Wp):
    """Nearest-neighbour resize of the patch part of pos_embed (vision_tf.py:103-115)."""
    cls_pos = pos[:, 0:1]
    other = pos[0, 1:].t().unsqueeze(0)                       # (1, D, Nold)
    Pold = int(other.shape[2] ** 0.5)
    other = other.reshape(1, pos.shape[2], Pold, Pold)
    new = F.interpolate(other, size=(Hp, Wp), mode='nearest').flatten(2).transpose(1, 2)
    return torch.cat([cls_pos, new], dim=1)


def _nearest_resize_time(te, T):
    """Nearest-neighbour resize of time_embed (vision_tf.py:127-132)."""
    return F.interpolate(te.transpose(1, 2), size=T, mode='nearest').transpose(1, 2)


def drop_path_scale(keep_mask, rate):
    """DropPath (vit_utils.py:139-154) given an explicit 0/1 keep mask: x / keep_prob * mask."""
    return keep_mask / (1.0 - rate)


# ----------------------------------------------------------------------------- the forward

def seeker_forward(sd, cfg, input_frames, query_mask, drop_masks=None, taps=None):
    """Forward of Seeker / QueryMaskTracker (model/mask_tracker.py:92-142).

    sd: state dict (reference keys) of CPU tensors; cfg: tcow_amd.synth.seeker_config dict.
    drop_masks: None (eval) or dict {(block, 'temporal'|'spatial'|'mlp'): (keep 0/1 tensor, rate)} with
      shapes (B, N), (B, T), (B,) -- the per-row draws of DropPath at vit.py:172,186/208,216.
    taps: optional dict that receives intermediate tensors ('tokens_in', 'block{i}', 'features', 'pooled').
    Returns (output_mask (B,Cout,T,H,W) logits, output_flags (B,T,F) or None).
    """
    dt = sd[PREFIX + 'pos_embed'].dtype
    B, _, T, Hf, Wf = input_frames.shape
    P = cfg['patch_size']; D = cfg['embed_dim']; heads = cfg['num_heads']; ca = cfg['causal_attention']
    Hp, Wp = Hf // P, Wf // P
    N = Hp * Wp
    assert Hf % P == 0 and Wf % P == 0                                   # mask_tracker.py:89-90
    assert query_mask.shape[1] == 1                                      # mask_tracker.py:105
    assert T == cfg['num_total_frames']                                  # vision_tf.py:96

    # K0: input assembly (mask_tracker.py:102-108) + optional rgb normalisation (vision_tf.py:81-89).
    x_in = torch.cat([input_frames.to(dt), query_mask.to(dt)], dim=1)     # (B, 4, T, H, W)
    if cfg.get('pretrained_norm', False):
        x_in = x_in.clone()
        x_in[:, 0:3] = (x_in[:, 0:3] - TIMESFORMER_MEAN) / TIMESFORMER_STD
    Ci = x_in.shape[1]

    # K1: patch embed = GEMM over flattened patches (vit.py:233-241).
    Wpe = sd[PREFIX + 'patch_embed.proj.weight'].reshape(D, Ci * P * P)
    patches = x_in.reshape(B, Ci, T, Hp, P, Wp, P).permute(0, 2, 3, 5, 1, 4, 6).reshape(B, T, N, Ci * P * P)
    X = patches @ Wpe.t() + sd[PREFIX + 'patch_embed.proj.bias']         # (B, T, N, D)

    # K2: embeddings (vision_tf.py:99-138).
    pos = sd[PREFIX + 'pos_embed']
    if pos.shape[1] != N + 1:
        pos = _nearest_resize_pos(pos, (N + 1) // Wp, Wp)
    te = sd[PREFIX + 'time_embed']
    if te.shape[1] != T:
        te = _nearest_resize_time(te, T)
    CLS = (sd[PREFIX + 'cls_token'][0, 0] + pos[0, 0]).unsqueeze(0).expand(B, D).clone()   # (B, D)
    X = X + pos[0, 1:][None, None] + te[0][None, :, None]
    if taps is not None:
        taps['tokens_in'] = X.clone(); taps['cls_in'] = CLS.clone()

    def dp(i, kind, shape_ones):
        if drop_masks is None or (i, kind) not in drop_masks:
            return None
        keep, rate = drop_masks[(i, kind)]
        return drop_path_scale(keep.to(dt), rate)

    joint = cfg.get('attention_type', 'divided_space_time') == 'joint_space_time'
    for i in range(cfg['depth']):
        b = PREFIX + f'blocks.{i}.'
        if joint:
            # ---- joint space-time attention (vit.py:159-162): one sequence (cls, all N*T patch tokens) per clip, no mask, DropPath per sample
            assert ca == 0                                                   # vit.py:160
            allt = torch.cat([CLS[:, None, :], X.reshape(B, T * N, D)], dim=1)
            Y = attention(layer_norm(allt, sd[b + 'norm1.weight'], sd[b + 'norm1.bias']), sd[b + 'attn.qkv.weight'], sd[b + 'attn.qkv.bias'],
                          sd[b + 'attn.proj.weight'], sd[b + 'attn.proj.bias'], heads, 0)
            s = dp(i, 'spatial', None)
            if s is not None:
                Y = Y * s.reshape(B, 1, 1)
            allt = allt + Y
            CLS = allt[:, 0]
            X = allt[:, 1:].reshape(B, T, N, D)
        else:
            # ---- temporal half (vit.py:169-176)
            U = layer_norm(X, sd[b + 'temporal_norm1.weight'], sd[b + 'temporal_norm1.bias'])
            Ut = U.permute(0, 2, 1, 3).reshape(B * N, T, D)                   # '(b h w) t m'
            R = attention(Ut, sd[b + 'temporal_attn.qkv.weight'], sd[b + 'temporal_attn.qkv.bias'],
                          sd[b + 'temporal_attn.proj.weight'], sd[b + 'temporal_attn.proj.bias'], heads, ca)
            s = dp(i, 'temporal', None)
            if s is not None:
                R = R * s.reshape(B * N, 1, 1)                                # DropPath before temporal_fc
            R = R @ sd[b + 'temporal_fc.weight'].t() + sd[b + 'temporal_fc.bias']
            Xt = X + R.reshape(B, N, T, D).permute(0, 2, 1, 3)
            # ---- spatial half (vit.py:179-210)
            use_cls = ca in (0, 1)
            if use_cls:
                Vin = torch.cat([CLS[:, None, None, :].expand(B, T, 1, D), Xt], dim=2)       # (B,T,S,D)
            else:                                                              # ca>=2 or ca==-1 (vit.py:202-208)
                Vin = Xt
            S = Vin.shape[2]
            Y = attention(layer_norm(Vin, sd[b + 'norm1.weight'], sd[b + 'norm1.bias']).reshape(B * T, S, D),
                          sd[b + 'attn.qkv.weight'], sd[b + 'attn.qkv.bias'],
                          sd[b + 'attn.proj.weight'], sd[b + 'attn.proj.bias'], heads, 0).reshape(B, T, S, D)
            s = dp(i, 'spatial', None)
            if s is not None:
                Y = Y * s.reshape(B, T, 1, 1)
            if use_cls:
                cls_rows = Y[:, :, 0]                                          # (B, T, D)
                cls_out = cls_rows.mean(dim=1) if ca == 0 else cls_rows[:, 0]  # vit.py:193-198
                res = Y[:, :, 1:]
            else:
                cls_out = torch.zeros_like(CLS)                                # vit.py:205
                res = Y
            X = Xt + res                                                       # vit.py:215
            CLS = CLS + cls_out
        # ---- MLP (vit.py:216, 55-61) over cls + all patch tokens
        allt = torch.cat([CLS[:, None, :], X.reshape(B, T * N, D)], dim=1)
        h = layer_norm(allt, sd[b + 'norm2.weight'], sd[b + 'norm2.bias'])
        h = gelu_erf(h @ sd[b + 'mlp.fc1.weight'].t() + sd[b + 'mlp.fc1.bias'])
        h = h @ sd[b + 'mlp.fc2.weight'].t() + sd[b + 'mlp.fc2.bias']
        s = dp(i, 'mlp', None)
        if s is not None:
            h = h * s.reshape(B, 1, 1)
        allt = allt + h
        CLS = allt[:, 0]
        X = allt[:, 1:].reshape(B, T, N, D)
        if taps is not None:
            taps[f'block{i}'] = X.clone(); taps[f'cls{i}'] = CLS.clone()

    if cfg['norm_embeddings']:                                             # vision_tf.py:152-153
        X = layer_norm(X, sd[PREFIX + 'norm.weight'], sd[PREFIX + 'norm.bias'])
    if taps is not None:
        taps['features'] = X.clone()

    # K10: per-patch head + un-patchify (mask_tracker.py:112-115).
    Co = cfg['output_channels']
    Pm = X @ sd['seeker.tracker_post_linear.weight'].t() + sd['seeker.tracker_post_linear.bias']
    M = Pm.reshape(B, T, Hp, Wp, Co, P, P).permute(0, 4, 1, 2, 5, 3, 6).reshape(B, Co, T, Hf, Wf)
    # K11: coarsen (mask_tracker.py:118-132).
    st = cfg['track_map_stride']
    if st > 1:
        Mf = M.permute(0, 2, 1, 3, 4).reshape(B * T, Co, Hf, Wf)
        Mf = F.avg_pool2d(Mf, st, st)
        if taps is not None:
            taps['pooled'] = Mf.reshape(B, T, Co, Hf // st, Wf // st).clone()
        if cfg['track_map_resize'] == 'nearest':
            Mf = F.interpolate(Mf, scale_factor=st, mode='nearest')
        else:
            Mf = F.interpolate(Mf, scale_factor=st, mode='bilinear', align_corners=True)
        M = Mf.reshape(B, T, Co, Hf, Wf).permute(0, 2, 1, 3, 4).contiguous()
    # K12: flags (mask_tracker.py:135-137).
    flags = None
    if cfg['flag_channels'] > 0:
        fl = X @ sd['seeker.flag_post_linear.weight'].t() + sd['seeker.flag_post_linear.bias']
        flags = fl.mean(dim=2)                                             # mean over (H', W') -> (B, T, F)
    return M, flags


def to_torch_state_dict(np_sd, dtype=torch.float32):
    return {k: torch.from_numpy(v).to(dtype) for k, v in np_sd.items()}


def grad_sample(g):
    """Strided sample of a large gradient (numpy): every 7th row x every 5th column of its 2-D view [shape[0] (or, for a leading
    1, the flattened middle dims), last dims] -- co-prime strides so that no tile / lane pattern of a kernel is systematically
    skipped.  Used by oracle/make_golden_r2.py (g7) and by the GPU test that checks against it."""
    import numpy as np
    g = np.asarray(g)
    g2 = g.reshape(g.shape[0], -1) if g.shape[0] > 1 else g.reshape(-1, g.shape[-1])
    return np.ascontiguousarray(g2[::7, ::5])
