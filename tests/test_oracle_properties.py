"""CPU: known-answer properties of the path checked on the oracle restatement."""
import numpy as np
import pytest
import torch

from oracle import seeker_oracle as so
from tcow_amd import synth

BASE = dict(num_total_frames=6, frame_height=32, frame_width=48, embed_dim=64, depth=2, num_heads=4)


def _fwd(cfg, rgb, qm, sd=None, **kw):
    sd = sd or so.to_torch_state_dict(synth.make_state_dict(cfg, 7))
    with torch.no_grad():
        return so.seeker_forward(sd, cfg, rgb, qm, **kw)


def _clip(B, cfg):
    c = synth.make_clip(B, cfg['num_total_frames'], cfg['frame_height'], cfg['frame_width'], seed=11)
    return torch.from_numpy(c['rgb']), torch.from_numpy(synth.make_query_mask(c, 0, 0))


@pytest.mark.parametrize('ca,leak_back', [(1, 0), (2, 0), (3, 2), (0, None)])
def test_causality(ca, leak_back):
    """vit.py:93-99,193-198: with causal_attention in {1,2} a change in frame t0 never reaches frames < t0 (bit-exact);
    3 lets every block look one frame ahead, so a 2-block model leaks exactly two frames back; 0 leaks everywhere."""
    cfg = synth.seeker_config(**BASE, causal_attention=ca)
    rgb, qm = _clip(1, cfg)
    t0 = 4
    rgb2 = rgb.clone(); rgb2[:, :, t0] += 0.25
    a, _ = _fwd(cfg, rgb, qm); b, _ = _fwd(cfg, rgb2, qm)
    diff = (a - b).abs().amax(dim=(0, 1, 3, 4))
    if leak_back is None:
        assert (diff > 0).all()
    else:
        assert float(diff[: t0 - leak_back].max()) == 0.0
        assert (diff[t0 - leak_back:] > 0).all()


def test_stock_init_ignores_temporal_attention():
    """vit.py:289-297 zero-initialises every temporal_fc: the temporal path then has no effect on the output."""
    cfg = synth.seeker_config(**BASE, causal_attention=1)
    sd = so.to_torch_state_dict(synth.make_state_dict(cfg, 7))
    for k in sd:
        if 'temporal_fc' in k:
            sd[k] = torch.zeros_like(sd[k])
    rgb, qm = _clip(1, cfg)
    a, _ = _fwd(cfg, rgb, qm, sd)
    for k in sd:
        if 'temporal_attn' in k:
            sd[k] = sd[k] * 3.0 + 0.1
    b, _ = _fwd(cfg, rgb, qm, sd)
    assert float((a - b).abs().max()) == 0.0


def test_flags_equal_linear_of_mean_feature():
    cfg = synth.seeker_config(**BASE, causal_attention=1)
    sd = so.to_torch_state_dict(synth.make_state_dict(cfg, 7))
    rgb, qm = _clip(2, cfg)
    taps = {}
    _, fl = _fwd(cfg, rgb, qm, sd, taps=taps)
    alt = taps['features'].mean(dim=2) @ sd['seeker.flag_post_linear.weight'].t() + sd['seeker.flag_post_linear.bias']
    assert float((fl - alt).abs().max()) < 1e-6


def test_avgpool_folds_into_head_weights():
    """avg_pool2d(4) of the un-patchified head output == head with 4x4-averaged weights (mask_tracker.py:113-122)."""
    cfg = synth.seeker_config(**BASE, causal_attention=1)
    sd = so.to_torch_state_dict(synth.make_state_dict(cfg, 7))
    rgb, qm = _clip(1, cfg)
    taps = {}
    _fwd(cfg, rgb, qm, sd, taps=taps)
    P, st, Co, D = 16, 4, 3, cfg['embed_dim']
    Wh = sd['seeker.tracker_post_linear.weight'].reshape(Co, P // st, st, P // st, st, D).mean(dim=(2, 4)).reshape(-1, D)
    bh = sd['seeker.tracker_post_linear.bias'].reshape(Co, P // st, st, P // st, st).mean(dim=(2, 4)).reshape(-1)
    X = taps['features']
    B, T, N, _ = X.shape
    Hp, Wp = cfg['frame_height'] // P, cfg['frame_width'] // P
    y = (X @ Wh.t() + bh).reshape(B, T, Hp, Wp, Co, P // st, P // st).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, T, Co, Hp * P // st, Wp * P // st)
    assert float((y - taps['pooled']).abs().max()) < 1e-6


def test_batch_rows_are_independent():
    cfg = synth.seeker_config(**BASE, causal_attention=1)
    rgb, qm = _clip(2, cfg)
    a, fa = _fwd(cfg, rgb, qm)
    b, fb = _fwd(cfg, rgb[1:], qm[1:])
    assert float((a[1:] - b).abs().max()) < 1e-6 and float((fa[1:] - fb).abs().max()) < 1e-6


def test_drop_path_semantics():
    """vit.py:172-176: a dropped temporal row still receives temporal_fc's bias; dropped spatial / mlp rows get nothing."""
    cfg = synth.seeker_config(**BASE, causal_attention=1)
    rgb, qm = _clip(1, cfg)
    N = 6
    keep_all = {(i, k): (torch.ones(s), 0.5) for i in range(2) for k, s in (('temporal', (1, N)), ('spatial', (1, 6)), ('mlp', (1,)))}
    scaled, _ = _fwd(cfg, rgb, qm, drop_masks=keep_all)                   # everything kept but scaled by 1/keep = 2
    plain, _ = _fwd(cfg, rgb, qm)
    assert float((scaled - plain).abs().max()) > 1e-4
    none = {k: (torch.zeros_like(v[0]), 0.5) for k, v in keep_all.items()}
    dropped, _ = _fwd(cfg, rgb, qm, drop_masks=none)
    assert torch.isfinite(dropped).all()


def test_token_count_and_shapes():
    cfg = synth.seeker_config()
    shapes = synth.state_dict_shapes(cfg)
    assert len(shapes) == 251 and sum(int(np.prod(s)) for s in shapes.values()) == 122145027   # SURVEY.md section 6 probe
