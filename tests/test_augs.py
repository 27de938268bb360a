"""Index half of the input pipeline (data/augs.py:50-210): parameter sampling and index tables against what the REFERENCE
produced (tests/golden/g13_augs.npz, oracle/make_golden_r2.py::g13), and the HIP gather against the same goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.make_golden_r2 import augs_inputs
from oracle.make_golden_r3 import smooth_inputs
from tcow_amd import augs

KEYS = ['palindrome', 'reverse', 'frame_stride_factor', 'offset', 'color_jitter', 'rgb_blur', 'rgb_grayscale', 'horz_flip']


def _cases(g):
    i = 0
    while f'params{i}::scalars' in g:
        yield i
        i += 1


def test_sample_augs_params_reproduces_the_reference_draw_for_draw():
    _, g = load_golden('g13_augs')
    n = 0
    for i in _cases(g):
        seed, nl, nc, fs, rnd, a2d, rp, pp = g['param_cases'][i]
        np.random.seed(1000 + int(seed))
        p = augs.sample_augs_params(int(nl), int(nc), int(fs), bool(rnd), bool(a2d), float(rp), float(pp))
        assert [float(p[k]) for k in KEYS] == g[f'params{i}::scalars'].tolist(), i
        assert np.array_equal(p['frame_inds_load'], g[f'params{i}::frame_inds_load']) and np.array_equal(p['frame_inds_clip'], g[f'params{i}::frame_inds_clip']), i
        assert np.array_equal(np.asarray(p['crop_rect'], dtype=np.float64), g[f'params{i}::crop_rect']), i
        n += 1
    assert n == 48


def _aug_cases(g):
    for k in g:
        if k.endswith('::cfg') and k.startswith('aug'):
            yield k[:-5], [int(v) for v in g[k]]


def _params(cfg):
    H, W, oh, ow, cc, rnd, a2d, seed = cfg
    np.random.seed(seed)
    return augs.sample_augs_params(14, 10, 1, bool(rnd), bool(a2d), 0.3, 0.4)


def test_index_maps_equal_the_reference_chain_on_integer_modalities():
    """frame selection -> centre crop -> flip -> crop -> NEAREST resize of the reference == one gather through our three index tables."""
    _, g = load_golden('g13_augs')
    n = 0
    for tag, cfg in _aug_cases(g):
        H, W, oh, ow, cc = cfg[:5]
        p = _params(cfg)
        segm, div = augs_inputs(tag, H, W)
        fi, sy, sx = augs.index_maps(p, H, W, oh, ow, center_crop=bool(cc))
        assert fi.dtype == np.int32 and len(fi) == 10 and len(sy) == oh and len(sx) == ow
        for src, key in ((segm, 'segm_out'), (div, 'div_out')):
            got = src.numpy()[:, fi][:, :, sy][:, :, :, sx]
            assert np.array_equal(got, g[f'{tag}::{key}']), (tag, key)
        n += 1
    assert n == 18


@pytest.mark.gpu
def test_gather_frames_kernel_bit_exact(cuda):
    _, g = load_golden('g13_augs')
    for tag, cfg in _aug_cases(g):
        H, W, oh, ow, cc = cfg[:5]
        p = _params(cfg)
        segm, div = augs_inputs(tag, H, W)
        out = augs.apply_augs_index({'segm': segm.cuda(), 'div_segm': div.cuda(), 'scalar_thing': torch.zeros(3).cuda()}, p, oh, ow, center_crop=bool(cc))
        assert np.array_equal(out['segm'].cpu().numpy(), g[f'{tag}::segm_out']) and np.array_equal(out['div_segm'].cpu().numpy(), g[f'{tag}::div_out']), tag
        assert out['segm'].dtype == torch.uint8 and tuple(out['div_segm'].shape) == (4, 10, oh, ow)
    # f32 frames ride the same kernel when no smooth resize is involved (temporal sub-sampling + flip at native size)
    rgb = torch.rand(3, 14, 48, 64, device=cuda)
    p = dict(frame_inds_clip=np.arange(2, 12)[::-1].copy(), horz_flip=True, crop_rect=-np.ones(4))
    got = augs.apply_augs_index({'rgb': rgb}, p, 48, 64)['rgb']
    assert torch.equal(got, torch.flip(rgb[:, torch.arange(11, 1, -1, device=cuda)], dims=[-1]))
    with pytest.raises(NotImplementedError):
        augs.apply_augs_index({'rgb': rgb}, p, 24, 32)                      # the index path alone cannot resize a float modality: apply_augs does


def _smooth_cases(g):
    for k in g:
        if k.endswith('::cfg') and k.startswith('sm'):
            yield k[:-5], [int(v) for v in g[k]]


def _smooth_params(cfg):
    H, W, oh, ow, cc, rnd, a2d, seed = cfg
    np.random.seed(seed)
    p = augs.sample_augs_params(14, 10, 1, bool(rnd), bool(a2d), 0.3, 0.4)
    p['color_jitter'] = False; p['rgb_blur'] = False; p['rgb_grayscale'] = False       # (the fixture ran the reference with the photometric flags off)
    return p


def test_antialias_tables_equal_aten_interpolate():
    """aa_tables (host side of tcow_resize_aa) against torch.nn.functional.interpolate(mode='bilinear', antialias=True) -- the operator
    torchvision's tensor Resize dispatches to (data/augs.py:40-43) -- applied as ATen applies it (width pass, then height pass)."""
    g = torch.Generator().manual_seed(3)
    for (H, W, oh, ow) in [(96, 128, 60, 80), (77, 131, 40, 56), (64, 64, 120, 160), (100, 99, 33, 200), (37, 53, 37, 20), (9, 7, 2, 3)]:
        x = torch.rand(2, 3, H, W, generator=g)
        want = torch.nn.functional.interpolate(x, size=(oh, ow), mode='bilinear', antialias=True, align_corners=False).numpy()
        ymin, ysz, wy, ky = augs.aa_tables(H, oh); xmin, xsz, wx, kx = augs.aa_tables(W, ow)
        xn = x.numpy()
        tmp = np.stack([sum(wx[X, i] * xn[..., xmin[X] + i] for i in range(xsz[X])) for X in range(ow)], axis=-1).astype(np.float32)
        got = np.stack([sum(wy[Y, j] * tmp[..., ymin[Y] + j, :] for j in range(ysz[Y])) for Y in range(oh)], axis=-2).astype(np.float32)
        assert np.abs(got - want).max() < 1e-6, (H, W, oh, ow)
        assert np.allclose(wy.sum(1), 1.0, atol=1e-6) and np.allclose(wx.sum(1), 1.0, atol=1e-6)


@pytest.mark.gpu
def test_resize_aa_kernel_vs_reference_pipeline(cuda):
    """G15: rgb / depth clips through the REFERENCE's apply_augs_2d_frames (frame selection, centre crop, flip, crop, antialiased bilinear
    resize) against ONE pass of tcow_resize_aa.  Floating point: 1e-5 absolute on values in [0, 1) (depth: [0, 20) -> 2e-4), the
    difference being the summation order of <= 7 x 7 taps."""
    _, g = load_golden('g15_augs_smooth')
    n = 0
    for tag, cfg in _smooth_cases(g):
        H, W, oh, ow, cc = cfg[:5]
        p = _smooth_params(cfg)
        rgb, depth = smooth_inputs(tag, H, W)
        out = augs.apply_augs({'rgb': rgb.cuda(), 'depth': depth.cuda(), 'scalar_thing': torch.zeros(3).cuda()}, p, oh, ow, center_crop=bool(cc))
        o = out['rgb'].cpu().numpy(); dd = out['depth'].cpu().numpy()
        assert o.shape == (3, 10, oh, ow) and dd.shape == (1, 10, oh, ow)
        big = H >= 400
        assert np.abs((o[:, ::3, ::7, ::5] if big else o[:, ::3]) - g[f'{tag}::rgb_frames']).max() < 1e-5, tag
        assert np.abs((dd[:, 4, ::7, ::5] if big else dd[:, 4]) - g[f'{tag}::depth_frame']).max() < 2e-4, tag
        assert np.abs(o.astype(np.float64).sum(axis=(2, 3)) - g[f'{tag}::rgb_sum']).max() < 1e-5 * oh * ow
        assert np.abs(dd.astype(np.float64).sum(axis=(2, 3)) - g[f'{tag}::depth_sum']).max() < 2e-4 * oh * ow
        n += 1
    assert n == 14
    # the photometric operators are torchvision's: asking for one is an error, not a silent skip
    p = _smooth_params([96, 128, 60, 80, 0, 1, 1, 3000]); p['color_jitter'] = True
    with pytest.raises(NotImplementedError):
        augs.apply_augs({'rgb': torch.rand(3, 14, 96, 128, device=cuda)}, p, 60, 80)
