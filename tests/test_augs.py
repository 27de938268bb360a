"""Index half of the input pipeline (data/augs.py:50-210): parameter sampling and index tables against what the REFERENCE
produced (tests/golden/g13_augs.npz, oracle/make_golden_r2.py::g13), and the HIP gather against the same goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.make_golden_r2 import augs_inputs
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
        augs.apply_augs_index({'rgb': rgb}, p, 24, 32)                      # would need the antialiased bilinear resize (torchvision): out of scope
