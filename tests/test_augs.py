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
    # integer rgb frames with a photometric operator: an error, not a silent skip (the reference feeds float frames in [0, 1])
    p = _smooth_params([96, 128, 60, 80, 0, 1, 1, 3000]); p['color_jitter'] = True
    with pytest.raises(NotImplementedError):
        augs.apply_augs({'rgb': torch.zeros(3, 14, 96, 128, dtype=torch.uint8, device=cuda)}, p, 60, 80)


def test_photometric_operators_follow_the_published_definitions():
    """augs.py:33-35,175-181: torchvision ColorJitter / GaussianBlur / Grayscale (not installed here: PARITY UNPINNED against the reference).
    Checked against independent float64 restatements of torchvision's documented tensor semantics: blends written out in numpy, python's
    colorsys for the RGB -> HSV -> RGB hue path, scipy's sampled Gaussian with mirror boundary for the blur."""
    import colorsys
    from scipy import ndimage
    g = torch.Generator().manual_seed(5)
    img = torch.rand(3, 3, 24, 20, generator=g)
    img[0, :, :4, :4] = 0.5                                     # grey patch: max == min, the hue path's degenerate branch
    x = img.double().numpy()
    gray = (0.2989 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2])[:, None]
    assert np.abs(augs.adjust_brightness(img, 1.15).numpy() - np.clip(1.15 * x, 0, 1)).max() < 1e-6
    assert np.abs(augs.adjust_contrast(img, 0.85).numpy() - np.clip(0.85 * x + 0.15 * gray.mean(axis=(1, 2, 3), keepdims=True), 0, 1)).max() < 1e-6
    assert np.abs(augs.adjust_saturation(img, 1.2).numpy() - np.clip(1.2 * x - 0.2 * gray, 0, 1)).max() < 1e-6
    assert np.abs(augs.grayscale3(img).numpy() - np.repeat(gray, 3, axis=1)).max() < 1e-6
    for shift in (0.07, -0.1):
        got = augs.adjust_hue(img, shift).numpy()
        want = np.empty_like(x)
        for t in range(3):
            for i in range(24):
                for j in range(20):
                    h, sat, v = colorsys.rgb_to_hsv(*x[t, :, i, j])
                    want[t, :, i, j] = colorsys.hsv_to_rgb((h + shift) % 1.0, sat, v)
        assert np.abs(got - want).max() < 2e-6, shift
    for sigma in (0.1, 0.9, 3.5):
        want = ndimage.gaussian_filter1d(ndimage.gaussian_filter1d(x, sigma, axis=-1, mode='mirror', radius=2), sigma, axis=-2, mode='mirror', radius=2)
        assert np.abs(augs.gaussian_blur5(img, sigma).numpy() - want).max() < 1e-6, sigma
    # the order of the four adjustments is part of the draw; one parameter set serves every frame of the clip
    a = augs.color_jitter(img, [1, 0, 3, 2], 1.1, 0.9, 1.15, 0.05)
    b = augs.adjust_saturation(augs.adjust_hue(augs.adjust_brightness(augs.adjust_contrast(img, 0.9), 1.1), 0.05), 1.15)
    assert torch.equal(a, b)
    # draws: torch's global generator in torchvision's order (ColorJitter.get_params, GaussianBlur.get_params), inside the constructor's ranges
    torch.manual_seed(11)
    d = augs.photometric_draws({'color_jitter': True, 'rgb_blur': True, 'rgb_grayscale': True})
    torch.manual_seed(11)
    order = torch.randperm(4).tolist(); u = [float(torch.empty(1).uniform_(lo, hi)) for lo, hi in ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1), (0.1, 3.5))]
    assert d['color_jitter'] == (order, u[0], u[1], u[2], u[3]) and d['rgb_blur'] == u[4] and d['rgb_grayscale'] is True
    assert augs.photometric_draws({'color_jitter': False, 'rgb_blur': False, 'rgb_grayscale': False}) == {}


def _photometric_f64(x, draws):
    """Independent float64 restatement of the three operators on a (T, 3, h, w) numpy stack: blends written out, python's colorsys for the
    RGB -> HSV -> RGB hue path, scipy's sampled Gaussian with mirror boundary for the blur (no code shared with tcow_amd.augs)."""
    import colorsys
    from scipy import ndimage
    x = x.astype(np.float64).copy()
    gray = lambda a: (0.2989 * a[:, 0] + 0.587 * a[:, 1] + 0.114 * a[:, 2])[:, None]
    if 'color_jitter' in draws:
        order, fb, fc, fs, fh = draws['color_jitter']
        for k in order:
            if k == 0: x = np.clip(fb * x, 0, 1)
            elif k == 1: x = np.clip(fc * x + (1 - fc) * gray(x).mean(axis=(1, 2, 3), keepdims=True), 0, 1)
            elif k == 2: x = np.clip(fs * x + (1 - fs) * gray(x), 0, 1)
            else:
                y = np.empty_like(x)
                for t in range(x.shape[0]):
                    for i in range(x.shape[2]):
                        for j in range(x.shape[3]):
                            hh, sat, v = colorsys.rgb_to_hsv(*x[t, :, i, j])
                            y[t, :, i, j] = colorsys.hsv_to_rgb((hh + fh) % 1.0, sat, v)
                x = y
    if 'rgb_blur' in draws:
        sg = draws['rgb_blur']
        x = ndimage.gaussian_filter1d(ndimage.gaussian_filter1d(x, sg, axis=-1, mode='mirror', radius=2), sg, axis=-2, mode='mirror', radius=2)
    if draws.get('rgb_grayscale'):
        x = np.repeat(gray(x), 3, axis=1)
    return x


@pytest.mark.gpu
def test_photometric_kernel_vs_independent_float64(cuda):
    """csrc/photometric.hip (ColorJitter in the drawn order + 5-tap reflect-padded blur + grayscale, frame selection and centre crop folded in)
    against the INDEPENDENT float64 restatements above -- not against tcow_amd.augs' own tensor functions -- and, as a second reference, against
    those tensor functions.  Frame sizes that are not multiples of the kernel's 32 x 32 tile, every operator alone and all orders of the jitter
    chain that put contrast first / in the middle / last (its mean is taken on the image as the adjustments in front of it left it)."""
    from tcow_amd import ops
    g = torch.Generator().manual_seed(21)
    cases = [
        ((3, 6, 40, 52), (1, 3, 37, 45), [4, 0, 2], {'color_jitter': ([1, 0, 3, 2], 1.1, 0.9, 1.15, 0.05), 'rgb_blur': 1.3, 'rgb_grayscale': True}),
        ((3, 5, 33, 70), (0, 2, 33, 64), [1, 1, 3], {'color_jitter': ([0, 2, 1, 3], 0.85, 1.18, 0.8, -0.1)}),
        ((3, 4, 64, 64), (0, 0, 64, 64), [3, 0], {'color_jitter': ([3, 2, 0, 1], 1.2, 0.8, 1.2, 0.1), 'rgb_blur': 0.1}),
        ((3, 3, 24, 20), (2, 1, 20, 17), [0, 2, 1], {'rgb_blur': 3.5}),
        ((3, 3, 24, 20), (2, 1, 20, 17), [2], {'rgb_grayscale': True}),
        ((3, 3, 35, 35), (1, 1, 33, 33), [1, 2], {'color_jitter': ([2, 3, 1, 0], 1.0, 1.0, 1.0, 0.0)}),
    ]
    for shape, rect, fi, draws in cases:
        fr = torch.rand(*shape, generator=g)
        fr[:, 0, :5, :5] = 0.5                                  # grey patch: max == min, the hue path's degenerate branch
        y0, x0, h, w = rect
        got = augs.photometric_hip(fr.to(cuda), fi, rect, draws).cpu()
        assert tuple(got.shape) == (3, len(fi), h, w)
        sel = fr[:, fi][:, :, y0:y0 + h, x0:x0 + w].permute(1, 0, 2, 3)                       # (T, 3, h, w)
        want = _photometric_f64(sel.numpy(), draws).transpose(1, 0, 2, 3)
        assert np.abs(got.numpy() - want).max() < 3e-6, (shape, draws)
        second = augs.apply_photometric(sel, draws).permute(1, 0, 2, 3)
        assert float((got - second).abs().max()) < 2e-6, (shape, draws)
    # argument errors surface as exceptions
    with pytest.raises(ops.L.TcowError):
        ops.photometric(torch.zeros(3, 2, 8, 8, device=cuda), torch.zeros(1, dtype=torch.int32, device=cuda), (0, 0, 9, 8), [], (1, 1, 1, 0), None, False)


@pytest.mark.gpu
def test_apply_augs_with_photometric_operators(cuda):
    """The rgb modality through apply_augs with all three photometric operators on: equal to the operators applied to the selected,
    centre-cropped frames at source resolution followed by flip / crop / ATen's antialiased resize (the reference's order, augs.py:175-201)."""
    g = torch.Generator().manual_seed(8)
    for (H, W, oh, ow, cc, flip) in [(96, 128, 60, 80, 0, 1), (120, 200, 60, 80, 1, 0), (64, 64, 64, 64, 0, 1)]:
        p = _smooth_params([H, W, oh, ow, cc, flip, 1, 4000 + H]); p.update(color_jitter=True, rgb_blur=True, rgb_grayscale=(H == 120))
        rgb = torch.rand(3, 14, H, W, generator=g)
        draws = {'color_jitter': ([2, 0, 1, 3], 1.12, 0.88, 1.07, -0.04), 'rgb_blur': 1.3}
        if H == 120: draws['rgb_grayscale'] = True
        got = augs.apply_augs({'rgb': rgb.to(cuda)}, p, oh, ow, center_crop=bool(cc), draws_in=draws)['rgb'].cpu()
        fi, ys, xs = augs.crop_maps(p, H, W, oh, ow, bool(cc))
        y0, x0, h, w = augs.center_rect(H, W, oh, ow, bool(cc))
        sel = rgb[:, torch.as_tensor(np.asarray(fi, dtype=np.int64))][:, :, y0:y0 + h, x0:x0 + w]
        img = augs.apply_photometric(sel.permute(1, 0, 2, 3), draws)
        img = img[:, :, torch.as_tensor(ys - y0)][:, :, :, torch.as_tensor(xs - x0)]
        want = torch.nn.functional.interpolate(img, size=(oh, ow), mode='bilinear', antialias=True, align_corners=False).permute(1, 0, 2, 3)
        assert got.shape == want.shape and float((got - want).abs().max()) < 1e-5, (H, W)
