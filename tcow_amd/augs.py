"""The reference's clip augmentation pipeline on the GPU (data/augs.py:50-210, data/data_kubric.py:341-434):
temporal sub-sampling (palindrome / reverse / stride factor / offset), centre crop, horizontal flip, random crop and the final resize.

The index part is integer arithmetic and composes into three small tables (source frame per clip frame, source row / column per row /
column of the cropped image); with them
  * integer / mask modalities ('segm', 'div_segm', query / target masks: augs.py:196-198, NEAREST) are ONE gather pass
    (`tcow_gather_frames`), bit-exact;
  * rgb / depth / coordinate modalities (augs.py:199-201: BILINEAR with antialias=True) are ONE pass of `tcow_resize_aa`: the separable
    triangle filter of torch.nn.functional.interpolate(mode='bilinear', antialias=True, align_corners=False) -- the ATen operator
    torchvision's tensor Resize dispatches to -- with its index / weight tables computed on the host (`aa_tables`, a restatement of
    ATen's _compute_indices_weights_aa in float32) and the crop / flip / frame selection folded into the source addressing.
The reference runs slicing + torch.flip + slicing + torchvision Resize as four full-tensor passes per modality on the CPU loader workers.

The photometric operators of the rgb modality (augs.py:33-35,175-181: torchvision ColorJitter(0.2, 0.2, 0.2, 0.1), GaussianBlur(5, sigma in
[0.1, 3.5]), Grayscale(3)) run as ONE fused device kernel (`tcow_photometric`, csrc/photometric.hip; + a per-frame mean pass when contrast is
drawn) on the selected frames at source resolution and BEFORE flip / crop / resize like the reference (the contrast mean is taken over the
whole centre-cropped frame).  The tensor functions below (adjust_*, gaussian_blur5, grayscale3) are the same definitions in torch: the CPU-side
statement the tests compare with float64 restatements, and the second reference of the kernel's GPU test.  torchvision is not installed here, so no vector
of the reference's can pin them: they follow torchvision's published tensor definitions (functional_tensor: _blend, rgb_to_grayscale
0.2989 / 0.587 / 0.114, _rgb2hsv / _hsv2rgb, the 5-tap Gaussian with reflect padding) and `photometric_draws` consumes torch's global RNG in
torchvision's order (randperm(4), then one uniform_ per jitter factor; one uniform_ for sigma).  PARITY UNPINNED against the reference for
these three operators; tests/test_augs.py checks them against an independent float64 restatement (python's colorsys for the hue path).

`sample_augs_params` draws from numpy's GLOBAL generator in exactly the reference's order, so `np.random.seed(s)` reproduces the
reference's parameters draw for draw (pinned by tests/golden/g13_augs.npz).
"""
import numpy as np

# The reference's draw sequence (data/augs.py:66-133) as DATA: (name, kind, argument, condition on the values drawn so far).
#   'lt'   value = rand() < arg          'unit' value = rand() * 0.2 + arg          'offset' value = randint(0, avail - clip + 1)
# Conditions see the dict of earlier draws; a draw whose condition is false is NOT made (the generator does not advance) and takes `default`.
_TEMPORAL_DRAWS = [
    ('palindrome', 'lt', 'palindrome_prob', None, False),
    ('reverse_p', 'lt', 0.35, lambda d: d['palindrome'], False),
    ('stride2', 'lt', 0.35, lambda d: d['palindrome'], False),
    ('reverse_n', 'lt', 'reverse_prob', lambda d: not d['palindrome'], False),
]
_SPATIAL_DRAWS = [
    ('color_jitter', 'lt', 0.9, None, False),
    ('rgb_blur', 'lt', 0.2, None, False),
    ('rgb_grayscale', 'lt', 0.05, None, False),
    ('horz_flip', 'lt', 0.5, lambda d: d['augs_2d'], False),
    ('crop_y1', 'unit', 0.0, lambda d: d['augs_2d'], -1.0),
    ('crop_y2', 'unit', 0.8, lambda d: d['augs_2d'], -1.0),
    ('crop_x1', 'unit', 0.0, lambda d: d['augs_2d'], -1.0),
    ('crop_x2', 'unit', 0.8, lambda d: d['augs_2d'], -1.0),
]


def _draw(table, d, probs):
    for name, kind, arg, cond, default in table:
        if cond is not None and not cond(d):
            d[name] = default
            continue
        u = np.random.rand()
        a = probs[arg] if isinstance(arg, str) else arg
        d[name] = bool(u < a) if kind == 'lt' else u * 0.2 + a


def sample_augs_params(num_frames_load, num_frames_clip, frame_stride, do_random_augs, augs_2d, reverse_prob, palindrome_prob):
    """data/augs.py:50-135: same keys, same values, same consumption of numpy's global generator."""
    d = dict(augs_2d=bool(augs_2d), palindrome=False, reverse_p=False, stride2=False, reverse_n=False)
    probs = dict(palindrome_prob=palindrome_prob, reverse_prob=reverse_prob)
    clip = list(range(num_frames_clip))
    offset = (num_frames_load - num_frames_clip) // 2
    if do_random_augs:
        _draw(_TEMPORAL_DRAWS, d, probs)
        if d['palindrome']:
            clip = clip + clip[::-1][1:]
        if d['reverse_p'] or d['reverse_n']:
            clip = clip[::-1]
        if d['stride2']:
            clip = clip[::2]
        assert len(clip) >= num_frames_clip
        offset = np.random.randint(0, len(clip) - num_frames_clip + 1)
        clip = clip[offset:offset + num_frames_clip]
        _draw(_SPATIAL_DRAWS, d, probs)
    else:
        for name, _, _, _, default in _SPATIAL_DRAWS:
            d[name] = default
    return dict(palindrome=d['palindrome'], reverse=bool(d['reverse_p'] or d['reverse_n']), frame_stride_factor=2 if d['stride2'] else 1, offset=offset,
                frame_inds_load=np.arange(0, num_frames_load * frame_stride, frame_stride), frame_inds_clip=np.array(clip),
                color_jitter=d['color_jitter'], rgb_blur=d['rgb_blur'], rgb_grayscale=d['rgb_grayscale'], horz_flip=d['horz_flip'],
                crop_rect=np.array([d['crop_y1'], d['crop_y2'], d['crop_x1'], d['crop_x2']]) if (do_random_augs and augs_2d) else -np.ones(4))


def center_rect(H, W, out_h, out_w, center_crop):
    """(y0, x0, h, w) of the centre crop to the output aspect ratio (augs.py:163-171; torchvision CenterCrop: top = round((H - h) / 2))."""
    y0, x0, h, w = 0, 0, H, W
    if center_crop:
        cur, want = W / H, out_w / out_h
        if cur > want:
            cw = int(H * want); x0 = int(round((W - cw) / 2.0)); w = cw
        elif cur < want:
            ch = int(W / want); y0 = int(round((H - ch) / 2.0)); h = ch
    return y0, x0, h, w


def crop_maps(augs_params, H, W, out_h, out_w, center_crop=False):
    """Source (frame, row, column) of every (frame, row, column) of the image the final resize sees: centre crop to the output aspect
    ratio (torchvision CenterCrop: top = round((H - h) / 2)), horizontal flip, fractional crop rectangle (int(y1 * H) : int(y2 * H) on the
    post-flip image) -- augs.py:150-194.  Returns (frame_idx [Tc] int32, ys [h] int64, xs [w] int64)."""
    frame_idx = np.asarray(augs_params['frame_inds_clip'], dtype=np.int32)
    y0, x0, h, w = center_rect(H, W, out_h, out_w, center_crop)
    ys = np.arange(h, dtype=np.int64) + y0                                 # source row of each row of the current image
    xs = np.arange(w, dtype=np.int64) + x0
    if bool(augs_params['horz_flip']):                                     # augs.py:184-185
        xs = xs[::-1]
    cr = augs_params['crop_rect']
    if cr is not None and np.all(np.asarray(cr) >= 0.0):                   # augs.py:189-194
        # NB the reference multiplies the fractions by the size BEFORE the centre crop ((C,T,H,W) = raw_frames.shape, augs.py:159) and
        # slices the centre-cropped image with them; python slicing clamps what falls off the end
        y1, y2, x1, x2 = [float(v) for v in cr]
        ys = ys[int(y1 * H):int(y2 * H)]
        xs = xs[int(x1 * W):int(x2 * W)]
    return frame_idx, ys, xs


def aa_tables(n_in, n_out):
    """Index / weight table of the antialiased bilinear resize along one axis: ATen's _compute_indices_weights_aa (UpSampleKernel.cpp) for
    the triangle filter, align_corners=False, in float32.  Returns (first source index [n_out] int32, tap count [n_out] int32,
    weights [n_out, K] float32 zero-padded, K)."""
    f = np.float32
    scale = f(n_in) / f(n_out)
    support = f(scale) if scale >= 1.0 else f(1.0)
    invscale = f(1.0) / scale if scale >= 1.0 else f(1.0)
    K = int(np.ceil(support)) * 2 + 1
    i = np.arange(n_out, dtype=np.float32)
    center = scale * (i + f(0.5))
    xmin = np.maximum((center - support + f(0.5)).astype(np.int64), 0)
    xsize = np.minimum((center + support + f(0.5)).astype(np.int64), n_in) - xmin
    j = np.arange(K, dtype=np.float32)[None, :]
    arg = (j + xmin[:, None].astype(np.float32) - center[:, None] + f(0.5)) * invscale
    w = np.maximum(f(1.0) - np.abs(arg), f(0.0)).astype(np.float32)
    w[np.arange(K)[None, :] >= xsize[:, None]] = 0.0
    tot = w.sum(axis=1, dtype=np.float32)
    w = (w / tot[:, None]).astype(np.float32)
    return xmin.astype(np.int32), xsize.astype(np.int32), np.ascontiguousarray(w), K


def index_maps(augs_params, H, W, out_h, out_w, center_crop=False, return_resized=False):
    """Source (frame, row, column) of every output (frame, row, column) for the spatial chain of augs.py:150-203 with NEAREST resize
    (source = floor(dst * in / out), torch's 'nearest') on top of `crop_maps`.
    Returns int32 arrays (frame_idx [Tc], src_y [out_h], src_x [out_w]) (+ whether the last step changes the size)."""
    frame_idx, ys, xs = crop_maps(augs_params, H, W, out_h, out_w, center_crop)
    # nearest resize: torch.nn.functional.interpolate(mode='nearest'): src = floor(dst * (in / out)) with the scale in float32
    def nearest(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        return np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64), n_in - 1)
    resized = (len(ys) != out_h) or (len(xs) != out_w)
    ys = ys[nearest(len(ys), out_h)]
    xs = xs[nearest(len(xs), out_w)]
    if return_resized:
        return frame_idx, ys.astype(np.int32), xs.astype(np.int32), resized
    return frame_idx, ys.astype(np.int32), xs.astype(np.int32)


def gather_clip(frames, frame_idx, src_y, src_x):
    """frames (C, Tv, H, W) uint8 / float32 CUDA tensor -> (C, Tc, h, w) through the index tables, one HIP pass (tcow_gather_frames)."""
    import torch
    from . import ops
    dev = frames.device
    fi = torch.as_tensor(frame_idx, dtype=torch.int32).to(dev); sy = torch.as_tensor(src_y, dtype=torch.int32).to(dev); sx = torch.as_tensor(src_x, dtype=torch.int32).to(dev)
    return ops.gather_frames(frames, fi, sy, sx)


def resize_clip_aa(frames, frame_idx, ys, xs, out_h, out_w):
    """frames (C, Tv, H, W) f32 CUDA tensor -> (C, Tc, out_h, out_w): frame selection + crop / flip (ys, xs = source row / column of each
    row / column of the cropped image) + antialiased bilinear resize, one HIP pass (tcow_resize_aa)."""
    import torch
    from . import ops
    dev = frames.device
    ymin, ysz, wy, ky = aa_tables(len(ys), out_h)
    xmin, xsz, wx, kx = aa_tables(len(xs), out_w)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev)
    return ops.resize_aa(frames, t(frame_idx, torch.int32), t(ys, torch.int32), t(xs, torch.int32), t(ymin, torch.int32), t(ysz, torch.int32), t(wy, torch.float32), ky,
                         t(xmin, torch.int32), t(xsz, torch.int32), t(wx, torch.float32), kx, out_h, out_w)


# ---- photometric operators of the rgb modality (augs.py:175-181), torchvision's tensor definitions on (T, 3, h, w) float images in [0, 1]
def rgb_to_gray(img):
    return (0.2989 * img[:, 0] + 0.587 * img[:, 1] + 0.114 * img[:, 2]).unsqueeze(1)


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0.0, 1.0)


def adjust_brightness(img, f):
    return _blend(img, 0.0 * img, f)


def adjust_contrast(img, f):
    return _blend(img, rgb_to_gray(img).mean(dim=(-3, -2, -1), keepdim=True), f)       # one mean per frame


def adjust_saturation(img, f):
    return _blend(img, rgb_to_gray(img), f)


def adjust_hue(img, shift):
    import torch
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    maxc = torch.maximum(torch.maximum(r, g), b); minc = torch.minimum(torch.minimum(r, g), b)
    eq = maxc == minc
    cr = maxc - minc
    one = torch.ones_like(maxc)
    sat = cr / torch.where(eq, one, maxc)
    div = torch.where(eq, one, cr)
    rc, gc, bc = (maxc - r) / div, (maxc - g) / div, (maxc - b) / div
    is_r, is_g = maxc == r, maxc == g
    h = is_r * (bc - gc) + (is_g & ~is_r) * (2.0 + rc - bc) + (~is_g & ~is_r) * (4.0 + gc - rc)
    h = torch.fmod(h / 6.0 + 1.0, 1.0)
    h = (h + shift) % 1.0
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    v = maxc
    p = (v * (1.0 - sat)).clamp(0.0, 1.0); q = (v * (1.0 - sat * f)).clamp(0.0, 1.0); t = (v * (1.0 - sat * (1.0 - f))).clamp(0.0, 1.0)
    pick = lambda opts: sum((i == k) * o for k, o in enumerate(opts))                   # (select by sextant, as torchvision's mask einsum)
    return torch.stack([pick((v, q, p, p, t, v)), pick((t, v, v, q, p, p)), pick((p, p, t, v, v, q))], dim=1)


def color_jitter(img, order, brightness, contrast, saturation, hue):
    """torchvision ColorJitter.forward with its drawn parameters made explicit: the four adjustments in `order` (0 brightness, 1 contrast,
    2 saturation, 3 hue), one parameter set for all frames of the clip (the reference passes the (T, 3, H, W) stack in one call)."""
    for k in order:
        k = int(k)
        if k == 0: img = adjust_brightness(img, brightness)
        elif k == 1: img = adjust_contrast(img, contrast)
        elif k == 2: img = adjust_saturation(img, saturation)
        else: img = adjust_hue(img, hue)
    return img


def gaussian_blur5(img, sigma):
    """torchvision gaussian_blur(kernel_size=5, sigma): taps exp(-0.5 (x / sigma)^2) on x = -2..2, normalised, separable, reflect padding."""
    import torch
    x = torch.linspace(-2.0, 2.0, 5, dtype=img.dtype, device=img.device)
    k = torch.exp(-0.5 * (x / sigma) ** 2); k = k / k.sum()
    pad = torch.nn.functional.pad(img, (2, 2, 2, 2), mode='reflect')
    H, W = img.shape[-2:]
    hor = sum(k[j] * pad[..., :, j:j + W] for j in range(5))                            # (T, 3, H + 4, W)
    return sum(k[j] * hor[..., j:j + H, :] for j in range(5))


def grayscale3(img):
    return rgb_to_gray(img).expand(-1, 3, -1, -1).contiguous()


def photometric_draws(augs_params):
    """The parameters torchvision's transforms draw from torch's GLOBAL generator when the reference calls them (augs.py:175-181), in the
    same order: ColorJitter.get_params = randperm(4) then uniform_ for brightness, contrast, saturation in [0.8, 1.2] and hue in [-0.1, 0.1];
    GaussianBlur.get_params = uniform_(0.1, 3.5)."""
    import torch
    out = {}
    if augs_params.get('color_jitter'):
        order = torch.randperm(4).tolist()
        u = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi))
        out['color_jitter'] = (order, u(0.8, 1.2), u(0.8, 1.2), u(0.8, 1.2), u(-0.1, 0.1))
    if augs_params.get('rgb_blur'):
        out['rgb_blur'] = float(torch.empty(1).uniform_(0.1, 3.5))
    if augs_params.get('rgb_grayscale'):
        out['rgb_grayscale'] = True
    return out


def apply_photometric(img, draws):
    """(T, 3, h, w) float image stack -> the same after ColorJitter / GaussianBlur / Grayscale, in the reference's order (augs.py:176-181)."""
    if 'color_jitter' in draws:
        img = color_jitter(img, *draws['color_jitter'])
    if 'rgb_blur' in draws:
        img = gaussian_blur5(img, draws['rgb_blur'])
    if draws.get('rgb_grayscale'):
        img = grayscale3(img)
    return img


def blur_taps(sigma):
    """torchvision's _get_gaussian_kernel1d(5, sigma) in float32: taps exp(-0.5 (x / sigma)^2) on x = -2 .. 2, normalised."""
    import torch
    x = torch.linspace(-2.0, 2.0, 5, dtype=torch.float32)
    k = torch.exp(-0.5 * (x / float(sigma)) ** 2)
    return (k / k.sum()).tolist()


def photometric_hip(fr, frame_idx, rect, draws):
    """The three photometric operators as ONE fused device kernel (csrc/photometric.hip): fr (3, Tv, H, W) f32 CUDA frames in [0, 1], the clip's
    source-frame table and centre-crop rectangle (y0, x0, h, w), the draws of `photometric_draws` -> (3, T, h, w)."""
    import torch
    from . import ops
    order, factors = [], (1.0, 1.0, 1.0, 0.0)
    if 'color_jitter' in draws:
        o, fb, fc, fs, fh = draws['color_jitter']
        order, factors = [int(k) for k in o], (fb, fc, fs, fh)
    taps = blur_taps(draws['rgb_blur']) if 'rgb_blur' in draws else None
    fi = torch.as_tensor(np.asarray(frame_idx, dtype=np.int32), device=fr.device)
    return ops.photometric(fr.contiguous(), fi, rect, order, factors, taps, bool(draws.get('rgb_grayscale')))


def apply_augs(modalities, augs_params, out_h, out_w, center_crop=False, draws_in=None):
    """augs.py:137-207: dict name -> (C, Tv, H, W) CUDA tensor (uint8 / float32); tensors with fewer than 4 dimensions pass through.
    `draws_in`: the photometric parameters (photometric_draws) when the caller wants them fixed; drawn from torch's global RNG otherwise."""
    out = {}
    for name, fr in modalities.items():
        if fr.dim() < 4:
            out[name] = fr.clone()
            continue
        C, Tv, H, W = fr.shape
        if 'rgb' in name and (augs_params.get('color_jitter') or augs_params.get('rgb_blur') or augs_params.get('rgb_grayscale')):
            # photometric operators act on the selected, centre-cropped frames at source resolution, before flip / crop / resize (augs.py:175-201)
            import torch
            if not fr.is_floating_point():
                raise NotImplementedError('photometric augmentation of integer rgb frames (the reference feeds float frames in [0, 1], data_kubric.py)')
            draws = draws_in if draws_in is not None else photometric_draws(augs_params)
            fi, ys, xs = crop_maps(augs_params, H, W, out_h, out_w, center_crop)
            y0, x0, h, w = center_rect(H, W, out_h, out_w, center_crop)
            ident = np.arange(len(fi), dtype=np.int32)
            if fr.dtype == torch.float32 and fr.is_contiguous():
                img = photometric_hip(fr, fi, (y0, x0, h, w), draws)               # frame selection + centre crop + the three operators: one kernel
            else:
                # a half / double / strided source: convert only the selected frames and the cropped window, never the whole clip
                sel = fr[:, torch.as_tensor(np.asarray(fi, dtype=np.int64), device=fr.device), y0:y0 + h, x0:x0 + w].to(torch.float32).contiguous()
                img = photometric_hip(sel, ident, (0, 0, h, w), draws)
            if len(ys) == out_h and len(xs) == out_w:
                out[name] = gather_clip(img, ident, (ys - y0).astype(np.int32), (xs - x0).astype(np.int32))
            else:
                out[name] = resize_clip_aa(img, ident, ys - y0, xs - x0, out_h, out_w)
            continue
        if 'segm' in name or 'mask' in name:                                # integer valued: NEAREST (augs.py:196-198)
            fi, sy, sx = index_maps(augs_params, H, W, out_h, out_w, center_crop)
            out[name] = gather_clip(fr.contiguous(), fi, sy, sx)
        else:                                                               # rgb, depth, coordinates: BILINEAR, antialias=True (augs.py:199-201)
            fi, ys, xs = crop_maps(augs_params, H, W, out_h, out_w, center_crop)
            if len(ys) == out_h and len(xs) == out_w:
                out[name] = gather_clip(fr.contiguous(), fi, ys.astype(np.int32), xs.astype(np.int32))      # (the filter is the identity at scale 1)
            else:
                out[name] = resize_clip_aa(fr.float().contiguous(), fi, ys, xs, out_h, out_w)
    return out


def apply_augs_index(modalities, augs_params, out_h, out_w, center_crop=False):
    """augs.py:137-207 for the modalities the index path covers (see module docstring): dict name -> (C, Tv, H, W) CUDA tensor."""
    out = {}
    for name, fr in modalities.items():
        if fr.dim() < 4:
            out[name] = fr.clone()
            continue
        C, Tv, H, W = fr.shape
        fi, sy, sx, resized = index_maps(augs_params, H, W, out_h, out_w, center_crop, return_resized=True)
        smooth = not ('segm' in name or 'mask' in name)
        if smooth and resized:
            raise NotImplementedError(f"'{name}' needs the antialiased bilinear resize of augs.py:199-201: use apply_augs")
        out[name] = gather_clip(fr.contiguous(), fi, sy, sx)
    return out
