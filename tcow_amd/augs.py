"""Index half of the reference's clip augmentation pipeline (data/augs.py:50-210, data/data_kubric.py:341-434):
temporal sub-sampling (palindrome / reverse / stride factor / offset), centre crop, horizontal flip, random crop and NEAREST resize.
All of it is integer index arithmetic, so it composes into three small index tables (source frame per clip frame, source row per output
row, source column per output column) and ONE gather pass on the GPU (`tcow_gather_frames`) for every modality at once, bit-exact --
the reference runs slicing + torch.flip + slicing + torchvision Resize as four full-tensor passes per modality on the CPU loader
workers.  Applies to the integer / mask modalities ('segm', 'div_segm', query / target masks: augs.py:196-198) at any size and to rgb
whenever no smooth resize is involved (same size after cropping, i.e. the test-time path of a pre-sized dataset).

Out of scope: the photometric half (ColorJitter / GaussianBlur / Grayscale, augs.py:175-181) and the antialiased bilinear resize of rgb
(augs.py:199-201) are torchvision operators; torchvision is not installed here, so no golden vectors can pin them ("parity unpinned").

`sample_augs_params` draws from numpy's GLOBAL generator in exactly the reference's order, so `np.random.seed(s)` reproduces the
reference's parameters draw for draw (pinned by tests/golden/g13_augs.npz).
"""
import numpy as np


def sample_augs_params(num_frames_load, num_frames_clip, frame_stride, do_random_augs, augs_2d, reverse_prob, palindrome_prob):
    """data/augs.py:50-135."""
    palindrome = False
    reverse = False
    frame_stride_factor = 1
    offset = (num_frames_load - num_frames_clip) // 2
    frame_inds_load = list(range(0, num_frames_load * frame_stride, frame_stride))
    frame_inds_clip = list(range(0, num_frames_clip))
    if do_random_augs:
        palindrome = (np.random.rand() < palindrome_prob)
        if palindrome:
            reverse = (np.random.rand() < 0.35)
            frame_stride_factor = (2 if np.random.rand() < 0.35 else 1)
        else:
            reverse = (np.random.rand() < reverse_prob)
            frame_stride_factor = 1
        if palindrome:
            frame_inds_clip = frame_inds_clip + frame_inds_clip[::-1][1:]
        if reverse:
            frame_inds_clip = frame_inds_clip[::-1]
        if frame_stride_factor > 1:
            frame_inds_clip = frame_inds_clip[::frame_stride_factor]
        avail = len(frame_inds_clip)
        assert avail >= num_frames_clip
        offset = np.random.randint(0, avail - num_frames_clip + 1)
        frame_inds_clip = frame_inds_clip[offset:offset + num_frames_clip]
    p = dict(palindrome=palindrome, reverse=reverse, frame_stride_factor=frame_stride_factor, offset=offset,
             frame_inds_load=np.array(frame_inds_load), frame_inds_clip=np.array(frame_inds_clip))
    color_jitter = rgb_blur = rgb_grayscale = horz_flip = False
    crop_rect = -np.ones(4)
    if do_random_augs:
        color_jitter = (np.random.rand() < 0.9)
        rgb_blur = (np.random.rand() < 0.2)
        rgb_grayscale = (np.random.rand() < 0.05)
        if augs_2d:
            horz_flip = (np.random.rand() < 0.5)
            y1 = np.random.rand() * 0.2; y2 = np.random.rand() * 0.2 + 0.8
            x1 = np.random.rand() * 0.2; x2 = np.random.rand() * 0.2 + 0.8
            crop_rect = np.array([y1, y2, x1, x2])
    p.update(color_jitter=color_jitter, rgb_blur=rgb_blur, rgb_grayscale=rgb_grayscale, horz_flip=horz_flip, crop_rect=crop_rect)
    return p


def index_maps(augs_params, H, W, out_h, out_w, center_crop=False, return_resized=False):
    """Source (frame, row, column) of every output (frame, row, column) for the spatial chain of augs.py:150-203 with NEAREST resize:
    centre crop to the output aspect ratio (torchvision CenterCrop: top = round((H - h) / 2)), horizontal flip, fractional crop rectangle
    (int(y1 * H) : int(y2 * H) on the post-flip image), nearest resize (source = floor(dst * in / out), torch's 'nearest').
    Returns int32 arrays (frame_idx [Tc], src_y [out_h], src_x [out_w]) (+ whether the last step changes the size)."""
    frame_idx = np.asarray(augs_params['frame_inds_clip'], dtype=np.int32)
    y0, x0, h, w = 0, 0, H, W
    if center_crop:                                                        # augs.py:163-171
        cur, want = W / H, out_w / out_h
        if cur > want:
            cw = int(H * want); x0 = int(round((W - cw) / 2.0)); w = cw
        elif cur < want:
            ch = int(W / want); y0 = int(round((H - ch) / 2.0)); h = ch
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
            raise NotImplementedError(f"'{name}' needs the antialiased bilinear resize of augs.py:199-201 (torchvision): out of scope of the index path")
        out[name] = gather_clip(fr.contiguous(), fi, sy, sx)
    return out
