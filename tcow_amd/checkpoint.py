"""Checkpoint interop for the drop-in Seeker.

* `load_tcow_checkpoint`: build a Seeker from a reference `checkpoint.pth` (`seeker_args` + `net_seeker`,
  written at train.py:269-304 and consumed at eval/inference.py:38-54).
* `load_pretrained_vit`: the ViT-B/16 -> TimeSformer weight surgery the reference applies for
  `tracker_pretrained` (third_party/TimeSformer/timesformer/models/helpers.py:100-205): inflate the 3-channel
  patch conv to 3+query channels (repeat, scale by 3/C), nearest-resize pos/time embeddings, copy
  attn -> temporal_attn and norm1 -> temporal_norm1, drop the classifier, load non-strictly.
  The ImageNet URL of vit.py:35 needs network access; offline only a file path works.
"""
import math

import torch
import torch.nn.functional as F

from ._lib import TcowError


def pretrained_surgery(state_dict, in_chans, num_patches, num_frames):
    """helpers.py:117-199 on a plain ViT state dict (keys relative to the VisionTransformer)."""
    sd = dict(state_dict)
    if 'model' in sd and isinstance(sd['model'], dict):
        sd = dict(sd['model'])                                            # helpers.py:113-114
    w = sd.get('patch_embed.proj.weight')
    if w is not None and in_chans != 3:
        w = w.float()
        if w.shape[1] != 3:
            del sd['patch_embed.proj.weight']                             # helpers.py:141-144
        else:
            rep = int(math.ceil(in_chans / 3))
            w = w.repeat(1, rep, 1, 1)[:, :in_chans] * (3 / float(in_chans))   # helpers.py:146-150
            sd['patch_embed.proj.weight'] = w
    for k in ('head.weight', 'head.bias'):                                # num_classes == 0 -> dropped (helpers.py:160-165)
        sd.pop(k, None)
    if 'pos_embed' in sd and num_patches + 1 != sd['pos_embed'].size(1):  # helpers.py:169-176
        pe = sd['pos_embed']
        cls_pe = pe[0, 0, :].unsqueeze(0).unsqueeze(1)
        other = pe[0, 1:, :].unsqueeze(0).transpose(1, 2)
        new = F.interpolate(other, size=(num_patches), mode='nearest').transpose(1, 2)
        sd['pos_embed'] = torch.cat((cls_pe, new), 1)
    if 'time_embed' in sd and num_frames != sd['time_embed'].size(1):     # helpers.py:179-182
        te = sd['time_embed'].transpose(1, 2)
        sd['time_embed'] = F.interpolate(te, size=(num_frames), mode='nearest').transpose(1, 2)
    out = dict(sd)
    for key in sd:                                                        # helpers.py:185-199
        if 'blocks' in key and 'attn' in key:
            nk = key.replace('attn', 'temporal_attn')
            if nk not in sd:
                out[nk] = sd[key]
        if 'blocks' in key and 'norm1' in key:
            nk = key.replace('norm1', 'temporal_norm1')
            if nk not in sd:
                out[nk] = sd[key]
    return out


def load_pretrained_vit(tracker, path, logger=None):
    if not path:
        raise TcowError('tracker_pretrained=True needs the ImageNet ViT-B/16 weights from the network (vit.py:35); '
                        'pass a checkpoint file path as tracker_pretrained instead, or False')
    sd = torch.load(path, map_location='cpu')
    if 'state_dict' in sd:
        sd = sd['state_dict']
    vit = tracker.vit
    n_patches = vit.pos_embed.shape[1] - 1
    sd = pretrained_surgery(sd, tracker.input_channels, n_patches, tracker.num_total_frames)
    missing, unexpected = vit.load_state_dict(sd, strict=False)           # helpers.py:202
    if logger is not None:
        logger.info(f'(tcow_amd) pretrained ViT loaded: {len(missing)} missing, {len(unexpected)} unexpected keys')
    return missing, unexpected


def load_tcow_checkpoint(path, logger=None, device='cuda', precision='bf16'):
    """eval/inference.py:38-54: checkpoint['seeker_args'] -> Seeker(**args); load_state_dict(net_seeker)."""
    from .seeker import Seeker
    ck = torch.load(path, map_location='cpu')
    args = dict(ck['seeker_args'])
    args['tracker_pretrained'] = False        # weights come from the checkpoint itself (inference.py:46-47 does the same)
    net = Seeker(logger, precision=precision, **args)
    net.load_state_dict(ck['net_seeker'], strict=True)
    return net.to(device)
