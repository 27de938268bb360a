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


def _conv_filter(sd, patch_size=16):
    """vit.py:381-390: a patch-embedding weight stored as a flattened linear projection becomes a (D, 3, P, P) conv weight."""
    out = {}
    for k, v in sd.items():
        if 'patch_embed.proj.weight' in k:
            if v.shape[-1] != patch_size:
                patch_size = v.shape[-1]
            v = v.reshape((v.shape[0], 3, patch_size, patch_size))
        out[k] = v
    return out


def pretrained_surgery(state_dict, in_chans, num_patches, num_frames, patch_size=16):
    """helpers.py:100-205 (`load_pretrained` as TimeSformer.__init__ calls it, vit.py:462-464: num_classes = 0, filter_fn =
    _conv_filter, attention_type = 'divided_space_time') on a plain image-ViT state dict; returns the dict that is then loaded
    non-strictly.  Pinned key-for-key against the reference's own function in tests/golden/g10_pretrained.npz."""
    sd = dict(state_dict)
    if 'model' in sd and isinstance(sd['model'], dict):
        sd = dict(sd['model'])                                            # helpers.py:113-114
    sd = _conv_filter(sd, patch_size)                                     # helpers.py:116-117
    w = sd.get('patch_embed.proj.weight')
    if w is not None and in_chans != 3:
        conv_type = w.dtype
        w = w.float()
        if w.shape[1] != 3:
            del sd['patch_embed.proj.weight']                             # helpers.py:141-144
        else:
            rep = int(math.ceil(in_chans / 3))
            w = w.repeat(1, rep, 1, 1)[:, :in_chans] * (3 / float(in_chans))   # helpers.py:146-150
            sd['patch_embed.proj.weight'] = w.to(conv_type)
    for k in ('head.weight', 'head.bias'):                                # num_classes == 0 != 1000 -> classifier dropped (helpers.py:160-165)
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


def _torch_load(path, trusted=False):
    """Tensor-only loading first (torch.load(weights_only=True) with argparse.Namespace allow-listed: train.py:269-283 stores `train_args` as
    one next to the tensors); only a file the caller declares TRUSTED -- a TCOW checkpoint.pth written by train.py or save_tcow_checkpoint,
    which may hold further plain-Python objects -- falls back to the full unpickler, which executes code from the file.  Downloaded image-ViT
    weights (`tracker_pretrained=<path>`) are never loaded that way."""
    import argparse
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location='cpu', weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError):
        if not trusted:
            raise
    return torch.load(path, map_location='cpu', weights_only=False)


def load_state_dict_file(path):
    """helpers.py:24-52: unwrap 'state_dict' (stripping a leading 'module') / 'model_state' (stripping a leading 'model')."""
    ck = _torch_load(path)
    if isinstance(ck, dict) and 'state_dict' in ck:
        return {(k[7:] if k.startswith('module') else k): v for k, v in ck['state_dict'].items()}
    if isinstance(ck, dict) and 'model_state' in ck:
        return {(k[6:] if k.startswith('model') else k): v for k, v in ck['model_state'].items()}
    return ck


def load_pretrained_vit(tracker, path, logger=None):
    if not path:
        raise TcowError('tracker_pretrained=True needs the ImageNet ViT-B/16 weights from the network (vit.py:35); '
                        'pass a checkpoint file path as tracker_pretrained instead, or False')
    sd = load_state_dict_file(path)
    vit = tracker.vit
    n_patches = vit.pos_embed.shape[1] - 1
    sd = pretrained_surgery(sd, tracker.input_channels, n_patches, tracker.num_total_frames, tracker.patch_size)
    missing, unexpected = vit.load_state_dict(sd, strict=False)           # helpers.py:202
    if logger is not None:
        logger.info(f'(tcow_amd) pretrained ViT loaded: {len(missing)} missing, {len(unexpected)} unexpected keys')
    return missing, unexpected


def parse_tracker_pretrained(value):
    """mask_tracker.py:55-67: (flag, path) of a `tracker_pretrained` constructor argument."""
    if isinstance(value, bool):
        return value, ''
    if isinstance(value, str):
        if value.lower() in ['1', 'y', 'yes', 't', 'true']:
            return True, ''
        if len(value) <= 5:
            return False, ''
        return True, value
    raise ValueError(f'Invalid tracker_pretrained value: {value}.')


def load_tcow_checkpoint(path, logger=None, device='cuda', precision='bf16'):
    """eval/inference.py:38-54: checkpoint['seeker_args'] -> Seeker(**args); load_state_dict(net_seeker).

    The reference passes `seeker_args` through unchanged, so a model trained from the ImageNet ViT (`tracker_pretrained='1'`, the CLI
    default args.py:150) is rebuilt with `pretrained=True`: that re-downloads weights which the checkpoint then overwrites, and --
    the part that matters -- keeps the (rgb - 0.45) / 0.225 input normalisation of vision_tf.py:81-89 switched on.  Here the module is
    constructed without the (offline-impossible, redundant) download and the parsed flag is restored afterwards."""
    from .seeker import Seeker
    ck = _torch_load(path, trusted=True)
    args = dict(ck['seeker_args'])
    flag, _ = parse_tracker_pretrained(args.get('tracker_pretrained', False))
    args['tracker_pretrained'] = False
    net = Seeker(logger, precision=precision, **args)
    net.load_state_dict(ck['net_seeker'], strict=True)
    net.seeker.tracker_pretrained = flag
    return net.to(device)


def save_tcow_checkpoint(directory, epoch, net, optimizer=None, lr_scheduler=None, seeker_args=None, train_args=None, dset_args=None, name='tcow_amd',
                         checkpoint_every=0):
    """train.py:269-304 (`save_model_checkpoint`): the reference's checkpoint dictionary and side files, so that its own `--resume`
    (train.py:246-257) and eval/inference.py:38-54 read what this framework trained.  `net` is the Seeker (un-wrapped: the reference
    saves `networks_nodp`); optimizer / lr_scheduler state dicts use torch's layout (FusedAdamWClip keeps torch.optim.AdamW's).
    seeker_args defaults to the constructor keywords the Seeker recorded; train_args (the reference's argparse.Namespace: eval/inference.py
    deep-copies it and reads .num_queries etc.) must be supplied by the caller for checkpoints the reference's tools are to read.
    checkpoint_every > 0 also writes the periodic copy model_{epoch}.pth of train.py:297-300, which inference.py reads for epoch >= 0."""
    import os
    import shutil
    import numpy as np
    os.makedirs(directory, exist_ok=True)
    if seeker_args is None:
        seeker_args = getattr(net, 'seeker_args', None)
        if not seeker_args:
            raise TcowError('save_tcow_checkpoint: no seeker_args (the module did not record its constructor keywords): pass seeker_args=')
    tracker = getattr(net, 'seeker', None)
    checkpoint = {'epoch': epoch, 'train_args': train_args, 'dset_args': dset_args if dset_args is not None else {},
                  'seeker_args': dict(seeker_args),
                  'net_seeker': {k: v.detach().cpu() for k, v in net.state_dict().items()}}
    if tracker is not None and getattr(tracker, 'precision', None) == 'fp16':
        checkpoint['tcow_amd_loss_scale_log2'] = float(tracker.ls_log2)          # (extra key: ignored by the reference's readers)
    if optimizer is not None:
        checkpoint['optim_seeker'] = optimizer.state_dict()
    if lr_scheduler is not None:
        checkpoint['lr_sched_seeker'] = lr_scheduler.state_dict()
    path = os.path.join(directory, 'checkpoint.pth')
    torch.save(checkpoint, path)
    np.savetxt(os.path.join(directory, 'checkpoint_epoch.txt'), np.array([epoch], dtype=np.int32), fmt='%d')       # train.py:292-295
    np.savetxt(os.path.join(directory, 'checkpoint_name.txt'), np.array([name]), fmt='%s')
    if checkpoint_every > 0 and epoch % checkpoint_every == 0:                                                      # train.py:297-300
        shutil.copy(path, os.path.join(directory, f'model_{epoch}.pth'))
    return path


def resume_tcow_checkpoint(path, net, optimizer=None, lr_scheduler=None):
    """train.py:246-257: weights, optimizer and scheduler state of a checkpoint.pth into live objects; returns the epoch to start from."""
    ck = _torch_load(path, trusted=True)
    net.load_state_dict(ck['net_seeker'], strict=True)
    tracker = getattr(net, 'seeker', None)
    if tracker is not None and hasattr(tracker, 'invalidate_weight_cache'):
        tracker.invalidate_weight_cache()
    if tracker is not None and 'tcow_amd_loss_scale_log2' in ck and 'ls_log2' in tracker._buffers:
        with torch.no_grad():
            tracker.ls_log2.fill_(float(ck['tcow_amd_loss_scale_log2']))
    if optimizer is not None and ck.get('optim_seeker'):
        optimizer.load_state_dict(ck['optim_seeker'])
    if lr_scheduler is not None and ck.get('lr_sched_seeker'):
        lr_scheduler.load_state_dict(ck['lr_sched_seeker'])
    return int(ck['epoch']) + 1
