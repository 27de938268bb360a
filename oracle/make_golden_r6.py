"""ORACLE tooling (round 6): the forward-time embedding resize, from the REAL reference (imported in place by oracle/ref_shim.py).

    python -m oracle.make_golden_r6

  g17_resize_a / _b   DenseTimeSformer.forward with a STORED pos_embed of a different square grid and a time_embed of a different length than the
                   clip needs (vision_tf.py:103-115: nearest-neighbour resize of the patch rows to (H, W) with H = x.size(1) // W -- the cls row counted --;
                   vision_tf.py:127-132: nearest resize of the time table).  The reference is built for the clip's geometry, then the two tables are
                   grafted after construction (a checkpoint trained at another resolution / clip length), forward + backward through the reference's own
                   autograd: outputs, the gradients of both stored tables (a scatter-add through the resize), cls_token and the patch-embed bias, and
                   the gradient norms of every parameter.  (tcow_amd/engine.py::_effective_embeddings had no reference vector.)
                     a: T = 4, 64x64 (4x4 patches), stored grid 3x3 (up-sampling, ratio 0.75), stored time table of 6 (down-sampling), causal_attention 1
                     b: T = 7, 32x64 (2x4 patches: H != W), stored grid 5x5 (down-sampling), stored time table of 3 (up-sampling), causal_attention 0
Build container only; nothing from the reference is stored except numbers it computed.  Inputs are regenerated from seeds (tcow_amd.synth); the grafted
tables are drawn from synth._rng(seed, 'resize_pos' / 'resize_time').
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, seeker_oracle as so        # noqa: E402
from oracle.make_golden import OUT                      # noqa: E402
from tcow_amd import synth                              # noqa: E402

SEED = 900
PREFIX = 'seeker.tracker_backbone.timesformer.model.'
KEEP = [PREFIX + 'pos_embed', PREFIX + 'time_embed', PREFIX + 'cls_token', PREFIX + 'patch_embed.proj.bias', 'seeker.tracker_post_linear.weight',
        PREFIX + 'blocks.0.temporal_attn.qkv.bias']


def stored_tables(seed, D, grid, t_len):
    """The grafted tables: pos_embed (1, 1 + grid^2, D), time_embed (1, t_len, D), N(0, 0.5) so that the resize visibly matters."""
    pos = synth._rng(seed, 'resize_pos').standard_normal(size=(1, 1 + grid * grid, D), dtype=np.float32) * 0.5
    te = synth._rng(seed, 'resize_time').standard_normal(size=(1, t_len, D), dtype=np.float32) * 0.5
    return pos.astype(np.float32), te.astype(np.float32)


def case(name, cfg, grid, t_len):
    sd = synth.make_state_dict(cfg, SEED)
    D = cfg['embed_dim']
    pos, te = stored_tables(SEED, D, grid, t_len)
    T, H, W = cfg['num_total_frames'], cfg['frame_height'], cfg['frame_width']
    clip = synth.make_clip(2, T, H, W, seed=SEED)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
    net = ref_shim.build_reference_seeker(cfg, sd)
    model = net.seeker.tracker_backbone.timesformer.model
    model.pos_embed = torch.nn.Parameter(torch.from_numpy(pos.copy()))            # grafted after construction: another resolution / clip length
    model.time_embed = torch.nn.Parameter(torch.from_numpy(te.copy()))
    net.eval()
    om, fl = net(rgb, qm)
    Gm = torch.from_numpy(synth._rng(SEED, 'gradprobe_mask').standard_normal(size=tuple(om.shape), dtype=np.float32))
    Gf = torch.from_numpy(synth._rng(SEED, 'gradprobe_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32))
    ((om * Gm).sum() + (fl * Gf).sum()).backward()
    named = dict(net.named_parameters())
    # the restatement on the same grafted state dict, through its own autograd
    sd2 = dict(sd); sd2[PREFIX + 'pos_embed'] = pos; sd2[PREFIX + 'time_embed'] = te
    osd = {k: torch.from_numpy(np.asarray(v).copy()).requires_grad_(True) for k, v in sd2.items()}
    o2, f2 = so.seeker_forward(osd, cfg, rgb, qm)
    ((o2 * Gm).sum() + (f2 * Gf).sum()).backward()
    d_mask = (om - o2).abs().max().item(); d_flags = (fl - f2).abs().max().item()
    assert d_mask < 1e-5 and d_flags < 1e-5, f'oracle deviates from the reference: {d_mask} {d_flags}'
    for k in KEEP:
        e = (osd[k].grad - named[k].grad).abs().max().item() / (named[k].grad.abs().max().item() + 1e-12)
        assert e < 1e-4, f'oracle gradient deviates from the reference for {k}: {e}'
    # the resize really happened: a forward with the tables the constructor made differs
    with torch.no_grad():
        o3, _ = so.seeker_forward(so.to_torch_state_dict(sd), cfg, rgb, qm)
    assert (o3 - om).abs().max().item() > 1e-3
    norms = {k: (float(p.grad.norm()) if p.grad is not None else None) for k, p in named.items()}
    meta = dict(cfg=cfg, B=2, seed=SEED, stored_grid=grid, stored_time=t_len, d_mask=d_mask, d_flags=d_flags, grad_norms=norms,
                generator='oracle/make_golden_r6.py', reference='basilevh/tcow @ /root/reference', torch=torch.__version__)
    arrays = dict(output_mask=om.detach().numpy(), output_flags=fl.detach().numpy())
    for k in KEEP:
        arrays['grad::' + k] = named[k].grad.numpy().copy()
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f'  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)  oracle-vs-reference: mask {d_mask:.2e} flags {d_flags:.2e}')


if __name__ == '__main__':
    assert ref_shim.available(), 'needs /root/reference (build container only)'
    case('g17_resize_a', synth.seeker_config(num_total_frames=4, frame_height=64, frame_width=64, embed_dim=128, depth=2, num_heads=2, causal_attention=1), grid=3, t_len=6)
    case('g17_resize_b', synth.seeker_config(num_total_frames=7, frame_height=32, frame_width=64, embed_dim=128, depth=2, num_heads=2, causal_attention=0), grid=5, t_len=3)
