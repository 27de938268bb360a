"""ORACLE tooling: generate the golden fixtures under tests/golden/ from the REAL reference (imported in place
from /root/reference by oracle/ref_shim.py) and check our CPU restatement against it on the way.

Run in the build container only:  python -m oracle.make_golden [--full]
The fixtures hold data only (seeds, configs, reference outputs); weights and inputs are regenerated from the seeds by
tcow_amd.synth on whichever machine runs the tests.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, seeker_oracle as so   # noqa: E402
from tcow_amd import synth                          # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
SEED = 900


def probe_grad_tensor(shape, name):
    r = synth._rng(SEED, 'gradprobe_' + name)
    return r.standard_normal(size=shape, dtype=np.float32)


def run_reference(cfg, B, with_grad=False, inst=0):
    sd = synth.make_state_dict(cfg, SEED)
    net = ref_shim.build_reference_seeker(cfg, sd)
    T, H, W = cfg['num_total_frames'], cfg['frame_height'], cfg['frame_width']
    clip = synth.make_clip(B, T, H, W, seed=SEED)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, inst, 0))
    t0 = time.time()
    if with_grad:
        om, fl = net(rgb, qm)
    else:
        with torch.no_grad():
            om, fl = net(rgb, qm)
    t_ref = time.time() - t0
    with torch.no_grad():
        om2, fl2 = so.seeker_forward(so.to_torch_state_dict(sd), cfg, rgb, qm)
    d_mask = (om.detach() - om2).abs().max().item(); d_flags = (fl.detach() - fl2).abs().max().item()
    assert d_mask < 1e-5 and d_flags < 1e-5, f'oracle restatement deviates from the reference: {d_mask} {d_flags}'
    extra = {}
    if with_grad:
        Gm = torch.from_numpy(probe_grad_tensor(tuple(om.shape), 'mask')); Gf = torch.from_numpy(probe_grad_tensor(tuple(fl.shape), 'flags'))
        loss = (om * Gm).sum() + (fl * Gf).sum()
        loss.backward()
        named = dict(net.named_parameters())
        norms = {k: float(p.grad.norm()) if p.grad is not None else None for k, p in named.items()}
        extra['grad_norms'] = norms
        keep = ['seeker.tracker_post_linear.weight', 'seeker.tracker_backbone.timesformer.model.cls_token',
                'seeker.tracker_backbone.timesformer.model.time_embed', 'seeker.tracker_backbone.timesformer.model.pos_embed',
                'seeker.tracker_backbone.timesformer.model.blocks.0.temporal_attn.qkv.bias',
                'seeker.tracker_backbone.timesformer.model.blocks.1.attn.proj.weight',
                'seeker.tracker_backbone.timesformer.model.blocks.0.temporal_fc.weight',
                'seeker.tracker_backbone.timesformer.model.blocks.1.norm2.weight',
                'seeker.tracker_backbone.timesformer.model.patch_embed.proj.bias']
        extra['grads'] = {k: named[k].grad.numpy().copy() for k in keep}
        # the oracle's own autograd must agree with the reference's
        osd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd.items()}
        o3, f3 = so.seeker_forward(osd, cfg, rgb, qm)
        ((o3 * Gm).sum() + (f3 * Gf).sum()).backward()
        for k in keep:
            e = (osd[k].grad - named[k].grad).abs().max().item() / (named[k].grad.abs().max().item() + 1e-12)
            assert e < 1e-4, f'oracle gradient deviates from the reference for {k}: {e}'
    return om.detach().numpy(), fl.detach().numpy(), dict(d_mask=d_mask, d_flags=d_flags, t_ref=t_ref), extra


def save(name, cfg, B, arrays, meta):
    os.makedirs(OUT, exist_ok=True)
    meta = dict(meta); meta.update(cfg=cfg, B=B, seed=SEED, generator='oracle/make_golden.py', reference='basilevh/tcow @ /root/reference',
                                   torch=torch.__version__)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f'  wrote {name}.npz  ({os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024:.1f} KiB)  oracle-vs-reference: mask {meta["d_mask"]:.2e} flags {meta["d_flags"]:.2e}')


def pooled(om, st):
    B, C, T, H, W = om.shape
    return torch.nn.functional.avg_pool2d(torch.from_numpy(om).permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), st, st).numpy() if st > 1 else None


def index_maps():
    """G6: integer index maps obtained by pushing index tensors through the reference's own einops patterns."""
    from einops import rearrange
    out = {}
    for (B, T, Hp, Wp) in [(1, 30, 15, 20), (2, 4, 4, 4)]:
        N = Hp * Wp
        # token order: '(b t) n m -> (b n) t m' then '(b n) t m -> b (n t) m' (vision_tf.py:124,137); value = t*N + n of the source
        idx = torch.arange(B * T * N).reshape(B * T, N, 1)
        y = rearrange(idx, '(b t) n m -> (b n) t m', b=B, t=T)
        y = rearrange(y, '(b n) t m -> b (n t) m', b=B, t=T)            # (B, N*T, 1): position p holds source (b, t, n)
        out[f'token_src_{B}_{T}_{Hp}_{Wp}'] = y[..., 0].numpy().astype(np.int32)
        # final 'B (H W T) D -> B D T H W' (vision_tf.py:161): value at (b, t, h, w) = position in the token list
        pos = torch.arange(B * N * T).reshape(B, N * T, 1)
        z = rearrange(pos, 'B (H W T) D -> B D T H W', B=B, T=T, H=Hp, W=Wp, D=1)
        out[f'token_pos_{B}_{T}_{Hp}_{Wp}'] = z[:, 0].numpy().astype(np.int32)
    for (C, P, Hp, Wp) in [(3, 16, 2, 3), (3, 4, 2, 2)]:
        # un-patchify 'B T H W (C h w) -> B C T (H h) (W w)' (mask_tracker.py:114): value = flat (H,W,(C h w)) source index
        src = torch.arange(Hp * Wp * C * P * P).reshape(1, 1, Hp, Wp, C * P * P)
        m = rearrange(src, 'B T H W (C h w) -> B C T (H h) (W w)', C=C, h=P, w=P)
        out[f'unpatchify_{C}_{P}_{Hp}_{Wp}'] = m[0, :, 0].numpy().astype(np.int32)
    # patchify: Conv2d(k=P, s=P) weight flattening c*P*P + py*P + px (vit.py:233): checked through a one-hot conv
    P, C = 4, 4
    conv = torch.nn.Conv2d(C, C * P * P, P, P, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.eye(C * P * P).reshape(C * P * P, C, P, P))
        img = torch.arange(C * 8 * 12, dtype=torch.float32).reshape(1, C, 8, 12)
        pat = conv(img).flatten(2).transpose(1, 2)                      # (1, N, C*P*P): entry = source pixel flat index
    out['patchify_4_4_2_3'] = pat[0].numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, 'g6_index_maps.npz'), **out)
    print('  wrote g6_index_maps.npz')


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--full', action='store_true'); args = ap.parse_args()
    assert ref_shim.available(), 'reference tree not found'
    torch.manual_seed(0)
    base = dict(num_total_frames=4, frame_height=64, frame_width=64)
    # G1: BASELINE configs[0]: 2-layer / 4-head at D=64 (head_dim 16: oracle pin only) and its D=256 twin (head_dim 64: HIP path)
    for name, kw in [('g1_cfg1_d64', dict(embed_dim=64, depth=2, num_heads=4, causal_attention=1)),
                     ('g1_cfg1_d256', dict(embed_dim=256, depth=2, num_heads=4, causal_attention=1))]:
        cfg = synth.seeker_config(**base, **kw)
        om, fl, meta, extra = run_reference(cfg, 2, with_grad=True)
        arrays = dict(output_mask=om, output_flags=fl)
        for k, v in extra['grads'].items():
            arrays['grad::' + k] = v
        meta['grad_norms'] = extra['grad_norms']
        save(name, cfg, 2, arrays, meta)
    # G2: behaviour switches
    variants = {'ca0': dict(causal_attention=0), 'ca2': dict(causal_attention=2), 'ca3': dict(causal_attention=3), 'cam1': dict(causal_attention=-1),
                'normemb_nearest': dict(causal_attention=1, norm_embeddings=True, track_map_resize='nearest'),
                'stride1_prenorm': dict(causal_attention=1, track_map_stride=1, pretrained_norm=True),
                'stride2': dict(causal_attention=1, track_map_stride=2)}
    for vn, kw in variants.items():
        cfg = synth.seeker_config(**base, embed_dim=256, depth=2, num_heads=4, **kw)
        om, fl, meta, _ = run_reference(cfg, 1)
        save('g2_' + vn, cfg, 1, dict(output_mask=om, output_flags=fl), meta)
    # G3: mid-size, native 12-layer Seeker
    cfg = synth.seeker_config(num_total_frames=8, frame_height=96, frame_width=128, causal_attention=1)
    om, fl, meta, _ = run_reference(cfg, 1)
    save('g3_mid_T8_96x128', cfg, 1, dict(pooled=pooled(om, 4), output_flags=fl, frame_sum=om.sum(axis=(3, 4)), frame_absmax=np.abs(om).max(axis=(3, 4)),
                                          logit_std=np.float32(om.std())), meta)
    if args.full:
        # G4: BASELINE configs[1] geometry at full size (one query forward, fp32 reference on CPU)
        cfg = synth.seeker_config(causal_attention=1)
        om, fl, meta, _ = run_reference(cfg, 1)
        save('g4_cfg2_T30_240x320', cfg, 1, dict(pooled=pooled(om, 4), output_flags=fl, frame_sum=om.sum(axis=(3, 4)), frame_absmax=np.abs(om).max(axis=(3, 4)),
                                                  logit_std=np.float32(om.std())), meta)
    index_maps()


if __name__ == '__main__':
    main()
