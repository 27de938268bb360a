"""ORACLE tooling (round 2): more golden fixtures from the REAL reference (imported in place by oracle/ref_shim.py).

    python -m oracle.make_golden_r2 [--only g7,g8,g9,g10,g11]

  g7_cfg2_grads     BASELINE configs[1] at full size, the Qs = 3 queries of one clip: reference forward + backward per query
                    (pipeline.py:134-158 calls the Seeker once per query; one .backward() sums the three graphs, train.py:98),
                    loss = sum_q <output_mask_q, Gm_q> + <output_flags_q, Gf_q> with seeded probe tensors.  Stores every
                    parameter's gradient norm, small gradient tensors in full and strided samples of the large ones.
  g8_cfg3_long      BASELINE configs[3]: T=60 480x640 inference forward (S = 1201): pooled logits of six frames + per-frame sums.
  g9_cfg4_eval      BASELINE configs[4]: 4 queries x 4 temporal strides of one synthetic plugin-shaped video through the
                    reference's MyTrainPipeline.forward_plugin + calculate_metrics_mask_track (pipeline.py:202-240,
                    eval/metrics.py:9-113) on 6 of the 16 items, and data_utils.get_usage_modes (data_utils.py:301-342).
  g10_pretrained    helpers.load_pretrained (helpers.py:100-205) run offline on a toy ViT checkpoint file: the resulting
                    TimeSformer state dict (pins tcow_amd.checkpoint.pretrained_surgery) + a forward golden with the
                    pretrained rgb normalisation (vision_tf.py:81-89) on.
  g11_depth18/24    V0: the reference's native depth-18 (D=896, 14 heads) and depth-24 (D=1024, 16 heads) Seekers, small clip.
Build container only; nothing from the reference is stored except numbers it computed.
"""
import argparse
import os
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, seeker_oracle as so        # noqa: E402
from oracle.make_golden import OUT, SEED, pooled, save   # noqa: E402
from tcow_amd import synth                                # noqa: E402

G7_FULL = ['seeker.tracker_backbone.timesformer.model.cls_token', 'seeker.tracker_backbone.timesformer.model.time_embed',
           'seeker.tracker_backbone.timesformer.model.patch_embed.proj.bias', 'seeker.tracker_post_linear.bias',
           'seeker.tracker_backbone.timesformer.model.blocks.0.temporal_attn.qkv.bias',
           'seeker.tracker_backbone.timesformer.model.blocks.5.attn.qkv.bias',
           'seeker.tracker_backbone.timesformer.model.blocks.11.mlp.fc1.bias',
           'seeker.tracker_backbone.timesformer.model.blocks.3.norm2.weight',
           'seeker.tracker_backbone.timesformer.model.blocks.7.temporal_norm1.bias',
           'seeker.tracker_backbone.timesformer.model.blocks.0.temporal_fc.bias']
G7_SAMPLED = ['seeker.tracker_post_linear.weight', 'seeker.tracker_backbone.timesformer.model.pos_embed',
              'seeker.tracker_backbone.timesformer.model.patch_embed.proj.weight',
              'seeker.tracker_backbone.timesformer.model.blocks.0.temporal_attn.qkv.weight',
              'seeker.tracker_backbone.timesformer.model.blocks.0.temporal_fc.weight',
              'seeker.tracker_backbone.timesformer.model.blocks.5.attn.qkv.weight',
              'seeker.tracker_backbone.timesformer.model.blocks.5.attn.proj.weight',
              'seeker.tracker_backbone.timesformer.model.blocks.8.temporal_attn.proj.weight',
              'seeker.tracker_backbone.timesformer.model.blocks.11.mlp.fc1.weight',
              'seeker.tracker_backbone.timesformer.model.blocks.11.mlp.fc2.weight',
              'seeker.tracker_backbone.timesformer.model.blocks.0.mlp.fc2.weight']


sample_grad = so.grad_sample


def g7():
    cfg = synth.seeker_config(causal_attention=1)
    sd = synth.make_state_dict(cfg, SEED)
    net = ref_shim.build_reference_seeker(cfg, sd)
    net.train()                                              # drop_path_rate is 0 in build_reference_seeker: train == eval arithmetic
    clip = synth.make_clip(1, 30, 240, 320, seed=SEED)
    rgb = torch.from_numpy(clip['rgb'])
    Qs = 3
    oms, fls = [], []
    t0 = time.time()
    for q in range(Qs):
        qm = torch.from_numpy(synth.make_query_mask(clip, q, 0))
        om, fl = net(rgb.clone(), qm)
        Gm = torch.from_numpy(synth._rng(SEED, f'g7_mask_{q}').standard_normal(size=tuple(om.shape), dtype=np.float32))
        Gf = torch.from_numpy(synth._rng(SEED, f'g7_flags_{q}').standard_normal(size=tuple(fl.shape), dtype=np.float32))
        ((om * Gm).sum() * 1e-3 + (fl * Gf).sum()).backward()          # gradients accumulate over the three graphs
        oms.append(om.detach().numpy()); fls.append(fl.detach().numpy())
        print(f'  g7 query {q}: {time.time() - t0:.0f} s', flush=True)
    named = dict(net.named_parameters())
    arrays = {'output_flags': np.concatenate(fls, 0), 'pooled': np.concatenate([pooled(o, 4) for o in oms], 0),
              'logit_std': np.float32(np.concatenate(oms, 0).std())}
    norms = {k: (float(p.grad.norm()) if p.grad is not None else None) for k, p in named.items()}
    for k in G7_FULL:
        arrays['grad::' + k] = named[k].grad.numpy().copy()
    for k in G7_SAMPLED:
        arrays['gsample::' + k] = sample_grad(named[k].grad.detach().numpy())
    meta = dict(d_mask=0.0, d_flags=0.0, t_ref=time.time() - t0, grad_norms=norms, queries=Qs, mask_probe_scale=1e-3)
    save('g7_cfg2_grads', cfg, 1, arrays, meta)


def g8():
    cfg = synth.seeker_config(num_total_frames=60, frame_height=480, frame_width=640, causal_attention=1)
    sd = synth.make_state_dict(cfg, SEED)
    net = ref_shim.build_reference_seeker(cfg, sd)
    clip = synth.make_clip(1, 60, 480, 640, seed=SEED)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
    t0 = time.time()
    with torch.no_grad():
        om, fl = net(rgb, qm)
    t_ref = time.time() - t0
    om = om.numpy()
    frames = [0, 1, 29, 30, 58, 59]
    pl = pooled(om, 4).reshape(60, 3, 120, 160)
    save('g8_cfg3_long', cfg, 1, dict(pooled_frames=pl[frames], frames=np.asarray(frames), output_flags=fl.numpy(), frame_sum=om.sum(axis=(3, 4)),
                                     frame_absmax=np.abs(om).max(axis=(3, 4)), logit_std=np.float32(om.std()),
                                     positive_frac=np.float32((om > 0).mean())), dict(d_mask=0.0, d_flags=0.0, t_ref=t_ref))


def g9():
    from tcow_amd import plugin_data as pd
    mods = ref_shim.load()
    # (1) usage modes: the reference's own function on several availability patterns
    cases = [(list(range(120)), [0], list(range(0, 120, 5)), 30, 0, 0), (list(range(120)), [0, 10, 40], list(range(0, 120, 5)), 30, 0, 2),
             (list(range(64)), [12, 13], [12, 40, 41, 63], 16, 3, 1), (list(range(50)), [49], [], 8, 7, 0), (list(range(35)), [5], [5, 6], 30, 0, 1)]
    arrays = {}
    for i, c in enumerate(cases):
        ref = mods['data_utils'].get_usage_modes(c[0], c[1], c[2], c[3], c[4], min_target_frames_covered=c[5])
        ours = pd.get_usage_modes(c[0], c[1], c[2], c[3], c[4], min_target_frames_covered=c[5])
        assert ref == ours, (i, ref, ours)
        arrays[f'modes_{i}'] = np.asarray(ref, dtype=np.float64).reshape(-1, 3)
        arrays[f'modes_{i}_args'] = np.asarray([len(c[0]), c[3], c[4], c[5]] + [-1] + c[1] + [-1] + c[2], dtype=np.int64)
    # (2) forward_plugin + metrics on 6 of the 16 (query, stride) items of the config-2 shape
    cfg = synth.seeker_config(causal_attention=1)
    net = ref_shim.build_reference_seeker(cfg, synth.make_state_dict(cfg, SEED))
    video = synth.make_plugin_video(120, 240, 320, seed=SEED)
    items = pd.eval_items(video, num_frames=30, query_time_idx=0, queries=(0, 1, 2, 3), strides=(1, 2, 3, 4))
    assert len(items) == 16
    args = SimpleNamespace(num_frames=30, num_queries=1)
    pipe = mods['pipeline'].MyTrainPipeline(args, ref_shim.NullLogger(), {'seeker': net}, 'cpu')
    pipe.set_phase('test')
    pick = [0, 3, 5, 10, 12, 15]
    t0 = time.time()
    for i in pick:
        it = items[i]
        data = {'source_name': ['plugin'], 'within_batch_idx': torch.arange(1), 'pv_rgb_tf': torch.from_numpy(it['pv_rgb_tf'])[None],
                'pv_query_tf': torch.from_numpy(it['pv_query_tf'])[None], 'pv_target_tf': torch.from_numpy(it['pv_target_tf'])[None]}
        with torch.no_grad():
            mr, lr = pipe(data, 0, 0, 0, 0.0, True, True)                  # include_loss, metrics_only (eval/inference.py:75)
        om = mr['output_mask'].numpy()
        arrays[f'item{i}::pooled'] = pooled(om, 4)[::4]                      # every 4th frame (8 of 30)
        arrays[f'item{i}::output_flags'] = mr['output_flags'].numpy()
        arrays[f'item{i}::frame_sum'] = om.sum(axis=(3, 4))
        for k, v in lr['metrics'].items():
            arrays[f'item{i}::metric::{k}'] = np.asarray(v.numpy() if torch.is_tensor(v) else v)
        print(f'  g9 item {i}: {time.time() - t0:.0f} s', flush=True)
    torch.set_grad_enabled(True)
    arrays['picked'] = np.asarray(pick)
    arrays['item_query_stride'] = np.asarray([[it['query'], it['frame_stride']] for it in items])
    save('g9_cfg4_eval', cfg, 1, arrays, dict(d_mask=0.0, d_flags=0.0, t_ref=time.time() - t0, video_seed=SEED, video_frames=120))


def toy_vit_checkpoint(D, depth, P, n_patches, seed):
    """A plain image-ViT state dict as timm / the ImageNet checkpoint of vit.py:35 lays it out (3-channel conv, head, no temporal keys)."""
    r = lambda name, shape, s=0.05: torch.from_numpy((synth._rng(seed, 'toyvit.' + name).standard_normal(size=shape, dtype=np.float32) * s).astype(np.float32))
    sd = {'cls_token': r('cls', (1, 1, D)), 'pos_embed': r('pos', (1, n_patches + 1, D)), 'patch_embed.proj.weight': r('pe.w', (D, 3, P, P)),
          'patch_embed.proj.bias': r('pe.b', (D,)), 'norm.weight': 1 + r('n.w', (D,)), 'norm.bias': r('n.b', (D,)),
          'head.weight': r('h.w', (10, D)), 'head.bias': r('h.b', (10,))}
    for i in range(depth):
        b = f'blocks.{i}.'
        sd.update({b + 'norm1.weight': 1 + r(b + 'n1w', (D,)), b + 'norm1.bias': r(b + 'n1b', (D,)), b + 'attn.qkv.weight': r(b + 'qkvw', (3 * D, D)),
                   b + 'attn.qkv.bias': r(b + 'qkvb', (3 * D,)), b + 'attn.proj.weight': r(b + 'pw', (D, D)), b + 'attn.proj.bias': r(b + 'pb', (D,)),
                   b + 'norm2.weight': 1 + r(b + 'n2w', (D,)), b + 'norm2.bias': r(b + 'n2b', (D,)), b + 'mlp.fc1.weight': r(b + 'f1w', (4 * D, D)),
                   b + 'mlp.fc1.bias': r(b + 'f1b', (4 * D,)), b + 'mlp.fc2.weight': r(b + 'f2w', (D, 4 * D)), b + 'mlp.fc2.bias': r(b + 'f2b', (D,))})
    return sd


def g10():
    """helpers.load_pretrained through the shim, on a geometry that exercises every branch: 3 -> 4 input channels, pos_embed
    16 -> 24 patches (nearest), no time_embed in the file, attn -> temporal_attn / norm1 -> temporal_norm1 copies, head dropped."""
    from functools import partial
    mods = ref_shim.load()
    import timesformer.models.helpers as helpers
    vit = mods['vit']
    D, depth, heads, P, T, H, W = 128, 2, 2, 16, 4, 64, 96
    cfg = synth.seeker_config(num_total_frames=T, frame_height=H, frame_width=W, embed_dim=D, depth=depth, num_heads=heads, causal_attention=1, pretrained_norm=True)
    toy = toy_vit_checkpoint(D, depth, P, 16, SEED)
    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        for wrap in ('plain', 'state_dict', 'model'):
            path = os.path.join(td, f'vit_{wrap}.pth')
            torch.save(toy if wrap == 'plain' else {wrap: toy}, path)
            torch.manual_seed(SEED)
            model = vit.VisionTransformer(img_size=(H, W), patch_size=P, in_chans=4, num_classes=0, embed_dim=D, depth=depth, num_heads=heads, mlp_ratio=4,
                                          qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), drop_path_rate=0., num_frames=T,
                                          attention_type='divided_space_time', causal_attention=1)
            model.default_cfg = vit.default_cfgs['catchall']
            before = {k: v.clone() for k, v in model.state_dict().items()}
            dc = dict(model.default_cfg); dc['url'] = 'offline'          # helpers.py:103 insists on a non-empty url even for file loads
            helpers.load_pretrained(model, cfg=dc, num_classes=0, in_chans=4, filter_fn=vit._conv_filter, img_size=(H, W), num_frames=T,
                                    num_patches=(H // P) * (W // P), attention_type='divided_space_time', pretrained_model=path)
            after = model.state_dict()
            if wrap == 'plain':
                for k, v in after.items():
                    arrays['sd::' + k] = v.numpy().copy()
                    arrays['changed::' + k] = np.bool_(not torch.equal(v, before[k]))
                ref_after = {k: v.clone() for k, v in after.items()}
            else:
                assert all(torch.equal(after[k], ref_after[k]) for k in after), wrap      # the wrappers load_state_dict() unwraps give the same result
    # forward golden: a Seeker whose backbone carries these weights (heads from synth), rgb normalisation ON
    sd = synth.make_state_dict(cfg, SEED)
    for k, v in ref_after.items():
        sd[so.PREFIX + k] = v.numpy().copy()
    net = ref_shim.build_reference_seeker(cfg, sd)
    clip = synth.make_clip(1, T, H, W, seed=SEED)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
    with torch.no_grad():
        om, fl = net(rgb.clone(), qm)
        om2, fl2 = so.seeker_forward(so.to_torch_state_dict(sd), cfg, rgb, qm)
    d = (om - om2).abs().max().item()
    assert d < 1e-5, d
    arrays.update(output_mask=om.numpy(), output_flags=fl.numpy())
    for k in ('seeker.tracker_post_linear.weight', 'seeker.tracker_post_linear.bias', 'seeker.flag_post_linear.weight', 'seeker.flag_post_linear.bias'):
        arrays['head::' + k] = sd[k]
    save('g10_pretrained', cfg, 1, arrays, dict(d_mask=d, d_flags=(fl - fl2).abs().max().item(), t_ref=0.0, toy=dict(D=D, depth=depth, P=P, n_patches=16, seed=SEED)))


def g11():
    for depth in (18, 24):
        D, heads = {18: (896, 14), 24: (1024, 16)}[depth]
        cfg = synth.seeker_config(num_total_frames=4, frame_height=64, frame_width=96, embed_dim=D, depth=depth, num_heads=heads, causal_attention=1)
        sd = synth.make_state_dict(cfg, SEED)
        net = ref_shim.build_reference_seeker(cfg, sd)               # native construction through Seeker(network_depth=18/24) (vit.py:433-447)
        clip = synth.make_clip(1, 4, 64, 96, seed=SEED)
        rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
        with torch.no_grad():
            om, fl = net(rgb.clone(), qm)
            om2, fl2 = so.seeker_forward(so.to_torch_state_dict(sd), cfg, rgb, qm)
        d = (om - om2).abs().max().item()
        assert d < 2e-5, d
        save(f'g11_depth{depth}', cfg, 1, dict(output_mask=om.numpy(), output_flags=fl.numpy()), dict(d_mask=d, d_flags=(fl - fl2).abs().max().item(), t_ref=0.0))


def g12():
    """K9b: DropPath row semantics (vit_utils.py:139-164; vit.py:172-174,186,216,272-273) from the reference in TRAIN mode.  The
    reference draws torch.rand((rows,1,1)) per DropPath call -- temporal rows = (b,h,w) sites, spatial rows = (b,t) frames, mlp rows =
    samples -- in block order (block 0 has rate 0 = Identity: no draw); re-seeding and replaying the same draws recovers the keep
    masks it used, which are stored next to its outputs and gradients."""
    rate, depth, B, T, H, W = 0.3, 3, 2, 4, 64, 64
    for ca in (1, 0):
        cfg = synth.seeker_config(num_total_frames=T, frame_height=H, frame_width=W, embed_dim=256, depth=depth, num_heads=4, causal_attention=ca)
        sd = synth.make_state_dict(cfg, SEED)
        net = ref_shim.build_reference_seeker(cfg, sd, drop_path_rate=rate)
        net.train()
        clip = synth.make_clip(B, T, H, W, seed=SEED)
        rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
        N = (H // 16) * (W // 16)
        rates = torch.linspace(0, rate, depth).tolist()
        seed = 1234 + ca
        torch.manual_seed(seed)
        om, fl = net(rgb.clone(), qm)
        torch.manual_seed(seed)
        masks = {}
        for i in range(depth):
            if rates[i] > 0.:
                for kind, rows in (('temporal', B * N), ('spatial', B * T), ('mlp', B)):
                    keep = (1.0 - rates[i] + torch.rand((rows, 1, 1), dtype=torch.float32)).floor_()
                    masks[(i, kind)] = (keep.reshape({'temporal': (B, N), 'spatial': (B, T), 'mlp': (B,)}[kind]), rates[i])
        assert any(float(k.min()) == 0.0 for k, _ in masks.values()) and any(float(k.max()) == 1.0 for k, _ in masks.values())
        osd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd.items()}
        om2, fl2 = so.seeker_forward(osd, cfg, rgb, qm, drop_masks=masks)
        d = (om.detach() - om2.detach()).abs().max().item(); dfl = (fl.detach() - fl2.detach()).abs().max().item()
        assert d < 1e-5 and dfl < 1e-5, f'oracle DropPath semantics deviate from the reference: {d} {dfl}'
        Gm = torch.from_numpy(synth._rng(SEED, 'g12_mask').standard_normal(size=tuple(om.shape), dtype=np.float32))
        Gf = torch.from_numpy(synth._rng(SEED, 'g12_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32))
        ((om * Gm).sum() + (fl * Gf).sum()).backward()
        ((om2 * Gm).sum() + (fl2 * Gf).sum()).backward()
        named = dict(net.named_parameters())
        arrays = dict(output_mask=om.detach().numpy(), output_flags=fl.detach().numpy())
        for (i, kind), (keep, r) in masks.items():
            arrays[f'keep::{i}::{kind}'] = keep.numpy(); arrays[f'rate::{i}'] = np.float32(r)
        norms = {}
        for k, p in named.items():
            norms[k] = float(p.grad.norm()) if p.grad is not None else None
            if p.grad is not None:
                e = (osd[k].grad - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-12)
                assert e < 2e-4, f'oracle DropPath gradient deviates from the reference for {k}: {e}'
        for k in ('seeker.tracker_backbone.timesformer.model.blocks.1.temporal_fc.bias', 'seeker.tracker_backbone.timesformer.model.blocks.1.temporal_attn.proj.weight',
                  'seeker.tracker_backbone.timesformer.model.blocks.2.attn.proj.bias', 'seeker.tracker_backbone.timesformer.model.blocks.2.mlp.fc2.weight',
                  'seeker.tracker_backbone.timesformer.model.blocks.0.norm1.weight', 'seeker.tracker_backbone.timesformer.model.cls_token'):
            arrays['grad::' + k] = named[k].grad.numpy().copy()
        save(f'g12_droppath_ca{ca}', cfg, B, arrays, dict(d_mask=d, d_flags=dfl, t_ref=0.0, grad_norms=norms, drop_path_rate=rate, torch_seed=seed))


def augs_inputs(tag, H, W):
    """Deterministic integer test videos of the g13 fixture: segm (1,14,H,W) and div_segm (4,14,H,W) uint8 (regenerated by the tests)."""
    r = synth._rng(SEED, 'g13_' + tag)
    return torch.from_numpy(r.integers(0, 7, size=(1, 14, H, W), dtype=np.uint8)), torch.from_numpy(r.integers(0, 2, size=(4, 14, H, W), dtype=np.uint8))


def g13():
    """(f)4: the reference's augmentation INDEX logic (data/augs.py:50-210).  sample_augs_params is pure numpy (global RNG): run as is for
    many seeds / settings.  apply_augs_2d_frames is run on the integer modalities ('segm', 'div_segm') only; torchvision is not installed,
    so its two index operators used there get functional stand-ins written from torchvision's documented semantics -- CenterCrop (top =
    round((H - h) / 2), left = round((W - w) / 2)) and Resize(NEAREST) (torch.nn.functional.interpolate(mode='nearest')) -- everything
    else (frame selection, crop arithmetic, flip, op order) is the reference's own code."""
    import importlib
    ref_shim.load()
    cwd = os.getcwd(); os.chdir(ref_shim.REF)
    try:
        augs = importlib.import_module('augs')
    finally:
        os.chdir(cwd)

    class CenterCrop:
        def __init__(self, size): self.size = size
        def __call__(self, x):
            h, w = self.size; H, W = x.shape[-2:]
            top = int(round((H - h) / 2.0)); left = int(round((W - w) / 2.0))
            return x[..., top:top + h, left:left + w]

    class Resize:
        def __init__(self, size, interpolation=None, antialias=None): self.size, self.antialias = size, antialias
        def __call__(self, x):
            assert not self.antialias, 'smooth resize is out of scope'
            return torch.nn.functional.interpolate(x.float(), size=self.size, mode='nearest').to(x.dtype)

    augs.torchvision.transforms.CenterCrop = CenterCrop
    augs.torchvision.transforms.Resize = Resize
    arrays = {}
    cases = []
    for seed in range(12):
        for (nl, nc, fs, rnd, a2d, rp, pp) in [(36, 30, 1, True, True, 0.2, 0.1), (42, 30, 2, True, False, 0.5, 0.6), (30, 30, 1, False, False, 0.0, 0.0), (24, 8, 3, True, True, 0.0, 1.0)]:
            cases.append((seed, nl, nc, fs, rnd, a2d, rp, pp))
    keys = ['palindrome', 'reverse', 'frame_stride_factor', 'offset', 'color_jitter', 'rgb_blur', 'rgb_grayscale', 'horz_flip']
    for i, (seed, nl, nc, fs, rnd, a2d, rp, pp) in enumerate(cases):
        pipe = augs.MyAugmentationPipeline(ref_shim.NullLogger(), nl, nc, 240, 320, fs, rnd, a2d, rp, pp, False)
        np.random.seed(1000 + seed)
        p = pipe.sample_augs_params()
        arrays[f'params{i}::scalars'] = np.asarray([float(p[k]) for k in keys], dtype=np.float64)
        arrays[f'params{i}::frame_inds_load'] = np.asarray(p['frame_inds_load']); arrays[f'params{i}::frame_inds_clip'] = np.asarray(p['frame_inds_clip'])
        arrays[f'params{i}::crop_rect'] = np.asarray(p['crop_rect'], dtype=np.float64)
    arrays['param_cases'] = np.asarray(cases, dtype=np.float64)
    # index chain on integer modalities: several source sizes (incl. aspect ratios that trigger either centre-crop branch)
    k = 0
    for (H, W, oh, ow, cc, rnd, a2d) in [(120, 160, 60, 80, False, True, True), (100, 180, 48, 64, True, False, False), (150, 120, 60, 80, True, True, True),
                                          (96, 128, 96, 128, False, True, True), (77, 131, 40, 56, True, True, True), (64, 64, 120, 160, True, True, True)]:
        for seed in range(3):
            pipe = augs.MyAugmentationPipeline(ref_shim.NullLogger(), 14, 10, oh, ow, 1, rnd, a2d, 0.3, 0.4, cc)
            np.random.seed(2000 + 10 * k + seed)
            p = pipe.sample_augs_params()
            tag = f'aug{k}_{seed}'
            segm, div = augs_inputs(tag, H, W)
            out = pipe.apply_augs_2d_frames({'segm': segm, 'div_segm': div}, p)
            arrays[tag + '::cfg'] = np.asarray([H, W, oh, ow, int(cc), int(rnd), int(a2d), 2000 + 10 * k + seed], dtype=np.int64)
            arrays[tag + '::segm_out'] = out['segm'].numpy(); arrays[tag + '::div_out'] = out['div_segm'].numpy()
        k += 1
    np.savez_compressed(os.path.join(OUT, 'g13_augs.npz'), **arrays)
    print('  wrote g13_augs.npz', os.path.getsize(os.path.join(OUT, 'g13_augs.npz')) // 1024, 'KiB')


def g14():
    """A0: attention_type='joint_space_time' (vit.py:159-163; CLI-selectable at args.py:154-156): forward, gradients and train-mode
    DropPath (one draw per sample for the attention and one for the MLP) from the reference."""
    rate, depth, B, T, H, W = 0.3, 3, 2, 3, 48, 64
    cfg = synth.seeker_config(num_total_frames=T, frame_height=H, frame_width=W, embed_dim=256, depth=depth, num_heads=4, causal_attention=0,
                              attention_type='joint_space_time')
    sd = synth.make_state_dict(cfg, SEED)
    clip = synth.make_clip(B, T, H, W, seed=SEED)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
    net = ref_shim.build_reference_seeker(cfg, sd, drop_path_rate=rate)
    assert len(net.state_dict()) == len(sd)
    Gm = Gf = None
    arrays = {}
    for mode in ('eval', 'train'):
        net.train(mode == 'train'); net.zero_grad()
        seed = 4321
        torch.manual_seed(seed)
        om, fl = net(rgb.clone(), qm)
        masks = None
        if mode == 'train':
            torch.manual_seed(seed)
            rates = torch.linspace(0, rate, depth).tolist()
            masks = {}
            for i in range(depth):
                if rates[i] > 0.:
                    for kind in ('spatial', 'mlp'):
                        masks[(i, kind)] = ((1.0 - rates[i] + torch.rand((B, 1, 1), dtype=torch.float32)).floor_().reshape(B), rates[i])
        osd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd.items()}
        om2, fl2 = so.seeker_forward(osd, cfg, rgb, qm, drop_masks=masks)
        d = (om.detach() - om2.detach()).abs().max().item(); dfl = (fl.detach() - fl2.detach()).abs().max().item()
        assert d < 1e-5 and dfl < 1e-5, (mode, d, dfl)
        if Gm is None:
            Gm = torch.from_numpy(synth._rng(SEED, 'g14_mask').standard_normal(size=tuple(om.shape), dtype=np.float32))
            Gf = torch.from_numpy(synth._rng(SEED, 'g14_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32))
        ((om * Gm).sum() + (fl * Gf).sum()).backward(); ((om2 * Gm).sum() + (fl2 * Gf).sum()).backward()
        named = dict(net.named_parameters())
        for k, p in named.items():
            if p.grad is not None:
                e = (osd[k].grad - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-12)
                assert e < 2e-4, (mode, k, e)
        arrays[f'{mode}::output_mask'] = om.detach().numpy(); arrays[f'{mode}::output_flags'] = fl.detach().numpy()
        arrays[f'{mode}::grad_norms'] = np.asarray([float(p.grad.norm()) if p.grad is not None else -1.0 for p in named.values()])
        for k in ('seeker.tracker_backbone.timesformer.model.blocks.1.attn.qkv.weight', 'seeker.tracker_backbone.timesformer.model.blocks.2.attn.proj.bias',
                  'seeker.tracker_backbone.timesformer.model.cls_token', 'seeker.tracker_backbone.timesformer.model.blocks.0.norm1.weight',
                  'seeker.tracker_backbone.timesformer.model.patch_embed.proj.weight'):
            gk = named[k].grad.numpy()
            arrays[f'{mode}::' + ('gsample::' if gk.size > 4096 else 'grad::') + k] = so.grad_sample(gk) if gk.size > 4096 else gk.copy()
        if masks:
            for (i, kind), (keep, r) in masks.items():
                arrays[f'keep::{i}::{kind}'] = keep.numpy(); arrays[f'rate::{i}'] = np.float32(r)
    arrays['param_names'] = np.asarray(list(dict(net.named_parameters()).keys()))
    save('g14_joint', cfg, B, arrays, dict(d_mask=d, d_flags=dfl, t_ref=0.0, drop_path_rate=rate))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--only', default='g10,g11,g9,g7,g8'); args = ap.parse_args()
    assert ref_shim.available(), 'reference tree not found'
    torch.manual_seed(0)
    for name in args.only.split(','):
        print('==', name, flush=True)
        globals()[name]()


if __name__ == '__main__':
    main()
