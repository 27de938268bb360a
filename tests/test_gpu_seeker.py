"""GPU: the drop-in Seeker on libtcow_hip against the golden vectors of the real reference, and against the properties
the domain offers at full size.  Stated tolerances (north_star: mask-logit max|d| < 1e-3 vs the fp32 reference):
  fp32 mode : max|d| < 1e-4 asserted (measured ~1e-6) -- the parity mode.
  bf16 mode : max|d| < 0.08 * logit_std + 1e-3 asserted (bf16 operands cannot reach 1e-3: the reference itself under
              bf16 autocast deviates by 8e-3 at logit std 0.137)."""
import numpy as np
import pytest
import torch

from conftest import build_hip_seeker, golden_inputs, load_golden
from test_oracle_golden import summarise
from tcow_amd import synth

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4


def bf16_tol(ref):
    return 0.08 * float(np.std(ref)) + 1e-3


def _run(name, precision, grad=False):
    meta, g = load_golden(name)
    cfg, sd, rgb, qm = golden_inputs(meta)
    net = build_hip_seeker(cfg, sd, precision).cuda()
    net.train(grad)
    if grad:
        om, fl = net(rgb.cuda(), qm.cuda())
    else:
        with torch.no_grad():
            om, fl = net(rgb.cuda(), qm.cuda())
    return meta, g, net, om, fl


@pytest.mark.parametrize('name', ['g1_cfg1_d256', 'g2_ca0', 'g2_ca2', 'g2_ca3', 'g2_cam1', 'g2_normemb_nearest', 'g2_stride1_prenorm', 'g2_stride2'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_forward_vs_reference_golden(cuda, name, precision):
    meta, g, net, om, fl = _run(name, precision)
    assert om.dtype == torch.float32 and tuple(om.shape) == g['output_mask'].shape and tuple(fl.shape) == g['output_flags'].shape
    d = np.abs(om.cpu().numpy() - g['output_mask']).max()
    df = np.abs(fl.cpu().numpy() - g['output_flags']).max()
    if precision == 'fp32':
        assert d < FP32_TOL and df < FP32_TOL
    else:
        assert d < bf16_tol(g['output_mask']) and df < bf16_tol(g['output_flags']) + 5e-3


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16', 4e-2)])
def test_gradients_vs_reference_golden(cuda, precision, tol):
    meta, g, net, om, fl = _run('g1_cfg1_d256', precision, grad=True)
    Gm = torch.from_numpy(synth._rng(meta['seed'], 'gradprobe_mask').standard_normal(size=tuple(om.shape), dtype=np.float32)).cuda()
    Gf = torch.from_numpy(synth._rng(meta['seed'], 'gradprobe_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32)).cuda()
    ((om * Gm).sum() + (fl * Gf).sum()).backward()
    named = dict(net.named_parameters())
    for k, ref in g.items():
        if k.startswith('grad::'):
            got = named[k[6:]].grad.cpu().numpy()
            assert np.abs(got - ref).max() <= tol * np.abs(ref).max() + 1e-7, k
    for k, n in meta['grad_norms'].items():
        if n is None:
            assert named[k].grad is None, k                                 # model.norm.* unused (vision_tf.py:152) -> no grad, like the reference
        else:
            assert abs(float(named[k].grad.norm()) - n) <= tol * n + 1e-7, k


@pytest.mark.parametrize('name', ['g3_mid_T8_96x128', 'g4_cfg2_T30_240x320'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_large_geometries_vs_reference_golden(cuda, name, precision):
    """Native 12-layer ViT-B Seeker at T=8 96x128 and at the full BASELINE configs[1] size (T=30, 240x320)."""
    meta, g, net, om, fl = _run(name, precision)
    pooled, fsum, fmax = summarise(om.cpu())
    d = np.abs(pooled - g['pooled']).max(); df = np.abs(fl.cpu().numpy() - g['output_flags']).max()
    if precision == 'fp32':
        assert d < FP32_TOL and df < FP32_TOL and np.abs(fmax - g['frame_absmax']).max() < FP32_TOL
        assert np.abs(fsum - g['frame_sum']).max() < 0.5
    else:
        tol = 0.08 * float(g['logit_std']) + 1e-3
        assert d < tol and df < tol + 5e-3


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('ca,leak', [(1, 0), (2, 0), (3, 2)])
def test_causality_is_bit_exact_on_gpu(cuda, precision, ca, leak):
    """Perturbing frame t0 leaves every earlier output frame bit-identical (masked keys contribute exactly zero)."""
    cfg = synth.seeker_config(num_total_frames=6, frame_height=32, frame_width=48, embed_dim=128, depth=2, num_heads=2, causal_attention=ca)
    sd = synth.make_state_dict(cfg, 7)
    net = build_hip_seeker(cfg, sd, precision).cuda().eval()
    clip = synth.make_clip(1, 6, 32, 48, seed=11)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    t0 = 4
    rgb2 = rgb.clone(); rgb2[:, :, t0] += 0.25
    with torch.no_grad():
        a, _ = net(rgb, qm); b, _ = net(rgb2, qm)
    diff = (a - b).abs().amax(dim=(0, 1, 3, 4))
    assert float(diff[: t0 - leak].max()) == 0.0 and bool((diff[t0:] > 0).all())


def test_full_size_properties(cuda):
    """BASELINE configs[1] geometry, bf16: determinism, batch independence (the Qs queries of pipeline.py:134 batched as
    B=3 equal three B=1 calls), inputs untouched, eval == train when DropPath is off."""
    cfg = synth.seeker_config(causal_attention=1)
    sd = synth.make_state_dict(cfg, 900)
    net = build_hip_seeker(cfg, sd, 'bf16').cuda().eval()
    clip = synth.make_clip(1, 30, 240, 320, seed=900)
    rgb = torch.from_numpy(clip['rgb']).cuda().expand(3, -1, -1, -1, -1).contiguous()
    qm = torch.cat([torch.from_numpy(synth.make_query_mask(clip, q, 0)) for q in range(3)], 0).cuda()
    rgb_copy, qm_copy = rgb.clone(), qm.clone()
    with torch.no_grad():
        a, fa = net(rgb, qm); b, fb = net(rgb, qm)
        singles = [net(rgb[q:q + 1], qm[q:q + 1])[0] for q in range(3)]
    assert torch.equal(a, b) and torch.equal(fa, fb)
    assert torch.equal(rgb, rgb_copy) and torch.equal(qm, qm_copy)         # mask_tracker.py:107 clones; inputs never mutated
    for q in range(3):
        assert torch.equal(a[q:q + 1], singles[q])
    assert float((a[0] - a[1]).abs().max()) > 0                              # different queries -> different masks
    net.train()
    c, _ = net(rgb[:1], qm[:1])
    assert torch.equal(c.detach(), a[:1])


def test_train_step_reduces_loss_and_droppath(cuda):
    from tcow_amd.loss import mask_loss
    cfg = synth.seeker_config(num_total_frames=4, frame_height=64, frame_width=64, embed_dim=256, depth=4, num_heads=4, causal_attention=1)
    sd = synth.make_state_dict(cfg, 900)
    net = build_hip_seeker(cfg, sd, 'bf16', drop_path_rate=0.3).cuda().train()
    clip = synth.make_clip(2, 4, 64, 64, seed=3)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    target = torch.zeros(2, 3, 4, 64, 64, device=cuda); target[:, 0] = torch.from_numpy(clip['div_segm'][:, 0]).float().cuda()
    o1, _ = net(rgb, qm); o2, _ = net(rgb, qm)
    assert not torch.equal(o1, o2)                                           # stochastic depth active in train mode (vit_utils.py:139-154)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        out, _ = net(rgb, qm)
        loss = mask_loss(out, target)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 0.3)
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and min(losses[-3:]) < losses[0]
    assert net.seeker.flag_post_linear.weight.grad is None                  # output_flags unused -> no grad (pipeline.py:157)


def test_errors_surface_as_exceptions(cuda):
    from tcow_amd._lib import TcowError
    cfg = synth.seeker_config(num_total_frames=4, frame_height=32, frame_width=32, embed_dim=128, depth=2, num_heads=2)
    net = build_hip_seeker(cfg, synth.make_state_dict(cfg, 1), 'bf16').cuda()
    with pytest.raises(AssertionError):
        net(torch.zeros(1, 3, 5, 32, 32, device=cuda), torch.zeros(1, 1, 5, 32, 32, device=cuda))      # vision_tf.py:96
    with pytest.raises(TcowError):
        net(torch.zeros(1, 3, 4, 32, 32), torch.zeros(1, 1, 4, 32, 32))                                  # CPU tensors: no fallback
    out, fl = net(torch.zeros(1, 3, 4, 32, 32, device=cuda, dtype=torch.float16), torch.zeros(1, 1, 4, 32, 32, device=cuda, dtype=torch.uint8))
    assert out.dtype == torch.float32 and tuple(out.shape) == (1, 3, 4, 32, 32) and tuple(fl.shape) == (1, 4, 3)   # any input dtype is cast (mask_tracker.py:103-104)


def test_fused_adamw_clip_matches_torch(cuda):
    """tcow_adamw_clip_step == torch.nn.utils.clip_grad_norm_(0.3) + torch.optim.AdamW (train.py:99-102), incl. params without grad."""
    from tcow_amd.optim import FusedAdamWClip
    g = torch.Generator(device='cuda').manual_seed(0)
    shapes = [(768, 768), (3, 5), (70001,), (1, 1, 768), (2304,)]
    pa = [torch.nn.Parameter(torch.randn(s, device=cuda, generator=g)) for s in shapes] + [torch.nn.Parameter(torch.zeros(7, device=cuda))]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ref = torch.optim.AdamW(pb, lr=1e-3)
    opt = FusedAdamWClip(pa, lr=1e-3, max_norm=0.3)
    for it in range(3):
        for a, b in zip(pa[:-1], pb[:-1]):                      # the last parameter never gets a gradient
            gr = torch.randn(a.shape, device=cuda, generator=g) * (10.0 if it == 0 else 0.001)   # clipped on step 0, not afterwards
            a.grad = gr.clone(); b.grad = gr.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(pb, 0.3)
        ref.step(); opt.step()
        assert abs(float(opt.grad_norm()) - float(n_ref)) <= 1e-4 * float(n_ref)
        for a, b in zip(pa, pb):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))


def test_persistent_gradient_buckets(cuda):
    """persistent_grads=True: same gradient values, delivered in storage that is stable across steps (no autograd copy)."""
    cfg = synth.seeker_config(num_total_frames=4, frame_height=32, frame_width=32, embed_dim=128, depth=2, num_heads=2, causal_attention=1)
    sd = synth.make_state_dict(cfg, 5)
    clip = synth.make_clip(1, 4, 32, 32, seed=2)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    grads = {}
    for persistent in (False, True):
        net = build_hip_seeker(cfg, sd, 'bf16').cuda().train()
        net.seeker.persistent_grads = persistent
        ptrs = []
        for _ in range(2):
            for p in net.parameters():
                p.grad = None
            om, fl = net(rgb, qm)
            (om.square().mean() + fl.square().mean()).backward()
            ptrs.append([p.grad.data_ptr() for p in net.parameters() if p.grad is not None])
        grads[persistent] = [p.grad.clone() for p in net.parameters() if p.grad is not None]
        if persistent:
            assert ptrs[0] == ptrs[1]
    assert all(torch.equal(a, b) for a, b in zip(grads[False], grads[True]))


def test_training_is_bitwise_reproducible(cuda):
    """Two runs of the full training step (pipeline, HIP forward / backward, TCOW objective, fused clip + AdamW, seeded DropPath)
    give bit-identical loss curves: no kernel on the path accumulates floating-point data with atomics, and a race in the
    hand-synchronised kernels (counted waits, barrier-free epilogues) would show up here as a difference."""
    from tcow_amd.optim import FusedAdamWClip
    from tcow_amd.pipeline import SeekerPipeline
    from tcow_amd.seeker import Seeker
    from tcow_amd.tcow_loss import default_args
    T, H, W, depth, steps = 30, 240, 320, 4, 5

    def run():
        torch.manual_seed(0)
        cfg = synth.seeker_config(num_total_frames=T, frame_height=H, frame_width=W, depth=depth, causal_attention=1)
        net = Seeker(None, num_total_frames=T, frame_height=H, frame_width=W, causal_attention=1, drop_path_rate=0.1, network_depth=depth, embed_dim=768, num_heads=12,
                     precision='bf16')
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()})
        net = net.cuda().train()
        opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3)
        opt.on_step.append(net.seeker.invalidate_weight_cache); net.seeker.persistent_grads = True
        data = synth.to_torch_tree(synth.make_kubric_batch(1, T, H, W, seed=900, n_objects=5), 'cuda', host_keys=synth.HOST_KEYS)
        pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device='cuda', rng=np.random.default_rng(0))
        losses = []
        for i in range(steps):
            mr = pipe.forward_kubric(data)
            loss = pipe.step_losses(data, mr, 0.1 + i / 100.0)['total_seeker']          # progress > 0: the radix top-k path is active
            loss.backward(); opt.step()
            losses.append(loss.detach())
        w = net.seeker.tracker_post_linear.weight.detach().clone()
        return torch.stack(losses).cpu().numpy(), w.cpu().numpy()

    (la, wa), (lb, wb) = run(), run()
    assert np.isfinite(la).all()
    assert np.array_equal(la, lb), (la, lb)
    assert np.array_equal(wa, wb)
