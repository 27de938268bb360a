"""GPU: the drop-in Seeker on libtcow_hip against the golden vectors of the real reference, and against the properties
the domain offers at full size.  Stated tolerances (north_star: mask-logit max|d| < 1e-3 vs the fp32 reference):
  fp32 mode : max|d| < 1e-5 asserted (measured 7e-8 ... 3e-6 over all goldens) -- the parity mode, 100x inside the north-star bound.
  bf16x3    : max|d| < 1e-4 asserted (measured 5e-7 ... 1.1e-5): the fp32 mode with split-bf16 GEMM products, 10x inside the bound.
  fp16 mode : the bf16 kernels built for binary16 storage: 1/8 of the bf16 bounds (0.00625 x logit std; measured 0.003 ... 0.0042 x std) and,
              at BASELINE configs[1], the north-star bound itself: max|d| < 1e-3 asserted (measured ~6e-4).
  bf16 mode : max|d| < 0.05 x the golden's logit std asserted = 1.5 x the worst measured ratio (0.013 ... 0.034 x std over 13 goldens,
              tools/dev_bf16_ratios.py; 3.4e-3 absolute at BASELINE configs[1], logit std 0.154); flags < 0.012 x their std (measured
              <= 0.0077).  bf16 operands cannot reach 1e-3 at these logit scales: profiles/r02_bf16_error_budget.txt (weight copies alone
              2.7e-3; every activation class 0.2 ... 2.0e-3); the reference itself under bf16 autocast deviates by 8e-3."""
import math

import numpy as np
import pytest
import torch

from conftest import build_hip_seeker, golden_inputs, load_golden
from test_oracle_golden import summarise
from tcow_amd import synth

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-5
X3_TOL = 1e-4        # precision='bf16x3' (f32 storage, GEMM products as three bf16 MFMAs): measured 5e-7 ... 1.1e-5 over the goldens
EXACT = {'fp32': FP32_TOL, 'bf16x3': X3_TOL}


def bf16_tol(ref):
    """1.3 x the worst deviation measured over the forward goldens (tools/dev_bf16_ratios.py, round 3): bf16 0.0381 x std (g11_depth24; configs[1]
    goldens 0.020 / 0.030), binary16 0.0045 x std against the 0.00625 this gives through h16()."""
    return 0.05 * float(np.std(ref))


def bf16_flags_tol(ref):
    """Worst measured 0.0063 x std (g4b); T x 3 samples per clip move by 2x between equivalent roundings (see h16f), hence 1.9 x."""
    return 0.012 * float(np.std(ref))


def h16(precision):
    """Rounding-error scale of a 16-bit mode relative to bf16: binary16 carries three more significand bits."""
    return 0.125 if precision == 'fp16' else 1.0


def h16f(precision):
    """The same for the occlusion flags: a clip has only T x 3 of them, and the maximum over so few samples moves by 2x between equivalent
    roundings of the same network (g11_depth24 in fp16: 0.99e-3 with the temporal projection as two GEMMs, 2.2e-3 folded into one; in bf16
    the other way round, 9.9e-3 vs 4.1e-3 -- measured with and without the fold in round 3, tools/dev_bf16_ratios.py), so the binary16 bound keeps a factor 1.6 of slack."""
    return 0.2 if precision == 'fp16' else 1.0


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
@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'bf16x3', 'fp16'])
def test_forward_vs_reference_golden(cuda, name, precision):
    meta, g, net, om, fl = _run(name, precision)
    assert om.dtype == torch.float32 and tuple(om.shape) == g['output_mask'].shape and tuple(fl.shape) == g['output_flags'].shape
    d = np.abs(om.cpu().numpy() - g['output_mask']).max()
    df = np.abs(fl.cpu().numpy() - g['output_flags']).max()
    if precision in EXACT:
        assert d < EXACT[precision] and df < EXACT[precision]
    else:
        assert d < h16(precision) * bf16_tol(g['output_mask']) and df < h16f(precision) * bf16_flags_tol(g['output_flags'])


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16', 4e-2), ('bf16x3', 2e-4), ('fp16', 5e-3)])
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


@pytest.mark.parametrize('name', ['g17_resize_a', 'g17_resize_b'])
@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16', 4e-2), ('bf16x3', 2e-4), ('fp16', 5e-3)])
def test_forward_time_embedding_resize_vs_reference_golden(cuda, name, precision, tol):
    """vision_tf.py:103-115,127-132 (engine._effective_embeddings): a stored pos_embed of another square grid (3x3 -> 4x4 patches; 5x5 -> 2x4, H != W, with
    H = x.size(1) // W counting the cls row) and a time_embed of another length (6 -> 4, 3 -> 7) are nearest-resized in the forward.  G17: the REFERENCE ran
    with the tables grafted after construction (oracle/make_golden_r6.py); outputs, the gradients of the STORED tables (the adjoint scatters through the
    index maps), cls_token / patch-embed bias, and every parameter's gradient norm."""
    from test_oracle_golden import resize_tables
    meta, g = load_golden(name)
    cfg, sd, rgb, qm = golden_inputs(meta)
    net = build_hip_seeker(cfg, sd, precision)
    pos, te = resize_tables(meta)
    net.seeker.vit.pos_embed = torch.nn.Parameter(torch.from_numpy(pos.copy()))           # grafted after construction, as the reference run did
    net.seeker.vit.time_embed = torch.nn.Parameter(torch.from_numpy(te.copy()))
    net = net.cuda().train()                                                             # (drop_path_rate 0: train == eval arithmetic, gradients enabled)
    om, fl = net(rgb.cuda(), qm.cuda())
    d = np.abs(om.detach().cpu().numpy() - g['output_mask']).max(); df = np.abs(fl.detach().cpu().numpy() - g['output_flags']).max()
    if precision in EXACT:
        assert d < EXACT[precision] and df < EXACT[precision]
    else:
        assert d < h16(precision) * bf16_tol(g['output_mask']) and df < h16f(precision) * bf16_flags_tol(g['output_flags'])
    Gm = torch.from_numpy(synth._rng(meta['seed'], 'gradprobe_mask').standard_normal(size=tuple(om.shape), dtype=np.float32)).cuda()
    Gf = torch.from_numpy(synth._rng(meta['seed'], 'gradprobe_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32)).cuda()
    ((om * Gm).sum() + (fl * Gf).sum()).backward()
    named = dict(net.named_parameters())
    for k, ref in g.items():
        if k.startswith('grad::'):
            got = named[k[6:]].grad.cpu().numpy()
            assert got.shape == ref.shape and np.abs(got - ref).max() <= tol * np.abs(ref).max() + 1e-7, k
    for k, n in meta['grad_norms'].items():
        if n is None:
            assert named[k].grad is None, k
        else:
            assert abs(float(named[k].grad.norm()) - n) <= tol * n + 1e-7, k


@pytest.mark.parametrize('name', ['g3_mid_T8_96x128', 'g4_cfg2_T30_240x320', 'g4b_cfg2_seed2'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'bf16x3', 'fp16'])
def test_large_geometries_vs_reference_golden(cuda, name, precision):
    """Native 12-layer ViT-B Seeker at T=8 96x128 and at the full BASELINE configs[1] size (T=30, 240x320) -- the latter with two
    independent weight / clip seeds (g4, g4b): the binary16 mode must stay inside the north-star bound of 1e-3 on both."""
    meta, g, net, om, fl = _run(name, precision)
    pooled, fsum, fmax = summarise(om.cpu())
    d = np.abs(pooled - g['pooled']).max(); df = np.abs(fl.cpu().numpy() - g['output_flags']).max()
    if precision in EXACT:
        assert d < EXACT[precision] and df < EXACT[precision] and np.abs(fmax - g['frame_absmax']).max() < EXACT[precision]
        assert np.abs(fsum - g['frame_sum']).max() < 0.5
    else:
        assert d < h16(precision) * 0.05 * float(g['logit_std']) and df < h16f(precision) * bf16_flags_tol(g['output_flags'])
        if precision == 'fp16' and name.startswith('g4'):
            assert d < 1e-3                                # north_star: mask-logit max|d| < 1e-3 at BASELINE configs[1]


# max|d| / std(logits) allowed at trained-checkpoint magnitudes: today's ratios at the init-scale goldens (fp32 1e-5 / 0.154, bf16x3 1e-4 / 0.154,
# the 16-bit modes 0.05 and 0.00625 x std) -- 16-bit error is relative, so the same ratios must hold when the logits are 30x larger
TRAINED_REL = {'fp32': 6.5e-5, 'bf16x3': 6.5e-4, 'fp16': 0.00625, 'bf16': 0.05}
TRAINED_AGREE = {'fp32': 0.99999, 'bf16x3': 0.9999, 'fp16': 0.9995, 'bf16': 0.996}


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'fp16', 'bf16'])
def test_parity_at_trained_checkpoint_logit_scale(cuda, precision):
    """g16 (the REFERENCE's output at BASELINE configs[1] geometry with weights of trained magnitude: q / k rows x3 -> peaked attention rows,
    block weights x1.5, mask head scaled to logit std 5 -- every other golden is at trunc-normal(0.02), logit std 0.154, where `< 1e-3`
    absolute is a statement about small logits).  Asserted: max|d| / std stays at the per-precision ratio of the init-scale goldens, and
    the binary masks (logit > 0, what IoU is computed from) of three frames agree with the reference's."""
    from test_oracle_golden import mask_bits, trained_scale_inputs
    meta, g = load_golden('g16_cfg2_trained_scale')
    cfg, sd, rgb, qm = trained_scale_inputs(meta)
    net = build_hip_seeker(cfg, sd, precision).cuda().eval()
    with torch.no_grad():
        om, fl = net(rgb.cuda(), qm.cuda())
    om = om.cpu()
    std = float(g['logit_std'])
    pooled, fsum, fmax = summarise(om)
    d = float(np.abs(pooled - g['pooled']).max()); dmax = float(np.abs(fmax - g['frame_absmax']).max())
    agree = 1.0 - float(np.unpackbits(mask_bits(om.numpy(), meta['mask_frames']) ^ g['mask_bits']).mean())
    print(f'trained scale {precision}: logit std {std:.3f}  max|d| (4x4 pooled) {d:.3e} = {d / std:.2e} x std  frame abs-max d {dmax:.3e}  binary-mask agreement {agree:.6f}')
    assert np.isfinite(om.numpy()).all()
    assert d < TRAINED_REL[precision] * std, (d, std)
    assert agree >= TRAINED_AGREE[precision], agree


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'bf16x3', 'fp16'])
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


def _droppath_masks(g):
    masks = {}
    for k in g:
        if k.startswith('keep::'):
            _, i, kind = k.split('::')
            masks[(int(i), kind)] = (torch.from_numpy(g[k]), float(g[f'rate::{i}']))
    return masks


@pytest.mark.parametrize('name', ['g12_droppath_ca1', 'g12_droppath_ca0'])
@pytest.mark.parametrize('precision,tol,gtol', [('fp32', FP32_TOL, 2e-4), ('bf16', None, 4e-2), ('bf16x3', X3_TOL, 2e-4), ('fp16', None, 5e-3)])
def test_droppath_forced_masks_vs_reference_train_mode(cuda, name, precision, tol, gtol):
    """K9b (vit_utils.py:139-164; vit.py:172-174,186,216,272-273): the keep masks the REFERENCE drew in train mode (golden g12) are
    forced into the HIP engine; outputs and gradients must equal the reference's -- temporal DropPath per site before temporal_fc (a
    dropped site still receives + b_fc), spatial per frame (incl. the cls row), MLP per sample."""
    meta, g = load_golden(name)
    cfg, sd, rgb, qm = golden_inputs(meta)
    net = build_hip_seeker(cfg, sd, precision, drop_path_rate=meta['drop_path_rate']).cuda().train()
    net.seeker.forced_drop_masks = _droppath_masks(g)
    om, fl = net(rgb.cuda(), qm.cuda())
    d = np.abs(om.detach().cpu().numpy() - g['output_mask']).max(); df = np.abs(fl.detach().cpu().numpy() - g['output_flags']).max()
    if precision in EXACT:
        assert d < tol and df < tol
    else:
        assert d < h16(precision) * bf16_tol(g['output_mask']) and df < h16f(precision) * bf16_flags_tol(g['output_flags'])
    Gm = torch.from_numpy(synth._rng(meta['seed'], 'g12_mask').standard_normal(size=tuple(om.shape), dtype=np.float32)).cuda()
    Gf = torch.from_numpy(synth._rng(meta['seed'], 'g12_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32)).cuda()
    ((om * Gm).sum() + (fl * Gf).sum()).backward()
    named = dict(net.named_parameters())
    for k, ref in g.items():
        if k.startswith('grad::'):
            assert np.abs(named[k[6:]].grad.cpu().numpy() - ref).max() <= gtol * np.abs(ref).max() + 1e-7, k
    for k, n in meta['grad_norms'].items():
        if n is not None:
            assert abs(float(named[k].grad.norm()) - n) <= gtol * n + 1e-7, k
    # and the masks matter: without them the same module gives a different answer
    net.seeker.forced_drop_masks = None; net.eval()
    with torch.no_grad():
        plain, _ = net(rgb.cuda(), qm.cuda())
    assert float((plain - om.detach()).abs().max()) > 1e-3


@pytest.mark.parametrize('precision,tol', [('fp32', 3e-4), ('bf16', 5e-2), ('bf16x3', 6e-4), ('fp16', 8e-3)])
def test_full_size_gradients_vs_reference_golden(cuda, precision, tol):
    """BASELINE configs[1] at full size with the Qs = 3 queries batched (M = 27 090 rows: the 320-tile GEMMs, streaming attention
    and 256-tile weight-gradient kernels the benchmark runs) against gradients of the REAL reference (golden g7: three sequential
    query forwards + one backward, pipeline.py:134-158 / train.py:98): every parameter's gradient norm, ten small gradients in full
    and strided samples of eleven large ones."""
    from oracle.seeker_oracle import grad_sample
    meta, g = load_golden('g7_cfg2_grads')
    cfg = meta['cfg']
    sd = synth.make_state_dict(cfg, meta['seed'])
    net = build_hip_seeker(cfg, sd, precision).cuda().train()
    clip = synth.make_clip(1, 30, 240, 320, seed=meta['seed'])
    Qs = meta['queries']
    rgb = torch.from_numpy(clip['rgb']).cuda().expand(Qs, -1, -1, -1, -1).contiguous()
    qm = torch.cat([torch.from_numpy(synth.make_query_mask(clip, q, 0)) for q in range(Qs)], 0).cuda()
    om, fl = net(rgb, qm)
    pooled, _, _ = summarise(om.detach().cpu())
    d = np.abs(pooled - g['pooled']).max(); df = np.abs(fl.detach().cpu().numpy() - g['output_flags']).max()
    if precision in EXACT:
        assert d < EXACT[precision] and df < EXACT[precision]
    else:
        assert d < h16(precision) * 0.05 * float(g['logit_std'])
        assert precision != 'fp16' or d < 1e-3             # the north-star bound itself at BASELINE configs[1]
    Gm = torch.cat([torch.from_numpy(synth._rng(meta['seed'], f'g7_mask_{q}').standard_normal(size=(1,) + tuple(om.shape[1:]), dtype=np.float32)) for q in range(Qs)]).cuda()
    Gf = torch.cat([torch.from_numpy(synth._rng(meta['seed'], f'g7_flags_{q}').standard_normal(size=(1,) + tuple(fl.shape[1:]), dtype=np.float32)) for q in range(Qs)]).cuda()
    ((om * Gm).sum() * meta['mask_probe_scale'] + (fl * Gf).sum()).backward()
    named = dict(net.named_parameters())
    worst = {}
    for k, ref in g.items():
        if k.startswith('grad::'):
            got = named[k[6:]].grad.cpu().numpy()
        elif k.startswith('gsample::'):
            got = grad_sample(named[k[9:]].grad.cpu().numpy())
        else:
            continue
        worst[k] = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
        assert worst[k] <= tol, (k, worst[k])
    for k, n in meta['grad_norms'].items():
        if n is None:
            assert named[k].grad is None, k
        else:
            assert abs(float(named[k].grad.norm()) - n) <= tol * n + 1e-7, (k, float(named[k].grad.norm()), n)


@pytest.mark.parametrize('precision', ['bf16', 'fp32', 'bf16x3', 'fp16'])
def test_config3_long_clip_vs_reference_golden(cuda, precision):
    """BASELINE configs[3]: T=60, 480x640 (1200 spatial x 60 temporal tokens, S = 1201), inference forward, against the reference's
    output on the same synthetic clip (golden g8: pooled logits of six frames, per-frame sums / abs-max, flags)."""
    meta, g = load_golden('g8_cfg3_long')
    cfg, sd, rgb, qm = golden_inputs(meta)
    net = build_hip_seeker(cfg, sd, precision).cuda().eval()
    with torch.no_grad():
        om, fl = net(rgb.cuda(), qm.cuda())
    assert tuple(om.shape) == (1, 3, 60, 480, 640) and bool(torch.isfinite(om).all())
    pooled, fsum, fmax = summarise(om.cpu())
    pooled = pooled.reshape(60, 3, 120, 160)[g['frames']]
    d = np.abs(pooled - g['pooled_frames']).max(); df = np.abs(fl.cpu().numpy() - g['output_flags']).max()
    if precision in EXACT:
        assert d < EXACT[precision] and df < EXACT[precision] and np.abs(fmax - g['frame_absmax']).max() < EXACT[precision]
    else:
        assert d < h16(precision) * 0.05 * float(g['logit_std']) and df < h16f(precision) * bf16_flags_tol(g['output_flags'])
    assert abs(float((om > 0).float().mean()) - float(g['positive_frac'])) < ({'fp32': 1e-5, 'bf16x3': 1e-4, 'fp16': 1e-3}.get(precision, 5e-3))


@pytest.mark.parametrize('precision', ['bf16', 'fp32', 'fp16'])
def test_config4_batched_eval_vs_reference_golden(cuda, precision):
    """BASELINE configs[4]: num_queries = 4 x 4 temporal strides of one plugin-shaped video = 16 clips through ONE batched Seeker call
    (the reference runs 16 sequential B = 1 forwards, eval/test.py + data_plugin.py:141-156); logits, flags and the IoU metrics of
    eval/metrics.py:9-113 against what the reference's MyTrainPipeline.forward_plugin produced on 6 of the 16 items (golden g9)."""
    from tcow_amd import plugin_data as pd
    from tcow_amd.metrics import calculate_metrics_mask_track
    from tcow_amd.pipeline import SeekerPipeline
    meta, g = load_golden('g9_cfg4_eval')
    cfg = meta['cfg']
    net = build_hip_seeker(cfg, synth.make_state_dict(cfg, meta['seed']), precision).cuda().eval()
    video = synth.make_plugin_video(meta['video_frames'], 240, 320, seed=meta['video_seed'])
    items = pd.eval_items(video, num_frames=30, query_time_idx=0, queries=(0, 1, 2, 3), strides=(1, 2, 3, 4))
    assert [[it['query'], it['frame_stride']] for it in items] == g['item_query_stride'].tolist()
    pipe = SeekerPipeline(net, num_queries=1, phase='test', device='cuda')
    with torch.no_grad():
        mr = pipe.forward_plugin_items(items)
    om, fl = mr['output_mask'], mr['output_flags']
    assert tuple(om.shape) == (16, 3, 30, 240, 320)
    for i in g['picked'].tolist():
        pooled, fsum, _ = summarise(om[i:i + 1].cpu())
        logit_tol = FP32_TOL if precision == 'fp32' else h16(precision) * bf16_tol(g[f'item{i}::pooled'])
        flag_tol = FP32_TOL if precision == 'fp32' else h16(precision) * bf16_flags_tol(g[f'item{i}::output_flags'])
        assert np.abs(pooled[::4] - g[f'item{i}::pooled']).max() < logit_tol, i
        assert np.abs(fl[i:i + 1].cpu().numpy() - g[f'item{i}::output_flags']).max() < flag_tol, i
        m = calculate_metrics_mask_track(om[i:i + 1], mr['target_mask'][i:i + 1], plugin=True)
        for k in m:
            ref = g[f'item{i}::metric::{k}']
            if k.startswith('count_'):
                assert int(m[k]) == int(ref), (i, k)                        # which frames carry annotations: exact
            else:
                assert abs(float(m[k]) - float(ref)) < (1e-6 if precision == 'fp32' else 3e-3 * h16(precision)), (i, k, float(m[k]), float(ref))
    # batched == sequential, bit for bit (batch rows are independent)
    with torch.no_grad():
        one = pipe.forward_plugin_items(items[5:6])['output_mask']
    assert torch.equal(one[0], om[5])


@pytest.mark.parametrize('name', ['g11_depth18', 'g11_depth24'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'bf16x3', 'fp16'])
def test_depth_18_and_24_vs_reference_golden(cuda, name, precision):
    """V0 (vit.py:433-447): D = 896 / 14 heads / 18 blocks and D = 1024 / 16 heads / 24 blocks through Seeker(network_depth=...)."""
    meta, g = load_golden(name)
    cfg, sd, rgb, qm = golden_inputs(meta)
    from tcow_amd.seeker import Seeker
    net = Seeker(None, num_total_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'], frame_width=cfg['frame_width'], causal_attention=1,
                 network_depth=cfg['depth'], drop_path_rate=0.0, precision=precision)
    assert net.seeker.embed_dim == cfg['embed_dim'] and net.seeker.num_heads == cfg['num_heads']
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        om, fl = net(rgb.cuda(), qm.cuda())
    d = np.abs(om.cpu().numpy() - g['output_mask']).max(); df = np.abs(fl.cpu().numpy() - g['output_flags']).max()
    if precision in EXACT:
        assert d < EXACT[precision] and df < EXACT[precision]
    else:
        assert d < h16(precision) * bf16_tol(g['output_mask']) and df < h16f(precision) * bf16_flags_tol(g['output_flags'])


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'fp16'])
def test_pretrained_checkpoint_forward_vs_reference_golden(cuda, precision, tmp_path):
    """(f)1 end to end on the GPU: image-ViT file -> tracker_pretrained=<path> (weight surgery of helpers.py:100-205) -> a reference-format
    checkpoint.pth (train.py:269-304) -> load_tcow_checkpoint (eval/inference.py:38-54) -> HIP forward with the rgb normalisation of
    vision_tf.py:81-89, against the reference's own forward with its own load_pretrained weights (golden g10)."""
    import argparse
    from oracle.make_golden_r2 import toy_vit_checkpoint
    from tcow_amd.checkpoint import load_tcow_checkpoint
    from tcow_amd.seeker import Seeker
    meta, g = load_golden('g10_pretrained')
    cfg, sd, rgb, qm = golden_inputs(meta)
    vit_path = tmp_path / 'vit.pth'
    torch.save({'state_dict': toy_vit_checkpoint(**meta['toy'])}, vit_path)
    seeker_args = dict(num_total_frames=cfg['num_total_frames'], num_visible_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'], frame_width=cfg['frame_width'],
                       tracker_pretrained=str(vit_path), attention_type='divided_space_time', patch_size=16, causal_attention=1, norm_embeddings=False, drop_path_rate=0.1,
                       network_depth=cfg['depth'], track_map_stride=4, track_map_resize='bilinear', query_channels=1, output_channels=3, flag_channels=3,
                       embed_dim=cfg['embed_dim'], num_heads=cfg['num_heads'])
    trained = Seeker(None, **seeker_args)
    heads = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('head::')}
    trained.load_state_dict({**trained.state_dict(), **heads}, strict=True)
    ck_path = tmp_path / 'checkpoint.pth'
    torch.save({'epoch': 0, 'train_args': argparse.Namespace(name='t'), 'dset_args': {}, 'seeker_args': seeker_args, 'net_seeker': trained.state_dict(),
                'optim_seeker': {}, 'lr_sched_seeker': {}}, ck_path)
    vit_path.unlink()                                                      # the eval-side load must not need the image-ViT file again
    net = load_tcow_checkpoint(str(ck_path), device='cuda', precision=precision).eval()
    assert net.seeker.tracker_pretrained is True
    with torch.no_grad():
        om, fl = net(rgb.cuda(), qm.cuda())
    d = np.abs(om.cpu().numpy() - g['output_mask']).max(); df = np.abs(fl.cpu().numpy() - g['output_flags']).max()
    if precision == 'fp32':
        assert d < FP32_TOL and df < FP32_TOL
    else:
        assert d < h16(precision) * bf16_tol(g['output_mask']) and df < h16f(precision) * bf16_flags_tol(g['output_flags'])


@pytest.mark.parametrize('precision,tol,gtol', [('fp32', 2e-5, 2e-4), ('bf16', 1.5e-2, 4e-2)])
def test_shared_rgb_queries_equal_expanded_batch(cuda, precision, tol, gtol):
    """Qs query masks per clip with the frames passed ONCE ((B,3,..) + (B*Qs,1,..)) == the same frames repeated per query
    (what pipeline.py:134-158 feeds): outputs, and the gradients of the patch-embedding weight whose rgb / mask column blocks are now
    produced by two different GEMMs."""
    cfg = synth.seeker_config(num_total_frames=4, frame_height=64, frame_width=96, embed_dim=256, depth=2, num_heads=4, causal_attention=1)
    sd = synth.make_state_dict(cfg, 11)
    B, Qs = 2, 3
    clip = synth.make_clip(B, 4, 64, 96, seed=5)
    rgb = torch.from_numpy(clip['rgb']).cuda()
    qm = torch.stack([torch.from_numpy(synth.make_query_mask(clip, q, 0)) for q in range(Qs)], 1).reshape(B * Qs, 1, 4, 64, 96).cuda()   # clip-major rows
    res = {}
    for shared in (True, False):
        net = build_hip_seeker(cfg, sd, precision).cuda().train()
        frames = rgb if shared else rgb[:, None].expand(B, Qs, -1, -1, -1, -1).reshape(B * Qs, 3, 4, 64, 96).contiguous()
        om, fl = net(frames, qm)
        assert tuple(om.shape) == (B * Qs, 3, 4, 64, 96) and tuple(fl.shape) == (B * Qs, 4, 3)
        g = torch.Generator(device='cuda').manual_seed(1)
        (om * torch.randn(om.shape, device=cuda, generator=g)).sum().backward()
        pe = net.seeker.vit.patch_embed.proj
        res[shared] = (om.detach(), fl.detach(), pe.weight.grad.clone(), pe.bias.grad.clone(), net.seeker.vit.blocks[0].attn.qkv.weight.grad.clone())
    a, b = res[True], res[False]
    assert float((a[0] - b[0]).abs().max()) < tol and float((a[1] - b[1]).abs().max()) < tol
    for x, y in zip(a[2:], b[2:]):
        assert float((x - y).abs().max()) <= gtol * float(y.abs().max()) + 1e-7
    assert float((a[0][0] - a[0][1]).abs().max()) > 0                      # different queries of one clip -> different masks


@pytest.mark.parametrize('precision,tol,gtol', [('fp32', FP32_TOL, 3e-4), ('bf16', None, 4e-2), ('bf16x3', X3_TOL, 3e-4), ('fp16', None, 5e-3)])
def test_joint_space_time_vs_reference_golden(cuda, precision, tol, gtol):
    """A0 (vit.py:159-163, args.py:154-156): attention_type='joint_space_time' -- one attention over (cls, all N*T patch tokens) per clip
    through the streaming MFMA kernels -- forward, gradients, and train-mode DropPath (one draw per sample) against the reference."""
    from oracle.seeker_oracle import grad_sample
    from tcow_amd.seeker import Seeker
    meta, g = load_golden('g14_joint')
    cfg, sd, rgb, qm = golden_inputs(meta)
    names = [str(n) for n in g['param_names']]
    for mode in ('eval', 'train'):
        net = Seeker(None, num_total_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'], frame_width=cfg['frame_width'], causal_attention=0,
                     attention_type='joint_space_time', drop_path_rate=meta['drop_path_rate'], network_depth=cfg['depth'], embed_dim=cfg['embed_dim'],
                     num_heads=cfg['num_heads'], precision=precision)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        net = net.cuda().train()
        net.seeker.forced_drop_masks = _droppath_masks(g) if mode == 'train' else {}     # {} = train graph with every DropPath forced open
        om, fl = net(rgb.cuda(), qm.cuda())
        d = np.abs(om.detach().cpu().numpy() - g[f'{mode}::output_mask']).max(); df = np.abs(fl.detach().cpu().numpy() - g[f'{mode}::output_flags']).max()
        if precision in EXACT:
            assert d < tol and df < tol, (mode, d, df)
        else:
            assert d < h16(precision) * bf16_tol(g[f'{mode}::output_mask']) and df < h16(precision) * bf16_flags_tol(g[f'{mode}::output_flags'])
        Gm = torch.from_numpy(synth._rng(meta['seed'], 'g14_mask').standard_normal(size=tuple(om.shape), dtype=np.float32)).cuda()
        Gf = torch.from_numpy(synth._rng(meta['seed'], 'g14_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32)).cuda()
        ((om * Gm).sum() + (fl * Gf).sum()).backward()
        named = dict(net.named_parameters())
        for k, ref in g.items():
            if k.startswith(f'{mode}::grad::'):
                got = named[k.split('::', 2)[2]].grad.cpu().numpy()
            elif k.startswith(f'{mode}::gsample::'):
                got = grad_sample(named[k.split('::', 2)[2]].grad.cpu().numpy())
            else:
                continue
            assert np.abs(got - ref).max() <= gtol * np.abs(ref).max() + 1e-7, (mode, k)
        for name, n in zip(names, g[f'{mode}::grad_norms']):
            if n >= 0:
                assert abs(float(named[name].grad.norm()) - n) <= gtol * n + 1e-7, (mode, name)
    from tcow_amd._lib import TcowError
    with pytest.raises(TcowError):
        Seeker(None, attention_type='space_only')                          # not runnable in the reference either (vision_tf.py:127 vs vit.py:263-265)


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
    import torch.nn.functional as F

    def mask_loss(logits, target):                                           # a plain objective for the step test (the TCOW objective: test_pipeline_loss_metrics.py)
        return F.binary_cross_entropy_with_logits(logits, target)
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
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [1, 2], gamma=0.3); sched_ref = torch.optim.lr_scheduler.MultiStepLR(ref, [1, 2], gamma=0.3)   # train.py:236-243 attaches unchanged
    for it in range(3):
        for a, b in zip(pa[:-1], pb[:-1]):                      # the last parameter never gets a gradient
            gr = torch.randn(a.shape, device=cuda, generator=g) * (10.0 if it == 0 else 0.001)   # clipped on step 0, not afterwards
            a.grad = gr.clone(); b.grad = gr.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(pb, 0.3)
        v0 = pa[0]._version
        ref.step(); opt.step(); sched.step(); sched_ref.step()
        assert pa[0]._version > v0                              # raw-pointer update made visible to version-keyed caches
        assert abs(float(opt.grad_norm()) - float(n_ref)) <= 1e-4 * float(n_ref)
        for a, b in zip(pa, pb):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    # state in torch.optim.AdamW's layout: it loads into AdamW and back (train.py:246-257 resumes 'optim_seeker' this way)
    ref2 = torch.optim.AdamW(pb, lr=1e-3); ref2.load_state_dict(opt.state_dict())
    assert all(torch.equal(ref2.state[b]['exp_avg'], opt.state[a]['exp_avg']) for a, b in zip(pa[:-1], pb[:-1]))
    opt.load_state_dict(ref.state_dict())
    assert opt.step_count == 3


@pytest.mark.parametrize('precision', ['bf16', 'fp16', 'fp32'])
def test_inference_forward_is_hipgraph_capturable(cuda, precision):
    """The forward makes no host synchronisation and no allocation outside torch's pool, so it can be captured once and replayed as a
    hipGraph (torch.cuda.CUDAGraph): replay == eager bit for bit, and the replay follows the static input buffers."""
    cfg = synth.seeker_config(num_total_frames=4, frame_height=32, frame_width=48, embed_dim=128, depth=2, num_heads=2, causal_attention=1)
    net = build_hip_seeker(cfg, synth.make_state_dict(cfg, 5), precision).cuda().eval()
    clip = synth.make_clip(1, 4, 32, 48, seed=2)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    with torch.no_grad():
        ref, ref_f = net(rgb, qm)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(rgb, qm)                                      # warm-up on the capture stream (weight copies, LDS attributes)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out, fl = net(rgb, qm)
        graph.replay(); torch.cuda.synchronize()
        assert torch.equal(out, ref) and torch.equal(fl, ref_f)
        rgb.mul_(0.5); graph.replay(); torch.cuda.synchronize()
        again, _ = net(rgb, qm)
        assert torch.equal(out, again) and not torch.equal(again, ref)


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_optimizer_writes_the_operand_copies_itself(cuda, precision):
    """FusedAdamWClip(module=net) in the 16-bit modes: the update kernel of the GEMM weights (64 x 64 tiles, tcow_adamw_clip_step_cast) also writes their 16-bit
    W / W^T operand copies, and the module's batched re-cast only covers the folded products.  Bit-identical training against fuse_cast=False (the separate
    re-cast of rounds 1-5); after a step every cached copy equals a cast of its f32 master weight and its transpose."""
    from tcow_amd.optim import FusedAdamWClip
    cfg = synth.seeker_config(num_total_frames=4, frame_height=64, frame_width=64, embed_dim=128, depth=2, num_heads=2, causal_attention=1)
    sd = synth.make_state_dict(cfg, 7)
    clip = synth.make_clip(2, 4, 64, 64, seed=3)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    dt = torch.bfloat16 if precision == 'bf16' else torch.float16
    res = {}
    for fuse in (True, False):
        net = build_hip_seeker(cfg, sd, precision, drop_path_rate=0.0).cuda().train()
        net.seeker.persistent_grads = True
        opt = FusedAdamWClip(list(net.parameters()), lr=1e-3, max_norm=0.3, module=net, fuse_cast=fuse)
        losses = []
        for it in range(4):
            om, fl = net(rgb, qm)
            loss = om.square().mean() * 1e-2 + fl.square().mean() * 1e-2
            loss.backward(); opt.step()
            losses.append(float(loss.detach()))
        assert (opt._tiles is not None) == fuse
        if fuse:
            n_tiles = sum((p.shape[0] // 64) * (p.reshape(p.shape[0], -1).shape[1] // 64) for p in net.parameters() if id(p) in opt._cast_keys)
            assert opt._tiles.shape[0] == n_tiles and n_tiles > 0
            reg = net.seeker.__dict__['_wreg']
            for k, (p, Wc, Wt, N, K) in reg.items():
                if isinstance(k, int):                                           # plain GEMM weights (fold entries: W' = Wfc Wproj, cast by the module)
                    w = p.detach().reshape(N, K).to(dt)
                    assert torch.equal(Wc, w) and torch.equal(Wt, w.t().contiguous()), (N, K, id(p) in opt._cast_keys)
        res[fuse] = (losses, [p.detach().clone() for p in net.parameters()], float(opt.grad_norm()))
    assert res[True][0] == res[False][0] and res[True][2] == res[False][2]
    assert all(torch.equal(a, b) for a, b in zip(res[True][1], res[False][1]))


def test_fp16_overflow_skips_the_step_and_lowers_the_loss_scale(cuda):
    """precision='fp16': a non-finite gradient (an overflow of the scaled binary16 backward) must not poison the weights: the fused
    clip + AdamW kernels skip the update (parameters and moments untouched) and the module's loss-scale exponent drops by 4, on the device;
    the next good step applies normally and the exponent creeps back."""
    from tcow_amd.optim import FusedAdamWClip
    cfg = synth.seeker_config(num_total_frames=4, frame_height=32, frame_width=32, embed_dim=128, depth=2, num_heads=2, causal_attention=1)
    net = build_hip_seeker(cfg, synth.make_state_dict(cfg, 5), 'fp16').cuda().train()
    opt = FusedAdamWClip(list(net.parameters()), lr=1e-3, max_norm=0.3, module=net)
    clip = synth.make_clip(1, 4, 32, 32, seed=2)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    opt.zero_grad(set_to_none=True); om, fl = net(rgb, qm); (om.square().mean() + fl.square().mean()).backward(); opt.step()          # a normal step creates the scale state
    assert float(net.seeker.ls_log2) == -2.0 and all(bool(torch.isfinite(p).all()) for p in net.parameters())
    before = [p.detach().clone() for p in net.parameters()]
    m_before = opt.state[next(iter(net.parameters()))]['exp_avg'].clone()
    opt.zero_grad(set_to_none=True); om, fl = net(rgb, qm); (om.square().mean() + fl.square().mean()).backward()
    next(p for p in net.parameters() if p.grad is not None).grad.view(-1)[0] = float('inf')        # what an overflow looks like to the optimizer
    opt.step()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))                # skipped
    assert torch.equal(m_before, opt.state[next(iter(net.parameters()))]['exp_avg'])
    assert float(net.seeker.ls_log2) == -6.0
    opt.zero_grad(set_to_none=True); om, fl = net(rgb, qm); (om.square().mean() + fl.square().mean()).backward(); opt.step()          # scale 2^-4 of before: still a good step
    assert not all(torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters()) and -6.0 < float(net.seeker.ls_log2) < -5.9


def test_fp16_deferred_unscale_only_with_one_backward_per_step(cuda):
    """ADVICE r5 (engine.py deferred unscale): the loss scale is chosen per backward, so leaving gradients scaled for the optimizer is only sound when ONE
    backward feeds a step.  (a) default module (persistent_grads False) + FusedAdamWClip(module=): two model calls in one graph (the reference's per-query
    loop, pipeline.py:134-174) -- nothing is deferred, param.grad = the sum of the two calls' TRUE gradients; (b) persistent_grads: deferred, and
    unscale_() turns param.grad into the true gradient, after which step() gives bit-identical parameters; (c) a discarded optimizer no longer leaves
    gradients scaled; (d) a DataParallel-style replica never defers."""
    import gc
    from tcow_amd.optim import FusedAdamWClip
    cfg = synth.seeker_config(num_total_frames=4, frame_height=32, frame_width=32, embed_dim=128, depth=2, num_heads=2, causal_attention=1)
    sd = synth.make_state_dict(cfg, 5)
    clip = synth.make_clip(2, 4, 32, 32, seed=2)
    rgb = torch.from_numpy(clip['rgb']).cuda()
    qms = [torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda(), torch.from_numpy(synth.make_query_mask(clip, 1, 1)).cuda()]
    loss_of = lambda om, fl, k: om.square().mean() * (1e-4 * (1 + 300 * k)) + fl.square().mean() * 1e-4      # very different seed magnitudes -> different scales

    def grads(net):
        return [None if p.grad is None else p.grad.detach().clone().float() for p in net.parameters()]

    # (a) two calls, one graph
    net = build_hip_seeker(cfg, sd, 'fp16').cuda().train()
    opt = FusedAdamWClip(list(net.parameters()), lr=1e-3, max_norm=0.3, module=net)
    singles = []
    for k in range(2):
        opt.zero_grad(set_to_none=True)
        om, fl = net(rgb, qms[k]); loss_of(om, fl, k).backward()
        assert net.seeker.__dict__.get('pending_inv_scale') is None
        singles.append(grads(net))
    opt.zero_grad(set_to_none=True)
    (om0, fl0), (om1, fl1) = net(rgb, qms[0]), net(rgb, qms[1])
    (loss_of(om0, fl0, 0) + loss_of(om1, fl1, 1)).backward()
    assert net.seeker.__dict__.get('pending_inv_scale') is None
    both = grads(net)
    for g, a, b in zip(both, singles[0], singles[1]):
        if g is not None:
            assert torch.allclose(g, a + b, rtol=1e-5, atol=1e-9 + 1e-6 * float((a + b).abs().max()))
    opt.step()
    assert math.isfinite(float(opt.grad_norm())) and float(opt.skipped_steps) == 0

    # (b) persistent_grads: deferred; unscale_() -> true gradients; step after unscale_ == step without it, bit for bit
    res = {}
    for use_unscale in (False, True):
        net = build_hip_seeker(cfg, sd, 'fp16').cuda().train(); net.seeker.persistent_grads = True
        opt = FusedAdamWClip(list(net.parameters()), lr=1e-3, max_norm=0.3, module=net)
        om, fl = net(rgb, qms[0]); loss_of(om, fl, 0).backward()
        inv = net.seeker.__dict__.get('pending_inv_scale')
        assert inv is not None and float(inv) != 1.0
        scaled = grads(net)
        if use_unscale:
            opt.unscale_()
            assert net.seeker.__dict__.get('pending_inv_scale') is None
            for g, s_ in zip(grads(net), scaled):
                if g is not None:
                    assert torch.equal(g, s_ * float(inv))
            n = float(torch.nn.utils.clip_grad_norm_(net.parameters(), 1e9))      # what INTEGRATION.md's isfinite(clip_grad_norm_) check sees: the true norm
        opt.step()
        if use_unscale:
            assert abs(n - float(opt.grad_norm())) <= 1e-4 * n
        res[use_unscale] = [p.detach().clone() for p in net.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(res[False], res[True]))

    # (c) the optimizer is gone: the next backward unscales itself again
    del opt; gc.collect()
    om, fl = net(rgb, qms[0])
    with pytest.warns(UserWarning, match='no FusedAdamWClip'):
        loss_of(om, fl, 0).backward()
    assert net.seeker.__dict__.get('pending_inv_scale') is None
    true0 = grads(net)
    # (d) a replica (torch.nn.DataParallel's shallow copy, flagged by torch) never defers, whatever its dict inherited
    opt = FusedAdamWClip(list(net.parameters()), lr=1e-3, max_norm=0.3, module=net)
    net.seeker._is_replica = True
    net.zero_grad(set_to_none=True)
    try:
        om, fl = net(rgb, qms[0]); loss_of(om, fl, 0).backward()
        assert net.seeker.__dict__.get('pending_inv_scale') is None
    finally:
        del net.seeker._is_replica
    for g, t in zip(grads(net), true0):
        if g is not None:
            assert torch.equal(g, t)


def test_fp16_unscaling_inside_the_optimizer_is_bit_identical(cuda):
    """precision='fp16' with FusedAdamWClip(module=net) and no data-parallel hook: the backward leaves the gradient buckets loss-scaled and the
    optimizer folds the inverse scale (a power of two) into its clip coefficient (tcow_adamw_clip_step_scaled) instead of a multiplication pass
    over every bucket.  Same parameters, moments and gradient norm, bit for bit, as the path that unscales in the backward; param.grad of the
    deferred path = scaled gradient, pending_inv_scale = the factor."""
    from tcow_amd.optim import FusedAdamWClip
    cfg = synth.seeker_config(num_total_frames=4, frame_height=32, frame_width=32, embed_dim=128, depth=2, num_heads=2, causal_attention=1)
    sd = synth.make_state_dict(cfg, 5)
    clip = synth.make_clip(1, 4, 32, 32, seed=2)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    out = {}
    for deferred in (True, False):
        net = build_hip_seeker(cfg, sd, 'fp16').cuda().train()
        net.seeker.persistent_grads = True
        opt = FusedAdamWClip(list(net.parameters()), lr=1e-3, max_norm=0.3, module=net)
        if not deferred:
            net.seeker.__dict__['_defer_unscale'] = False
        norms = []
        for it in range(3):
            om, fl = net(rgb, qm)
            (om.square().mean() * 1e-4 + fl.square().mean() * 1e-4).backward()
            inv = net.seeker.__dict__.get('pending_inv_scale')
            assert (inv is not None) == deferred
            if it == 0:
                g0 = [p.grad.clone() * (inv if deferred else 1.0) for p in net.parameters() if p.grad is not None]
            opt.step()
            norms.append(float(opt.grad_norm()))
        assert not torch.equal(net.seeker.tracker_post_linear.weight.detach().cpu().float(), torch.as_tensor(np.asarray(sd['seeker.tracker_post_linear.weight'])).float())      # the steps did move the parameters
        out[deferred] = ([p.detach().clone() for p in net.parameters()], [opt.state[p]['exp_avg_sq'].clone() for p in net.parameters() if p in opt.state and 'exp_avg_sq' in opt.state[p]], norms, g0)
    (pa, va, na, ga), (pb, vb, nb, gb) = out[True], out[False]
    assert na == nb and all(math.isfinite(x) and x > 0 for x in na)
    assert all(torch.equal(a, b) for a, b in zip(ga, gb))                 # scaled gradient x inverse scale == the unscaled gradient (power of two)
    assert all(torch.equal(a, b) for a, b in zip(pa, pb)) and all(torch.equal(a, b) for a, b in zip(va, vb))


def test_two_threads_two_streams_match_sequential_calls(cuda):
    """The torch.nn.DataParallel thread model (train.py:222-223) on ONE device: two host threads, each with its own stream and its own replica (same
    weights, different clips), run forward + backward at the same time.  Outputs and every gradient must equal the sequential calls bit for bit --
    which needs the library's scratch (attention / weight-gradient / LayerNorm / mask-head workspaces, ops.workspace) to be private to a stream:
    with the per-device cache of rounds 1-5 both threads wrote the same attention and weight-gradient workspaces."""
    import threading
    cfg = synth.seeker_config(num_total_frames=8, frame_height=96, frame_width=128, embed_dim=256, depth=3, num_heads=4, causal_attention=1)
    sd = synth.make_state_dict(cfg, 11)
    clips = [synth.make_clip(2, 8, 96, 128, seed=40 + i) for i in range(2)]
    ins = [(torch.from_numpy(c['rgb']).cuda(), torch.from_numpy(synth.make_query_mask(c, 0, 0)).cuda()) for c in clips]

    def one(net, rgb, qm):
        om, fl = net(rgb, qm)
        (om.square().mean() + fl.square().mean()).backward()
        return om.detach().clone(), fl.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]

    for precision in ('bf16', 'fp32'):
        nets = [build_hip_seeker(cfg, sd, precision).cuda().train() for _ in range(2)]
        seq = [one(nets[i], *ins[i]) for i in range(2)]                     # sequential, default stream
        torch.cuda.synchronize()
        for rep in range(3):
            for n in nets:
                n.zero_grad(set_to_none=True)
            streams = [torch.cuda.Stream() for _ in range(2)]
            res, errs = [None, None], []
            gate = threading.Barrier(2)

            def worker(i):
                try:
                    torch.cuda.set_device(0)
                    with torch.cuda.stream(streams[i]):
                        gate.wait()
                        res[i] = one(nets[i], *ins[i])
                    streams[i].synchronize()
                except BaseException as e:       # noqa: BLE001
                    errs.append(e)

            th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
            [t.start() for t in th]; [t.join() for t in th]
            assert not errs, errs
            torch.cuda.synchronize()
            for i in range(2):
                assert torch.equal(res[i][0], seq[i][0]) and torch.equal(res[i][1], seq[i][1]), (precision, rep, i)
                for a, b in zip(res[i][2], seq[i][2]):
                    assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), (precision, rep, i)
    from tcow_amd import ops
    assert len({k[1] for k in ops._ws_cache if k[2] == 'attn'}) >= 3        # the default stream's and the two side streams' attention scratch are distinct buffers


def test_torch_dataparallel_wrapper_matches_the_plain_module(cuda):
    """The reference's own multi-GPU mode is single-process torch.nn.DataParallel (train.py:222-223).  On this one-GPU box both replicas are placed on
    device 0 (device_ids=[0, 0]): scatter -> replicate (shallow copies, `_is_replica`) -> two host threads on ONE stream -> gather, and the backward
    through Broadcast / ReduceAddCoalesced.  Outputs = the plain module's on the full batch; parameter gradients = the plain module's (the loss is a sum
    over clips).  fp32 so that the comparison is tight."""
    cfg = synth.seeker_config(num_total_frames=4, frame_height=64, frame_width=64, embed_dim=128, depth=2, num_heads=2, causal_attention=1)
    sd = synth.make_state_dict(cfg, 21)
    clip = synth.make_clip(2, 4, 64, 64, seed=5)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    net = build_hip_seeker(cfg, sd, 'fp32').cuda().train()
    om, fl = net(rgb, qm)
    (om.square().sum() + fl.square().sum()).backward()
    ref = [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    dp = torch.nn.DataParallel(net, device_ids=[0, 0])
    for rep in range(2):
        om2, fl2 = dp(rgb, qm)
        assert om2.shape == om.shape and torch.allclose(om2, om.detach(), rtol=0, atol=1e-6) and torch.allclose(fl2, fl.detach(), rtol=0, atol=1e-6)
        (om2.square().sum() + fl2.square().sum()).backward()
        for p, r in zip(net.parameters(), ref):
            if r is None:           # parameters the step does not touch (model.norm.*): DataParallel's Broadcast adjoint hands back zeros where the plain module leaves None
                assert p.grad is None or float(p.grad.abs().max()) == 0.0
            else:
                assert float((p.grad - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-9
        net.zero_grad(set_to_none=True)


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


def test_moving_the_module_drops_every_operand_cache(cuda):
    """.to() / .cuda() swap the parameters' .data without changing id() or ._version, the keys the operand caches are validated by --
    so Module._apply must drop the 16-bit weight copies, their registries / pointer table, the folded products and the persistent gradient
    buffers (there were two `_apply` definitions, the second shadowing the one that did).  After cuda -> cpu -> cuda a trained module must
    behave bit for bit like a fresh one built from its state dict, through a forward, a backward and a fused optimizer step."""
    from tcow_amd.optim import FusedAdamWClip
    cfg = synth.seeker_config(num_total_frames=4, frame_height=32, frame_width=32, embed_dim=128, depth=2, num_heads=2, causal_attention=1)
    sd = synth.make_state_dict(cfg, 5)
    clip = synth.make_clip(1, 4, 32, 32, seed=2)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()

    def step(net, opt):
        om, fl = net(rgb, qm)
        (om.square().mean() + fl.square().mean()).backward()
        opt.step()
        return om.detach().clone()

    def trainer(state):
        net = build_hip_seeker(cfg, state, 'bf16').cuda().train()
        net.seeker.persistent_grads = True
        return net, FusedAdamWClip(list(net.parameters()), lr=1e-3, max_norm=0.3, module=net)

    net, opt = trainer(sd)
    step(net, opt); step(net, opt)
    sk = net.seeker
    assert sk._wcache and sk.__dict__.get('_wreg') and sk._gbufs                 # the caches the move must drop exist
    state = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}
    net.cpu()
    assert not sk._wcache and not sk._gbufs and '_wreg' not in sk.__dict__ and '_wtab' not in sk.__dict__ and '_foldreg' not in sk.__dict__
    assert sk.__dict__.get('_param_list_cache') is None
    net.cuda()
    fresh, fopt = trainer(state)
    # the moved module's optimizer state (moments) lives on; compare forward + gradients (the part the caches feed), then an eval forward after
    # an in-place weight update through raw pointers on both
    for p in list(net.parameters()) + list(fresh.parameters()):
        p.grad = None
    o1, f1 = net(rgb, qm); o2, f2 = fresh(rgb, qm)
    assert torch.equal(o1, o2) and torch.equal(f1, f2)
    (o1.square().mean() + f1.square().mean()).backward(); (o2.square().mean() + f2.square().mean()).backward()
    for (k, a), (_, b) in zip(net.named_parameters(), fresh.named_parameters()):
        assert (a.grad is None) == (b.grad is None) and (a.grad is None or torch.equal(a.grad, b.grad)), k
    with torch.no_grad():
        for a, b in zip(net.parameters(), fresh.parameters()):
            a.data.mul_(1.01); b.data.mul_(1.01)                                 # (.data: no version bump -- only the epoch below tells the caches)
    net.seeker.invalidate_weight_cache(); fresh.seeker.invalidate_weight_cache()
    net.eval(); fresh.eval()
    with torch.no_grad():
        assert torch.equal(net(rgb, qm)[0], fresh(rgb, qm)[0])


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
        opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net)
        net.seeker.persistent_grads = True
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


@pytest.mark.parametrize('seed', list(range(16)))
def test_random_geometries_vs_oracle(cuda, seed):
    """Sixteen random small geometries (frames 1..9, 1..5 x 1..6 patches, D in {64, 128, 192}, depth 1..3, every causal_attention value,
    track-map stride / resize, norm_embeddings, rgb normalisation, 1..3 clips): forward AND gradients of the HIP module against the
    oracle (the CPU restatement pinned to the reference by tests/test_oracle_golden.py) -- fp32 to 1e-5 / 2e-4, fp16 and bf16 to their
    rounding bounds.  Catches what fixed goldens cannot: single-frame clips, one-patch frames, sequences of length 2, ragged tiles."""
    from oracle import seeker_oracle as so
    rng = np.random.default_rng(1000 + seed)
    T = int(rng.integers(1, 10)); Hp = int(rng.integers(1, 6)); Wp = int(rng.integers(1, 7)); D = int(rng.choice([64, 128, 192]))
    st = int(rng.choice([1, 2, 4]))
    cfg = synth.seeker_config(num_total_frames=T, frame_height=16 * Hp, frame_width=16 * Wp, embed_dim=D, depth=int(rng.integers(1, 4)), num_heads=D // 64,
                              causal_attention=int(rng.choice([-1, 0, 1, 2, 3, 4])), norm_embeddings=bool(rng.integers(0, 2)), track_map_stride=st,
                              track_map_resize=str(rng.choice(['bilinear', 'nearest'])), pretrained_norm=bool(rng.integers(0, 2)))
    sd = synth.make_state_dict(cfg, 2000 + seed)
    B = int(rng.integers(1, 4))
    clip = synth.make_clip(B, T, 16 * Hp, 16 * Wp, seed=3000 + seed)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.cat([torch.from_numpy(synth.make_query_mask(clip, 0, 0))] * 1, 0)
    if qm.shape[0] != B:
        qm = qm.expand(B, -1, -1, -1, -1).contiguous()
    tsd = so.to_torch_state_dict(sd)
    for v in tsd.values():
        v.requires_grad_(True)
    om_r, fl_r = so.seeker_forward(tsd, cfg, rgb, qm)
    Gm = torch.from_numpy(rng.standard_normal(size=tuple(om_r.shape)).astype(np.float32)); Gf = torch.from_numpy(rng.standard_normal(size=tuple(fl_r.shape)).astype(np.float32))
    ((om_r * Gm).sum() + (fl_r * Gf).sum()).backward()
    std = float(om_r.detach().std()) + 1e-6; fstd = float(fl_r.detach().std()) + 1e-6
    for precision, tol, ftol, gtol in (('fp32', 1e-5, 1e-5, 2e-4), ('fp16', 0.00625 * std + 1e-5, 0.0015 * fstd + 2e-5, 5e-3), ('bf16', 0.05 * std + 1e-4, 0.012 * fstd + 2e-4, 4e-2)):
        net = build_hip_seeker(cfg, sd, precision).cuda().train()           # drop_path_rate 0: train mode only to get gradients
        om, fl = net(rgb.cuda(), qm.cuda())
        assert float((om.detach().cpu() - om_r.detach()).abs().max()) < tol, (precision, cfg)
        assert float((fl.detach().cpu() - fl_r.detach()).abs().max()) < ftol, (precision, cfg)
        ((om * Gm.cuda()).sum() + (fl * Gf.cuda()).sum()).backward()
        for k, p_ in net.named_parameters():
            ref = tsd[k].grad
            if ref is None or float(ref.abs().max()) == 0.0:
                assert p_.grad is None or float(p_.grad.abs().max()) <= gtol, (precision, k)
                continue
            assert p_.grad is not None, (precision, k)
            assert float((p_.grad.cpu() - ref).abs().max()) <= gtol * float(ref.abs().max()) + 1e-7, (precision, k, cfg)
