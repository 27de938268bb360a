"""Caller rows P / L / M (pipeline, loss, metrics) against golden values captured from the real reference
(tests/golden/g5_pipeline_cfg1.npz, written by oracle/make_golden_pipeline.py)."""
import numpy as np
import pytest
import torch

from conftest import build_hip_seeker, load_golden
from tcow_amd import synth
from tcow_amd.metrics import calculate_metrics_mask_track
from tcow_amd.pipeline import SeekerPipeline, sample_query_inds
from tcow_amd.tcow_loss import default_args

B, Qs, T, H, W = 2, 3, 4, 64, 64


def _data(device=None):
    return synth.to_torch_tree(synth.make_kubric_batch(B, T, H, W, seed=900), device)


class _Replay(torch.nn.Module):
    """Stands in for the model on CPU: replays the reference's own output_mask (B,Qs,3,T,H,W)."""
    def __init__(self, out): super().__init__(); self.out = out
    def forward(self, rgb, qm):
        assert tuple(rgb.shape) == (B * Qs, 3, T, H, W) and tuple(qm.shape) == (B * Qs, 1, T, H, W)
        return self.out.reshape(B * Qs, 3, T, H, W), None


@pytest.mark.parametrize('phase', ['test', 'train'])
def test_query_selection_masks_losses_metrics_match_reference(phase):
    _, g = load_golden('g5_pipeline_cfg1')
    data = _data()
    ref_out = torch.from_numpy(g[f'{phase}::output_mask'])
    pipe = SeekerPipeline(_Replay(ref_out), num_queries=Qs, train_args=default_args(hard_negative_factor=1.0), phase=phase, device='cpu')
    sel = torch.from_numpy(g[f'{phase}::sel_query_inds'])
    if phase == 'test':                                                     # deterministic top-Qs by desirability (my_utils.py:287-292)
        kr = data['kubric_retval']
        assert torch.equal(sample_query_inds(B, Qs, kr['pv_inst_count'], kr['traject_retval_tf']['desirability_tf'], 'test'), sel)
    mr = pipe.forward_kubric(data, sel_query_inds=sel)
    for k in ('seeker_query_mask', 'snitch_occl_by_ptr', 'full_occl_cont_id', 'target_mask', 'sel_occl_fracs'):   # bit-exact (integer / mask work)
        assert np.array_equal(mr[k].numpy(), g[f'{phase}::{k}']), k
    assert np.allclose(mr['sel_desirability'].numpy(), g[f'{phase}::sel_desirability'])
    for tag, progress in (('p0', 0.0), ('p5', 0.5)):
        res = pipe.step_losses(data, mr, progress)
        for k in ('track', 'occl_mask', 'cont_mask'):
            assert abs(res[k] - float(g[f'{phase}_{tag}::{k}'])) < 2e-6, (k, tag)
        assert abs(float(res['total_seeker']) - float(g[f'{phase}_{tag}::total_seeker'])) < 2e-6
        for k, v in res['metrics'].items():
            ref = g[f'{phase}_{tag}::metric::{k}']
            assert (int(v) == int(ref)) if 'count' in k else abs(float(v) - float(ref)) < 1e-6, k
    assert np.allclose(mr['snitch_weights'].numpy(), g[f'{phase}::snitch_weights'], atol=1e-6)


def test_metrics_edge_cases():
    out = torch.full((1, 1, 3, 2, 4, 4), -1.0); tgt = torch.zeros(1, 1, 3, 2, 4, 4)
    m = calculate_metrics_mask_track(out, tgt)                              # nothing annotated -> means -1, counts 0 (metrics.py:85-96)
    assert all(float(v) == -1.0 for k, v in m.items() if k.startswith('mean')) and all(int(v) == 0 for k, v in m.items() if k.startswith('count'))
    tgt[0, 0, 0, 0, :2] = 1; out[0, 0, 0, 0, :1] = 1.0
    m = calculate_metrics_mask_track(out, tgt)
    assert int(m['count_snitch_iou']) == 1 and abs(float(m['mean_snitch_iou']) - 0.5) < 1e-6 and int(m['count_snitch_during_vis_iou']) == 1
    mp = calculate_metrics_mask_track(out[:, 0], tgt[:, 0] - 0.0, plugin=True)
    assert abs(float(mp['mean_snitch_iou']) - 0.5) < 1e-6


def test_focal_loss_follows_the_published_definition():
    """loss.py:49-51 switches the per-pixel term to torchvision.ops.sigmoid_focal_loss (alpha 0.25, gamma 2); torchvision is not installed,
    so this pins the restatement against the formula written out independently in float64 (aot_loss = 0: the term enters the loss directly)."""
    import argparse
    from tcow_amd.tcow_loss import TcowLosses, default_args
    a = argparse.Namespace(**{**vars(default_args()), 'focal_loss': True, 'aot_loss': 0.0})
    g = torch.Generator().manual_seed(1)
    lo = torch.randn(2, 3, 4, 8, 8, generator=g) * 2; tg = (torch.rand(2, 3, 4, 8, 8, generator=g) > 0.6).float(); w = torch.rand(2, 3, 4, 8, 8, generator=g) + 0.1
    got = TcowLosses(a, fused=False).mask_loss(lo, tg, w, 0.0, False)
    x, y = lo.double(), tg.double()
    p = 1.0 / (1.0 + torch.exp(-x))
    ce = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
    fl = (0.25 * y + 0.75 * (1 - y)) * ce * (1 - (p * y + (1 - p) * (1 - y))) ** 2
    assert abs(float(got) - float((fl * w.double()).mean())) < 1e-6


def test_hard_negative_band_matches_reference_blur():
    from tcow_amd.tcow_loss import hard_negative_band
    _, g = load_golden('g5_pipeline_cfg1')
    tm = torch.from_numpy(g['test::target_mask'])[:, :, 0]
    assert int(hard_negative_band(tm, H, W).sum()) == int(g['hard_negative_band_sum'])   # equality with gaussian_blur>0 asserted at generation


def test_hard_negative_band_equals_an_independent_gaussian_blur():
    """loss.py:136-146 calls torchvision's gaussian_blur(target, k, sigma=k) (not installed here: no vector of the reference's can pin it).
    Its published definition -- taps exp(-x^2 / 2 sigma^2) on x = -(k-1)/2 .. (k-1)/2, normalised, separable, reflect padding -- is what
    scipy.ndimage.gaussian_filter(sigma=k, radius=k//2, mode='mirror') computes, so the box-dilation shortcut of hard_negative_band is checked
    against that independent implementation on random masks, thin structures and border cases."""
    from scipy.ndimage import gaussian_filter
    from tcow_amd.tcow_loss import hard_negative_band
    rng = np.random.default_rng(5)
    for Hh, Ww in ((64, 64), (48, 80), (240, 320)):
        k = int(np.sqrt(Hh * Ww) / 12.0); k += (k % 2 == 0)
        for density in (0.0005, 0.01, 0.3):
            t = (rng.random((2, Hh, Ww)) < density).astype(np.float32)
            t[0, 0, :3] = 1.0; t[1, -1, -1] = 1.0; t[1, Hh // 2, :] = 1.0            # corners / borders / a line
            blur = np.stack([gaussian_filter(x.astype(np.float64), sigma=float(k), radius=k // 2, mode='mirror') for x in t])
            want = (blur > 0.0) & ~(t >= 0.5)
            got = hard_negative_band(torch.from_numpy(t), Hh, Ww).numpy()
            assert np.array_equal(got, want), (Hh, Ww, density)


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('bf16x3', 1e-4), ('fp16', 2.5e-3), ('bf16', 2e-2)])
def test_full_step_on_gpu_matches_reference(cuda, precision, tol):
    """Synthetic Kubric batch -> HIP Seeker (3 queries batched) -> TCOW loss -> backward, vs the reference's scalars."""
    _, g = load_golden('g5_pipeline_cfg1')
    cfg = synth.seeker_config(num_total_frames=T, frame_height=H, frame_width=W, embed_dim=256, depth=2, num_heads=4, causal_attention=1)
    net = build_hip_seeker(cfg, synth.make_state_dict(cfg, 900), precision).cuda().train()
    pipe = SeekerPipeline(net, num_queries=Qs, train_args=default_args(hard_negative_factor=1.0), phase='train', device='cuda')
    data = _data('cuda')
    mr = pipe.forward_kubric(data, sel_query_inds=torch.from_numpy(g['train::sel_query_inds']))
    assert float((mr['output_mask'].detach().cpu() - torch.from_numpy(g['train::output_mask'])).abs().max()) < {'fp32': 1e-4, 'bf16x3': 1e-4, 'fp16': 6.3e-4}.get(precision, 5e-3)
    res = pipe.step_losses(data, mr, 0.0)
    assert abs(float(res['total_seeker']) - float(g['train_p0::total_seeker'])) < tol
    res['total_seeker'].backward()
    gn = sum(float(p.grad.norm()) ** 2 for p in net.parameters() if p.grad is not None) ** 0.5
    assert abs(gn - float(g['train::grad_norm_total'])) < {'fp32': 1e-3, 'bf16x3': 1e-3, 'fp16': 6.3e-3}.get(precision, 5e-2) * float(g['train::grad_norm_total'])
    gb = net.seeker.tracker_post_linear.bias.grad.cpu().numpy()
    ref = g['train::grad::seeker.tracker_post_linear.bias']
    assert np.abs(gb - ref).max() < {'fp32': 1e-4, 'bf16x3': 2e-4, 'fp16': 6.3e-3}.get(precision, 5e-2) * np.abs(ref).max() + 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize('phase', ['test', 'train'])
def test_fused_mask_objective_matches_reference_scalars(cuda, phase):
    """tcow_mask_loss (HIP) on the reference's own logits: the three loss terms and the total, early (top-k = all) and
    late (top-k < all) in training, against the reference's scalars; its logit gradient against autograd of the tensor path."""
    _, g = load_golden('g5_pipeline_cfg1')
    data = _data('cuda')
    ref_out = torch.from_numpy(g[f'{phase}::output_mask']).cuda()
    sel = torch.from_numpy(g[f'{phase}::sel_query_inds'])
    args = default_args(hard_negative_factor=1.0)
    for tag, progress in (('p0', 0.0), ('p5', 0.5), ('p9', 0.09)):
        grads = {}
        for fused in (True, False):
            out = ref_out.clone().requires_grad_(True)
            pipe = SeekerPipeline(_Replay(out), num_queries=Qs, train_args=args, phase=phase, device='cuda')
            pipe.losses.fused = fused
            mr = pipe.forward_kubric(data, sel_query_inds=sel)
            res = pipe.step_losses(data, mr, progress)
            if tag != 'p9':
                for k in ('track', 'occl_mask', 'cont_mask', 'total_seeker'):
                    assert abs(float(res[k]) - float(g[f'{phase}_{tag}::{k}'])) < 3e-6, (k, tag, fused)
            res['total_seeker'].backward()
            grads[fused] = out.grad.clone()
        # identical up to rounding, except that a value within an ulp of the k-th largest may fall on the other side of the
        # top-k cut (torch's bce and the kernel's differ in the last bit): allow a handful of such boundary pixels
        scale = float(grads[False].abs().max())
        d = (grads[True] - grads[False]).abs()
        assert int((d > 2e-5 * scale).sum()) <= 4, (tag, int((d > 2e-5 * scale).sum()))
        assert float(d.max()) <= 1.01 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('aot', [0.8, 0.0])
def test_fused_mask_objective_with_focal_loss(cuda, aot):
    """train_args.focal_loss = True (loss.py:49-51: torchvision's sigmoid focal loss, alpha 0.25, gamma 2, replaces the BCE term) through the HIP objective
    (mask_loss.hip::pix_loss: value and closed-form derivative) against the tensor restatement + autograd: the three terms, the total and d(total)/d(logits),
    early (top-k = all) and late (top-k < all) in training.  (The reference's own number for this branch cannot be produced here -- torchvision is absent --;
    the tensor path restates torchvision's published definition, test_focal_loss_follows_the_published_definition.)"""
    import argparse
    _, g = load_golden('g5_pipeline_cfg1')
    data = _data('cuda')
    ref_out = torch.from_numpy(g['train::output_mask']).cuda() * 4.0           # wider logits: (1 - p_t)^2 spans its range
    sel = torch.from_numpy(g['train::sel_query_inds'])
    args = argparse.Namespace(**{**vars(default_args(hard_negative_factor=1.0)), 'focal_loss': True, 'aot_loss': aot})
    for progress in (0.0, 0.09, 0.5):
        got = {}
        for fused in (True, False):
            out = ref_out.clone().requires_grad_(True)
            pipe = SeekerPipeline(_Replay(out), num_queries=Qs, train_args=args, phase='train', device='cuda')
            pipe.losses.fused = fused
            mr = pipe.forward_kubric(data, sel_query_inds=sel)
            res = pipe.step_losses(data, mr, progress)
            res['total_seeker'].backward()
            got[fused] = ({k: float(res[k]) for k in ('track', 'occl_mask', 'cont_mask', 'total_seeker')}, out.grad.clone())
        for k, v in got[False][0].items():
            assert abs(got[True][0][k] - v) <= 3e-6 + 2e-5 * abs(v), (k, progress, got[True][0][k], v)
        scale = float(got[False][1].abs().max())
        d = (got[True][1] - got[False][1]).abs()
        assert scale > 0 and int((d > 3e-5 * scale).sum()) <= 4 and float(d.max()) <= 1.01 * scale, (progress, float(d.max()), scale)
    # ... and it is not the BCE objective
    args0 = argparse.Namespace(**{**vars(args), 'focal_loss': False})
    out = ref_out.clone().requires_grad_(True)
    pipe = SeekerPipeline(_Replay(out), num_queries=Qs, train_args=args0, phase='train', device='cuda'); pipe.losses.fused = True
    res0 = pipe.step_losses(data, pipe.forward_kubric(data, sel_query_inds=sel), 0.5)
    assert abs(float(res0['total_seeker']) - got[True][0]['total_seeker']) > 1e-3


@pytest.mark.gpu
def test_fused_mask_loss_edge_cases(cuda):
    """Frames without weight are left out (loss.py:176-181) and scale the loss by sqrt(selected fraction); ties at the top-k
    threshold; an empty target switches the Jaccard term off (loss.py:21); a negligible mean weight gives zero (loss.py:184)."""
    from tcow_amd.tcow_loss import TcowLosses
    torch.manual_seed(5)
    dev = 'cuda'
    BQ, Tn, Hn, Wn = 3, 5, 12, 20
    L = TcowLosses(default_args())

    def both(logits, target, weights, progress, weighted, check_grad=True):
        outs = []
        for fused in (True, False):
            x = logits.clone().requires_grad_(True)
            if fused:
                from tcow_amd import ops
                lo = torch.zeros(BQ, 3, Tn, Hn, Wn, device=dev); tg = torch.zeros_like(lo); dl = torch.full_like(lo, 7.0)
                lo[:, 1] = x.detach(); tg[:, 1] = target
                loss = torch.zeros(1, device=dev); total = torch.zeros((), device=dev)
                ops.mask_loss(lo, tg, 1, pixel_w=weights.contiguous(), weighted_aot=weighted, aot_loss=0.8,
                              topk_frac=min(max(1.0 - progress * 8.5, 0.15), 1.0), loss_weight=0.5, loss_out=loss, total=total, dlogits=dl)
                assert float((dl[:, 0] - 7.0).abs().max()) == 0.0 and float((dl[:, 2] - 7.0).abs().max()) == 0.0   # other channels untouched
                assert abs(float(total) - 0.5 * float(loss)) < 1e-7
                outs.append((float(loss), dl[:, 1] / 0.5))
            else:
                l = L.mask_loss(x[:, None], target[:, None], weights[:, None], progress, weighted)      # (B, Q=1, T, H, W)
                if l.requires_grad:
                    l.backward()
                outs.append((float(l), x.grad if x.grad is not None else torch.zeros_like(x)))
        (lf, gf), (lt, gt) = outs
        assert abs(lf - lt) < 2e-6 * max(1.0, abs(lt)), (lf, lt)
        if check_grad:
            d = (gf - gt).abs(); sc = max(float(gt.abs().max()), 1e-12)
            assert int((d > 2e-5 * sc).sum()) <= 2 and float(d.max()) <= 1.01 * sc, (int((d > 2e-5 * sc).sum()), float(d.max()), sc)
        return lf

    x = torch.randn(BQ, Tn, Hn, Wn, device=dev) * 3
    t = (torch.rand(BQ, Tn, Hn, Wn, device=dev) > 0.7).float()
    w = torch.rand(BQ, Tn, Hn, Wn, device=dev) + 0.5
    for weighted in (False, True):
        for progress in (0.0, 0.05, 0.5):
            both(x, t, w, progress, weighted)
    w2 = w.clone(); w2[0, 1] = 0; w2[2, 3:] = 0                             # three frames carry no weight
    for weighted in (False, True):
        assert both(x, t, w2, 0.06, weighted) > 0
    xq = torch.round(x)                                                     # few distinct values: many ties at the threshold
    lf = both(xq, t, torch.ones_like(w), 0.06, False, check_grad=False)      # (torch.topk breaks ties arbitrarily; the kernel shares them)
    assert lf > 0
    assert both(x, torch.zeros_like(t), w, 0.06, False) > 0                 # empty target: Jaccard term is 0
    assert both(x, t, torch.full_like(w, 5e-5), 0.06, True) == 0.0         # mean weight below 1e-4
    assert both(x, t, torch.zeros_like(w), 0.06, True) == 0.0              # no frame selected


@pytest.mark.gpu
@pytest.mark.parametrize('topk_frac', [1.0, 0.3])
def test_mask_objective_channels_as_one_batch_are_bit_identical(cuda, topk_frac):
    """tcow_mask_loss_batch: the three channels of the objective as one set of launches (job = blockIdx.z) against three tcow_mask_loss calls --
    terms, total (accumulated in channel order) and every logit gradient bit for bit; also with one channel left out (loss weight 0)."""
    from tcow_amd import ops
    torch.manual_seed(11)
    dev = 'cuda'
    BQ, Tn, Hn, Wn = 3, 5, 24, 40
    lo = torch.randn(BQ, 3, Tn, Hn, Wn, device=dev) * 2; tg = (torch.rand(BQ, 3, Tn, Hn, Wn, device=dev) > 0.7).float()
    pw = torch.rand(BQ, Tn, Hn, Wn, device=dev) + 0.5
    fw1 = torch.rand(BQ * Tn, device=dev) + 0.5; fw1[3] = 0.0
    fw2 = torch.rand(BQ * Tn, device=dev) + 0.5
    chans = ((0, pw, None, False, 1.0), (1, None, fw1, True, 0.3), (2, None, fw2, True, 0.2))
    for active in ((0, 1, 2), (0, 2)):
        single = dict(terms=torch.zeros(3, device=dev), total=torch.zeros((), device=dev), dl=torch.full_like(lo, 7.0))
        for c, p_, f_, wt, lw in chans:
            if c in active:
                ops.mask_loss(lo, tg, c, pixel_w=p_, frame_w=f_, weighted_aot=wt, aot_loss=0.8, topk_frac=topk_frac, loss_weight=lw,
                              loss_out=single['terms'][c:c + 1], total=single['total'], dlogits=single['dl'])
        batch = dict(terms=torch.zeros(3, device=dev), total=torch.zeros((), device=dev), dl=torch.full_like(lo, 7.0))
        ops.mask_loss_channels(lo, tg, [(c, p_, f_, wt, lw, batch['terms'][c:c + 1]) for c, p_, f_, wt, lw in chans if c in active],
                               aot_loss=0.8, topk_frac=topk_frac, total=batch['total'], dlogits=batch['dl'])
        assert torch.equal(single['terms'], batch['terms']) and torch.equal(single['total'], batch['total']) and torch.equal(single['dl'], batch['dl'])
        assert float(batch['total']) > 0 and all(float(batch['terms'][c]) > 0 for c in active)
        if 1 not in active:
            assert float((batch['dl'][:, 1] - 7.0).abs().max()) == 0.0            # the channel that was left out is untouched


@pytest.mark.gpu
def test_loss_gradient_is_scaled_on_the_device_only_when_needed(cuda):
    """tcow_scale_unless_one (the fused objective's backward): x *= s decided on the device -- s == 1 leaves every bit alone, any other s scales all
    n elements (n % 4 != 0 included); through autograd: (3 * total).backward() gives three times the gradient of total.backward()."""
    from tcow_amd import ops
    from tcow_amd.tcow_loss import FusedMaskObjective
    torch.manual_seed(2)
    for n in (1, 7, 4096, 100003):
        x = torch.randn(n, device='cuda')
        ref = x.clone()
        ops.scale_unless_one(x, torch.ones((), device='cuda'))
        assert torch.equal(x, ref)
        ops.scale_unless_one(x, torch.full((), 0.5, device='cuda'))
        assert torch.equal(x, ref * 0.5)
    BQ, Tn, Hn, Wn = 2, 3, 8, 16
    lo0 = torch.randn(1, BQ, 3, Tn, Hn, Wn, device='cuda'); tg = (torch.rand(1, BQ, 3, Tn, Hn, Wn, device='cuda') > 0.6).float()
    sw = torch.rand(1, BQ, Tn, Hn, Wn, device='cuda') + 0.5; fw = torch.rand(BQ * Tn, device='cuda') + 0.5
    grads = []
    for mult in (1.0, 3.0):
        lo = lo0.clone().requires_grad_(True)
        total, _ = FusedMaskObjective.apply(lo, tg, sw.reshape(BQ, Tn, Hn, Wn).contiguous(), fw, fw, (1.0, 0.5, 0.25), 0.8, 0.5)
        (total * mult).backward()
        grads.append(lo.grad.clone())
    assert float(grads[0].abs().max()) > 0 and torch.equal(grads[1], grads[0] * 3.0)
    # a second backward through the same node fails loudly (the gradient image was scaled in place and handed on; it used to come back as zeros): ADVICE r5
    from tcow_amd._lib import TcowError
    lo = lo0.clone().requires_grad_(True)
    total, _ = FusedMaskObjective.apply(lo, tg, sw.reshape(BQ, Tn, Hn, Wn).contiguous(), fw, fw, (1.0, 0.5, 0.25), 0.8, 0.5)
    total.backward(retain_graph=True)
    with pytest.raises(TcowError, match='backward called twice'):
        total.backward()
    # the channel workspace holds the 4 B / pixel bit-pattern image only when the radix select runs (topk_frac < 1 and aot_loss > 0)
    lib = ops.L.lib()
    full, small = lib.tcow_mask_loss_workspace_bytes(6, 128), lib.tcow_mask_loss_workspace_bytes_for(6, 128, 1.0, 0.8)
    assert full == lib.tcow_mask_loss_workspace_bytes_for(6, 128, 0.5, 0.8) and full - small == 6 * 128 * 4 and small == lib.tcow_mask_loss_workspace_bytes_for(6, 128, 0.5, 0.0)


@pytest.mark.gpu
def test_iou_counts_kernel_matches_tensor_path(cuda):
    """tcow_iou_counts (integer areas, exact) behind calculate_metrics_mask_track vs the tensor reductions on the CPU."""
    torch.manual_seed(3)
    out = torch.randn(2, 3, 3, 5, 24, 40); tgt = (torch.rand(2, 3, 3, 5, 24, 40) > 0.6).float()
    tgt[0, 1] = 0; tgt[1, :, 2, 2:] = 0                                     # un-annotated instances / frames -> counts drop, means may be -1
    from tcow_amd import ops
    cnt = ops.iou_counts(out.cuda(), tgt.cuda()).cpu()
    ob, tb = out > 0, tgt > 0.5
    assert torch.equal(cnt[..., 0], tb.sum((-1, -2)).int()) and torch.equal(cnt[..., 1], (ob & tb).sum((-1, -2)).int()) and torch.equal(cnt[..., 2], (ob | tb).sum((-1, -2)).int())
    a = calculate_metrics_mask_track(out.cuda(), tgt.cuda()); b = calculate_metrics_mask_track(out, tgt)
    for k in b:
        assert (int(a[k]) == int(b[k])) if 'count' in k else abs(float(a[k]) - float(b[k])) < 1e-6, k
    ap = calculate_metrics_mask_track(out[:, 0].cuda(), tgt[:, 0].cuda(), plugin=True); bp = calculate_metrics_mask_track(out[:, 0], tgt[:, 0], plugin=True)
    for k in bp:
        assert (int(ap[k]) == int(bp[k])) if 'count' in k else abs(float(ap[k]) - float(bp[k])) < 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize('phase', ['test', 'train'])
def test_mask_builder_and_weight_kernels_match_tensor_path(cuda, phase):
    """tcow_build_masks / tcow_snitch_weights (HIP) vs the tensor-op restatement that the goldens pin: query / target masks, occluder
    tags and ids bit-exact; pixel weights (class balancing, x2 occluded, 7x7 dilation band at 64x64, frame weights) to rounding."""
    _, g = load_golden('g5_pipeline_cfg1')
    sel = torch.from_numpy(g[f'{phase}::sel_query_inds'])
    ref_out = torch.from_numpy(g[f'{phase}::output_mask'])
    args = default_args()                                                   # hard_negative_factor = 3: the band is exercised
    res = {}
    for dev in ('cpu', 'cuda'):
        pipe = SeekerPipeline(_Replay(ref_out.to(dev)), num_queries=Qs, train_args=args, phase=phase, device=dev)
        mr = pipe.forward_kubric(_data(dev if dev == 'cuda' else None), sel_query_inds=sel)
        out = pipe.step_losses(_data(dev if dev == 'cuda' else None), mr, 0.3)
        res[dev] = (mr, out)
    mc, mg = res['cpu'][0], res['cuda'][0]
    assert mg['_target_pos_count'] is not None and mc['_target_pos_count'] is None      # the HIP builder ran on the GPU side only
    for k in ('seeker_query_mask', 'snitch_occl_by_ptr', 'full_occl_cont_id', 'target_mask', 'sel_occl_fracs'):
        assert mg[k].shape == mc[k].shape and mg[k].dtype == mc[k].dtype and torch.equal(mg[k].cpu(), mc[k]), k
    assert int(mg['_target_pos_count'][0]) == int((mc['target_mask'][:, :, 0] == 1).sum())
    wc, wg = mc['snitch_weights'], mg['snitch_weights'].cpu()
    assert wc.shape == wg.shape and float((wc - wg).abs().max()) < 1e-5 * float(wc.abs().max())
    assert int((wc != wc[0, 0, 0, 0, 0]).sum()) > 0                          # (not a constant map)
    for k in ('track', 'occl_mask', 'cont_mask', 'total_seeker'):
        assert abs(float(res['cuda'][1][k]) - float(res['cpu'][1][k])) < 5e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize('class_balancing,factor', [(True, 3.0), (False, 3.0), (True, 1.0), (False, 1.0)])
def test_snitch_weight_kernel_options(cuda, class_balancing, factor):
    """tcow_snitch_weights with class balancing / the hard-negative band switched on and off vs TcowLosses.pixel_weights * frame weights."""
    from tcow_amd import ops
    from tcow_amd.tcow_loss import TcowLosses
    torch.manual_seed(11)
    Bn, Qn, Tn, Hn, Wn = 1, 2, 3, 48, 80
    tgt = torch.zeros(Bn, Qn, 3, Tn, Hn, Wn)
    tgt[:, :, 0, :, 10:30, 20:50] = 1.0; tgt[0, 1, 0, 1] = 0.0              # one frame without the snitch
    ptr = (torch.rand(Bn, Qn, 1, Tn, Hn, Wn) > 0.9).to(torch.uint8) * 3
    fw = torch.rand(Bn, Qn, Tn) + 0.5
    L = TcowLosses(default_args(class_balancing=class_balancing, hard_negative_factor=factor))
    want = fw[..., None, None] * L.pixel_weights(tgt[:, :, 0], ptr[:, :, 0])
    pos = (tgt[:, :, 0] == 1).sum().to(torch.int32).reshape(1)
    got = ops.snitch_weights(tgt.cuda(), ptr.cuda(), fw.cuda(), pos.cuda(), class_balancing, factor).cpu()
    assert got.shape == want.shape and float((got - want).abs().max()) < 1e-5 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_query_table_kernel_matches_frame_decisions(cuda, seed):
    """tcow_build_query_masks' table pass vs the tensor restatement of data_utils.py:455-492 (frame_decisions) and of the frame weights
    (loss.py:55-83, 285-308) on random occlusion fractions / containment DAGs: frames with no, one and several container candidates, with and
    without a frontmost occluder.  Indices, ids, flags bit-exact; weights equal to the tensor expressions' f32 results."""
    from tcow_amd import ops
    from tcow_amd.pipeline import frame_decisions
    from tcow_amd.tcow_loss import TcowLosses
    g = torch.Generator().manual_seed(seed)
    Bn, Kn, Tn, Qn, Hn, Wn = 2, 7, 5, 3, 16, 32
    occl = torch.rand(Bn, Kn, Tn, 3, generator=g)
    occl[:, :, 1, 0] = 0.99                                                  # a frame above the occlusion threshold for everyone
    dag = torch.rand(Bn, Tn, Kn, Kn, 3, generator=g)
    dag[:, 0, :, :, 0] *= 0.5                                                # frame 0: no container candidate at all
    dag[:, 2, :, 3, 0] = 0.9; dag[:, 2, :, :3, 0] *= 0.5; dag[:, 2, :, 4:, 0] *= 0.5      # frame 2: exactly one candidate
    dag[:, 3, :, :, 2] *= 0.3                                                # frame 3: occluded-by maxima below thres / 2
    sel = torch.stack([torch.randperm(Kn, generator=g)[:Qn] for _ in range(Bn)])
    segm = torch.randint(0, Kn + 1, (Bn, 1, Tn, Hn, Wn), generator=g, dtype=torch.uint8)
    div = (torch.rand(Bn, Kn, Tn, Hn, Wn, generator=g) > 0.5).to(torch.uint8)
    div[:, 3, 4] = 0                                                         # instance 3 has no pixels in frame 4: a chosen but empty occluder / container
    args = default_args(); qt = 1
    tab = ops.build_query_masks(segm.cuda(), div.cuda(), occl.cuda(), dag.cuda(), sel.cuda(), qt, args.front_occl_thres, args.outer_cont_thres,
                                args.occluded_weight, args.occl_cont_zero_weight)
    fi, ci, ids, flags = frame_decisions(occl, dag, sel, args)
    assert torch.equal(tab['front_idx'].cpu().long(), fi) and torch.equal(tab['cont_idx'].cpu().long(), ci)
    assert torch.equal(tab['ids'].cpu(), ids) and torch.equal(tab['flags'].cpu(), flags)
    assert int((fi >= 0).sum()) > 0 and int((fi < 0).sum()) > 0 and int((ci >= 0).sum()) > 0 and int((ci < 0).sum()) > 0
    bi = torch.arange(Bn)[:, None].expand(Bn, Qn)
    sel_of = occl[bi, sel]
    assert torch.equal(tab['sel_occl_fracs'].cpu(), sel_of)
    qm, tg, pt, counts = ops.build_masks(segm.cuda(), div.cuda(), sel.cuda(), fi.cuda(), ci.cuda(), qt)
    assert torch.equal(tab['query_mask'], qm) and torch.equal(tab['target'], tg) and torch.equal(tab['snitch_occl_by_ptr'], pt) and torch.equal(tab['counts'], counts)
    L = TcowLosses(args)
    fw = tab['frame_w'].cpu()
    assert torch.equal(fw[0], L.frame_weights(sel_of, qt))
    tgc = tg.cpu()
    for ch in (1, 2):
        has = tgc[:, :, ch].flatten(-2).any(dim=-1).to(torch.float32)
        assert torch.equal(fw[ch], has * (1.0 - args.occl_cont_zero_weight) + args.occl_cont_zero_weight), ch
        assert 0 < int(has.sum()) < has.numel()


@pytest.mark.gpu
def test_droppath_rows_kernel(cuda):
    """tcow_droppath_rows vs the broadcast expressions of engine._row_vectors (vit_utils.py:139-154): bit-exact."""
    from tcow_amd import ops
    torch.manual_seed(5)
    depth, Bn, Tn, Sn = 4, 3, 5, 9
    N = Sn - 1
    keep_p = 1.0 - torch.linspace(0, 0.3, depth)
    u = torch.rand(depth, Bn * N + Bn * Tn + Bn)
    mask0 = torch.ones(Bn, Tn, Sn); mask0[:, :, 0] = 0; mask0 = mask0.reshape(-1)
    kp = keep_p[:, None]
    sc = (u + kp).floor() / kp
    kt = sc[:, :Bn * N].reshape(depth, Bn, 1, N); ks = sc[:, Bn * N:Bn * N + Bn * Tn].reshape(depth, Bn, Tn, 1); km = sc[:, Bn * N + Bn * Tn:].reshape(depth, Bn, 1)
    rt = torch.ones(depth, Bn, Tn, Sn); rt[:, :, :, 1:] = kt; rt = rt.reshape(depth, -1)
    want = torch.stack([rt, ks.expand(depth, Bn, Tn, Sn).reshape(depth, -1), km.expand(depth, Bn, Tn * Sn).reshape(depth, -1), rt * mask0[None]])
    got = ops.droppath_rows(u.cuda(), keep_p.cuda(), mask0.cuda(), Bn, Tn, Sn).cpu()
    assert torch.equal(got, want)
    assert 0 < int((got == 0).sum()) < got.numel()
