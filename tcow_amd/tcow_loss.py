"""Caller row L: the TCOW mask-tracking objective of loss.py:55-421 in tensor ops.

Not a kernel target: it supplies grad_output for the hand-written Seeker backward and completes the training step.
Every rule of the reference is kept, including its quirks (cited inline), so that the scalars match golden values
captured from the reference (tests/golden/g5_*.npz)."""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from .metrics import calculate_metrics_mask_track

DEFAULT_ARGS = dict(track_lw=1.0, occl_mask_lw=0.5, cont_mask_lw=0.5, occluded_weight=5, occl_cont_zero_weight=0.02,
                    class_balancing=True, focal_loss=False, aot_loss=0.8, hard_negative_factor=3.0,
                    front_occl_thres=0.95, outer_cont_thres=0.75)      # args.py:182-212


def default_args(**over):
    d = dict(DEFAULT_ARGS); d.update(over)
    return SimpleNamespace(**d)


def bootstrap_warmup_loss(loss_pixels, topk_frac):                         # loss.py:13-17
    k = int(topk_frac * loss_pixels.numel())
    if k == loss_pixels.numel():
        return loss_pixels.mean()          # top-k of everything is everything (early training: topk_frac == 1)
    return torch.topk(loss_pixels.flatten(), k=k)[0].mean()


def tversky_loss(logits, target, alpha=1.0, beta=1.0, eps=0.1):            # loss.py:20-32
    if target.mean() >= 1e-6:
        p0 = torch.sigmoid(logits); p1 = 1.0 - p0
        g0 = target; g1 = 1.0 - target
        num = torch.sum(p0 * g0)
        den = num + alpha * torch.sum(p0 * g1) + beta * torch.sum(p1 * g0)
        return 1.0 - (num / (den + eps))
    return torch.tensor(0.0, device=logits.device)


def hard_negative_band(target, H, W):
    """loss.py:136-146: gaussian_blur(target, k, sigma=k) > 0 with k = odd(int(sqrt(HW)/12)) is a k x k box dilation
    (all Gaussian taps are > 0.89 and the reflect padding only mirrors pixels that are inside the window anyway)."""
    k = int(np.sqrt(H * W) / 12.0)
    if k % 2 == 0:
        k += 1
    shp = target.shape
    x = target.reshape(-1, 1, H, W)
    x = F.max_pool2d(x, kernel_size=(k, 1), stride=1, padding=(k // 2, 0))        # box dilation is separable
    d = F.max_pool2d(x, kernel_size=(1, k), stride=1, padding=(0, k // 2)).reshape(shp) > 0.0
    return d & ~(target >= 0.5)


class FusedMaskObjective(torch.autograd.Function):
    """total = sum_c lw[c] * mask_loss_c over the three channels of (B,Q,3,T,H,W) logits, value and logit gradient
    from libtcow_hip's tcow_mask_loss in the forward (no host synchronisation).  Returns (total, terms[3])."""

    @staticmethod
    def forward(ctx, logits, target, snitch_w, occl_fw, cont_fw, lws, aot_loss, topk_frac, focal=False):
        from . import ops
        B, Q, C, T, H, W = logits.shape
        lo = logits.detach().reshape(B * Q, C, T, H, W).contiguous(); tg = target.reshape(B * Q, C, T, H, W).contiguous()
        acc = torch.zeros(4, dtype=torch.float32, device=lo.device)
        terms = acc[:3]; total = acc[3]
        need_grad = logits.requires_grad
        dl = torch.empty_like(lo) if need_grad else None
        jobs = []
        for c, (pw, fw, weighted) in enumerate(((snitch_w, None, False), (None, occl_fw, True), (None, cont_fw, True))):
            if lws[c] > 0.0:
                jobs.append((c, pw, fw, weighted, lws[c], terms[c:c + 1]))
            elif dl is not None:
                dl[:, c].zero_()
        if jobs:                                   # the active channels as ONE set of launches (tcow_mask_loss_batch)
            ops.mask_loss_channels(lo, tg, jobs, aot_loss=aot_loss, topk_frac=topk_frac, total=total, dlogits=dl, focal=focal)
        ctx.dl = dl; ctx.shape = logits.shape
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None, None, None, None
        dl = ctx.dl; ctx.dl = None
        if dl is None:         # the gradient image is scaled IN PLACE below and handed on: a second backward through this node has nothing left to scale
            from ._lib import TcowError
            raise TcowError('FusedMaskObjective: backward called twice on the same objective (retain_graph=True is not supported: '
                            'd(loss)/d(logits) is scaled in place by the first call) -- recompute the loss, or use TcowLosses(fused=False)')
        if g_total.dtype == torch.float32 and g_total.numel() == 1 and g_total.device == dl.device:
            from . import ops
            ops.scale_unless_one(dl, g_total.contiguous())          # in place, and not at all when the upstream gradient is 1 (decided on the device)
            return dl.reshape(ctx.shape), None, None, None, None, None, None, None, None
        return (dl * g_total).reshape(ctx.shape), None, None, None, None, None, None, None, None


class TcowLosses:
    def __init__(self, train_args=None, phase='train', fused=None):
        """fused: None = use libtcow_hip's fused mask objective whenever the logits live on the GPU."""
        self.args = train_args if train_args is not None else default_args()
        self.phase = phase
        self.fused = fused

    def frame_weights(self, sel_occl_fracs, query_time):                   # loss.py:55-83
        fw = (sel_occl_fracs[..., 0] * float(self.args.occluded_weight)).to(torch.float32).clip(min=1.0)
        fw = fw.clone()
        fw[-1, :, query_time] *= 0.2      # loss.py:79 indexes with the leaked loop variable b == B-1: only the last clip
        return fw

    def pixel_weights(self, target, snitch_occl_by_ptr, no_hard_negatives=False):   # loss.py:85-148
        (B, Q, T, H, W) = snitch_occl_by_ptr.shape
        pw = torch.ones((B, Q, T, H, W), dtype=torch.float32, device=target.device)
        if self.args.class_balancing:
            # loss.py:100-119 computes the two correction factors on the host in float64; the same arithmetic on
            # 0-dim float64 device tensors gives the same float32 factors without draining the stream
            pos = target == 1.0; neg = target == 0.0
            pf = (pos.sum() / pos.numel()).clip(min=0.05).double(); nf = (neg.sum() / neg.numel()).clip(min=0.05).double()
            more_pos = pf > nf
            ratio = torch.where(more_pos, nf / pf, pf / nf)
            pos_corr = torch.where(more_pos, ratio ** 0.7, ratio ** -0.3).float()
            neg_corr = torch.where(more_pos, ratio ** -0.3, ratio ** 0.7).float()
            pw = torch.where(neg, pw * neg_corr, pw)
            pw = torch.where(pos, pw * pos_corr, pw)
        pw = torch.where(snitch_occl_by_ptr != 0, pw * 2.0, pw)
        if self.args.hard_negative_factor > 1.0 and not no_hard_negatives:
            pw = torch.where(hard_negative_band(target, H, W), pw * float(self.args.hard_negative_factor), pw)
        return pw

    def mask_loss(self, logits, target, weights, progress, apply_weights_for_aot):   # loss.py:164-225
        which = weights
        while which.ndim > 3:
            which = which.any(dim=-1)
        which = which[..., None, None].expand_as(weights)
        n_sel = int(which.sum())                                              # one host sync, like the reference's which_frames.any()
        if n_sel > 0 and float(weights.mean()) >= 1e-4:
            if n_sel == which.numel():     # every frame carries weight (the usual case): skip the boolean gather
                lo = logits.reshape(-1); tg = target.reshape(-1); fw = weights.reshape(-1)
            else:
                lo = logits[which]; tg = target[which]; fw = weights[which]
            bce = F.binary_cross_entropy_with_logits(lo, tg, reduction='none')
            if self.args.focal_loss:
                # loss.py:49-51: torchvision.ops.sigmoid_focal_loss(x, y, reduction='none') with its defaults alpha = 0.25, gamma = 2 -- restated
                # from torchvision's published definition (torchvision is not installed here: no vector of the reference's pins this branch;
                # args.py:198 defaults it to False).  The fused kernels implement the same formula (mask_loss.hip::pix_loss).
                p = torch.sigmoid(lo)
                p_t = p * tg + (1.0 - p) * (1.0 - tg)
                bce = (0.25 * tg + 0.75 * (1.0 - tg)) * bce * (1.0 - p_t) ** 2
            custom = (bce * fw).mean()
            if self.args.aot_loss > 0.0:
                for_aot = bce * fw if apply_weights_for_aot else bce
                topk_frac = min(max(1.0 - progress * 8.5, 0.15), 1.0)
                boot = bootstrap_warmup_loss(for_aot, topk_frac)
                jac = boot if apply_weights_for_aot else tversky_loss(lo, tg, 1.0, 1.0, 0.1)
                loss = (boot + jac) / 2.0 * self.args.aot_loss + custom * (1.0 - self.args.aot_loss)
            else:
                loss = custom
            loss = loss * math.sqrt(n_sel / which.numel())
        else:
            loss = torch.tensor(0.0, device=logits.device)
        return loss

    def per_example(self, model_retval, query_time, progress, metrics_only=False):   # loss.py:238-329
        out = model_retval['output_mask']; tgt = model_retval['target_mask']
        if metrics_only:
            return {'metrics': calculate_metrics_mask_track(out, tgt)}
        a = self.args
        fused = (out.is_cuda and (out.shape[-1] * out.shape[-2]) % 4 == 0) if self.fused is None else self.fused
        if fused:
            return self._per_example_fused(model_retval, query_time, progress)
        res = {'track': None, 'occl_mask': None, 'cont_mask': None}
        if a.track_lw > 0.0:
            fw = self.frame_weights(model_retval['sel_occl_fracs'], query_time)
            pw = self.pixel_weights(tgt[:, :, 0], model_retval['snitch_occl_by_ptr'][:, :, 0])
            sw = fw[..., None, None] * pw
            model_retval['snitch_weights'] = sw
            res['track'] = self.mask_loss(out[:, :, 0], tgt[:, :, 0], sw, progress, False)
        for name, ch, lw in (('occl_mask', 1, a.occl_mask_lw), ('cont_mask', 2, a.cont_mask_lw)):
            if lw > 0.0:
                w = tgt[:, :, ch].any(dim=-1).any(dim=-1)[..., None, None].expand_as(tgt[:, :, ch]).to(torch.float32)
                w = w * (1.0 - a.occl_cont_zero_weight) + a.occl_cont_zero_weight
                res[name] = self.mask_loss(out[:, :, ch], tgt[:, :, ch], w, progress, True)
        res['metrics'] = calculate_metrics_mask_track(out, tgt)
        return res

    def _per_example_fused(self, model_retval, query_time, progress):
        """Same objective through tcow_mask_loss: weights are prepared with (synchronisation-free) tensor ops, the three
        channel losses and d(total)/d(logits) come from the HIP passes."""
        out = model_retval['output_mask']; tgt = model_retval['target_mask']
        a = self.args
        B, Q, C, T, H, W = out.shape
        sw = None; fws = [None, None]
        pre = model_retval.get('_frame_w')                                  # (3,B,Q,T) from tcow_build_query_masks, valid for the arguments it was made with
        if pre is not None and model_retval.get('_frame_w_key') != (float(a.occluded_weight), float(a.occl_cont_zero_weight), query_time):
            pre = None
        if a.track_lw > 0.0:
            fw = pre[0] if pre is not None else self.frame_weights(model_retval['sel_occl_fracs'], query_time)
            pos_count = model_retval.get('_target_pos_count')
            if pos_count is not None and tgt.is_contiguous():
                from . import ops                                           # class balancing, x2 occluded, dilation band and frame weights in two passes
                sw = ops.snitch_weights(tgt, model_retval['snitch_occl_by_ptr'], fw, pos_count, bool(a.class_balancing), float(a.hard_negative_factor))
            else:
                pw = self.pixel_weights(tgt[:, :, 0], model_retval['snitch_occl_by_ptr'][:, :, 0])
                sw = (fw[..., None, None] * pw).contiguous()
            model_retval['snitch_weights'] = sw
        for i, (ch, lw) in enumerate(((1, a.occl_mask_lw), (2, a.cont_mask_lw))):
            if lw > 0.0 and pre is not None:
                fws[i] = pre[ch]
            elif lw > 0.0:
                has = tgt[:, :, ch].flatten(-2).any(dim=-1).to(torch.float32)                       # (B,Q,T)
                fws[i] = (has * (1.0 - a.occl_cont_zero_weight) + a.occl_cont_zero_weight).contiguous()
        topk_frac = min(max(1.0 - progress * 8.5, 0.15), 1.0)
        lws = (float(a.track_lw), float(a.occl_mask_lw), float(a.cont_mask_lw))
        total, terms = FusedMaskObjective.apply(out, tgt, sw, fws[0], fws[1], lws, float(a.aot_loss), topk_frac, bool(a.focal_loss))
        res = {name: (terms[i] if lws[i] > 0.0 else None) for i, name in enumerate(('track', 'occl_mask', 'cont_mask'))}
        res['_fused_total'] = total
        res['metrics'] = calculate_metrics_mask_track(out, tgt)
        return res

    def entire_batch(self, loss_retval):                                   # loss.py:331-421 (logging omitted)
        """Scalars are returned as detached 0-dim tensors (float() them to log): no host synchronisation here."""
        a = self.args
        fused_total = loss_retval.get('_fused_total')
        terms = {k: ((v if v.dim() == 0 else torch.mean(v)) if torch.is_tensor(v) else -1.0) for k, v in loss_retval.items() if k not in ('metrics', '_fused_total')}
        if fused_total is not None:
            total = fused_total            # (terms are 0-dim per replica: their mean over replicas is the value itself)
        else:
            total = terms['track'] * a.track_lw + terms['occl_mask'] * a.occl_mask_lw + terms['cont_mask'] * a.cont_mask_lw
        out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in terms.items()}
        out['total_seeker'] = total
        out['metrics'] = loss_retval['metrics']
        return out
