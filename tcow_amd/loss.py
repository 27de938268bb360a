"""Loss side of the Seeker training step (caller row L of SURVEY.md 8a), in plain torch ops on the GPU.

The loss is not a kernel target: it supplies grad_output for the hand-written backward and completes the step
that bench.py times.  `mask_loss` is the class-balanced BCE term of the reference's mask losses
(loss.py:164-225 use 0.2*weighted-BCE + 0.8*bootstrapped/Jaccard terms; the full restatement lives in
tcow_amd/tcow_loss.py once row L is built -- see DESIGN.md).
"""
import torch
import torch.nn.functional as F


def mask_loss(logits, target, pos_weight_power=0.7):
    """Weighted BCE-with-logits over (B,3,T,H,W): positives re-weighted by (neg/pos)^0.7 per channel
    (class balancing in the spirit of loss.py:101-128), mean over all elements."""
    pos = target.sum(dim=(0, 2, 3, 4), keepdim=True)
    tot = float(target.numel() // target.shape[1])
    w_pos = ((tot - pos).clamp_(min=1.0) / pos.clamp_(min=1.0)).pow_(pos_weight_power).clamp_(max=20.0)
    weight = torch.where(target > 0.5, w_pos.expand_as(target), torch.ones_like(target))
    return F.binary_cross_entropy_with_logits(logits, target, weight=weight)
