"""Caller row M: per-frame mask IoU metrics of eval/metrics.py:9-113, as tensor reductions.

The reference binarises on the device, copies the (B,Q,3,T) area tables to the host and walks them in a Python triple
loop; here the same 12 numbers come out of masked tensor reductions (no host loop, works on CPU or GPU tensors)."""
import torch


def calculate_metrics_mask_track(output_mask, target_mask, plugin=False):
    """output_mask logits, target_mask in {0,1} (or -1 = unlabeled frame for plugin data), shape (B,Q,3,T,H,W)
    ((B,3,T,H,W) when plugin=True).  Returns the reference's dict: mean_* (f32, -1 when empty) and count_* (int32)."""
    lead = tuple(target_mask.shape[:-2])                                   # (B,Q,C,T) or (B,C,T)
    if plugin:
        lead = (lead[0], 1) + lead[1:]                                     # metrics.py:26-29
    Cmt = lead[2]
    if output_mask.is_cuda and output_mask.dtype == torch.float32 and target_mask.dtype == torch.float32 and (output_mask.shape[-1] * output_mask.shape[-2]) % 4 == 0:
        from . import ops                                                  # one pass over both tensors (tcow_iou_counts)
        cnt_i = ops.iou_counts(output_mask.detach(), target_mask).reshape(lead + (3,))
        mean6, n6 = ops.iou_means(cnt_i.reshape((-1,) + tuple(cnt_i.shape[2:])))            # the masked means below as one launch (tcow_iou_means)
        out = {}
        for i, k in enumerate(('snitch_iou', 'occl_mask_iou', 'cont_mask_iou', 'snitch_during_vis_iou', 'snitch_during_occl_iou', 'snitch_during_cont_iou')):
            out['mean_' + k] = mean6[i]; out['count_' + k] = n6[i]
        return out
    else:
        out_b = (output_mask > 0.0).reshape(lead + tuple(output_mask.shape[-2:]))      # metrics.py:19
        tgt_b = (target_mask > 0.5).reshape(lead + tuple(target_mask.shape[-2:]))      # metrics.py:20
        t_area = tgt_b.sum(dim=(-1, -2)).to(torch.float64)                 # (B,Q,C,T)
        inter = (out_b & tgt_b).sum(dim=(-1, -2)).to(torch.float64)
        union = (out_b | tgt_b).sum(dim=(-1, -2)).to(torch.float64)
    iou = inter / (union + 1e-7)                                           # metrics.py:55-66
    has = t_area > 0
    sn = has[:, :, 0]

    # the six masked means as ONE set of tensor reductions over a stacked selection (no host round trip; the reference loops on the host)
    names, sels, vals = [], [], []
    def want(name, sel, values):
        names.append(name); sels.append(sel); vals.append(values)
    want('snitch_iou', sn, iou[:, :, 0])
    if Cmt >= 2:
        want('occl_mask_iou', has[:, :, 1], iou[:, :, 1])
        want('snitch_during_vis_iou', sn & ~has[:, :, 1], iou[:, :, 0])                                   # metrics.py:70-72
        want('snitch_during_occl_iou', sn & has[:, :, 1], iou[:, :, 0])                                   # metrics.py:74-76
    if Cmt >= 3:
        want('cont_mask_iou', has[:, :, 2], iou[:, :, 2])
        want('snitch_during_cont_iou', sn & has[:, :, 2], iou[:, :, 0])                                   # metrics.py:78-80
    sel = torch.stack(sels).flatten(1); val = torch.stack(vals).flatten(1)
    n = sel.sum(dim=1)
    mean = torch.where(n > 0, (val * sel).sum(dim=1) / n.clamp(min=1), torch.full_like(n, -1.0, dtype=torch.float64)).to(torch.float32)
    n32 = n.to(torch.int32)
    res = {k: (-1.0, 0) for k in ('snitch_iou', 'occl_mask_iou', 'cont_mask_iou', 'snitch_during_vis_iou', 'snitch_during_occl_iou', 'snitch_during_cont_iou')}
    for i, k in enumerate(names):
        res[k] = (mean[i], n32[i])
    dev = output_mask.device
    out = {}
    for k, (m, n) in res.items():
        out['mean_' + k] = m if torch.is_tensor(m) else torch.tensor(m, dtype=torch.float32, device=dev)
        out['count_' + k] = n if torch.is_tensor(n) else torch.tensor(n, dtype=torch.int32, device=dev)
    return out
