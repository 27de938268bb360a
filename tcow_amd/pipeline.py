"""Caller row P: the per-batch query loop of pipeline.py:85-240, as tensor ops + ONE batched Seeker call.

`sample_query_inds` / `fill_kubric_query_target_mask_flags` restate utils/my_utils.py:265-305 and
data/data_utils.py:414-510 without their Python loops over (B, T); the Qs sequential forwards of pipeline.py:134-158
become a single forward over B*Qs rows (identical per row, tests/test_gpu_seeker.py::test_full_size_properties)."""
import numpy as np
import torch

from .tcow_loss import TcowLosses, default_args


def to_dev(t, dev):
    """Host -> device without a stream drain: pageable copies block until the stream is empty, pinned ones do not."""
    if t.is_cuda or torch.device(dev).type != 'cuda':
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


def sample_query_inds(B, Qs, inst_count, desirability, phase, rng=None):
    """my_utils.py:265-305. Test phases: the Qs most desirable valid instances. Train phases: elitist shuffle
    (inequality 9) plus the occasional uniformly random last query, drawn from `rng` (numpy Generator)."""
    sel = torch.zeros(B, Qs, dtype=torch.int64)
    for b in range(B):
        Qt = int(inst_count[b].item())
        to_rank = desirability[b, :Qt, 0].detach().cpu().numpy()
        exact = np.argsort(to_rank)[::-1]
        valid = exact[to_rank[exact] >= 0.0]
        assert len(valid) >= Qs, f'Not enough valid queries available for batch index {b}.'
        if 'test' not in phase:
            rng = rng or np.random.default_rng()
            w = np.power(np.linspace(1, 0, num=len(valid), endpoint=False), 9); w = w / np.abs(w).sum()
            rough = rng.choice(valid, size=len(valid), replace=False, p=w)          # my_utils.py:123-141
        else:
            rough = valid
        sel[b] = torch.as_tensor(np.ascontiguousarray(rough[:Qs]))
        if 'test' not in phase:
            if rng.random() < np.clip(0.2 + Qs * 0.1, 0.3, 0.5):
                sel[b, -1] = int(rough[rng.integers(Qs - 1, len(valid))])
    return sel


def fill_kubric_query_target_mask_flags(segm, div_segm, query_idx, qt_idx, occl_fracs, dag, args):
    """data_utils.py:414-510 for all (b, t) at once. segm (B,1,T,H,W) uint8, div_segm (B,M,T,H,W) uint8, query_idx (B),
    occl_fracs (B,M,T,3), dag (B,T,M,M,3). Returns (query_mask f32, snitch_occl_by_ptr u8, full_occl_cont_id u8,
    target_mask f32 (B,3,T,H,W), target_flags f32 (B,T,3))."""
    B, _, T, H, W = segm.shape
    dev = segm.device
    qi = to_dev(query_idx, dev).long()
    bidx = torch.arange(B, device=dev)
    seg = segm[:, 0]                                                       # (B,T,H,W)
    is_q = seg == (qi + 1).to(seg.dtype)[:, None, None, None]
    query_mask = torch.zeros_like(segm, dtype=torch.float32)
    query_mask[:, 0, qt_idx] = is_q[:, qt_idx].float()                     # data_utils.py:431: visible pixels at the query frame only
    div_q = div_segm[bidx, qi] == 1                                        # (B,T,H,W) amodal mask of the snitch
    occl = div_q & ~is_q
    snitch_occl_by_ptr = torch.where(occl, seg, torch.zeros_like(seg))[:, None]      # data_utils.py:435-437
    target = torch.zeros(B, 3, T, H, W, dtype=torch.float32, device=dev)
    target[:, 0] = div_q.float()                                           # data_utils.py:441
    of_q = occl_fracs.to(dev)[bidx, qi]                                    # (B,T,3)
    dag = dag.to(dev)
    dq = dag[bidx, :, qi]                                                  # (B,T,M,3): row of the snitch
    # frontmost occluder (data_utils.py:455-463)
    fmax, farg = dq[..., 2].max(dim=-1)
    has_front = (of_q[..., 0] >= args.front_occl_thres) & (fmax >= args.front_occl_thres / 2.0)
    tidx = torch.arange(T, device=dev)
    front_mask = div_segm[bidx[:, None], farg, tidx[None, :]] == 1         # (B,T,H,W)
    target[:, 1] = (front_mask & has_front[..., None, None]).float()
    # outermost container (data_utils.py:468-492)
    cont = dq[..., 0]
    cand = cont >= args.outer_cont_thres                                   # (B,T,M)
    has_cont = cand.any(dim=-1)
    self_contained = dag[..., 0].max(dim=-1)[0]                            # (B,T,M): how contained each instance l is by anything
    score = torch.where(cand, self_contained, torch.full_like(self_contained, float('inf')))
    outer = score.argmin(dim=-1)                                           # least-contained candidate, first on ties (python min)
    single = cand.sum(dim=-1) <= 1
    outer = torch.where(single, cont.argmax(dim=-1), outer)
    cont_mask = div_segm[bidx[:, None], outer, tidx[None, :]] == 1
    target[:, 2] = (cont_mask & has_cont[..., None, None]).float()
    ids = torch.zeros(B, T, 2, dtype=torch.uint8, device=dev)
    ids[..., 0] = torch.where(has_front, farg + 1, torch.zeros_like(farg)).to(torch.uint8)
    ids[..., 1] = torch.where(has_cont, outer + 1, torch.zeros_like(outer)).to(torch.uint8)
    flags = torch.stack([has_front.float(), has_cont.float(), of_q[..., 0].float()], dim=-1)
    return query_mask, snitch_occl_by_ptr, ids, target, flags


def frame_decisions(occl_fracs, dag, sel, args):
    """The per-frame decisions of data_utils.py:455-492 for all (b, q, t) at once (tiny tensors): sel (B,Q) query instances ->
    front_idx / cont_idx (B,Q,T) int64 (-1 = the frame has no frontmost occluder / outermost container), full_occl_cont_id
    (B,Q,T,2) u8 and target flags (B,Q,T,3) f32, identical to the per-query results of fill_kubric_query_target_mask_flags."""
    B, Q = sel.shape
    dev = dag.device
    bidx = torch.arange(B, device=dev)[:, None].expand(B, Q)
    of_q = occl_fracs[bidx, sel]                                            # (B,Q,T,3)
    dq = dag[bidx, :, sel]                                                  # (B,Q,T,M,3): row of the snitch
    fmax, farg = dq[..., 2].max(dim=-1)
    has_front = (of_q[..., 0] >= args.front_occl_thres) & (fmax >= args.front_occl_thres / 2.0)
    cont = dq[..., 0]
    cand = cont >= args.outer_cont_thres
    has_cont = cand.any(dim=-1)
    self_contained = dag[..., 0].max(dim=-1)[0][:, None].expand(cont.shape)
    score = torch.where(cand, self_contained, torch.full_like(self_contained, float('inf')))
    outer = score.argmin(dim=-1)
    outer = torch.where(cand.sum(dim=-1) <= 1, cont.argmax(dim=-1), outer)
    neg1 = torch.full_like(farg, -1)
    ids = torch.stack([torch.where(has_front, farg + 1, torch.zeros_like(farg)), torch.where(has_cont, outer + 1, torch.zeros_like(outer))], dim=-1).to(torch.uint8)
    flags = torch.stack([has_front.float(), has_cont.float(), of_q[..., 0].float()], dim=-1)
    return torch.where(has_front, farg, neg1), torch.where(has_cont, outer, neg1), ids, flags


class SeekerPipeline:
    """Counterpart of MyTrainPipeline (pipeline.py:15-258) around a tcow_amd (or any) Seeker module."""

    def __init__(self, seeker, num_queries=3, train_args=None, phase='train', device='cuda', rng=None):
        self.seeker = seeker
        self.Qs = num_queries
        self.args = train_args if train_args is not None else default_args()
        self.phase = phase
        self.device = device
        self.rng = rng
        self.losses = TcowLosses(self.args, phase)

    def forward_kubric(self, data_retval, sel_query_inds=None):
        kr = data_retval['kubric_retval']; tr = kr['traject_retval_tf']
        dev = self.device
        rgb = kr['pv_rgb_tf'].to(dev); segm = kr['pv_segm_tf'].to(dev); div = kr['pv_div_segm_tf'].to(dev)
        occl_fracs = tr['occl_fracs_tf'].to(dev); dag = tr['occl_cont_dag_tf'].to(dev); des = tr['desirability_tf']
        B, _, T, H, W = rgb.shape
        Qs = self.Qs
        qt = int(tr['query_time'][0].item())                               # pipeline.py:140: only [0] is used
        if sel_query_inds is None:
            sel_query_inds = sample_query_inds(B, Qs, kr['pv_inst_count'], des, self.phase, self.rng)
        pos_count = None; tab = None
        sel_host = sel_query_inds if not sel_query_inds.is_cuda else None
        if sel_host is not None and sel_host.numel() and (int(sel_host.min()) < 0 or int(sel_host.max()) >= min(occl_fracs.shape[1], dag.shape[2])):
            # the tensor path below would raise on such an index (occl_fracs[b, sel], dag[b, :, sel]); the table kernel reads through it
            raise IndexError(f'query instance index out of range: {sel_host.tolist()} with {occl_fracs.shape[1]} trajectories / {dag.shape[2]} instances')
        Mi = div.shape[1]
        # (the table kernel wants B*Qs*T >= 1 + 2 Qs threads for its counters and instance ids that fit a byte: otherwise the tensor path)
        if (segm.is_cuda and segm.dtype == torch.uint8 and div.dtype == torch.uint8 and (H * W) % 16 == 0 and B * Qs * T >= 1 + 2 * Qs and B * Qs * T < 65536
                and Mi <= 255):
            # one HIP pass over the segmentation maps for all queries (tcow_build_masks); the per-frame occluder / container choice
            # stays a handful of tensor ops on (B,Qs,T,M)-sized data
            from . import ops
            sel_d = to_dev(sel_query_inds, dev)
            a = self.args
            tab = ops.build_query_masks(segm.contiguous(), div.contiguous(), occl_fracs, dag, sel_d, qt, a.front_occl_thres, a.outer_cont_thres,
                                        a.occluded_weight, a.occl_cont_zero_weight)
            query_mask, target, counts = tab['query_mask'], tab['target'], tab['counts']
            nonzero = counts                                                # [1 + 2q], [2 + 2q] != 0: inspected on the host below
            pos_count = counts[0:1]
            snitch_ptr, ids_stack = tab['snitch_occl_by_ptr'], tab['ids']
        else:
            qms, ptrs, idss, tgts, nonzero = [], [], [], [], []
            for q in range(Qs):                                            # cheap tensor ops; the model call below is batched
                qm, ptr, ids, tgt, _ = fill_kubric_query_target_mask_flags(segm, div, sel_query_inds[:, q], qt, occl_fracs, dag, self.args)
                nonzero += [qm.any(), tgt.any()]
                qms.append(qm); ptrs.append(ptr); idss.append(ids); tgts.append(tgt)
            nonzero = torch.stack(nonzero)
            query_mask = torch.stack(qms, 1); target = torch.stack(tgts, 1)    # (B,Qs,1,T,H,W), (B,Qs,3,T,H,W)
            snitch_ptr, ids_stack = torch.stack(ptrs, 1), torch.stack(idss, 1)
        # pipeline.py:149-154 raises on an all-zero query / target mask.  The flags travel to pinned host memory behind the
        # mask kernels and are inspected after the model call has been queued: same error, before any loss / update, but the
        # host never waits on an empty stream.
        if nonzero.is_cuda:
            host_flags = torch.empty(nonzero.shape, dtype=nonzero.dtype, pin_memory=True)
            host_flags.copy_(nonzero, non_blocking=True)
            flags_ready = torch.cuda.Event(); flags_ready.record()
        else:
            host_flags, flags_ready = nonzero, None
        f0 = 1 if tab is not None else 0                                    # (the table's first entry is the positive-pixel count)
        if getattr(self.seeker, 'shares_rgb', False):                          # tcow_amd Seeker: the clip's frames go in once for its Qs queries
            out_mask, _ = self.seeker(rgb, query_mask.reshape(B * Qs, 1, T, H, W))
        else:
            rgb_rep = rgb[:, None].expand(B, Qs, 3, T, H, W).reshape(B * Qs, 3, T, H, W)
            out_mask, _ = self.seeker(rgb_rep, query_mask.reshape(B * Qs, 1, T, H, W))   # pipeline.py:157-158, Qs calls in one
        if flags_ready is not None:
            flags_ready.synchronize()
        for q in range(Qs):
            if not bool(host_flags[f0 + 2 * q]):
                raise RuntimeError(f'seeker_query_mask all zero? q: {q} query_idx: {sel_query_inds[:, q]} qt_idx: {qt}')   # pipeline.py:149-151
            if not bool(host_flags[f0 + 2 * q + 1]):
                raise RuntimeError(f'target_mask all zero? q: {q}')        # pipeline.py:152-154
        bi = torch.arange(B)
        sel_des = to_dev(torch.stack([des[bi, sel_query_inds[:, q], 0] for q in range(Qs)], 1), dev)       # (a host-side gather: desirability arrives on the host)
        if tab is not None:
            sel_dev, sel_of = sel_d, tab['sel_occl_fracs']
        else:
            sel_dev = to_dev(sel_query_inds, dev); bi_dev = to_dev(bi, dev)
            sel_of = torch.stack([occl_fracs[bi_dev, sel_dev[:, q]] for q in range(Qs)], 1)
        extra = {}
        if tab is not None:                                                  # frame weights of loss.py:55-83, 285-308 from the table kernel, with what they were made from
            extra = {'_frame_w': tab['frame_w'], '_frame_w_key': (float(self.args.occluded_weight), float(self.args.occl_cont_zero_weight), qt)}
        return {
            **extra,
            'sel_query_inds': sel_dev,
            'sel_occl_fracs': sel_of,                                                                     # (B,Qs,T,3)
            'sel_desirability': sel_des,
            'seeker_input': rgb, 'seeker_query_mask': query_mask,
            'snitch_occl_by_ptr': snitch_ptr, 'full_occl_cont_id': ids_stack, '_target_pos_count': pos_count,
            'target_mask': target, 'output_mask': out_mask.reshape(B, Qs, 3, T, H, W),
        }

    def forward_plugin(self, data_retval):                                 # pipeline.py:202-240
        dev = self.device
        rgb = data_retval['pv_rgb_tf'].to(dev)
        qm = data_retval['pv_query_tf'].to(dev).float(); tgt = data_retval['pv_target_tf'].to(dev).float()
        if not bool(qm.any()):
            raise RuntimeError('seeker_query_mask all zero?')
        out_mask, out_flags = self.seeker(rgb, qm)
        return {'seeker_input': rgb, 'seeker_query_mask': qm, 'target_mask': tgt, 'output_mask': out_mask, 'output_flags': out_flags}

    def forward_plugin_items(self, items, max_batch=32):
        """All (query, usage-mode) clips of a plugin video in as few Seeker calls as fit (`max_batch` clips each) instead of one
        B = 1 forward per item (eval/test.py forces batch_size 1, args.py:276; data_plugin.py:141-156 makes every (query frame,
        stride) pair its own dataset item).  `items`: dicts from tcow_amd.plugin_data.build_plugin_item / eval_items.  Batch rows
        are independent, so the result equals the sequential one bit for bit."""
        outs = []
        for i in range(0, len(items), max_batch):
            chunk = items[i:i + max_batch]
            batch = {k: torch.stack([torch.as_tensor(it[k]) for it in chunk]) for k in ('pv_rgb_tf', 'pv_query_tf', 'pv_target_tf')}
            outs.append(self.forward_plugin(batch))
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}

    def step_losses(self, data_retval, model_retval, progress=0.0):
        qt = int(data_retval['kubric_retval']['traject_retval_tf']['query_time'][0].item())
        return self.losses.entire_batch(self.losses.per_example(model_retval, qt, progress))
