"""Fused clip_grad_norm_ + AdamW for the Seeker training step (train.py:99-102, 239-243) on libtcow_hip.

Same arithmetic as torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.AdamW(params, lr).step()
(betas (0.9, 0.999), eps 1e-8, weight_decay 0.01 on every parameter -- the reference builds one parameter group with torch's
defaults), but as three kernel launches over a device-resident chunk table instead of ~250 per-tensor
foreach operations.  Parameters without a gradient are skipped, like torch does.

It IS a torch.optim.Optimizer: `param_groups` carries lr / betas / eps / weight_decay (read every step, so the reference's
MultiStepLR, train.py:236-243, attaches unchanged), and `state_dict()` / `load_state_dict()` use torch.optim.AdamW's layout
(per-parameter 'step', 'exp_avg', 'exp_avg_sq'), so a reference checkpoint's 'optim_seeker' (train.py:282) loads and what is
saved here loads into torch.optim.AdamW.

The kernels write the parameters through raw pointers; after every step the parameters' autograd version counters are bumped
(torch.autograd.graph.increment_version), so anything that caches derived copies keyed on `p._version` -- the Seeker's bf16
operand copies are -- can never serve stale weights.  `on_step` callbacks (e.g. QueryMaskTracker.invalidate_weight_cache, which
re-casts all GEMM operands in ONE launch) are an optimisation on top of that, not a correctness requirement.

One extension over torch's pair: a step whose total gradient norm is NaN / infinite is SKIPPED (parameters and moments untouched) instead
of poisoning the weights -- the overflow guard of the binary16 mode (precision='fp16'), whose loss-scale exponent is lowered in the same
step (device side, no synchronisation).  Skips are COUNTED on the device in every precision (`skipped_steps`, a device scalar; read it
-- one sync -- whenever you log: a NaN gradient in bf16 / fp32 training shows up there instead of silently freezing the weights), and the
update kernel subtracts them from the step number of its bias correction, so moments and correction stay in step; `state_dict()`
stores the applied-update count, i.e. what torch.optim.AdamW would have counted behind a GradScaler."""
import weakref

import numpy as np
import torch

from . import _lib as L

CHUNK = 65536


def _ops_stream():
    from .ops import _stream
    return _stream()


class FusedAdamWClip(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=0.3, module=None, fuse_cast=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise L.TcowError('FusedAdamWClip takes one parameter group (train.py:239-241 builds exactly one)')
        self.max_norm = max_norm
        self._table = None
        self._key = None
        self.scratch = None
        self.on_step = []          # callables run after every step
        self._tracker = None
        self._skip_carry = 0.0         # skipped steps of earlier pointer tables (the device counter lives in the scratch buffer)
        self._live = []; self._seen_grads = []; self._step_base = 0; self._steps_pending = 0
        # fuse_cast (16-bit modes, module= given): GEMM weights are updated tile by tile and the update writes their 16-bit W / W^T operand copies itself
        # (tcow_adamw_clip_step_cast) -- the module's batched re-cast then only covers what is left (the folded products)
        self.fuse_cast = bool(fuse_cast)
        self._tiles = None; self._flat_table = None; self._cast_keys = frozenset(); self._wreg_gen = None
        if module is not None:     # a Seeker / QueryMaskTracker: batch re-cast of its GEMM operand copies right after the update
            tracker = getattr(module, 'seeker', module)
            self._tracker = tracker
            if hasattr(tracker, 'invalidate_weight_cache'):
                self.on_step.append(tracker.invalidate_weight_cache)
            # A WEAK reference to the attached optimizer (engine._live_optim).  While it is alive, precision='fp16' has somebody who lowers the loss scale
            # after an overflow (run_backward warns otherwise), and -- with persistent_grads, one backward per step -- gradient buckets may stay loss-scaled:
            # step() folds the inverse scale into the clip coefficient.  Once this optimizer is discarded the module unscales in the backward again.
            tracker.__dict__['_optim_ref'] = weakref.ref(self)
        assert L.lib().tcow_adamw_chunk_bytes() == 40

    @property
    def params(self):
        return self.param_groups[0]['params']

    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @property
    def step_count(self):
        self._flush_steps()
        for p in self.params:
            st = self.state.get(p)
            if st:
                return int(st['step'])
        return 0

    def _flush_steps(self):
        """state[p]['step'] (CPU scalars, torch.optim.AdamW's layout) += the steps taken since the last flush: one foreach call when the state is
        read (state_dict, step_count, table rebuild) instead of one per step."""
        if self._steps_pending and self._live:
            torch._foreach_add_([self.state[p]['step'] for p in self._live], float(self._steps_pending))
            self._step_base += self._steps_pending
        self._steps_pending = 0

    def _cast_registry(self):
        """{id(parameter): (Wc, Wt, N, K)} of the GEMM weights whose 16-bit operand copies this optimizer may write (the module's registry of the current
        training forward: engine._get_weight), or {}."""
        trk = self._tracker
        if not self.fuse_cast or trk is None or getattr(trk, 'precision', None) not in ('bf16', 'fp16') or getattr(trk, '_is_replica', False):
            return {}
        out = {}
        for k, (p, Wc, Wt, N, K) in (trk.__dict__.get('_wreg') or {}).items():
            if isinstance(k, int) and isinstance(p, torch.nn.Parameter) and Wc is not None and Wt is not None and N % 64 == 0 and K % 64 == 0 and p.numel() == N * K:
                out[id(p)] = (Wc, Wt, N, K)
        return out

    def _lib(self):
        """The build of the library whose 16-bit format the module's operand copies have."""
        trk = self._tracker
        return L.lib('fp16') if (trk is not None and getattr(trk, 'precision', None) == 'fp16' and self._tiles is not None) else L.lib()

    def _build(self, live):
        reg = self._cast_registry()
        fused = [p for p in live if id(p) in reg]
        rows, flat_rows = [], []        # every parameter (gradient norm, in the order of `live`: the partial sums are folded in that order) / the flat-updated ones
        for p in live:
            st = self.state[p]
            if 'exp_avg' not in st:
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
            m, v = st['exp_avg'], st['exp_avg_sq']
            g = p.grad
            if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32
                    and m.dtype == torch.float32 and v.dtype == torch.float32):
                raise L.TcowError('FusedAdamWClip needs contiguous f32 parameters, gradients and moments')
            n = p.numel()
            for off in range(0, n, CHUNK):
                rows.append((p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off, min(CHUNK, n - off)))
                if id(p) not in reg:
                    flat_rows.append(rows[-1])
        if self.scratch is not None:
            self._skip_carry += float(self.scratch[-1])      # (rebuilds are rare: first step, re-allocated gradients, load_state_dict)
        self._table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(live[0].device)
        self.scratch = torch.zeros(len(rows) + 3, dtype=torch.float32, device=live[0].device)      # [chunk partials | coef, norm, skipped]
        self.scratch[-1] = self._skip_carry
        self._tiles = None; self._flat_table = None; self._cast_keys = frozenset()
        trk = self._tracker
        self._wreg_gen = trk.__dict__.get('_wreg_gen') if trk is not None else None
        if fused:
            assert L.lib().tcow_adamw_tile_bytes() == 56
            recs = []
            for p in fused:
                Wc, Wt, N, K = reg[id(p)]
                st = self.state[p]
                n0, k0 = np.meshgrid(np.arange(0, N, 64, dtype=np.int64), np.arange(0, K, 64, dtype=np.int64), indexing='ij')
                o = (n0 * K + k0).reshape(-1); ot = (k0 * N + n0).reshape(-1)
                r = np.empty((o.size, 7), dtype=np.int64)
                r[:, 0] = p.data_ptr() + 4 * o; r[:, 1] = p.grad.data_ptr() + 4 * o; r[:, 2] = st['exp_avg'].data_ptr() + 4 * o; r[:, 3] = st['exp_avg_sq'].data_ptr() + 4 * o
                r[:, 4] = Wc.data_ptr() + 2 * o; r[:, 5] = Wt.data_ptr() + 2 * ot; r[:, 6] = K | (N << 32)
                recs.append(r)
            self._tiles = torch.from_numpy(np.concatenate(recs, axis=0)).to(live[0].device)
            self._flat_table = torch.from_numpy(np.asarray(flat_rows, dtype=np.int64).reshape(-1, 5)).to(live[0].device)
            self._cast_keys = frozenset(id(p) for p in fused)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grp = self.param_groups[0]
        params = grp['params']
        # Fast path (every step but the first): the same Parameter list with the same gradient TENSOR OBJECTS as when the pointer table was
        # built (persistent gradient buckets), first / last gradient still at the recorded address -- one identity test per parameter instead of
        # two data_ptr() calls and a Tensor.__hash__ dictionary lookup each (0.7 ms of host time per step over ~250 parameters).
        fast = self._key is not None and len(params) == len(self._seen_grads)
        if fast:
            for p, g in zip(params, self._seen_grads):
                if p.grad is not g:
                    fast = False
                    break
        if fast and self._live:
            fast = self._live[0].grad.data_ptr() == self._key[0][1] and self._live[-1].grad.data_ptr() == self._key[-1][1]
        if fast and self._tracker is not None and self.fuse_cast:
            fast = self._tracker.__dict__.get('_wreg_gen') == self._wreg_gen       # the module's operand copies were re-allocated (or appeared): the tile table is stale
        if fast:
            live = self._live
        else:
            live = [p for p in params if p.grad is not None]
            if not live:
                return loss
            self._flush_steps()
            key = tuple((id(p), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr() if 'exp_avg' in self.state[p] else 0) for p in live)
            gen = self._tracker.__dict__.get('_wreg_gen') if self._tracker is not None else None
            if key != self._key or (self.fuse_cast and gen != self._wreg_gen):       # gradient / moment buffers moved (first step, re-allocated grads, load_state_dict) or the module's operand copies did: rebuild the pointer tables
                self._build(live)
                self._key = tuple((id(p), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr()) for p in live)
            self._live = live
            self._seen_grads = [p.grad for p in params]
            self._step_base = int(self.state[live[0]]['step'])
            self._steps_pending = 0
        if not live:
            return loss
        step = self._step_base + self._steps_pending + 1
        trk = self._tracker
        inv_scale = trk.__dict__.get('pending_inv_scale') if trk is not None else None      # binary16: the last backward left its gradients loss-scaled (it stays valid until the next backward rewrites them)
        if self._tiles is not None:
            lib = self._lib()
            L.check(lib.tcow_adamw_clip_step_cast(_ops_stream(), self._table.data_ptr(), self._table.shape[0], self._flat_table.data_ptr() if self._flat_table.shape[0] else None,
                                                  self._flat_table.shape[0], self._tiles.data_ptr(), self._tiles.shape[0],
                                                  float(grp['lr']), float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']), float(grp['weight_decay']), step,
                                                  float(self.max_norm or 0.0), self.scratch.data_ptr(), inv_scale.data_ptr() if inv_scale is not None else None),
                    'tcow_adamw_clip_step_cast', lib)
            trk.__dict__['_opt_cast_keys'] = self._cast_keys          # (consumed by the module's refresh_weights in the on_step callback below)
        else:
            L.check(L.lib().tcow_adamw_clip_step_scaled(_ops_stream(), self._table.data_ptr(), self._table.shape[0], float(grp['lr']),
                                                        float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']), float(grp['weight_decay']), step,
                                                        float(self.max_norm or 0.0), self.scratch.data_ptr(), inv_scale.data_ptr() if inv_scale is not None else None),
                    'tcow_adamw_clip_step_scaled')
        # precision='fp16': a non-finite gradient norm means the scaled backward overflowed binary16 -- the kernels above skipped the update
        # (clip coefficient -1); lower the module's loss-scale exponent by 4, otherwise let it creep back towards -2.  All on the device.
        ls = getattr(trk, 'ls_log2', None) if trk is not None and getattr(trk, 'precision', None) == 'fp16' and getattr(trk, 'loss_scale', None) == 'dynamic' else None
        if ls is not None and ls.device == self.scratch.device:
            ok = self.scratch[-3] >= 0                       # clip coefficient -1 = skipped
            ls.copy_(torch.minimum(ls + torch.where(ok, 1.0 / 256.0, -4.0), torch.full_like(ls, -2.0)))
        self._steps_pending += 1    # the per-parameter 'step' scalars of torch.optim.AdamW's state layout are brought up to date when somebody looks (_flush_steps)
        torch.autograd.graph.increment_version(live)   # the kernels wrote through raw pointers: make the update visible to version checks
        for cb in self.on_step:
            cb()
        return loss

    def unscale_(self):
        """precision='fp16' with persistent_grads: param.grad holds LOSS-SCALED gradients between backward and step() (torch.cuda.amp.GradScaler's
        convention; step() undoes the scale on the fly).  Call this first when something else must see true gradients in between -- clip_grad_norm_,
        gradient logging, another optimizer: multiplies every live gradient by the pending inverse scale (one pass over the gradients) and clears it.
        A no-op in every other mode."""
        trk = self._tracker
        inv = trk.__dict__.get('pending_inv_scale') if trk is not None else None
        if inv is None:
            return
        seen = set()
        for p in self.params:
            if p.grad is not None and p.grad.data_ptr() not in seen:
                seen.add(p.grad.data_ptr())
                p.grad.mul_(inv)
        trk.__dict__['pending_inv_scale'] = None

    def detach(self):
        """Detach from the module given as module= (its fp16 backwards unscale their gradients themselves again)."""
        trk = self._tracker
        if trk is not None:
            self.unscale_()
            ref = trk.__dict__.get('_optim_ref')
            if ref is not None and ref() is self:
                trk.__dict__.pop('_optim_ref', None)

    def grad_norm(self):
        """Total gradient norm of the last step (device tensor, no sync)."""
        return self.scratch[-2]

    @property
    def skipped_steps(self):
        """Steps skipped so far because their gradient norm was not finite (device f32 scalar, no sync; None before the first step)."""
        return None if self.scratch is None else self.scratch[-1]

    def state_dict(self):
        """torch.optim.AdamW's layout; 'step' = updates actually applied (calls minus skipped steps; one device read here, none per step)."""
        self._flush_steps()
        sd = super().state_dict()
        skipped = float(self.scratch[-1]) if self.scratch is not None else 0.0
        if skipped:
            # torch's Optimizer.state_dict() hands out the LIVE per-parameter dicts: adjust copies, never self.state (the device counter is
            # subtracted again inside the update kernel, so rewriting the live 'step' would double-count every skip after each save)
            sd['state'] = {k: dict(st) for k, st in sd['state'].items()}
            for st in sd['state'].values():
                if 'step' in st:
                    st['step'] = st['step'] - skipped
        return sd

    def load_state_dict(self, sd):
        self._steps_pending = 0           # (the loaded counters replace whatever was pending)
        super().load_state_dict(sd)
        self._key = None; self._live = []; self._seen_grads = []
        self._skip_carry = 0.0            # the loaded 'step' already counts applied updates only
        self.scratch = None
