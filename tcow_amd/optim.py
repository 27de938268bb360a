"""Fused clip_grad_norm_ + AdamW for the Seeker training step (train.py:99-102, 239-243) on libtcow_hip.

Same arithmetic as torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.AdamW(params, lr).step()
(betas (0.9, 0.999), eps 1e-8, weight_decay 0.01 on every parameter -- the reference builds one parameter group with torch's
defaults), but as three kernel launches over a device-resident chunk table instead of ~250 per-tensor
foreach operations.  Parameters without a gradient are skipped, like torch does."""
import numpy as np
import torch

from . import _lib as L

CHUNK = 65536


class FusedAdamWClip:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=0.3):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.state = {}            # id(p) -> (exp_avg, exp_avg_sq)
        self.step_count = 0
        self._table = None
        self._key = None
        self.scratch = None
        self.on_step = []          # callables run after every step (e.g. QueryMaskTracker.invalidate_weight_cache)
        assert L.lib().tcow_adamw_chunk_bytes() == 40

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _build(self, live):
        rows = []
        for p in live:
            if id(p) not in self.state:
                self.state[id(p)] = (torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32))
            m, v = self.state[id(p)]
            g = p.grad
            if not (p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32):
                raise L.TcowError('FusedAdamWClip needs contiguous f32 parameters and gradients')
            n = p.numel()
            for off in range(0, n, CHUNK):
                rows.append((p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off, min(CHUNK, n - off)))
        tab = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(live[0].device)
        self._table = tab
        self.scratch = torch.empty(len(rows) + 2, dtype=torch.float32, device=live[0].device)

    def step(self):
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        key = tuple((id(p), p.grad.data_ptr()) for p in live)
        if key != self._key:                     # gradient buffers are re-allocated by the backward: rebuild the pointer table
            self._build(live)
            self._key = key
        self.step_count += 1
        L.check(L.lib().tcow_adamw_clip_step(torch.cuda.current_stream().cuda_stream, self._table.data_ptr(), self._table.shape[0], self.lr,
                                             self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, float(self.max_norm or 0.0),
                                             self.scratch.data_ptr()), 'tcow_adamw_clip_step')
        # parameters were updated in place through raw pointers: tell whoever caches derived copies (the Seeker's bf16 operand
        # copies are keyed on this) -- cheaper than bumping 250 autograd version counters with dummy in-place ops
        for cb in self.on_step:
            cb()
    def grad_norm(self):
        """Total gradient norm of the last step (device tensor, no sync)."""
        return self.scratch[-1]

    def state_dict(self):
        return {'step': self.step_count, 'state': [(self.state[id(p)] if id(p) in self.state else None) for p in self.params],
                'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': self.weight_decay, 'max_norm': self.max_norm}

    def load_state_dict(self, sd):
        self.step_count = sd['step']
        for p, st in zip(self.params, sd['state']):
            if st is not None:
                self.state[id(p)] = (st[0].to(p.device), st[1].to(p.device))
        self._key = None
