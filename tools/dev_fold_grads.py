"""Dev: per-parameter gradient error of the golden g1_cfg1_d256 with the folded temporal projection on / off."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, golden_inputs, build_hip_seeker
from tcow_amd import synth
prec = os.environ.get('PREC', 'bf16')
meta, g = load_golden('g1_cfg1_d256')
cfg, sd, rgb, qm = golden_inputs(meta)
net = build_hip_seeker(cfg, sd, prec).cuda(); net.train(True)
om, fl = net(rgb.cuda(), qm.cuda())
Gm = torch.from_numpy(synth._rng(meta['seed'], 'gradprobe_mask').standard_normal(size=tuple(om.shape), dtype=np.float32)).cuda()
Gf = torch.from_numpy(synth._rng(meta['seed'], 'gradprobe_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32)).cuda()
((om * Gm).sum() + (fl * Gf).sum()).backward()
named = dict(net.named_parameters())
print('fold', os.environ.get('TCOW_FOLD', '1'), 'max|d| out', float(np.abs(om.detach().cpu().numpy() - g['output_mask']).max()))
for k, ref in g.items():
    if k.startswith('grad::'):
        got = named[k[6:]].grad.cpu().numpy()
        print(f'{np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12):9.2e}  {k[6:]}')
for k, n in meta['grad_norms'].items():
    if n is not None and ('temporal_fc' in k or 'temporal_attn.proj' in k):
        print(f'norm {float(named[k].grad.norm()):.5e} ref {n:.5e}  {k}')
