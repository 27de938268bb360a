"""Dev: mask-logit / flag / gradient error of precision='bf16x3' (and 'fp32') against the reference goldens."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import build_hip_seeker, golden_inputs, load_golden
from tcow_amd import synth
for name in ['g1_cfg1_d256', 'g2_ca3', 'g2_normemb_nearest', 'g2_stride1_prenorm', 'g11_depth18']:
    meta, g = load_golden(name)
    cfg, sd, rgb, qm = golden_inputs(meta)
    key = 'output_mask' if 'output_mask' in g else 'eval::output_mask'
    for prec in ('fp32', 'bf16x3', 'fp16', 'bf16'):
        try:
            net = build_hip_seeker(cfg, sd, prec).cuda().eval()
        except Exception as e:
            print(name, prec, 'build failed', e); continue
        with torch.no_grad():
            om, fl = net(rgb.cuda(), qm.cuda())
        d = np.abs(om.cpu().numpy() - g[key]).max()
        print(f'{name:22s} {prec:7s} max|d| {d:.2e}  (logit std {g[key].std():.3f})', flush=True)
# gradients at g1
meta, g = load_golden('g1_cfg1_d256')
cfg, sd, rgb, qm = golden_inputs(meta)
for prec in ('fp32', 'bf16x3', 'fp16', 'bf16'):
    net = build_hip_seeker(cfg, sd, prec).cuda().train()
    om, fl = net(rgb.cuda(), qm.cuda())
    Gm = torch.from_numpy(synth._rng(meta['seed'], 'gradprobe_mask').standard_normal(size=tuple(om.shape), dtype=np.float32)).cuda()
    Gf = torch.from_numpy(synth._rng(meta['seed'], 'gradprobe_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32)).cuda()
    ((om * Gm).sum() + (fl * Gf).sum()).backward()
    named = dict(net.named_parameters()); worst = 0
    for k, ref in g.items():
        if k.startswith('grad::'):
            worst = max(worst, np.abs(named[k[6:]].grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30))
    print(f'g1 gradients {prec}: worst rel {worst:.2e}')
