"""Dev: measured bf16-mode deviation of every forward golden, as a fraction of the golden's logit std (to set the test tolerance)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import build_hip_seeker, golden_inputs, load_golden
from test_oracle_golden import summarise
for name in ['g1_cfg1_d256', 'g2_ca0', 'g2_ca2', 'g2_ca3', 'g2_cam1', 'g2_normemb_nearest', 'g2_stride1_prenorm', 'g2_stride2', 'g11_depth18', 'g11_depth24', 'g3_mid_T8_96x128', 'g4_cfg2_T30_240x320', 'g4b_cfg2_seed2', 'g8_cfg3_long']:
    meta, g = load_golden(name)
    cfg, sd, rgb, qm = golden_inputs(meta)
    for prec in ('bf16', 'fp16'):
        net = build_hip_seeker(cfg, sd, prec).cuda().eval()
        with torch.no_grad():
            om, fl = net(rgb.cuda(), qm.cuda())
        if 'output_mask' in g:
            d = np.abs(om.cpu().numpy() - g['output_mask']).max(); std = float(np.std(g['output_mask']))
        else:
            pooled, _, _ = summarise(om.cpu())
            if 'pooled' in g: d = np.abs(pooled - g['pooled']).max()
            else: d = np.abs(pooled.reshape(60, 3, 120, 160)[g['frames']] - g['pooled_frames']).max()
            std = float(g['logit_std'])
        df = np.abs(fl.cpu().numpy() - g['output_flags']).max(); fstd = float(np.std(g['output_flags']))
        print(f'{name:24s} {prec}: mask max|d| {d:.3e} = {d / std:.4f} x std ({std:.4f});  flags max|d| {df:.3e} = {df / max(fstd, 1e-9):.4f} x std ({fstd:.4f})', flush=True)
        del net
