"""Dev: forward error of a golden with the folded temporal projection on / off."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, golden_inputs, build_hip_seeker
for name in os.environ.get('GOLD', 'g11_depth24,g11_depth18,g1_cfg1_d256').split(','):
    for prec in ('fp16', 'bf16'):
        meta, g = load_golden(name)
        cfg, sd, rgb, qm = golden_inputs(meta)
        net = build_hip_seeker(cfg, sd, prec).cuda(); net.train(False)
        with torch.no_grad():
            om, fl = net(rgb.cuda(), qm.cuda())
        print(f"fold {os.environ.get('TCOW_FOLD', '1')} {name} {prec}: mask max|d| {float(np.abs(om.cpu().numpy() - g['output_mask']).max()):.3e}  flags {float(np.abs(fl.cpu().numpy() - g['output_flags']).max()):.3e}  (logit std {g['output_mask'].std():.3f})", flush=True)
