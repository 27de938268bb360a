"""Dev: how much larger than the seed (max |d_mask|) do the 16-bit gradient operands of the backward get over a training run?
(bf16 run, unscaled values: the ratio is what a binary16 loss scale has to leave room for.)"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from tcow_amd import synth, ops
from tcow_amd.seeker import Seeker
from tcow_amd.pipeline import SeekerPipeline
from tcow_amd.tcow_loss import default_args
from tcow_amd.optim import FusedAdamWClip
import tcow_amd.engine as eng
dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cfg = synth.seeker_config(causal_attention=1)
net = Seeker(None, num_total_frames=30, frame_height=240, frame_width=320, causal_attention=1, drop_path_rate=0.1, precision='bf16')
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}); net = net.to(dev).train()
opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net); net.seeker.persistent_grads = True
data = synth.to_torch_tree(synth.make_kubric_batch(1, 30, 240, 320, seed=900, n_objects=5), dev, host_keys=synth.HOST_KEYS)
pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device=dev, rng=np.random.default_rng(0))
cur = {}
orig = ops.gemm_tn_grouped
def spy(mode, problems):
    for dY, X, dW, db in problems:
        cur['dy'] = torch.maximum(cur['dy'], dY.float().abs().amax())
        cur['x'] = torch.maximum(cur['x'], X.float().abs().amax())
    return orig(mode, problems)
eng.ops.gemm_tn_grouped = spy
orig_attn = ops.attn_bwd
def spy_attn(shape, spatial, qkv, out, dout, lse, dqkv):
    r = orig_attn(shape, spatial, qkv, out, dout, lse, dqkv)
    cur['dy'] = torch.maximum(cur['dy'], dqkv.float().abs().amax()); cur['x'] = torch.maximum(cur['x'], qkv.float().abs().amax())
    return r
eng.ops.attn_bwd = spy_attn
orig_bwd = eng.run_backward
def bwd(module, sv, params, d_mask, d_flags):
    cur['seed'] = d_mask.abs().amax()
    return orig_bwd(module, sv, params, d_mask, d_flags)
eng.run_backward = bwd
rows = []
for i in range(steps):
    cur['dy'] = torch.zeros((), device=dev); cur['x'] = torch.zeros((), device=dev)
    mr = pipe.forward_kubric(data); loss = pipe.step_losses(data, mr, i / 1000.0)['total_seeker']; loss.backward(); opt.step()
    rows.append((float(cur['seed']), float(cur['dy']), float(cur['x']), float(loss)))
r = np.array(rows)
ratio = r[:, 1] / r[:, 0]
print(f'{steps} steps: seed max {r[:,0].min():.2e} .. {r[:,0].max():.2e}; max|dY| / seed: first {ratio[0]:.2f}, max {ratio.max():.1f} at step {int(ratio.argmax())}, median {np.median(ratio):.2f}; '
      f'max |activation| {r[:,2].max():.1f}; loss {r[0,3]:.3f} -> {r[-1,3]:.3f}')
print('ratio every 10th step:', np.round(ratio[::10], 1).tolist())
print('max|activation| every 10th step:', np.round(r[::10, 2], 1).tolist())
