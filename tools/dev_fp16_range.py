"""Dev: range check of the fp16 mode's backward -- largest / typical magnitudes of the 16-bit gradient operands (scaled by loss_scale)
and of the forward activations in one benchmark training step.  binary16: max 65504, smallest normal 6.1e-5."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from tcow_amd import synth, ops
from tcow_amd.seeker import Seeker
from tcow_amd.pipeline import SeekerPipeline
from tcow_amd.tcow_loss import default_args
from tcow_amd.optim import FusedAdamWClip
dev = torch.device('cuda', 0)
cfg = synth.seeker_config(causal_attention=1)
net = Seeker(None, num_total_frames=30, frame_height=240, frame_width=320, causal_attention=1, drop_path_rate=0.1, precision='fp16')
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}); net = net.to(dev).train()
opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net); net.seeker.persistent_grads = True
data = synth.to_torch_tree(synth.make_kubric_batch(1, 30, 240, 320, seed=900, n_objects=5), dev, host_keys=synth.HOST_KEYS)
pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device=dev, rng=np.random.default_rng(0))
stats = {'dY': [], 'X': []}
orig = ops.gemm_tn_grouped
def spy(mode, problems):
    for dY, X, dW, db in problems:
        a = dY.float().abs(); b = X.float().abs()
        stats['dY'].append((float(a.max()), float(a[a > 0].median()), float((a > 0).float().mean()), float(((a > 0) & (a < 6.1e-5)).float().mean())))
        stats['X'].append((float(b.max()), float(b[b > 0].median())))
    return orig(mode, problems)
ops.gemm_tn_grouped = spy
import tcow_amd.engine as eng; eng.ops.gemm_tn_grouped = spy
orig_bwd = eng.run_backward
seed = {}
def bwd(module, sv, params, d_mask, d_flags):
    seed['amax'] = float(d_mask.abs().max()); seed['med'] = float(d_mask.abs()[d_mask != 0].median())
    return orig_bwd(module, sv, params, d_mask, d_flags)
eng.run_backward = bwd
for i in range(3):
    mr = pipe.forward_kubric(data); loss = pipe.step_losses(data, mr, i / 1000.0)['total_seeker']; loss.backward(); opt.step()
    if i < 2: stats = {'dY': [], 'X': []}
torch.cuda.synchronize()
d = np.array(stats['dY']); x = np.array(stats['X'])
print(f'loss_scale {net.seeker.loss_scale}: gradient operands (84 GEMMs): max |dY| {d[:,0].max():.3e} (smallest per-tensor max {d[:,0].min():.3e}), median |dY| {np.median(d[:,1]):.3e}, '
      f'nonzero fraction {d[:,2].mean():.4f}, subnormal fraction {d[:,3].mean():.4f}')
print(f'seed d_mask (unscaled): max {seed["amax"]:.3e} median {seed["med"]:.3e};  ')
print(f'activation operands: max |X| {x[:,0].max():.3e}, median {np.median(x[:,1]):.3e};  final loss {float(loss):.5f}, grad norm {float(opt.grad_norm()):.4e}')
