"""Dev: soak test -- many training steps on the bench workload, twice with the same seeds; the loss curves must be finite,
decreasing and IDENTICAL between the two runs (every kernel on the path is deterministic: no atomics on floating-point data),
which is how a race in the hand-synchronised kernels would show up."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from tcow_amd import synth
from tcow_amd.seeker import Seeker
from tcow_amd.pipeline import SeekerPipeline
from tcow_amd.tcow_loss import default_args
from tcow_amd.optim import FusedAdamWClip
dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
def run():
    torch.manual_seed(0)
    cfg = synth.seeker_config(causal_attention=1)
    net = Seeker(None, num_total_frames=30, frame_height=240, frame_width=320, causal_attention=1, drop_path_rate=0.1, precision=precision)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}); net = net.to(dev).train()
    opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net); net.seeker.persistent_grads = True
    data = synth.to_torch_tree(synth.make_kubric_batch(1, 30, 240, 320, seed=900, n_objects=5), dev, host_keys=synth.HOST_KEYS)
    pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device=dev, rng=np.random.default_rng(0))
    losses = []
    for i in range(steps):
        mr = pipe.forward_kubric(data); loss = pipe.step_losses(data, mr, i / 1000.0)['total_seeker']; loss.backward(); opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    if precision == 'fp16':
        print('fp16 loss-scale exponent at the end:', float(net.seeker.ls_log2), ' skipped steps:', int(opt.skipped_steps))
    return torch.stack(losses).cpu().numpy()
a = run(); b = run()
print(precision, 'finite:', bool(np.isfinite(a).all()), ' first/last loss: %.5f %.5f' % (a[0], a[-1]), ' identical runs:', bool((a == b).all()), ' max |a-b|: %.3e' % float(np.abs(a - b).max()))
print('losses:', np.round(a[::max(1, steps // 12)], 4).tolist())
assert np.isfinite(a).all() and a[-1] < a[0]
