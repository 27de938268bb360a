"""Dev: which Python lines issue the small fill / copy kernels of one bench step (torch profiler with stacks)."""
import sys, collections, torch
sys.path.insert(0, '.')
import numpy as np
from tcow_amd import synth
from tcow_amd.seeker import Seeker
from tcow_amd.pipeline import SeekerPipeline
from tcow_amd.tcow_loss import default_args
from tcow_amd.optim import FusedAdamWClip
dev = torch.device('cuda', 0)
cfg = synth.seeker_config(causal_attention=1)
net = Seeker(None, num_total_frames=30, frame_height=240, frame_width=320, causal_attention=1, drop_path_rate=0.1, precision='bf16')
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}); net = net.to(dev).train()
opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net); net.seeker.persistent_grads = True
data = synth.to_torch_tree(synth.make_kubric_batch(1, 30, 240, 320, seed=900, n_objects=5), dev, host_keys=synth.HOST_KEYS)
pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device=dev, rng=np.random.default_rng(0))
def step(i):
    mr = pipe.forward_kubric(data); loss = pipe.step_losses(data, mr, i / 1000.0)['total_seeker']; loss.backward(); opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3); torch.cuda.synchronize()
# which of our source lines launch the tiny fill / copy kernels
import re
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::zeros', 'aten::full', 'aten::ones', 'aten::sum', 'aten::any', 'aten::index', 'aten::eq', 'aten::ne', 'aten::gt', 'aten::lt', 'aten::mul', 'aten::add', 'aten::where', 'aten::cat', 'aten::stack', 'aten::clone', 'aten::contiguous', 'aten::index_put_', 'aten::_to_copy', 'aten::add_', 'aten::mul_', 'aten::div', 'aten::rand', 'aten::floor_', 'aten::expand'):
        fr = [f for f in (ev.stack or []) if '/tcow_amd/' in f or 'bench' in f]
        cnt[(ev.name, fr[0] if fr else (ev.stack[0] if ev.stack else '?'))] += 1
for (n, f), c in cnt.most_common(70):
    print(f'{c:4d} {n:14s} {f[:130]}')
print('---- fill / zero launches by innermost frames')
fc = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_'):
        fc[(ev.name, tuple((ev.stack or ['?'])[:5]))] += 1
for (n, st), c in fc.most_common(25):
    print(f'{c:4d} {n}'); [print('        ', f[:150]) for f in st]
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
for e in rows[:45]:
    print(f'{e.count:5d}  {e.key[:60]:60s} cpu_total {e.cpu_time_total/1e3:8.2f} ms')
