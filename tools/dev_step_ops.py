"""Dev: the torch (ATen) operators of one bench step that launch GPU kernels, grouped by operator and input shapes."""
import sys, collections, torch
sys.path.insert(0, '.')
import numpy as np
from tcow_amd import synth, ddp
from tcow_amd.seeker import Seeker
from tcow_amd.pipeline import SeekerPipeline
from tcow_amd.tcow_loss import default_args
from tcow_amd.optim import FusedAdamWClip
dev = torch.device('cuda', 0)
cfg = synth.seeker_config(causal_attention=1)
net = Seeker(None, num_total_frames=30, frame_height=240, frame_width=320, causal_attention=1, drop_path_rate=0.1, precision='bf16')
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}); net = net.to(dev).train()
opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net); net.seeker.persistent_grads = True
net.seeker.grad_hook = ddp.GradSync(1)
data = synth.to_torch_tree(synth.make_kubric_batch(1, 30, 240, 320, seed=900, n_objects=5), dev, host_keys=synth.HOST_KEYS)
pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device=dev, rng=np.random.default_rng(0))
def step(i):
    mr = pipe.forward_kubric(data); loss = pipe.step_losses(data, mr, i / 1000.0)['total_seeker']; loss.backward(); opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(3); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith('aten::')]
rows.sort(key=lambda e: -e.count)
tot = 0
for e in rows[:80]:
    tot += e.count
    print(f'{e.count:4d} {e.key:28s} dev {e.device_time_total:8.1f} us  {str(e.input_shapes)[:150]}')
print('aten ops with device time:', sum(e.count for e in rows))
