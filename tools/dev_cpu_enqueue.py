"""Dev: host time to ENQUEUE one training step (no synchronisation) vs the GPU time of the step."""
import sys, time, torch
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
t0 = time.perf_counter(); ts = []
for i in range(6):
    a = time.perf_counter(); step(3 + i); ts.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host enqueue per step (ms):', [round(x * 1e3, 1) for x in ts], ' loop', round((t1 - t0) * 1e3 / 6, 1), ' incl. drain', round((t2 - t0) * 1e3 / 6, 1))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(20); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
