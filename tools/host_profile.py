"""Host-side cost of enqueueing one bf16 training step (cProfile around single steps into an empty stream): where the Python / ctypes time goes.
usage (through gpurun): python3 tools/host_profile.py [n_rows]"""
import cProfile, io, pstats, sys, time
import torch
sys.path.insert(0, '.')
import bench
sys.argv = ['bench.py', '--steps', '1', '--warmup', '1']
args = bench.parse()
from tcow_amd import ddp, synth
from tcow_amd.seeker import Seeker
from tcow_amd.optim import FusedAdamWClip
from tcow_amd.pipeline import SeekerPipeline
from tcow_amd.tcow_loss import default_args
import numpy as np
dev = torch.device('cuda', 0)
cfg = synth.seeker_config(num_total_frames=args.frames, frame_height=args.height, frame_width=args.width, depth=args.depth, causal_attention=1)
data = synth.to_torch_tree(synth.make_kubric_batch(1, args.frames, args.height, args.width, seed=900, n_objects=5), dev, host_keys=synth.HOST_KEYS)
net = Seeker(None, num_total_frames=args.frames, frame_height=args.height, frame_width=args.width, tracker_pretrained=False, causal_attention=1,
             drop_path_rate=0.1, network_depth=args.depth, precision='bf16')
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}, strict=True)
net = net.to(dev).train()
opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net)
net.seeker.persistent_grads = True
net.seeker.grad_hook = ddp.GradSync(1)
pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device=dev, rng=np.random.default_rng(900))
def step():
    mr = pipe.forward_kubric(data)
    loss = pipe.step_losses(data, mr, 0.0)['total_seeker']
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); ts.append((time.perf_counter() - t0) * 1e3)
print('host enqueue per step (ms):', [round(t, 2) for t in ts])
pr = cProfile.Profile()
for _ in range(5):
    torch.cuda.synchronize(); pr.enable(); step(); pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30)
print(s.getvalue()[:6000])
