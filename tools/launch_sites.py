"""Which Python lines launch the torch / runtime kernels of a training step?  One profiled step of the bench configuration (after warm-up) under
torch.profiler with stacks; prints, per call site inside this repo, the device kernels launched from it (library kernels excluded).
usage (GPU box): python3 tools/launch_sites.py > gpurun_out/launch_sites.txt"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py', '--steps', '1', '--warmup', '0']
import bench
from torch.profiler import profile, ProfilerActivity

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
net = Seeker(None, num_total_frames=args.frames, frame_height=args.height, frame_width=args.width, tracker_pretrained=False, causal_attention=1, drop_path_rate=0.1,
             network_depth=args.depth, precision='bf16')
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}, strict=True)
net = net.to(dev).train()
opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net)
net.seeker.persistent_grads = True
net.seeker.grad_hook = ddp.GradSync(1)
pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device=dev, rng=np.random.default_rng(900))

def step():
    mr = pipe.forward_kubric(data)
    loss = pipe.step_losses(data, mr, 0.0)['total_seeker']
    loss.backward()
    opt.step()

for _ in range(3): step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sites = collections.Counter()

class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        flat = list(args) + list((kwargs or {}).values())
        on_gpu = any(torch.is_tensor(a) and a.is_cuda for a in flat) or (torch.is_tensor(out) and out.is_cuda)
        viewish = any(k in name for k in ('view', 'reshape', 'expand', 'permute', 'transpose', 'select', 'slice', 'unsqueeze', 'squeeze', 'detach', 'alias', 'as_strided', 't.default', 'unbind', 'split', '_unsafe_view', 'is_', 'size', 'stride', 'numel', 'empty', 'record_stream'))
        if on_gpu and not viewish:
            site = '?'
            for fr in reversed(traceback.extract_stack()):
                if root in fr.filename and 'launch_sites' not in fr.filename:
                    site = f'{fr.filename.replace(root + "/", "")}:{fr.lineno}'; break
            sites[(site, name.replace('aten.', ''))] += 1
        return out

with Rec():
    step()
torch.cuda.synchronize()
print(f'aten ops touching device tensors in one step (views excluded; each is ~one launch): {sum(sites.values())}')
by_site = collections.defaultdict(list)
for (site, op), c in sites.items(): by_site[site].append((c, op))
for site, lst in sorted(by_site.items(), key=lambda kv: -sum(c for c, _ in kv[1])):
    print(f'{sum(c for c, _ in lst):4d}  {site:40s} ' + ', '.join(f'{op} x{c}' if c > 1 else op for c, op in sorted(lst, reverse=True)))
