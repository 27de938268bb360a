"""Dev: wall-clock of the sections of one training step (with syncs between sections)."""
import sys, time, torch
sys.path.insert(0, '.')
import numpy as np
from tcow_amd import synth, ddp
from tcow_amd.seeker import Seeker
from tcow_amd.pipeline import SeekerPipeline
from tcow_amd.tcow_loss import default_args
dev = torch.device('cuda', 0)
cfg = synth.seeker_config(causal_attention=1)
net = Seeker(None, num_total_frames=30, frame_height=240, frame_width=320, causal_attention=1, drop_path_rate=0.1, precision='bf16')
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}); net = net.to(dev).train()
params = list(net.parameters()); opt = torch.optim.AdamW(params, lr=1e-4, fused=True)
data = synth.to_torch_tree(synth.make_kubric_batch(1, 30, 240, 320, seed=900), dev)
pipe = SeekerPipeline(net, num_queries=3, train_args=default_args(), phase='train', device=dev, rng=np.random.default_rng(0))
def S(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t0 = S(); opt.zero_grad(set_to_none=True)
    mr = pipe.forward_kubric(data); t1 = S()
    loss = pipe.step_losses(data, mr, 0.0)['total_seeker']; t2 = S()
    loss.backward(); t3 = S()
    torch.nn.utils.clip_grad_norm_(params, 0.3); opt.step(); t4 = S()
    print(f'it{it}: fwd(pipeline+model) {1e3*(t1-t0):.1f} ms | loss {1e3*(t2-t1):.1f} | backward {1e3*(t3-t2):.1f} | clip+adamw {1e3*(t4-t3):.1f} | total {1e3*(t4-t0):.1f}', flush=True)
# model-only forward / backward
rgb = data['kubric_retval']['pv_rgb_tf'].expand(3, -1, -1, -1, -1).contiguous(); qm = mr['seeker_query_mask'].reshape(3, 1, 30, 240, 320)
for it in range(3):
    t0 = S(); om, _ = net(rgb, qm); t1 = S(); om.mean().backward(); t2 = S()
    print(f'model only: fwd {1e3*(t1-t0):.1f} ms bwd {1e3*(t2-t1):.1f} ms')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); om, _ = net(rgb, qm); om.mean().backward(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
