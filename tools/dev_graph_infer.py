"""Dev: single-clip inference forward (BASELINE configs[1] geometry, one query) eager vs replayed as a hipGraph (torch.cuda.CUDAGraph)."""
import sys, time, torch
sys.path.insert(0, '.')
from tcow_amd import synth
from tcow_amd.seeker import Seeker
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
cfg = synth.seeker_config(causal_attention=1)
net = Seeker(None, num_total_frames=30, frame_height=240, frame_width=320, causal_attention=1, drop_path_rate=0.0, precision=prec)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}); net = net.cuda().eval()
clip = synth.make_clip(1, 30, 240, 320, seed=900)
rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
def wall(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    ref, ref_f = net(rgb, qm)
    t_eager = wall(lambda: net(rgb, qm))
    # host-only enqueue time
    torch.cuda.synchronize(); t0 = time.perf_counter(); net(rgb, qm); t_enq = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): net(rgb, qm)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, fl = net(rgb, qm)
    t_graph = wall(g.replay)
    g.replay(); torch.cuda.synchronize()
    print(f'{prec}: eager {t_eager:.2f} ms per forward (host enqueue alone {t_enq:.2f} ms), graph replay {t_graph:.2f} ms; identical outputs: {bool(torch.equal(out, ref) and torch.equal(fl, ref_f))}')
    rgb.mul_(0.5); g.replay(); torch.cuda.synchronize()
    chk, _ = net(rgb, qm)
    print('replay follows the static input buffers:', bool(torch.equal(out, chk)))
