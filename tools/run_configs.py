"""BASELINE.json configs[3] and configs[4] on the GPU box: long-clip inference (T=60, 480x640) and the batched eval path
(num_queries=4 x 4 temporal strides = 16 forwards of the config-2 shape, batched), with throughput and mask-IoU agreement
against the CPU oracle on a subset. Writes one JSON line per config."""
import json, sys, time, torch, numpy as np
sys.path.insert(0, '.')
from tcow_amd import synth, flops
from tcow_amd.seeker import Seeker
from tcow_amd.metrics import calculate_metrics_mask_track
from oracle import seeker_oracle as so

def build(cfg, precision='bf16'):
    net = Seeker(None, num_total_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'], frame_width=cfg['frame_width'], causal_attention=1,
                 drop_path_rate=0.0, precision=precision)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}, strict=True)
    return net.cuda().eval()

def timeit(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

# ---- config 4: long clip, inference only
cfg4 = synth.seeker_config(num_total_frames=60, frame_height=480, frame_width=640, causal_attention=1)
net = build(cfg4)
clip = synth.make_clip(1, 60, 480, 640, seed=900)
rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
with torch.no_grad():
    t = timeit(lambda: net(rgb, qm))
    out, _ = net(rgb, qm)
fl = flops.seeker_forward_flops(1, 60, 30, 40, 768, 12, 12)
print(json.dumps(dict(config='configs[3] long clip T=60 480x640 inference fwd (bf16, 1 GPU)', s_per_forward=t, clips_per_s=1 / t, forward_tflop=fl['total'] / 1e12,
                      achieved_tflops=fl['total'] / t / 1e12, attention_tflop=fl['attention'] / 1e12, finite=bool(torch.isfinite(out).all()),
                      peak_mem_gb=torch.cuda.max_memory_allocated() / 1e9)), flush=True)
del net, rgb, qm, out; torch.cuda.empty_cache()

# ---- config 5: batched eval: 4 queries x 4 strides = 16 forwards of the T=30 240x320 shape
cfg2 = synth.seeker_config(causal_attention=1)
net = build(cfg2)
kb = synth.to_torch_tree(synth.make_kubric_batch(1, 120, 240, 320, seed=900, n_objects=5))   # one long synthetic video, sub-sampled with strides 1..4
rgb_full = kb['kubric_retval']['pv_rgb_tf']; segm = kb['kubric_retval']['pv_segm_tf']; div = kb['kubric_retval']['pv_div_segm_tf']
rgbs, qms, tgts = [], [], []
for stride in (1, 2, 3, 4):                                             # data_utils.py:301-342 usage modes: frame_start 0, stride s
    idx = torch.arange(30) * stride
    for q in range(4):
        rgbs.append(rgb_full[0, :, idx]); m = torch.zeros(1, 30, 240, 320); m[0, 0] = (segm[0, 0, 0] == q + 1).float(); qms.append(m)
        t3 = torch.zeros(3, 30, 240, 320); t3[0] = div[0, q][idx].float(); tgts.append(t3)
rgb = torch.stack(rgbs).cuda(); qm = torch.stack(qms).cuda(); tgt = torch.stack(tgts).cuda()
with torch.no_grad():
    t = timeit(lambda: net(rgb, qm), n=3, w=1)
    out, _ = net(rgb, qm)
m_hip = calculate_metrics_mask_track(out, tgt, plugin=True)
# IoU agreement with the fp32 CPU oracle on the first 2 of the 16 forwards
sd = so.to_torch_state_dict(synth.make_state_dict(cfg2, 900))
torch.set_num_threads(64)
with torch.no_grad():
    ref, _ = so.seeker_forward(sd, cfg2, rgb[:2].cpu(), qm[:2].cpu())
m_ref = calculate_metrics_mask_track(ref, tgt[:2].cpu(), plugin=True); m_sub = calculate_metrics_mask_track(out[:2].cpu(), tgt[:2].cpu(), plugin=True)
agree = float(((out[:2].cpu() > 0) == (ref > 0)).float().mean())
print(json.dumps(dict(config='configs[4] batched eval: 4 queries x 4 strides = 16 forwards (T=30 240x320, bf16, 1 GPU)', s_per_batch=t, forwards_per_s=16 / t,
                      mean_snitch_iou_hip_all16=float(m_hip['mean_snitch_iou']), mean_snitch_iou_hip_first2=float(m_sub['mean_snitch_iou']),
                      mean_snitch_iou_oracle_first2=float(m_ref['mean_snitch_iou']), binary_mask_agreement_first2=agree,
                      max_abs_logit_diff_first2=float((out[:2].cpu() - ref).abs().max()), note='random-init weights: IoU values are low by construction; agreement is the parity figure')), flush=True)
