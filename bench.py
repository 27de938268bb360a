#!/usr/bin/env python3
"""Benchmark of the TCOW Seeker training step on MI355X (contract: see the build prompt / DESIGN.md section 4).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" on a rank = one synthetic clip (T=30, 240x320, BASELINE.json configs[1]) x num_queries=3 query forwards
(batched), the mask loss, one backward, gradient all-reduce (RCCL, overlapped with backward), grad-clip 0.3 and
an AdamW step -- mirroring pipeline.py:134-174 + train.py:89-102.  value = world_size * clips_per_rank / step time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp16', 'fp32', 'bf16x3'])
    ap.add_argument('--queries', type=int, default=3)
    ap.add_argument('--frames', type=int, default=30)
    ap.add_argument('--height', type=int, default=240)
    ap.add_argument('--width', type=int, default=320)
    ap.add_argument('--depth', type=int, default=12)
    ap.add_argument('--grad-dtype', default='f32', choices=['f32', 'bf16'], help='dtype of the all-reduced gradient buckets (N > 1)')
    ap.add_argument('--launch-selftest', action='store_true', help='only start the ranks, all-reduce one number and print the rank count (CPU-runnable check of the N > 1 launch path)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the fp32 parity-mode timing and the live max|d| check against the oracle')
    ap.add_argument('--no-config-legs', action='store_true', help='skip the configs[3] long-clip forward and configs[4] batched-eval legs')
    ap.add_argument('--parity-steps', type=int, default=10, help='timed steps of the bf16x3 / fp32 legs (the fp16 leg runs --steps)')
    ap.add_argument('--cpu-budget-s', type=float, default=25.0)
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of every NT-GEMM launch (the dominant kernel), recorded by the library itself on the launch stream
    (tcow_prof_gemm_begin/_end: two hipEventRecord per launch, no Python in the loop).  Used in a SECOND pass of --steps steps right after the
    headline loop: the headline and its own instrumentation do not share a loop."""

    def __init__(self, fmt='bf16'):
        from tcow_amd import _lib
        self.L = _lib
        self.fmt = fmt             # which build of the library launches the GEMMs of this run ('fp16': libtcow_hip_fp16.so)
        self.result = None

    def begin(self, max_launches):
        self.L.check(self.L.lib(self.fmt).tcow_prof_gemm_begin(int(max_launches)), 'tcow_prof_gemm_begin')

    def end(self):
        import ctypes
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
        self.L.check(self.L.lib(self.fmt).tcow_prof_gemm_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), 'tcow_prof_gemm_end')
        if n.value:
            self.result = dict(launches=n.value, avg_us=ms.value * 1e3 / n.value, flops_per_launch=fl.value / n.value,
                               tflops=fl.value / (ms.value * 1e-3) / 1e12)

    def summary(self):
        return self.result


class AttnTimer:
    """The same for the four attention entry points (tcow_prof_attn_begin/_end): the north-star's "achieved fraction of the attention roofline" -- per class
    the average HIP-event time of a call, its algorithmic TFLOP/s against the dense MFMA peak and its algorithmic bytes / time against 8 TB/s of HBM
    (spatial attention at configs[1] has 151 FLOP/B: left of the ridge, so the HBM fraction is the one that can approach 1)."""
    NAMES = ('spatial_fwd', 'spatial_bwd', 'temporal_fwd', 'temporal_bwd')

    def __init__(self, fmt='bf16'):
        from tcow_amd import _lib
        self.L = _lib; self.fmt = fmt; self.result = None

    def begin(self, max_launches):
        self.L.check(self.L.lib(self.fmt).tcow_prof_attn_begin(int(max_launches)), 'tcow_prof_attn_begin')

    def end(self, steps, peak_tflops):
        import ctypes
        ms, fl, by = (ctypes.c_double * 4)(), (ctypes.c_double * 4)(), (ctypes.c_double * 4)()
        n = (ctypes.c_long * 4)()
        self.L.check(self.L.lib(self.fmt).tcow_prof_attn_end(ms, fl, by, n), 'tcow_prof_attn_end')
        out = {}
        for c, name in enumerate(self.NAMES):
            if n[c]:
                t = ms[c] * 1e-3
                out[name] = dict(us=ms[c] * 1e3 / n[c], launches_per_step=n[c] / steps, tflops=fl[c] / t / 1e12, mfma_frac=fl[c] / t / 1e12 / peak_tflops,
                                 gbps=by[c] / t / 1e9, hbm_frac=by[c] / t / 1e9 / PEAK_HBM_GBS, flops_per_launch=fl[c] / n[c], bytes_per_launch=by[c] / n[c])
        if out:
            tot_ms = sum(ms[c] for c in range(4))
            out['ms_per_step'] = tot_ms / steps
            out['note'] = ('HIP events on the launch stream around every attention call of an untimed pass of --steps steps; FLOPs = 4 L^2 d per (sequence, head) '
                           'forward, x 2.5 backward; bytes = q, k, v, o (+ dO, dq, dk, dv; + O for spatial) once each; peaks: dense MFMA of the dtype, 8 TB/s HBM')
        self.result = out or None


def cpu_baseline(cfg, budget_s):
    """Times the oracle (our CPU port of the reference path, oracle/seeker_oracle.py) on this host's cores, as BASELINE.md section 3 plans it:
    1 warm-up + 3 timed query forwards and 1 timed forward + backward at the benchmark geometry (~60 s on the GPU box's 64 threads; a slower
    host stops after the forwards once `budget_s` x 4 is used up and extrapolates backward = 2 x forward, saying so).  Returns (the oracle's
    mask logits of the LAST forward -- the parity leg compares the HIP outputs with them --, the cpu_baseline record)."""
    from oracle import seeker_oracle as so
    from tcow_amd import synth
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    sd = so.to_torch_state_dict(synth.make_state_dict(cfg, 900))
    T, H, W = cfg['num_total_frames'], cfg['frame_height'], cfg['frame_width']
    clip = synth.make_clip(1, T, H, W, seed=900)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
    t_begin = time.time()
    fwd = []
    with torch.no_grad():
        t0 = time.time(); ref_mask, _ = so.seeker_forward(sd, cfg, rgb, qm); t_warm = time.time() - t0
        for _ in range(3):
            if fwd and time.time() - t_begin > 2.0 * budget_s:
                break
            t0 = time.time(); ref_mask, _ = so.seeker_forward(sd, cfg, rgb, qm); fwd.append(time.time() - t0)
    t_fwd = sum(fwd) / len(fwd)
    sample = f'warm-up forward {t_warm:.2f} s; {len(fwd)} timed query forwards (B=1) {", ".join(f"{t:.2f}" for t in fwd)} s (mean {t_fwd:.2f} s)'
    t_fb = None
    if time.time() - t_begin + 3.0 * t_fwd < 4.0 * budget_s:
        for v in sd.values():
            v.requires_grad_(True)
        t0 = time.time()
        om, fl = so.seeker_forward(sd, cfg, rgb, qm)
        (om.square().mean() + fl.square().mean()).backward()
        t_fb = time.time() - t0
        sample += f'; 1 timed query forward+backward {t_fb:.2f} s'
        for v in sd.values():
            v.requires_grad_(False); v.grad = None
    per_query = t_fb if t_fb is not None else 3.0 * t_fwd     # bwd ~ 2x fwd when not measured
    nq = 3
    return ref_mask, dict(value=1.0 / (nq * per_query), unit='clips/s', cores=threads, kind='port', s_per_forward=t_fwd, s_per_forward_backward=t_fb,
                sample=sample + f'; clips/s = 1 / ({nq} queries x {"measured forward+backward" if t_fb else "3 x forward"} time); optimizer step not included')


def pmc_traffic():
    """roofline.traffic: HBM bytes per launch of the dominant kernel family from the rocprofv3 --pmc passes of this command (FETCH_SIZE
    with the gfx950 x2 correction + WRITE_SIZE, tools/pmc_bench.sh -> profiles/rNN_pmc_gemm_nt.json).  Counters cannot be read
    inside the timed run, so the committed measurement is used ONLY while it still describes the kernel source that is running: the JSON
    records the sha256 of gemm_bf16.hip + gemm_nt_common.h it was taken with; any difference reports null rather than a stale number."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    srcs = [os.path.join(root, 'tcow_amd', 'csrc', f) for f in ('gemm_bf16.hip', 'gemm_nt_common.h')]      # the NT kernels: loop + shared epilogue
    sha = hashlib.sha256(b''.join(open(f, 'rb').read() for f in srcs)).hexdigest() if all(os.path.exists(f) for f in srcs) else None
    for path in sorted(glob.glob(os.path.join(root, 'profiles', 'r*_pmc_gemm_nt.json')), reverse=True):     # newest round first
        rec = json.load(open(path))
        if sha is not None and rec.get('gemm_bf16_sha256') == sha:
            return rec.get('traffic_bytes_per_launch')
    return None


def trained_scale_case(cfg, so, synth, args):
    """Inputs + oracle logits of the trained-magnitude parity case (one more oracle forward, ~7 s on the GPU box's host), or None when the
    run is not at the BASELINE configs[1] geometry the fixture's head gain was calibrated for."""
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'g16_cfg2_trained_scale.npz')
    if not os.path.exists(path) or (args.frames, args.height, args.width, args.depth) != (30, 240, 320, 12):
        return None
    meta = json.loads(bytes(np.load(path)['meta']).decode())
    sd = synth.trained_scale_state_dict(synth.make_state_dict(cfg, meta['seed']), meta['qk_gain'], meta['w_gain'], meta['head_gain'])
    clip = synth.make_clip(1, args.frames, args.height, args.width, seed=meta['seed'])
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
    with torch.no_grad():
        ref, _ = so.seeker_forward(so.to_torch_state_dict(sd), cfg, rgb, qm)
    return dict(sd=sd, rgb=rgb, qm=qm, ref=ref)


def rccl_knobs():
    """The RCCL / overlap settings an N > 1 run was taken with (environment as seen by rank 0): what the `ddp` block's numbers depend on."""
    keys = ('NCCL_MAX_NCHANNELS', 'NCCL_MIN_NCHANNELS', 'NCCL_ALGO', 'NCCL_PROTO', 'NCCL_BUFFSIZE', 'RCCL_MSCCL_ENABLE', 'TCOW_DDP_OVERLAP', 'TCOW_DDP_AVG',
            'TCOW_DIST_BACKEND', 'HSA_ENABLE_IPC_MODE_LEGACY')
    return {k: os.environ.get(k) for k in keys if os.environ.get(k) is not None}


def config_legs(args, dev):
    """BASELINE.json configs[3] and configs[4] in the bench line (a few seconds each; the same computations as tools/run_configs.py):
      config3_long_forward : T=60 480x640 inference forward (S = 1201 spatial x 60 temporal tokens), ms, achieved TFLOP/s, and the share of
                             the forward spent in the attention kernels (their launches timed alone with HIP events, x 12 blocks);
      config4_batched_eval : 4 queries x 4 temporal strides = 16 eval forwards of the configs[1] shape as ONE batched call: forwards/s, and
                             the binary-mask agreement + max|d| of the first forward against the CPU oracle."""
    from oracle import seeker_oracle as so
    from tcow_amd import flops, ops, synth
    from tcow_amd.seeker import Seeker
    out = {}

    def build(cfg):
        net = Seeker(None, num_total_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'], frame_width=cfg['frame_width'], causal_attention=1,
                     drop_path_rate=0.0, precision=args.precision)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}, strict=True)
        return net.to(dev).eval()

    def timeit(f, n, w):
        for _ in range(w):
            f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

    # ---- configs[3]
    cfg3 = synth.seeker_config(num_total_frames=60, frame_height=480, frame_width=640, causal_attention=1)
    net = build(cfg3)
    clip = synth.make_clip(1, 60, 480, 640, seed=900)
    rgb = torch.from_numpy(clip['rgb']).to(dev); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).to(dev)
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        t = timeit(lambda: net(rgb, qm), 5, 2)
        om, _ = net(rgb, qm)
    fl = flops.seeker_forward_flops(1, 60, 30, 40, 768, 12, 12)
    g = net.seeker.geometry(1)
    mode = net.seeker.mode
    qkv = torch.randn(g['M'], 3 * g['D'], device=dev).to(ops.tdtype(mode)); o = torch.empty(g['M'], g['D'], device=dev, dtype=ops.tdtype(mode))
    shp = ops.attn_shape(mode, 1, g['T'], g['S'], g['D'], g['heads'], 1)
    t_sp = timeit(lambda: ops.attn_fwd(shp, True, qkv, o, None), 10, 3); t_tm = timeit(lambda: ops.attn_fwd(shp, False, qkv, o, None), 10, 3)
    out['config3_long_forward'] = dict(workload='configs[3]: T=60 480x640 inference forward, 1 query, causal_attention=1', ms=t * 1e3, tflops=fl['total'] / t / 1e12,
                                       forward_tflop=fl['total'] / 1e12, attn_frac=12 * (t_sp + t_tm) / t, attn_spatial_us=t_sp * 1e6, attn_temporal_us=t_tm * 1e6,
                                       attn_tflops=fl['attention'] / (12 * (t_sp + t_tm)) / 1e12, finite=bool(torch.isfinite(om).all()),
                                       peak_mem_gb=torch.cuda.max_memory_allocated() / 1e9)
    del net, rgb, qm, om, qkv, o
    torch.cuda.empty_cache()

    # ---- configs[4]
    cfg1 = synth.seeker_config(causal_attention=1)
    net = build(cfg1)
    kb = synth.to_torch_tree(synth.make_kubric_batch(1, 120, 240, 320, seed=900, n_objects=5))   # one long synthetic video, sub-sampled with strides 1..4
    rgb_full = kb['kubric_retval']['pv_rgb_tf']; segm = kb['kubric_retval']['pv_segm_tf']
    rgbs, qms = [], []
    for stride in (1, 2, 3, 4):                                             # data_utils.py:301-342 usage modes: frame_start 0, stride s
        idx = torch.arange(30) * stride
        for q in range(4):
            rgbs.append(rgb_full[0, :, idx]); m = torch.zeros(1, 30, 240, 320); m[0, 0] = (segm[0, 0, 0] == q + 1).float(); qms.append(m)
    rgb = torch.stack(rgbs).to(dev); qm = torch.stack(qms).to(dev)
    with torch.no_grad():
        t = timeit(lambda: net(rgb, qm), 3, 1)
        om, _ = net(rgb, qm)
        ref, _ = so.seeker_forward(so.to_torch_state_dict(synth.make_state_dict(cfg1, 900)), cfg1, rgb[:1].cpu(), qm[:1].cpu())
    o1 = om[:1].cpu()
    out['config4_batched_eval'] = dict(workload='configs[4]: 4 queries x 4 temporal strides = 16 eval forwards (T=30 240x320) as one batched call', ms=t * 1e3,
                                       forwards_per_s=16 / t, mask_agreement=float(((o1 > 0) == (ref > 0)).float().mean()), max_abs_d=float((o1 - ref).abs().max()),
                                       against='oracle (CPU), first of the 16 forwards', finite=bool(torch.isfinite(om).all()))
    del net
    torch.cuda.empty_cache()
    return out


def parity_leg(make_trainer, bf16_net, ref_mask, args, bf16_rate=None):
    """The precision story in the bench line (north_star: mask-logit max|d| < 1e-3 vs the reference):
      * fp16_mode: the same training step with precision='fp16' (the benchmarked kernels compiled for IEEE binary16 storage: three more
        significand bits at the same speed) -- the fastest mode inside the 1e-3 bound;
      * fp32_parity_mode: the same training step with precision='fp32' (exact-f32 MFMA / FMA kernels) timed over --parity-steps steps;
      * bf16x3_mode: the same with precision='bf16x3' (f32 storage, GEMM products as three bf16 MFMAs on hi / lo operand splits) -- ~1e-5;
      * max_abs_d: eval forward of all four modes against the oracle's logits computed in this run, the MAXIMUM over 3 clips x 2 weight
        seeds (clip seeds 900 / 901 / 902 with weight seed 900 -- the first is the cpu_baseline forward --, and the same clips with weight
        seed 901: five more oracle forwards, ~35 s of CPU); `max_abs_d_cases` lists the six values of every mode."""
    from oracle import seeker_oracle as so
    from tcow_amd import synth
    out = {}
    cases = []                 # (weight seed, clip seed, rgb, qm, reference logits)
    sds = {}
    if ref_mask is not None:
        cfg = synth.seeker_config(num_total_frames=args.frames, frame_height=args.height, frame_width=args.width, depth=args.depth, causal_attention=1)
        for wseed in (900, 901):
            np_sd = synth.make_state_dict(cfg, wseed)
            sds[wseed] = {k: torch.from_numpy(v).cuda() for k, v in np_sd.items()}     # the weights the oracle runs with (the trainers have stepped)
            osd = so.to_torch_state_dict(np_sd)
            for cseed in (900, 901, 902):
                clip = synth.make_clip(1, args.frames, args.height, args.width, seed=cseed)
                rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
                if wseed == 900 and cseed == 900:
                    ref = ref_mask
                else:
                    with torch.no_grad():
                        ref, _ = so.seeker_forward(osd, cfg, rgb, qm)
                cases.append((wseed, cseed, rgb.cuda(), qm.cuda(), ref))
        # parity at TRAINED-checkpoint magnitudes (tests/golden/g16_cfg2_trained_scale.npz: q / k rows x3, block weights x1.5, mask head scaled
        # to logit std 5): 16-bit error is relative, so the figure that survives a real checkpoint is max|d| / std and the binary-mask agreement
        ts = trained_scale_case(cfg, so, synth, args)
        if ts is not None:
            sds['ts'] = {k: torch.from_numpy(v).cuda() for k, v in ts['sd'].items()}
            cases.append(('ts', 900, ts['rgb'].cuda(), ts['qm'].cuda(), ts['ref']))

    def max_abs_d(name, net):
        if ref_mask is None:
            return
        trained = {k: v.detach().clone() for k, v in net.state_dict().items()}
        net.eval()
        vals, loaded = [], None
        for (wseed, cseed, rgb, qm, ref) in cases:
            if loaded != wseed:
                net.load_state_dict(sds[wseed], strict=True); net.seeker.invalidate_weight_cache(); loaded = wseed
            with torch.no_grad():
                om, _ = net(rgb, qm)
            om = om.cpu()
            if wseed == 'ts':
                t = out.setdefault('trained_scale', dict(logit_std=float(ref.std()), weights='synth.trained_scale_state_dict(seed 900, qk_gain 3, w_gain 1.5, head scaled to logit std 5)',
                                                         max_abs_d={}, max_abs_d_rel={}, mask_agreement={}))
                d = float((om - ref).abs().max())
                t['max_abs_d'][name] = d; t['max_abs_d_rel'][name] = d / t['logit_std']; t['mask_agreement'][name] = float(((om > 0) == (ref > 0)).float().mean())
                continue
            vals.append(float((om - ref).abs().max()))
        out.setdefault('max_abs_d', {})[name] = max(vals)
        out.setdefault('max_abs_d_cases', {})[name] = vals
        out.setdefault('max_abs_d_rel', {})[name] = max(vals) / float(ref_mask.std())      # relative to the logits' std (0.154 at these weights)
        net.load_state_dict(trained, strict=True); net.seeker.invalidate_weight_cache(); net.train()

    max_abs_d('bf16', bf16_net)
    for key, precision, dtype in (('fp16_mode', 'fp16', 'f16 (the bf16 kernels built for binary16 storage, power-of-two loss scale chosen on the device)'),
                                  ('bf16x3_mode', 'bf16x3', 'f32 storage, GEMM and attention products as three bf16 MFMAs on hi / lo splits'), ('fp32_parity_mode', 'fp32', 'f32')):
        net, step = make_trainer(precision)
        nsteps = args.steps if precision == 'fp16' else max(args.parity_steps, 10)   # fp16 (the at-parity throughput): the headline's step count
        for _ in range(max(args.warmup, 2) if precision == 'fp16' else 2):      # (the at-parity leg warms up like the headline: first steps build optimizer state and allocator pools)
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / nsteps * 1e3
        out[key] = dict(ms_per_step=ms, clips_s=1e3 / ms, steps=nsteps, dtype=dtype)
        max_abs_d(precision, net)
        del net, step
        torch.cuda.empty_cache()
    if ref_mask is not None:
        # the fastest measured mode whose mask logits are within the north-star bound of 1e-3 of the reference forward
        rates = {'bf16': bf16_rate, 'fp16': out['fp16_mode']['clips_s'], 'bf16x3': out['bf16x3_mode']['clips_s'], 'fp32': out['fp32_parity_mode']['clips_s']}
        ok = [(rates[m], m) for m in rates if rates[m] is not None and out['max_abs_d'].get(m, 1.0) < 1e-3]
        if ok:
            r, m = max(ok)
            out['fastest_within_1e-3'] = dict(mode=m, clips_s=r, max_abs_d=out['max_abs_d'][m])
            # first-class fields: the throughput of the fastest mode INSIDE the north-star tolerance (mask-logit max|d| < 1e-3), measured over
            # the headline's step count in the same run
            out['value_at_parity'] = r
            out['dtype_at_parity'] = {'bf16': 'bf16', 'fp16': 'f16', 'bf16x3': 'f32 (bf16x3 products)', 'fp32': 'f32'}[m]
            out['max_abs_d_at_parity'] = out['max_abs_d'][m]
            out['steps_at_parity'] = args.steps if m in ('bf16', 'fp16') else max(args.parity_steps, 10)
        # ... and the same question at a TRAINED checkpoint's logit scale (G16, logit std 5.0 instead of the random-init 0.154): 16-bit error is relative, so
        # the absolute 1e-3 is a much harder bound there -- the rate a user with a real checkpoint gets inside the north-star tolerance
        ts = out.get('trained_scale')
        if ts is not None:
            names = {'bf16': 'bf16', 'fp16': 'f16', 'bf16x3': 'f32 (bf16x3 products)', 'fp32': 'f32'}
            ok = [(rates[m], m) for m in rates if rates[m] is not None and ts['max_abs_d'].get(m, 1.0) < 1e-3]
            r, m = max(ok) if ok else (None, None)
            out['value_at_parity_trained_scale'] = r
            out['dtype_at_parity_trained_scale'] = names.get(m)
            out['max_abs_d_at_parity_trained_scale'] = ts['max_abs_d'].get(m)
        out['max_abs_d']['logit_std'] = float(ref_mask.std())
        out['max_abs_d']['against'] = 'oracle (CPU restatement pinned to the reference): maximum over clip seeds 900-902 x weight seeds 900-901 (max_abs_d_cases)'
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves (one process per GPU, RCCL) as a CHILD
    process -- never exec: a process that may have touched the GPU must not replace itself -- relay its output and return its exit
    code.  Nothing in this process has initialised HIP at this point (torch is imported, torch.cuda is untouched)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


class RankErrors:
    """N > 1: no rank may die silently and no rank may hang on one that did.  Every rank writes an exception it meets into a directory all ranks of this
    launch share (/tmp/tcow_bench_<launcher pid>_<MASTER_PORT>/rank<r>.json) and waits a moment before it re-raises; a daemon thread on every rank polls
    that directory -- it works while the main thread is blocked inside a collective or a stream synchronisation -- and when a report appears rank 0's
    thread prints the ONE JSON line of the contract with `value: null` and `ddp.error` = the reports (rank, exception type, message, traceback tail),
    then every rank exits with code 3 instead of waiting for the collective timeout.  (torch.distributed.run would also tear the job down, but without
    a line on stdout, and only after the failing rank's process has ended.)"""

    def __init__(self, rank, world, args):
        import tempfile
        self.rank, self.world, self.args = rank, world, args
        self.dir = os.path.join(tempfile.gettempdir(), f'tcow_bench_{os.getppid()}_{os.environ.get("MASTER_PORT", "0")}')
        self.done = False
        if world > 1:
            import threading
            os.makedirs(self.dir, exist_ok=True)
            threading.Thread(target=self._watch, daemon=True).start()

    def report(self, exc):
        import traceback
        if self.world <= 1:
            return
        tb = traceback.format_exception(type(exc), exc, exc.__traceback__)
        rec = dict(rank=self.rank, type=type(exc).__name__, message=str(exc)[:2000], traceback_tail=''.join(tb)[-1500:])
        tmp = os.path.join(self.dir, f'.rank{self.rank}.tmp')
        with open(tmp, 'w') as f:
            json.dump(rec, f)
        os.replace(tmp, os.path.join(self.dir, f'rank{self.rank}.json'))
        time.sleep(5.0)                 # (rank 0's watcher prints the line before this process -- and with it the launcher's whole job -- goes away)

    def _reports(self):
        out = []
        for name in sorted(os.listdir(self.dir)):
            if name.startswith('rank') and name.endswith('.json'):
                try:
                    out.append(json.load(open(os.path.join(self.dir, name))))
                except (OSError, ValueError):
                    pass
        return out

    def _watch(self):
        while not self.done:
            time.sleep(0.5)
            try:
                found = self._reports()
            except OSError:
                continue
            if found:
                time.sleep(1.0 if self.rank == 0 else 3.0)         # rank 0: let other failing ranks finish writing; the others: let rank 0 print before any exit makes the launcher tear the job down
                found = self._reports() or found
                if self.rank == 0:
                    a = self.args
                    print(json.dumps(dict(metric='train clips/sec (T=30, 240x320)', value=None, unit='clips/s', n_gpus=self.world, steps=a.steps, warmup=a.warmup,
                                          ms_per_step=None, higher_is_better=True, scaling='weak', vs_baseline=None, data='synthetic',
                                          ddp=dict(error=found, ranks=self.world, note='a rank raised: the run was stopped instead of hanging in the next collective'))), flush=True)
                    sys.stderr.write(f'bench.py: stopped, rank(s) {[r["rank"] for r in found]} raised: {found[0]["type"]}: {found[0]["message"][:300]}\n')
                os._exit(3)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    errors = RankErrors(int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), args)
    try:
        run(args)
    except BaseException as e:          # noqa: BLE001 -- SystemExit of a rank-count mismatch included
        if not (isinstance(e, SystemExit) and e.code in (0, None)):
            errors.report(e)
        raise
    finally:
        errors.done = True
        if errors.world > 1 and errors.rank == 0:
            try:
                os.rmdir(errors.dir)            # (empty after a clean run; a directory with reports stays for the post-mortem)
            except OSError:
                pass


def run(args):
    from tcow_amd import ddp, engine, flops, ops, synth
    from tcow_amd.seeker import Seeker
    rank, local_rank, world = ddp.init_distributed()
    # the first collective of the run: small, checked, with a host-side timeout and a message that names the environment (ddp.first_contact)
    contact = ddp.first_contact(torch.device('cuda', local_rank % max(torch.cuda.device_count(), 1)) if world > 1 and torch.distributed.get_backend() == 'nccl' else None) if world > 1 else None
    if os.environ.get('TCOW_BENCH_FAIL_RANK') == str(rank):       # (tests: a rank that dies right after the rendezvous -- rank 0 must still print a line)
        raise RuntimeError(f'TCOW_BENCH_FAIL_RANK: injected failure on rank {rank}')
    if args.launch_selftest:
        if world > 1 and os.environ.get('TCOW_BENCH_FAIL_RANK') is not None:
            torch.distributed.barrier()        # (the surviving ranks sit in a collective the failed one never joins: the situation RankErrors exists for)
        if rank == 0:
            print(json.dumps(dict(selftest='launch', n_gpus=world, ranks_seen=(torch.distributed.get_world_size() if world > 1 else 1), gpus_arg=args.gpus,
                                  first_contact=contact)), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the rank count must match (torch.distributed.run --nproc-per-node {args.gpus})')
    ndev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank % ndev)              # (% ndev only matters for the single-GPU gloo dry run of the N>1 path)
    dev = torch.device('cuda', local_rank % ndev)

    cfg = synth.seeker_config(num_total_frames=args.frames, frame_height=args.height, frame_width=args.width,
                              depth=args.depth, causal_attention=1)
    from tcow_amd.optim import FusedAdamWClip
    from tcow_amd.pipeline import SeekerPipeline
    from tcow_amd.tcow_loss import default_args
    Qs = args.queries
    # synthetic Kubric-shaped batch for this rank: 1 clip, Qs queries chosen by desirability, query_time 0 (README.md:42)
    data = synth.to_torch_tree(synth.make_kubric_batch(1, args.frames, args.height, args.width, seed=ddp.shard_seed(900, rank), n_objects=5), dev,
                               host_keys=synth.HOST_KEYS)   # control-flow metadata stays on the host, as in the reference's DataLoader batch

    def make_trainer(precision):
        net = Seeker(None, num_total_frames=args.frames, frame_height=args.height, frame_width=args.width,
                     tracker_pretrained=False, causal_attention=1, drop_path_rate=0.1, network_depth=args.depth,
                     precision=precision)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 900).items()}, strict=True)
        net = net.to(dev).train()
        ddp.broadcast_parameters(net)
        opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3, module=net, fuse_cast=os.environ.get('TCOW_FUSE_CAST', '1') != '0')   # train.py:99-102,239-241: clip_grad_norm_(0.3) + AdamW(lr 1e-4), fused
        net.seeker.persistent_grads = True                                 # one backward per step: gradients live in persistent flat buckets
        net.seeker.grad_hook = ddp.GradSync(world, bucket_dtype=args.grad_dtype)
        pipe = SeekerPipeline(net, num_queries=Qs, train_args=default_args(), phase='train', device=dev,
                              rng=__import__('numpy').random.default_rng(ddp.shard_seed(900, rank)))
        state = {'step': 0}

        def step():
            model_retval = pipe.forward_kubric(data)                        # query sampling, query/target masks, ONE batched Seeker call (pipeline.py:85-200); persistent gradient buckets are overwritten, no zero_grad
            progress = state['step'] / 1000.0
            loss = pipe.step_losses(data, model_retval, progress)['total_seeker']   # loss.py:238-421: weighted BCE + bootstrapped BCE + soft Jaccard
            loss.backward()                                                # train.py:98 (bucketed RCCL all-reduce runs inside, overlapped)
            opt.step()                                                     # grad-clip 0.3 + AdamW in three launches
            state['step'] += 1
            return loss
        return net, step

    net, step = make_trainer(args.precision)
    timer = KernelTimer('fp16' if args.precision == 'fp16' else 'bf16')
    atimer = AttnTimer('fp16' if args.precision == 'fp16' else 'bf16')

    for _ in range(args.warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    net.seeker.grad_hook.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    ddp_stats = net.seeker.grad_hook.stats() if world > 1 else None
    # second, untimed pass of the same step count: per-launch HIP events around every NT GEMM (roofline.achieved)
    timer.begin(400 * args.steps)
    atimer.begin(120 * args.steps)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    timer.end()
    atimer.end(args.steps, {'bf16': PEAK_BF16_TFLOPS, 'fp16': PEAK_BF16_TFLOPS, 'fp32': PEAK_F32_TFLOPS, 'bf16x3': PEAK_BF16_TFLOPS / 3.0}[args.precision])
    # host time to enqueue ONE step into an empty stream (the loop above is throttled by the queue depth: the host runs ahead of the GPU)
    enq = []
    for _ in range(3):
        torch.cuda.synchronize()
        te = time.perf_counter(); step(); enq.append((time.perf_counter() - te) * 1e3)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    rank_ms = None
    if world > 1:
        mine = torch.tensor([dt / args.steps * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        rank_ms = [float(t.item()) for t in allr]
        dt = max(rank_ms) * args.steps / 1e3            # the contract's MAX over ranks
    ms_per_step = dt / args.steps * 1e3
    clips_per_s = world * 1.0 / (dt / args.steps)

    if rank == 0:
        g = net.seeker.geometry(Qs)
        fl = flops.seeker_forward_flops(1, g['T'], g['Hp'], g['Wp'], g['D'], g['heads'], args.depth)
        ks = timer.summary()
        peak = {'bf16': PEAK_BF16_TFLOPS, 'fp16': PEAK_BF16_TFLOPS, 'fp32': PEAK_F32_TFLOPS, 'bf16x3': PEAK_BF16_TFLOPS / 3.0}[args.precision]   # (dense f16 = dense bf16 peak)   # bf16x3: three MFMAs per product
        traffic = pmc_traffic() if args.precision in ('bf16', 'fp16') else None        # (same kernels, same bytes in both 16-bit builds)
        roof = dict(bound='mfma', kernel={'bf16': 'gemm_nt_bf16_320_kernel (+ gemm_nt_bf16_c2_kernel, gemm_nt_bf16_256_kernel, gemm_nt_bf16_kernel)', 'fp16': 'gemm_nt_bf16_320_kernel built for binary16 (libtcow_hip_fp16.so)',
                            'fp32': 'gemm_f32_kernel', 'bf16x3': 'gemm_x3_kernel'}[args.precision],
                    achieved=ks['tflops'], peak=peak, unit='TFLOP/s', frac=ks['tflops'] / peak, traffic=traffic,
                    launches_per_step=ks['launches'] / args.steps, avg_launch_us=ks['avg_us'], flops_per_launch=ks['flops_per_launch'])
        step_tflops = 3.0 * Qs * fl['total'] / (ms_per_step * 1e-3) / 1e12
        res = dict(metric='train clips/sec (T=30, 240x320)', value=clips_per_s, unit='clips/s', n_gpus=world, ranks_seen=(torch.distributed.get_world_size() if world > 1 else 1), steps=args.steps,
                   warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype={'bf16': 'bf16', 'fp16': 'f16', 'fp32': 'f32', 'bf16x3': 'f32 (bf16x3 products)'}[args.precision], data='synthetic',
                   config=dict(workload=f'TCOW Seeker train step: T={args.frames} {args.height}x{args.width} patch16, {args.depth}-layer divided '
                               f'space-time ViT (D={g["D"]}), num_queries={Qs}, causal_attention=1, 1 clip/GPU',
                               clips_per_gpu=1, num_queries=Qs, parallelism=f'dp{world}', optimizer='AdamW lr 1e-4, clip 0.3',
                               loss='TCOW mask losses (loss.py:238-421): class-balanced BCE + bootstrapped BCE + soft Jaccard on 3 channels'),
                   query_forwards_per_s=clips_per_s * Qs, step_model_tflops=step_tflops, step_mfma_frac=step_tflops / peak,
                   final_loss=float(loss.detach()), host_enqueue_ms=round(sorted(enq)[1], 2), roofline=roof, roofline_attention=atimer.result)
        if world > 1:
            # what the scaling curve needs to explain itself: how long the compute stream stood still for the gradient all-reduce (rank 0),
            # how much went over xGMI per step in how many collectives, and the spread of the per-rank step times
            res['ddp'] = dict(ddp_stats, ms_per_step_min=min(rank_ms), ms_per_step_max=max(rank_ms), ranks=world,
                              group_blocks=engine.group_sizes(args.depth),          # blocks per gradient group, top group first (TCOW_DDP_GROUP)
                              rccl=rccl_knobs(), rccl_version=contact.get('rccl_version'), backend=contact.get('backend'), first_contact_ms=round(contact.get('ms', 0.0), 1),
                              cpu_baseline='reported at N = 1 only (rank 0 of a single-GPU run)')
        ref_mask = None
        if not args.no_cpu_baseline and world == 1:
            try:
                ref_mask, res['cpu_baseline'] = cpu_baseline(cfg, args.cpu_budget_s)
            except Exception as e:  # the baseline is a reported aside; never fail the bench line over it
                res['cpu_baseline'] = dict(value=None, unit='clips/s', cores=os.cpu_count(), kind='port', sample=f'failed: {e}')
        if world == 1 and args.precision == 'bf16' and not args.no_parity:
            res.update(parity_leg(make_trainer, net, ref_mask, args, bf16_rate=clips_per_s))
        if world == 1 and not args.no_config_legs:
            try:
                res.update(config_legs(args, dev))
            except Exception as e:  # reported asides; never lose the headline over them
                res['config_legs_error'] = repr(e)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
