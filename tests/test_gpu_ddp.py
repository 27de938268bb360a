"""GPU: the data-parallel path of the step (tcow_amd/ddp.py + engine.run_backward; reference: train.py:222-223, loss.py:356-369) exercised on the ONE
GPU a test box has.

  * two processes share cuda:0 and talk over gloo; each runs the REAL engine forward + backward on its own clip (seed 900 + rank) with a
    GradSync attached: every parameter gradient must equal the mean of the two single-process gradients (the batch loss is the mean over
    replicas).  In the binary16 mode the two ranks use DIFFERENT static loss scales, so the mean only comes out right if every bucket is
    multiplied back before the hook sees it.
  * one process, init_process_group('nccl', world_size=1): loads RCCL and all-reduces the real buckets of a backward with ReduceOp.AVG,
    asynchronously, through the same GradSync code the N-GPU run uses -- values must be unchanged, the exposed-wait events must resolve.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

def _cfg():
    from tcow_amd import synth
    return synth.seeker_config(num_total_frames=4, frame_height=64, frame_width=64, depth=4, embed_dim=256, num_heads=4)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _net(precision):
    from conftest import build_hip_seeker
    from tcow_amd import synth
    net = build_hip_seeker(_cfg(), synth.make_state_dict(_cfg(), 4242), precision).cuda()
    net.train(True)
    return net


def _grads(net, seed, hook=None, loss_scale=None):
    """One forward + backward of the engine on the clip of `seed`; returns {name: gradient copy}."""
    from tcow_amd import synth
    cfg = _cfg()
    clip = synth.make_clip(1, cfg['num_total_frames'], cfg['frame_height'], cfg['frame_width'], seed=seed)
    rgb = torch.from_numpy(clip['rgb']).cuda(); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0)).cuda()
    net.seeker.grad_hook = hook
    if loss_scale is not None:
        net.seeker.loss_scale = loss_scale
    for p in net.parameters():
        p.grad = None
    om, fl = net(rgb, qm)
    Gm = torch.from_numpy(synth._rng(7, 'ddp_probe_mask').standard_normal(size=tuple(om.shape), dtype=np.float32)).cuda()
    Gf = torch.from_numpy(synth._rng(7, 'ddp_probe_flags').standard_normal(size=tuple(fl.shape), dtype=np.float32)).cuda()
    ((om * Gm).sum() + (fl * Gf).sum()).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


def _worker(rank, world, port, precision, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
        from tcow_amd import ddp
        torch.cuda.set_device(0)
        r, _, w = ddp.init_distributed(backend='gloo')
        net = _net(precision)
        torch.manual_seed(rank)
        with torch.no_grad():                                   # ranks start from different weights: the coalesced broadcast must make them equal
            for p in net.parameters():
                p.add_(torch.randn_like(p) * (0.01 if rank else 0.0))
        ncoll = ddp.broadcast_parameters(net)
        scales = {'fp16': (2.0 ** 6, 2.0 ** 10)}.get(precision)                                 # per-rank static loss scales (binary16 only)
        single = [_grads(net, ddp.shard_seed(900, k), hook=None, loss_scale=None if scales is None else scales[0]) for k in range(world)]
        sync = ddp.GradSync(world)
        got = _grads(net, ddp.shard_seed(900, rank), hook=sync, loss_scale=None if scales is None else scales[rank])
        tol = {'fp32': 1e-6, 'fp16': 4e-3}[precision]
        worst, bad = 0.0, []
        for k, g in got.items():
            ref = 0.5 * (single[0][k] + single[1][k])
            err = float((g - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
            worst = max(worst, err)
            if err > tol:
                bad.append((k, err))
        st = sync.stats()
        q.put((rank, len(got), worst, bad[:5], ncoll, st['buckets'], st['allreduce_bytes'], sorted(map(str, sync.launched))))
        torch.distributed.destroy_process_group()
    except Exception as e:                                       # noqa: BLE001 -- report instead of hanging the parent
        import traceback
        q.put((rank, -1, 0.0, [('exception', traceback.format_exc()[-1500:])], 0, 0, 0, []))


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_two_ranks_on_one_gpu_average_the_engine_gradients(cuda, precision, monkeypatch):
    monkeypatch.setenv('TCOW_DDP_GROUP', '2')      # two gradient groups in the 4-block test model (inherited by the spawned ranks)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, precision, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs: p.join(timeout=120)
    for rank, n, worst, bad, ncoll, buckets, nbytes, tags in res:
        assert n >= 85 and not bad, (rank, n, worst, bad)         # every trained tensor of the 4-block model (89), each == mean of the two single-process gradients
        assert ncoll == 1                                         # parameters + buffers of one dtype: ONE broadcast, not 251
        assert buckets >= 2 and nbytes > 0 and tags, (buckets, nbytes, tags)
    assert res[0][5:] == res[1][5:]


def test_rccl_world_of_one_runs_the_real_buckets(cuda, monkeypatch):
    """backend 'nccl' IS RCCL on ROCm: never initialised by any other test.  A group of one rank leaves the values unchanged, but the collective
    kernels run: ReduceOp.AVG, async work handles, wait() stream semantics and the event pair around the drain."""
    import torch.distributed as dist
    from tcow_amd import ddp
    monkeypatch.setenv('TCOW_DDP_GROUP', '2')      # the 4-block test model in two gradient groups (the default, 4 blocks per group, makes it one bucket)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    if dist.is_initialized():
        dist.destroy_process_group()
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    try:
        net = _net('fp32')
        ref = _grads(net, 900, hook=None)
        for bucket_dtype in ('f32', 'bf16'):
            sync = ddp.GradSync(1, force=True, bucket_dtype=bucket_dtype)
            assert sync.native_avg                                                  # RCCL averages inside the collective
            got = _grads(net, 900, hook=sync)
            st = sync.stats()
            assert st['buckets'] >= 2 and st['allreduce_bytes'] > 0 and st['allreduce_exposed_ms'] >= 0.0 and not sync.pending
            for k, g in got.items():
                if bucket_dtype == 'f32':
                    assert torch.equal(g, ref[k]), k                                # AVG over one rank: bit-identical
                else:
                    assert float((g - ref[k]).abs().max()) <= 2.0 ** -8 * float(ref[k].abs().max()) + 1e-30, k     # one bf16 rounding on the wire
        flat = torch.arange(1 << 20, dtype=torch.float32, device='cuda')
        ddp.broadcast_parameters(net)                                               # world of one: nothing to do, must not raise
        w = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True); w.wait(); torch.cuda.synchronize()
        assert float(flat[12345]) == 12345.0
    finally:
        dist.destroy_process_group()


def test_full_size_two_rank_bench_line_on_one_gpu(cuda):
    """The N > 1 bench line of BASELINE configs[2] (DDP, one clip per rank, T=30 240x320, 12 blocks) executed at HEAD: `python bench.py --gpus 2`
    starts its two ranks itself; both share this box's one GPU (bench.py maps local_rank % device_count) and reduce over gloo.  Exercises
    bench.py's world > 1 branches (barriers, max over ranks, the `ddp` block) and engine.run_backward's bucket publishing with the real
    buckets -- everything the driver's SCALE run executes except RCCL's wire.  The line is kept under gpurun_out/ (copied to profiles/)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    from tcow_amd import engine
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'TCOW_DDP_GROUP'):
        env.pop(k, None)
    env['TCOW_DIST_BACKEND'] = 'gloo'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-parity'],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_2rank_gloo_one_gpu.json'), 'w') as f:
            f.write(json.dumps(res) + '\n')
    except OSError:
        pass
    assert res['n_gpus'] == 2 and res['ranks_seen'] == 2 and res['steps'] == 3 and res['scaling'] == 'weak'
    assert res['value'] > 0 and np.isfinite(res['final_loss']) and res['config']['parallelism'] == 'dp2'
    d = res['ddp']
    sizes = engine.group_sizes(12)
    assert d['ranks'] == 2 and d['group_blocks'] == sizes and d['buckets'] == len(sizes) + 1          # block groups + the folded projection's late bucket
    assert abs(d['allreduce_bytes'] - 488.6e6) < 2e6                                                # every trained parameter once, f32
    assert d['allreduce_exposed_ms'] >= 0.0 and 0 < d['ms_per_step_min'] <= d['ms_per_step_max'] and abs(d['ms_per_step_max'] - res['ms_per_step']) < 1e-6
    assert 'cpu_baseline' not in res and 'roofline' in res
    # round 6: what a first SCALE run needs to explain itself -- the checked first collective, the backend / RCCL version, and per bucket when the backward
    # published it and when the compute stream got past its wait (launch order = publishing order of the backward: top group first, late bucket last)
    tl = d['bucket_timeline']
    assert d['backend'] == 'gloo' and d['first_contact_ms'] >= 0.0 and 'rccl_version' in d
    assert len(tl) == d['buckets'] and tl[0]['launch_ms'] == 0.0 and sum(b['bytes'] for b in tl) == d['allreduce_bytes']
    assert all(a['launch_ms'] <= b['launch_ms'] and a['done_ms'] <= b['done_ms'] for a, b in zip(tl, tl[1:])) and all(b['done_ms'] >= b['launch_ms'] for b in tl)
    ra = res['roofline_attention']
    assert all(ra[k]['us'] > 0 and 0 < ra[k]['hbm_frac'] < 1 and 0 < ra[k]['mfma_frac'] < 1 for k in ('spatial_fwd', 'spatial_bwd', 'temporal_fwd', 'temporal_bwd'))
    assert ra['spatial_fwd']['launches_per_step'] == 12 and ra['temporal_bwd']['launches_per_step'] == 12
