"""CPU, world_size 2 over gloo: the bucketed gradient averaging used by the data-parallel step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tcow_amd import ddp, synth
    r, lr, w = ddp.init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    sync = ddp.GradSync(world)
    buckets = {name: torch.full((n,), float(rank + 1) * (i + 1)) for i, (name, n) in enumerate([('head', 7), (1, 1000), (0, 1000), ('embed', 33)])}
    for name, flat in buckets.items():
        sync(name, flat)                       # engine.run_backward fires the hook in exactly this way, bucket by bucket
    sync.finish()
    ok = all(torch.allclose(flat, torch.full_like(flat, 1.5 * (i + 1))) for i, flat in enumerate(buckets.values()))
    late = ddp.GradSync(world, overlap=False)          # the non-overlapped fallback: nothing is sent before finish()
    b2 = {name: torch.full((n,), float(rank + 1) * (i + 1)) for i, (name, n) in enumerate([('head', 5), (0, 64)])}
    for name, flat in b2.items():
        late(name, flat)
    ok = ok and len(late.pending) == 0 and all(torch.equal(flat, torch.full_like(flat, float(rank + 1) * (i + 1))) for i, flat in enumerate(b2.values()))
    late.finish()
    ok = ok and all(torch.allclose(flat, torch.full_like(flat, 1.5 * (i + 1))) for i, flat in enumerate(b2.values()))
    half = ddp.GradSync(world, bucket_dtype='bf16')    # bf16 on the wire, f32 bucket in and out
    b3 = torch.full((300,), float(rank + 1) * 0.25)
    half('blk', b3); half.finish()
    ok = ok and b3.dtype == torch.float32 and torch.equal(b3, torch.full_like(b3, 0.375)) and half.bytes == 300 * 2
    lin = torch.nn.Linear(4, 4)
    torch.manual_seed(rank); torch.nn.init.normal_(lin.weight)
    ddp.broadcast_parameters(lin)
    gathered = [torch.zeros_like(lin.weight) for _ in range(world)]
    dist.all_gather(gathered, lin.weight.data)
    ok = ok and torch.equal(gathered[0], gathered[1])
    clip_a = synth.make_clip(1, 2, 16, 16, seed=ddp.shard_seed(900, rank))['rgb']
    ok = ok and sync.launched == ['head', 1, 0, 'embed'] and sync.bytes == (7 + 2000 + 33) * 4
    st = sync.stats()                              # the fields bench.py puts into its N > 1 line
    ok = ok and st['buckets'] == 4 and st['allreduce_bytes'] == (7 + 2000 + 33) * 4 and st['allreduce_exposed_ms'] >= 0.0 and st['bucket_dtype'] == 'f32' 
    q.put((rank, bool(ok), float(clip_a.sum())))
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] != res[1][2]              # ranks draw different clips (weak scaling: 1 clip per rank)


def test_single_process_is_a_noop():
    from tcow_amd import ddp
    sync = ddp.GradSync(1)
    t = torch.ones(5)
    sync('x', t); sync.finish()
    assert torch.equal(t, torch.ones(5)) and sync.pending == []


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (it silently ran one) --
    exercised on CPU through the launch self-test (gloo), which stops before any GPU work."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ); env.pop('WORLD_SIZE', None); env.pop('RANK', None); env['TCOW_DIST_BACKEND'] = 'gloo'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-selftest'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['ranks_seen'] == 2
    # a rank count that does not match --gpus is an error, not a silent single-GPU run
    env2 = dict(env, WORLD_SIZE='1', RANK='0')
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env2, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and 'WORLD_SIZE' in (bad.stderr + bad.stdout)
