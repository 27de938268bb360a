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


def test_a_hook_of_one_rank_says_it_is_inactive():
    """engine.run_backward leaves the binary16 loss scale to the optimizer only when no ACTIVE data-parallel hook is attached (ranks choose their own
    scales, so buckets must be unscaled before they are averaged): a GradSync of one rank is inactive unless forced."""
    from tcow_amd import ddp
    assert ddp.GradSync(world_size=1).active is False
    assert ddp.GradSync(world_size=1, force=True).active is True
    assert ddp.GradSync(world_size=2).active is True


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


def test_a_failing_rank_surfaces_in_rank_zeros_line_instead_of_a_hang():
    """a rank != 0 that raises after the rendezvous (here rank 1, injected) while rank 0 already sits in the next collective.  The run must
    end within seconds -- not at the collective timeout --, with ONE JSON line on stdout whose `ddp.error` names the rank and the exception, and a
    non-zero exit code.  Rank 0 failing itself gives the same line."""
    import json
    import subprocess
    import sys
    import time
    from conftest import ROOT
    for bad_rank in (1, 0):
        env = dict(os.environ); env.pop('WORLD_SIZE', None); env.pop('RANK', None)
        env.update(TCOW_DIST_BACKEND='gloo', TCOW_BENCH_FAIL_RANK=str(bad_rank), TCOW_DIST_TIMEOUT_S='600')
        t0 = time.time()
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-selftest'], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and time.time() - t0 < 120, (out.returncode, out.stderr[-1500:])
        lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, out.stdout[-1500:]
        res = json.loads(lines[0])
        assert res['value'] is None and res['n_gpus'] == 2
        err = res['ddp']['error']
        assert [e['rank'] for e in err] == [bad_rank] and err[0]['type'] == 'RuntimeError' and 'injected failure' in err[0]['message']


def test_first_contact_and_bucket_timeline_fields():
    """ddp.first_contact outside a process group is a no-op record; GradSync.stats() carries the per-bucket timeline key (None on the CPU path)."""
    from tcow_amd import ddp
    c = ddp.first_contact()
    assert c['world'] == 1 and c['ms'] == 0.0
    st = ddp.GradSync(1).stats()
    assert 'bucket_timeline' in st and st['bucket_timeline'] is None
    assert ddp.rccl_version() is None or isinstance(ddp.rccl_version(), str)


def test_group_sizes_spec():
    """TCOW_DDP_GROUP: one number or a comma list, top group first; the bottom group is small by default (its all-reduce cannot overlap)."""
    from tcow_amd.engine import group_sizes
    assert group_sizes(12, '4') == [4, 4, 4] and group_sizes(12, '5') == [5, 5, 2] and group_sizes(12, '5,5,2') == [5, 5, 2]
    assert group_sizes(12, '6,3') == [6, 3, 3] and group_sizes(4, '2') == [2, 2] and group_sizes(3, '8') == [3] and group_sizes(12, '12') == [12]
    assert sum(group_sizes(24, None if 'TCOW_DDP_GROUP' not in os.environ else '4')) == 24
    assert group_sizes(12, '5,5,2') == group_sizes(12, None) or 'TCOW_DDP_GROUP' in os.environ


def test_backward_publishes_only_finished_buckets():
    """CPU guard on engine.run_backward's ordering (the GPU tests check the values): every publish() of a block group must come after the
    flush_pending() that issues the group's deferred weight-gradient GEMMs and LayerNorm folds, the late fold bucket after
    finish_fold_group(), and `return grads` after grad_hook.finish().  Checked on the function's AST: for each publish call the nearest
    preceding statement that touches `pending` / `ln_jobs` work must be the flush."""
    import ast
    import inspect
    from tcow_amd import engine
    src = inspect.getsource(engine.run_backward)
    tree = ast.parse(src)
    fn = tree.body[0]
    calls = []                                       # (lineno, name) of the calls that matter, in source order, nested defs excluded
    class V(ast.NodeVisitor):
        def visit_FunctionDef(self, node):
            if node is fn:
                self.generic_visit(node)
        def visit_Call(self, node):
            f = node.func
            name = f.id if isinstance(f, ast.Name) else (f.attr if isinstance(f, ast.Attribute) else None)
            if name in ('publish', 'flush_pending', 'finish_fold_group', 'finish', 'linear_bwd', 'layernorm_bwd'):
                calls.append((node.lineno, name))
            self.generic_visit(node)
    V().visit(tree)
    calls.sort()
    names = [n for _, n in calls]
    pubs = [i for i, n in enumerate(names) if n == 'publish']
    assert len(pubs) == 3                                                   # block groups above the bottom one, the late fold bucket, the bottom group
    # (1) group publish inside the block loop: the last deferred producer before it is followed by a flush
    i = pubs[0]
    last_producer = max(j for j in range(i) if names[j] in ('linear_bwd', 'layernorm_bwd'))
    assert 'flush_pending' in names[last_producer:i], names
    # (2) the late bucket: behind finish_fold_group()
    assert names[pubs[1] - 1] == 'finish_fold_group', names
    # (3) the bottom bucket, then the drain, then nothing else
    assert names[pubs[2] + 1:] == ['finish'], names
    ret = [n.lineno for n in ast.walk(fn) if isinstance(n, ast.Return) and isinstance(n.value, ast.Name) and n.value.id == 'grads']
    assert ret and ret[-1] > calls[-1][0]
    # the bottom group's own weight gradients: the block loop's flush happens for i == 0 too (group_lo[0] == 0), before the loop ends
    assert 'if group_lo[i] == i or len(pending) + 8 > tn_group_max:' in src


def test_hook_sees_complete_buckets_with_a_fake_engine_run():
    """The same ordering, dynamically: run engine.run_backward's bucket bookkeeping with stub ops and a recording hook is not possible without
    the GPU library, so this drives GradSync the way the backward does and checks that finish() leaves nothing pending before gradients are
    handed out (single process: the hook is a no-op but its state machine still runs)."""
    from tcow_amd import ddp
    sync = ddp.GradSync(1, force=False)
    for tag in ('g7', 'g2', 'fold', 'g0'):
        sync(tag, torch.ones(3))
    sync.finish()
    assert sync.pending == [] and sync.deferred == [] and sync.steps == 1
