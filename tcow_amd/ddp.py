"""One-process-per-GPU data parallelism for the Seeker step: gradient all-reduce over RCCL (xGMI), overlapped
with the backward pass.

The reference uses single-process torch.nn.DataParallel (train.py:222-223): per step it re-broadcasts all
122 M parameters, gathers every replica's (B,Qs,3,T,H,W) outputs to GPU 0 and reduces gradients there.  Clips
are independent through the model and the per-example loss (pipeline.py:50-83), and the batch loss is the mean
over replicas (loss.py:356-369), so the MI355X counterpart is plain gradient averaging:

  * each rank owns B clips (1 in BASELINE configs[2]) and runs forward / loss / backward locally;
  * the hand-written backward (engine.run_backward) finishes its parameter gradients in GROUPS of transformer blocks
    (engine.group_sizes: TCOW_DDP_GROUP, default 5 / 5 / 2 blocks from the top at depth 12; the top group also carries the
    output heads, the bottom one the embeddings), each as ONE flat f32 buffer, plus one late bucket for the three tensors
    per block that come out of the folded temporal projection: 4 buckets per step at ViT-B -- 180 + 177 + 75 + 57 MB of the
    488.6 MB.  `GradSync` launches an asynchronous all-reduce on a bucket the moment the backward publishes it, so the two
    upper groups travel while the blocks below them are still being differentiated; the bottom group and the late bucket
    exist only when the backward is over and cannot overlap anything, which is why the bottom group is the small one
    (132 MB exposed by construction instead of the 203 MB of an even 4 / 4 / 4 split);
  * few, large collectives on purpose: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring sees ~300 GB/s of bus
    bandwidth, and every collective is a window in which resident RCCL workgroups push the one-workgroup-per-CU GEMMs of
    the backward into an extra round (profiles/r02_cu_contention.txt: 20-23 % on the 320-tile GEMMs while any CU is held).
    RCCL's channel count (NCCL_MAX_NCHANNELS) trades that window's length against its depth; the run records the knobs it
    saw in the `ddp.rccl` field of the bench line, none is set by this module;
  * parameters the step does not touch (model.norm.*, flag_post_linear.* in Kubric training) have no gradient on
    any rank and are simply not part of any bucket -- no unused-parameter handshake is needed.
"""
import collections
import datetime
import os

import torch
import torch.distributed as dist

STATS_WINDOW = 256      # finish() calls whose exposed-wait measurements are kept (a training run must not grow per-step state forever)


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract). Returns (rank, local_rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('TCOW_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')     # 'nccl' is RCCL on ROCm
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        # a finite collective timeout (TCOW_DIST_TIMEOUT_S, default 300 s): a rank that never arrives ends the run with an error instead of a hang
        tmo = datetime.timedelta(seconds=float(os.environ.get('TCOW_DIST_TIMEOUT_S', '300')))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=tmo)
    return rank, local_rank, world


def rccl_version():
    """Version of the collective library behind backend 'nccl' (RCCL on ROCm) as a string, or None."""
    try:
        v = torch.cuda.nccl.version()
        return '.'.join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:          # noqa: BLE001 -- a build without the binding
        return None


def first_contact(device=None, timeout_s=None, group=None):
    """The FIRST collective of a run, on purpose small and checked: every rank contributes its rank number, the sum must be world (world - 1) / 2.
    Waits on the host with a timeout, so that a broken fabric / IPC set-up (HSA_ENABLE_IPC_MODE_LEGACY, a rank on the wrong device, a firewall on the
    rendezvous port) is reported in seconds with the environment that matters, not as a hung job.  Returns dict(ms, world, backend, rccl_version)."""
    import time
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return dict(ms=0.0, world=1, backend=None, rccl_version=None)
    world, rank, backend = dist.get_world_size(group), dist.get_rank(group), dist.get_backend(group)
    timeout_s = float(os.environ.get('TCOW_FIRST_CONTACT_TIMEOUT_S', '120')) if timeout_s is None else float(timeout_s)
    t = torch.tensor([float(rank), 1.0], dtype=torch.float32, device=device if backend == 'nccl' else 'cpu')
    t0 = time.perf_counter()
    try:
        work = dist.all_reduce(t, group=group, async_op=True)
        done = work.wait(datetime.timedelta(seconds=timeout_s))
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        got = t.cpu().tolist()
    except Exception as e:      # noqa: BLE001
        raise RuntimeError(f'first collective ({backend}, {world} ranks) failed on rank {rank} after {time.perf_counter() - t0:.1f} s: {type(e).__name__}: {e} '
                           f'[MASTER_ADDR={os.environ.get("MASTER_ADDR")} MASTER_PORT={os.environ.get("MASTER_PORT")} LOCAL_RANK={os.environ.get("LOCAL_RANK")} '
                           f'HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")} (must be 0 on this pool); set NCCL_DEBUG=INFO for RCCL\'s own log]') from e
    if done is False or abs(got[0] - world * (world - 1) / 2.0) > 1e-3 or abs(got[1] - world) > 1e-3:
        raise RuntimeError(f'first collective ({backend}) returned {got} on rank {rank}: expected [{world * (world - 1) / 2.0}, {float(world)}] -- ranks missing or duplicated')
    return dict(ms=(time.perf_counter() - t0) * 1e3, world=world, backend=backend, rccl_version=rccl_version() if backend == 'nccl' else None)


class GradSync:
    """Bucketed, overlapped gradient averaging. Attach with `module.grad_hook = sync`: engine.run_backward calls it once
    per completed bucket and drains it (`finish()`) before handing the gradients to autograd, so after `loss.backward()`
    every param.grad is already the mean over ranks."""

    def __init__(self, world_size=None, group=None, overlap=None, bucket_dtype='f32', force=False):
        """overlap: launch each bucket's all-reduce as soon as the backward has produced it (default; `TCOW_DDP_OVERLAP=0` or
        overlap=False issues them all after the last bucket instead -- the fallback should RCCL kernels holding CUs during the
        backward cost more than they hide).
        bucket_dtype: 'f32' moves the f32 buckets as they are (488.6 MB per step at ViT-B); 'bf16' all-reduces a bf16 copy of each
        bucket (244.3 MB: half the per-link xGMI time) and widens the mean back into the f32 bucket -- master weights, moments and
        the clip norm stay f32 (AdamW reads the f32 bucket).
        force: run the collectives even in a group of one rank (the single-GPU RCCL check of tests/test_gpu_ddp.py: values unchanged, but the
        library is loaded and AVG / async wait() / the event bracketing of finish() execute on hardware)."""
        self.force = bool(force)
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.overlap = (os.environ.get('TCOW_DDP_OVERLAP', '1') != '0') if overlap is None else bool(overlap)
        # RCCL averages inside the collective (ReduceOp.AVG); gloo has no AVG: sum + one scale pass there (TCOW_DDP_AVG=0 forces that path)
        self.native_avg = os.environ.get('TCOW_DDP_AVG', '1') != '0' and dist.is_initialized() and dist.get_backend(group) == 'nccl'
        if bucket_dtype not in ('f32', 'bf16'):
            raise ValueError("bucket_dtype must be 'f32' or 'bf16'")
        self.bucket_dtype = bucket_dtype
        self.pending = []
        self.deferred = []
        self.bytes = 0
        self.launched = []              # bucket tags of the CURRENT / most recent backward (reset when the next one publishes its first bucket)
        self.steps = 0                  # finish() calls since reset_stats()
        self._nbuckets = 0              # buckets published since reset_stats()
        self._fresh = True
        self._exposed = collections.deque(maxlen=STATS_WINDOW)   # per finish(): (start event, end event) on the compute stream, or seconds on the CPU path
        self._timeline = None           # last step: [(tag, bytes, event at launch, event behind the wait)] on the compute stream (GPU buckets only)

    @property
    def active(self):
        """False when the hook moves nothing (one rank, not forced): engine.run_backward may then leave the binary16 loss scale for the optimizer to undo."""
        return self.world > 1 or self.force

    def _launch(self, flat, tag=None):
        op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
        wire = flat.to(torch.bfloat16) if self.bucket_dtype == 'bf16' else flat
        self.bytes += wire.numel() * wire.element_size()
        ev = None
        if flat.is_cuda:               # when the backward published this bucket, on the compute stream (per-bucket timeline of stats())
            ev = torch.cuda.Event(enable_timing=True); ev.record()
        self.pending.append((dist.all_reduce(wire, op=op, group=self.group, async_op=True), flat, wire, ev, tag))

    def __call__(self, name, flat):
        if (self.world <= 1 and not self.force) or flat.numel() == 0:
            return
        if self._fresh:
            self.launched = []; self._fresh = False
        self.launched.append(name)
        self._nbuckets += 1
        if self.overlap:
            self._launch(flat, name)
        else:
            self.deferred.append((flat, name))

    def finish(self):
        """Drain: the compute stream waits for every collective of this backward.  The wait is bracketed by two events on that stream, so
        `stats()` can report how long the stream actually stood still for communication (the EXPOSED part of the all-reduce) without any
        synchronisation inside the step."""
        for flat, name in self.deferred:
            self._launch(flat, name)
        self.deferred = []
        self.steps += 1
        self._fresh = True
        if not self.pending:
            return
        on_gpu = self.pending[0][1].is_cuda
        if on_gpu:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        else:
            import time
            t0 = time.perf_counter()
        self._drain()
        if on_gpu:
            e1.record(); self._exposed.append((e0, e1))
        else:
            self._exposed.append(time.perf_counter() - t0)

    def reset_stats(self):
        self.bytes = 0; self.launched = []; self.steps = 0; self._nbuckets = 0; self._fresh = True; self._exposed.clear(); self._timeline = None

    def stats(self):
        """Since the last reset_stats(): all-reduce bytes and buckets per step and the mean exposed wait per step in ms (synchronises)."""
        n = max(self.steps, 1)
        ms = 0.0
        for e in self._exposed:                                   # (the last STATS_WINDOW steps; bench runs are shorter than the window)
            if isinstance(e, tuple):
                e[1].synchronize(); ms += e[0].elapsed_time(e[1])
            else:
                ms += e * 1e3
        return dict(allreduce_exposed_ms=ms / max(len(self._exposed), 1), allreduce_bytes=self.bytes // n, buckets=self._nbuckets // n, bucket_dtype=self.bucket_dtype,
                    overlap=self.overlap, native_avg=self.native_avg, bucket_timeline=self.bucket_timeline())

    def _drain(self):
        line = []
        for work, flat, wire, ev, tag in self.pending:
            work.wait()                      # makes the current stream wait for the collective
            if ev is not None:               # the compute stream has passed this bucket's wait: the collective is complete
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                line.append((tag, wire.numel() * wire.element_size(), ev, e1))
            if wire is not flat:
                flat.copy_(wire)             # bf16 mean -> f32 bucket
            if not self.native_avg:
                flat.mul_(1.0 / self.world)  # sum -> mean (loss.py:356-369 averages the per-replica losses)
        self.pending = []
        if line:
            self._timeline = line

    def bucket_timeline(self):
        """The LAST drained step's buckets in launch order: tag, bytes, and -- on the compute stream, in ms relative to the first bucket's launch -- when the
        backward published the bucket (`launch_ms`) and when the stream got past its wait (`done_ms`: the collective is complete and everything before the
        wait has run).  done - launch of an overlapped bucket = wire time + whatever backward work ran meanwhile; consecutive `done_ms` of the buckets
        that cannot overlap (the bottom group, the late bucket) are their exposed wire times.  Synchronises; None before the first GPU step."""
        if not self._timeline:
            return None
        t0 = self._timeline[0][2]
        out = []
        for tag, nbytes, e0, e1 in self._timeline:
            e1.synchronize()
            out.append(dict(tag=tag, bytes=int(nbytes), launch_ms=round(t0.elapsed_time(e0), 3), done_ms=round(t0.elapsed_time(e1), 3)))
        return out


def broadcast_parameters(module, src=0, group=None):
    """One-time weight (and buffer) sync at start (DataParallel re-broadcasts every step; DDP does not need to): ONE collective per dtype over a
    flat copy of all tensors instead of one per tensor (251 at ViT-B).  Returns the number of collectives issued."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return 0
    by_dtype = collections.OrderedDict()
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault((t.dtype, t.device), []).append(t.data)
    n = 0
    for tensors in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()
        n += 1
    return n


def shard_seed(base_seed, rank):
    """Per-rank data seed (train.py:170-174 seeds 900; each rank draws different clips)."""
    return int(base_seed) + int(rank)
