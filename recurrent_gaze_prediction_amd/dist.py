"""One-process-per-GPU plumbing (torch.distributed; backend 'nccl' is RCCL on ROCm).

The gaze path shards on the clip (batch) axis only: every sample's
ConvGRU state is private, batch-norm is inference-mode, so inference needs no
data-path collective -- ranks are replicas over disjoint clips.  What the ranks do
share is control: a barrier around timed regions, a MAX over ranks of the elapsed
time, and (training) the gradient all-reduce issued before the global-norm clip.
"""
import os
import subprocess
import sys

import torch


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def spawn_ranks(n_ranks, script, argv):
    """Start `n_ranks` fresh ranks of `script` with torch.distributed.run (one per GPU of this node) and return
    the launcher's exit code.  For entry points called as ``python bench.py --gpus N`` without a launcher.  The
    CALLER must not have touched the GPU yet: a process that initialised HIP must never be replaced or forked into
    ranks; the children are ordinary new processes.  Rank 0's stdout (the JSON line) is inherited."""
    port = os.environ.get('MASTER_PORT') or str(29500 + os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(n_ranks)),
           '--master-addr', '127.0.0.1', '--master-port', port, script] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # this pool's driver supports dmabuf IPC only (RCCL)
    return subprocess.call(cmd, env=env)


def init(backend=None, device=None, force=False):
    """Initialise the default process group when WORLD_SIZE > 1; returns the module or None.  `force` builds the
    group for a single rank too (tests/test_dist_gpu.py: the RCCL path on a one-GPU box)."""
    rank, _, world = env_world()
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if (device is not None and torch.device(device).type == 'cuda') else 'gloo'
    kw = {}
    if backend == 'nccl' and device is not None:
        kw['device_id'] = torch.device(device)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def barrier(dist, device=None):
    if device is not None and torch.device(device).type == 'cuda':
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    if device is not None and torch.device(device).type == 'cuda':
        torch.cuda.synchronize(device)


def max_over_ranks(dist, value, device='cpu'):
    """MAX of a python float over all ranks (the timed-region clock)."""
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device='cpu'):
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def default_store():
    """The default process group's key-value store (c10d TCPStore / FileStore): a side channel that works while the
    collectives are stuck.  None when it cannot be had (bench.py's watchdog then has its deadline only)."""
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        return _get_default_store()
    except Exception:                                                      # noqa: BLE001 -- a private API: optional
        return None


def gather_over_ranks(dist, value, device='cpu'):
    """[value of rank 0, value of rank 1, ...] (python floats; one all-gather): which rank set the MAX."""
    if dist is None:
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    got = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    return [float(t.item()) for t in got]


def shard_clips(n_clips, rank, world):
    """Contiguous, balanced [lo, hi) range of clip indices owned by `rank`."""
    base, extra = divmod(int(n_clips), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _backend(dist):
    try:
        return dist.get_backend()
    except Exception:
        return None


def allreduce_mean_(dist, tensors):
    """In-place mean of each tensor over ranks (gradient averaging before the clip, base.py:286-292 semantics on
    the global batch).  Each tensor (a flat fp32 gradient buffer) is reduced where it lies: no concatenated
    temporary, no copy-back.  RCCL averages in the collective (ReduceOp.AVG); gloo sums, then scales."""
    if dist is None or not tensors:
        return tensors
    avg = _backend(dist) == 'nccl'
    works = [dist.all_reduce(t, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True) for t in tensors]
    for w in works:
        w.wait()
    if not avg:
        for t in tensors:
            t.div_(dist.get_world_size())
    return tensors


class GradBucketReducer(object):
    """Bucketed gradient all-reduce overlapped with the backward pass.

    A bucket is a contiguous slice of a flat fp32 gradient buffer (the head's 12 MB; one C3D layer's filter + bias,
    0.02 ... 28 MB).  ``reduce(bucket, ready=...)`` is called as soon as the kernels that produce the bucket are
    ENQUEUED: the collective is issued on a side stream that first waits for ``ready`` (a callable that makes the
    side stream wait for the producer, e.g. C3DEngine.wait_layer_grads), so RCCL moves late layers over xGMI while
    the earlier layers are still differentiating on the compute stream.  ``finish()`` makes the compute stream
    wait for every collective; after it the buffers hold the mean over ranks (fp32 reduction).
    With dist None it does nothing; with gloo (CPU tests) it reduces synchronously."""

    def __init__(self, dist, device=None):
        self.dist = dist
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(self.device) if (self.cuda and dist is not None) else None
        self.avg = dist is not None and _backend(dist) == 'nccl'
        self.pending = []
        self.bytes_reduced = 0
        self.buckets_reduced = 0

    def reduce(self, bucket, ready=None):
        if self.dist is None:
            return
        self.bytes_reduced += bucket.numel() * bucket.element_size()
        self.buckets_reduced += 1
        op = self.dist.ReduceOp.AVG if self.avg else self.dist.ReduceOp.SUM
        if self.stream is not None:
            if ready is not None:
                ready(self.stream)
            else:
                self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                work = self.dist.all_reduce(bucket, op=op, async_op=True)
            self.pending.append((work, bucket))
        else:
            self.dist.all_reduce(bucket, op=op)
            if not self.avg:
                bucket.div_(self.dist.get_world_size())

    def reduce_buckets(self, buckets):
        """buckets: [(slice, ready)] in completion order (GrcnEngine.grad_buckets()); `ready` is ignored off the GPU,
        where the producer has finished by the time reduce() runs."""
        for bucket, ready in buckets:
            self.reduce(bucket, ready=ready if self.stream is not None else None)

    def finish(self):
        if self.dist is None:
            return
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                for work, bucket in self.pending:
                    work.wait()                      # orders the side stream behind RCCL's own stream
                    if not self.avg:
                        bucket.div_(self.dist.get_world_size())
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self.pending = []


def ranks_seen(dist, rank, device='cpu'):
    """Number of distinct ranks that answer an all-gather (1 without a process group)."""
    if dist is None:
        return 1
    mine = torch.tensor([int(rank)], dtype=torch.int64, device=device)
    got = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    return len(set(int(t.item()) for t in got))


def all_ranks_ok(dist, ok, device='cpu'):
    """True iff `ok` holds on EVERY rank (one MIN all-reduce): lets all ranks leave together instead of one raising
    while its peers block in the next collective."""
    if dist is None:
        return bool(ok)
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def engine_status_all_ranks(dist, engine, device):
    """engine.status() on every rank, agreed: raises on ALL ranks if any rank's persistent ConvGRU launch timed out."""
    err = None
    try:
        engine.status()
    except Exception as exc:                    # RgpError(RGP_ETIMEOUT)
        err = exc
    if not all_ranks_ok(dist, err is None, device):
        raise err if err is not None else RuntimeError('a peer rank reported a failed persistent ConvGRU launch')


def engine_timeouts_all_ranks(dist, engine, device):
    """Number of ranks whose engine reports RGP_ETIMEOUT (a persistent ConvGRU launch that lost a group member) since
    the last check -- the same number on every rank (one SUM all-reduce), so all ranks fall back together; any other
    error is raised on all ranks."""
    from . import _lib
    mine, err = 0, None
    try:
        engine.status()
    except _lib.RgpError as exc:
        if exc.code == _lib.RGP_ETIMEOUT:
            mine = 1
        else:
            err = exc
    if not all_ranks_ok(dist, err is None, device):
        raise err if err is not None else RuntimeError('a peer rank reported a failed engine')
    return int(round(sum_over_ranks(dist, mine, device)))


def dp_train_probe(dist, device, rank=0, batch=8, n_steps=35, steps=5, warmup=2, dtype='bf16', seed=0, per_step=False,
                   inject_fault=None):
    """gaze_grcn's data-parallel TRAINING step at a per-GPU shape (default: BASELINE config 4's B = 8 clips x T = 35 per
    rank; bench.py also runs config 3's B = 64 x T = 16, whose persistent ConvGRU / BPTT launches fill the chip):
    forward + backward on the rank's own clips -> GradBucketReducer (the flat 12 MB fp32 gradient, RCCL AVG on a side
    stream) -> finish -> clip_by_global_norm(10) + TF-Adam (base.py:286-297).  Timed like the headline (barrier, `steps`
    steps, barrier, MAX over ranks; 'per_rank_ms' = every rank's own clock).  Every rank starts from the same weights and
    sees different clips, so after the steps the weights must still be identical on all ranks -- checked with a MAX/MIN
    all-reduce of a checksum.  Returns a dict (same on every rank).

    The gradient goes out as THREE buckets in the order the backward finishes them (GrcnEngine.grad_buckets: batch-norm +
    upsampling + output layer, the ConvGRU filters, the projection).  Whether the first leaves BEFORE the persistent BPTT
    launch or behind it is the library's co-residency rule (include/rgp.h, rgp_grcn_grads_top_early: before, when the
    launch leaves CUs to the collective -- B <= 24 clips per GPU); reported as 'top_bucket_release'.

    'convgru_fallbacks': if a persistent launch loses a group member (RGP_ETIMEOUT: another kernel held one of its CUs for
    the whole deadline), ALL ranks -- they agree through one all-reduce -- rebuild the engine on per-timestep launches,
    restore the initial weights (the poisoned gradient has been averaged into every replica by then) and repeat warm-up
    and timing; the count of ranks that reported a time-out is returned (0 expected) and 'convgru' says which plan the
    reported time belongs to.  per_step=True: per-timestep launches from the start -- REQUIRED when several ranks share
    one device (tests), where two persistent launches would compete for the CUs (include/rgp.h).
    inject_fault ('seq' | 'bptt', tests): the first timed step's launch loses a member on rank 0."""
    import time
    from . import synthetic as syn
    from .engine import GrcnEngine
    dev = torch.device(device)
    world = dist.get_world_size() if dist is not None else 1
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + int(rank))
    x = torch.relu(torch.randn(batch, n_steps, 1024, 7, 7, device=dev, generator=g))
    gt = torch.rand(batch, n_steps, 49, 49, device=dev, generator=g) + 1e-3
    gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
    logits = torch.empty(batch, n_steps, 49, 49, device=dev)
    probs = torch.empty_like(logits)
    reducer = GradBucketReducer(dist, dev)
    fallbacks = 0
    while True:
        head = GrcnEngine(batch, n_steps, dtype=dtype, save_for_backward=True, device=dev, per_step=per_step)
        head.set_weights(syn.grcn_params(seed + 1, n_steps))
        k = [0]
        gnorm = [None]

        def step():
            head.forward(x, out_logits=logits, out_probs=probs)
            head.backward(logits, probs, gt)
            reducer.reduce_buckets(head.grad_buckets())
            reducer.finish()
            gnorm[0] = head.adam_step(k[0], 1e-4 * 0.8 ** (k[0] // 500), max_grad_norm=10.0)
            k[0] += 1

        for _ in range(warmup):
            step()
        barrier(dist, dev)
        if inject_fault and not per_step and fallbacks == 0 and rank == 0:
            head.inject_fault(inject_fault)
        bytes0 = reducer.bytes_reduced
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier(dist, dev)
        mine = time.perf_counter() - t0
        elapsed = max_over_ranks(dist, mine, dev)
        per_rank = gather_over_ranks(dist, mine, dev)
        lost = engine_timeouts_all_ranks(dist, head, dev)
        if lost == 0:
            break
        if per_step:
            raise RuntimeError('dp_train_probe: RGP_ETIMEOUT from a per-step plan')
        fallbacks += lost
        per_step = True
    persistent = not per_step and head.persistent
    seen = ranks_seen(dist, rank, dev)
    # replicas stay replicas: same weights on every rank after the averaged steps
    chk = float(head.flat_params.double().abs().sum().item())
    hi = max_over_ranks(dist, chk, dev)
    lo = -max_over_ranks(dist, -chk, dev)
    return {'workload': 'gaze_grcn data-parallel TRAINING step (per-GPU shape %d clips x T = %d): fwd + bwd + bucketed '
                        'gradient all-reduce (mean, side stream) + clip_by_global_norm(10) + TF-Adam' % (batch, n_steps),
            'clips_per_gpu': batch, 'n_lstm_steps': n_steps, 'steps': steps, 'warmup': warmup, 'dtype': dtype,
            'ms_per_step': round(elapsed / steps * 1e3, 4),
            'per_rank_ms': [round(t / steps * 1e3, 4) for t in per_rank],
            'frames_per_s': round(world * batch * n_steps * steps / elapsed, 1),
            'allreduce_bytes_per_step': int((reducer.bytes_reduced - bytes0) // max(steps, 1)),
            'allreduce_buckets_per_step': 3,
            'convgru': ('persistent' if persistent else 'per-step launches') + (' (fallback after RGP_ETIMEOUT)' if fallbacks else ''),
            'convgru_workgroups': head.persistent_workgroups if persistent else 0,
            'convgru_fallbacks': fallbacks,
            'top_bucket_release': 'before the BPTT launch' if head.grads_top_early else 'behind the BPTT launch',
            'backend': _backend(dist), 'ranks_seen': seen, 'world': world,
            'grad_norm_last': float(gnorm[0].item()), 'replicas_in_sync': bool(abs(hi - lo) <= 1e-9 * max(abs(hi), 1.0))}


def dp_finetune_probe(dist, device, rank=0, batch=16, n_steps=35, steps=3, warmup=1, dtype='bf16', seed=0, c3d_chunk=None,
                      model='cascade', per_step=False):
    """BASELINE config 5 at its per-GPU shape (16 clips x T = 35 per rank = 560 C3D windows) as ONE data-parallel
    training step: C3D forward -> cascade forward -> l2 loss -> cascade backward -> conv-stack backward, the gradient
    leaving in NINE buckets through GradBucketReducer's side stream as their producers are queued (the cascade's
    216 MB first, then conv5b ... conv1a, 110.6 MB: SURVEY 8e) -> finish -> global-norm clip over ALL variables +
    TF-Adam (base.py:286-297).  model='grcn': the gaze_grcn head instead of the cascade (its three head buckets + the
    eight conv buckets), for boxes short of memory.

    Timed like the headline (barrier, `steps` steps, barrier, MAX over ranks), then -- after the replicas' weights
    have been compared -- the same steps once more WITHOUT the reducer ('ms_per_step_no_allreduce': the replicas
    drift apart there, which is why it runs last): the difference is the communication the overlap did not hide."""
    import time
    from .finetune import EndToEndCascade, EndToEndGaze
    dev = torch.device(device)
    world = dist.get_world_size() if dist is not None else 1
    F = batch * n_steps
    chunk = min(c3d_chunk or F, F)
    g = torch.Generator(device=dev)
    g.manual_seed(977 + int(rank))
    video = torch.rand(F, 16, 112, 112, 3, device=dev, generator=g) - 0.5
    gt = torch.rand(batch, n_steps, 49, 49, device=dev, generator=g) + 1e-3
    if model == 'cascade':
        m = EndToEndCascade(batch, n_steps, dtype=dtype, device=dev, max_windows=chunk, seed=seed + 1)
        frames = torch.rand(batch, n_steps, 98, 98, 3, device=dev, generator=g)
        run = lambda: m.train_step(video, frames, gt, 1e-4)
    else:
        # (per_step: ConvGRU as per-timestep launches -- required when several ranks share one device, tests only)
        m = EndToEndGaze(batch, n_steps, dtype=dtype, device=dev, max_windows=chunk, seed=seed + 1, per_step=per_step)
        gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
        run = lambda: m.train_step(video, gt, 1e-4)
    m.attach_process_group(dist)
    last = [None, None]
    per_rank = [None]

    def timed(k):
        barrier(dist, dev)
        t0 = time.perf_counter()
        for _ in range(k):
            last[0], last[1] = run()
        barrier(dist, dev)
        mine = time.perf_counter() - t0
        per_rank[0] = [round(t / k * 1e3, 3) for t in gather_over_ranks(dist, mine, dev)]
        return max_over_ranks(dist, mine, dev)

    for _ in range(warmup):
        run()
    bytes0, n0 = m.reducer.bytes_reduced, m.reducer.buckets_reduced
    elapsed = timed(steps)
    per_rank_ms = per_rank[0]
    per_step_bytes = int((m.reducer.bytes_reduced - bytes0) // max(steps, 1))
    per_step_buckets = int((m.reducer.buckets_reduced - n0) // max(steps, 1))
    finite = bool(torch.isfinite(last[0])) and bool(torch.isfinite(last[1]))
    # a persistent ConvGRU launch that lost a member (model='grcn' only: the cascade's cells run per step) shows up as a
    # non-finite loss as well; the count of ranks that saw it is reported, the probe does not re-run
    fallbacks = 0
    for e in m.engines:
        if callable(getattr(e, 'status', None)) and hasattr(e, 'grad_buckets'):
            fallbacks += engine_timeouts_all_ranks(dist, e, dev)
    chk = float(sum(e.flat_params.double().abs().sum().item() for e in m.engines))
    hi = max_over_ranks(dist, chk, dev)
    lo = -max_over_ranks(dist, -chk, dev)
    seen = ranks_seen(dist, rank, dev)
    # the same steps with the collectives switched off (replicas diverge from here on: nothing may follow that needs them)
    m.attach_process_group(None)
    run()
    elapsed_off = timed(steps)
    grad_bytes = int(sum(e.flat_grads.numel() * 4 for e in m.engines))
    return {'workload': ('C3D conv stack + %s, data-parallel JOINT TRAINING step (BASELINE config 5 per-GPU shape): fwd + bwd, '
                         'gradient buckets all-reduced (mean) on a side stream as their producers are queued, then '
                         'clip_by_global_norm(10) over all variables + TF-Adam'
                         % ('gaze_grcn_cascade' if model == 'cascade' else 'gaze_grcn head')),
            'clips_per_gpu': batch, 'n_lstm_steps': n_steps, 'windows_per_gpu': F, 'c3d_chunk': chunk, 'steps': steps,
            'warmup': warmup, 'dtype': dtype,
            'ms_per_step': round(elapsed / steps * 1e3, 3), 'per_rank_ms': per_rank_ms,
            'convgru_fallbacks': fallbacks,
            'ms_per_step_no_allreduce': round(elapsed_off / steps * 1e3, 3),
            'exposed_allreduce_ms_per_step': round((elapsed - elapsed_off) / steps * 1e3, 3),
            'frames_per_s': round(world * F * steps / elapsed, 1),
            'allreduce_bytes_per_step': per_step_bytes, 'allreduce_buckets_per_step': per_step_buckets,
            'gradient_bytes': grad_bytes, 'backend': _backend(dist), 'ranks_seen': seen, 'world': world,
            'finite': finite, 'replicas_in_sync': bool(abs(hi - lo) <= 1e-9 * max(abs(hi), 1.0))}
