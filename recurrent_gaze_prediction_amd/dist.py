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

    def reduce(self, bucket, ready=None):
        if self.dist is None:
            return
        self.bytes_reduced += bucket.numel() * bucket.element_size()
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


def dp_train_probe(dist, device, rank=0, batch=8, n_steps=35, steps=5, warmup=2, dtype='bf16', seed=0):
    """BASELINE config 4 at its per-GPU shape (B = 8 clips x T = 35 per rank) as ONE data-parallel training step:
    gaze_grcn forward + backward on the rank's own clips -> GradBucketReducer (the flat 12 MB fp32 gradient, RCCL AVG
    on a side stream) -> finish -> clip_by_global_norm(10) + TF-Adam (base.py:286-297).  Timed like the headline
    (barrier, `steps` steps, barrier, MAX over ranks).  Every rank starts from the same weights and sees different
    clips, so after the steps the weights must still be identical on all ranks -- checked with a MAX/MIN all-reduce of
    a checksum.  Returns a dict (same on every rank)."""
    import time
    from . import synthetic as syn
    from .engine import GrcnEngine
    dev = torch.device(device)
    world = dist.get_world_size() if dist is not None else 1
    head = GrcnEngine(batch, n_steps, dtype=dtype, save_for_backward=True, device=dev)
    head.set_weights(syn.grcn_params(seed + 1, n_steps))
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + int(rank))
    x = torch.relu(torch.randn(batch, n_steps, 1024, 7, 7, device=dev, generator=g))
    gt = torch.rand(batch, n_steps, 49, 49, device=dev, generator=g) + 1e-3
    gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
    logits = torch.empty(batch, n_steps, 49, 49, device=dev)
    probs = torch.empty_like(logits)
    reducer = GradBucketReducer(dist, dev)
    k = [0]
    gnorm = [None]

    def step():
        head.forward(x, out_logits=logits, out_probs=probs)
        head.backward(logits, probs, gt)
        reducer.reduce(head.flat_grads)
        reducer.finish()
        gnorm[0] = head.adam_step(k[0], 1e-4 * 0.8 ** (k[0] // 500), max_grad_norm=10.0)
        k[0] += 1

    for _ in range(warmup):
        step()
    barrier(dist, dev)
    bytes0 = reducer.bytes_reduced
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier(dist, dev)
    elapsed = max_over_ranks(dist, time.perf_counter() - t0, dev)
    head.status()
    seen = ranks_seen(dist, rank, dev)
    # replicas stay replicas: same weights on every rank after the averaged steps
    chk = float(head.flat_params.double().abs().sum().item())
    hi = max_over_ranks(dist, chk, dev)
    lo = -max_over_ranks(dist, -chk, dev)
    return {'workload': 'gaze_grcn data-parallel TRAINING step (BASELINE config 4 per-GPU shape): fwd + bwd + bucketed '
                        'gradient all-reduce (mean, side stream) + clip_by_global_norm(10) + TF-Adam',
            'clips_per_gpu': batch, 'n_lstm_steps': n_steps, 'steps': steps, 'warmup': warmup, 'dtype': dtype,
            'ms_per_step': round(elapsed / steps * 1e3, 4),
            'frames_per_s': round(world * batch * n_steps * steps / elapsed, 1),
            'allreduce_bytes_per_step': int((reducer.bytes_reduced - bytes0) // max(steps, 1)),
            'backend': _backend(dist), 'ranks_seen': seen, 'world': world,
            'grad_norm_last': float(gnorm[0].item()), 'replicas_in_sync': bool(abs(hi - lo) <= 1e-9 * max(abs(hi), 1.0))}
