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
