"""One-process-per-GPU plumbing (torch.distributed; backend 'nccl' is RCCL on ROCm).

The gaze path shards on the clip (batch) axis only: every sample's
ConvGRU state is private, batch-norm is inference-mode, so inference needs no
data-path collective -- ranks are replicas over disjoint clips.  What the ranks do
share is control: a barrier around timed regions, a MAX over ranks of the elapsed
time, and (training) the gradient all-reduce issued before the global-norm clip.
"""
import os

import torch


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init(backend=None, device=None):
    """Initialise the default process group when WORLD_SIZE > 1; returns the module or None."""
    rank, _, world = env_world()
    if world <= 1:
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


def allreduce_mean_(dist, tensors):
    """In-place mean of each tensor over ranks (gradient averaging before the clip,
    base.py:286-292 semantics on the global batch).  One flat bucket per call."""
    if dist is None or not tensors:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return tensors
