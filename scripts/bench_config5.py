#!/usr/bin/env python
"""BASELINE config 5 (models/gaze_grcn_cascade.py: stacked ConvGRU + end-to-end C3D fine-tune, 35-frame clips):
one training step = C3D forward -> cascade forward -> l2 loss -> cascade backward -> C3D backward ->
[gradient all-reduce] -> global-norm clip + TF-Adam on all variables.  Same launch contract as bench.py
(torch.distributed.run for N > 1); rank 0 prints one JSON line.  Not the driver's headline bench."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from recurrent_gaze_prediction_amd import dist as rdist                      # noqa: E402
from recurrent_gaze_prediction_amd.finetune import EndToEndCascade          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=16, help='clips per GPU (config 5: 128 global / 8 GPUs)')
    ap.add_argument('--n-steps', type=int, default=35)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--c3d-chunk', type=int, default=560)
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:      # no launcher: start the ranks ourselves (before any GPU call)
        raise SystemExit(rdist.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    rank, local_rank, world = rdist.env_world()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = rdist.init(backend='nccl', device=dev)
    B, T, F = args.batch, args.n_steps, args.batch * args.n_steps
    m = EndToEndCascade(B, T, dtype=args.dtype, device=dev, max_windows=min(args.c3d_chunk, F), seed=1)
    m.attach_process_group(dist)
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    video = torch.rand(F, 16, 112, 112, 3, device=dev, generator=g) - 0.5
    frames = torch.rand(B, T, 98, 98, 3, device=dev, generator=g)
    gt = torch.rand(B, T, 49, 49, device=dev, generator=g)
    for _ in range(args.warmup):
        m.train_step(video, frames, gt, 1e-4)
    rdist.barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, gnorm = m.train_step(video, frames, gt, 1e-4)
    rdist.barrier(dist, dev)
    el = rdist.max_over_ranks(dist, time.perf_counter() - t0, dev)
    assert bool(torch.isfinite(loss)) and bool(torch.isfinite(gnorm))
    if rank == 0:
        print(json.dumps({
            'metric': 'frames/sec, cascade end-to-end training step (BASELINE config 5)', 'value': round(world * F * args.steps / el, 2),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(el / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'dtype': args.dtype,
            'data': 'synthetic', 'config': {'workload': 'C3D + gaze_grcn_cascade joint training step', 'clips_per_gpu': B,
                                            'n_lstm_steps': T, 'windows_per_gpu': F, 'c3d_chunk': m.c3d.max_windows},
            'loss': float(loss), 'grad_norm': float(gnorm)}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
