"""Rank of tests/test_dist_gpu.py::test_two_ranks_on_half_batches_reproduce_the_full_batch_step (started by
torch.distributed.run, 2 ranks, backend gloo, both on cuda:0 -- two RCCL ranks cannot share a device, gloo can).

The REAL engines under world size 2: rank r runs EndToEndGaze on clip r of a 2-clip batch (forward, backward, the
bucketed reducer with the library's per-layer events, finish, global-norm clip + Adam); rank 0 also runs the
single-process step on both clips.  The loss is a sum over clips divided by the clip count (gaze_rnn.py:406-407) and no
operator couples clips (inference-mode batch-norm), so the mean of the two ranks' gradients is the full-batch gradient
(base.py:286-292 on the global batch) up to the summation order of the filter gradients.  Rank 0 prints one JSON line.

Both ranks sit on ONE device here, the configuration include/rgp.h rules out for the persistent ConvGRU kernels (two
such launches would compete for the CUs): every head in this file runs its recurrence per step (RGP_GRCN_PER_STEP)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def main():
    from recurrent_gaze_prediction_amd import dist as rdist
    from recurrent_gaze_prediction_amd import synthetic as syn
    from recurrent_gaze_prediction_amd.finetune import EndToEndGaze
    rank, _, world = rdist.env_world()
    assert world == 2
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    dist = rdist.init(backend='gloo', device=dev)
    B, T = 2, 2
    p3, ph = syn.c3d_params(71), syn.grcn_params(72, T, gru_std=0.05, random_bn=True)
    rs = np.random.RandomState(73)
    video = torch.tensor((rs.rand(B * T, 16, 112, 112, 3).astype(np.float32) - 0.5) * 2, device=dev)
    gt, _ = syn.gaze_maps(74, B, T)
    gt = torch.tensor((gt / gt.sum(axis=(2, 3), keepdims=True)).astype(np.float32), device=dev)

    def grads_of(m, v, lab, reduce):
        logits, probs = m.forward(v)
        loss = m.backward(v, logits, probs, lab)
        if reduce:
            m.reducer.finish()
        torch.cuda.synchronize()
        return float(loss), torch.cat([m.c3d.flat_grads, m.head.flat_grads]).double().clone()

    # this rank's clip
    m = EndToEndGaze(1, T, dtype='bf16', device=dev, c3d_params=p3, grcn_params=ph, per_step=True)
    m.attach_process_group(dist)
    v, lab = video[rank * T:(rank + 1) * T], gt[rank:rank + 1]
    loss_r, g_dp = grads_of(m, v, lab, True)
    loss2, gnorm = m.train_step(v, lab, 1e-4, max_grad_norm=10.0)
    torch.cuda.synchronize()
    par_dp = torch.cat([m.c3d.flat_params, m.head.flat_params]).double().clone()
    loss_mean = rdist.sum_over_ranks(dist, loss_r) / world
    rdist.barrier(dist, dev)
    out = None
    if rank == 0:
        ref = EndToEndGaze(B, T, dtype='bf16', device=dev, c3d_params=p3, grcn_params=ph, per_step=True)
        loss_ref, g_ref = grads_of(ref, video, gt, False)
        _, gnorm_ref = ref.train_step(video, gt, 1e-4, max_grad_norm=10.0)
        torch.cuda.synchronize()
        par_ref = torch.cat([ref.c3d.flat_params, ref.head.flat_params]).double()
        out = {'loss_dp_mean': loss_mean, 'loss_ref': loss_ref,
               'grad_rms_rel': float(((g_dp - g_ref) ** 2).mean().sqrt() / (g_ref ** 2).mean().sqrt()),
               'grad_max_rel': float((g_dp - g_ref).abs().max() / g_ref.abs().max()),
               'gnorm_dp': float(gnorm), 'gnorm_ref': float(gnorm_ref),
               # Adam's first step is lr * sign-like: parameters moved by +-1e-4 where the gradient is not noise
               'param_step_agree': float(((par_dp - par_ref).abs() < 2e-5).double().mean()),
               'bytes_reduced': int(m.reducer.bytes_reduced)}
    rdist.barrier(dist, dev)
    # bench.py's N > 1 leg (dist.dp_train_probe) under world size 2: both ranks must be seen and stay in sync
    probe = rdist.dp_train_probe(dist, dev, rank=rank, batch=2, n_steps=3, steps=2, warmup=1, per_step=True)
    # ... and bench.py's config-5 leg (dist.dp_finetune_probe) in its gaze_grcn form: 3 head + 8 conv buckets per step; every
    # rank trains on its own clip, so equal weights afterwards mean the averaged gradients were applied on both
    ft = rdist.dp_finetune_probe(dist, dev, rank=rank, batch=1, n_steps=2, steps=2, warmup=1, model='grcn', per_step=True)
    if rank == 0:
        out['probe'] = probe
        out['finetune_probe'] = ft
        print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
