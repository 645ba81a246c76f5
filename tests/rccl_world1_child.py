"""Child process of tests/test_dist_gpu.py: a ONE-rank RCCL process group on cuda:0.

The build boxes have a single GPU, and two RCCL ranks cannot share a device, so the N > 1 data path (dist.py) is
exercised here with world size 1: the `nccl` branch of dist.init (device_id=...), the barrier / MAX plumbing on
device tensors, the in-place AVG all-reduce of the flat gradient buffers and the GradBucketReducer's side stream
with the C3D library's per-layer events (rgp_c3d_wait_layer_grads) all run through RCCL on the hardware; with one
rank the mean is the identity, so the gradients with the process group attached must equal those without it (up to the
summation order of the filter gradients' fp32 atomics).  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def main():
    os.environ.update({'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1', 'MASTER_ADDR': '127.0.0.1',
                       'MASTER_PORT': str(29600 + os.getpid() % 1000)})
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from recurrent_gaze_prediction_amd import dist as rdist
    from recurrent_gaze_prediction_amd import synthetic as syn
    from recurrent_gaze_prediction_amd.finetune import EndToEndGaze
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    dist = rdist.init(backend='nccl', device=dev, force=True)
    out = {'backend': dist.get_backend(), 'world': dist.get_world_size()}
    rdist.barrier(dist, dev)
    out['max'] = rdist.max_over_ranks(dist, 1.25, dev)
    out['sum'] = rdist.sum_over_ranks(dist, 2.5, dev)
    t = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    rdist.allreduce_mean_(dist, [t, t[:1000]])
    out['avg_identity'] = bool(torch.equal(t, torch.arange(1 << 20, dtype=torch.float32, device=dev)))

    B, T = 1, 2
    p3, ph = syn.c3d_params(61), syn.grcn_params(62, T, gru_std=0.05, random_bn=True)
    rs = np.random.RandomState(63)
    video = torch.tensor((rs.rand(B * T, 16, 112, 112, 3).astype(np.float32) - 0.5) * 2, device=dev)
    gt, _ = syn.gaze_maps(64, B, T)
    gt = torch.tensor((gt / gt.sum(axis=(2, 3), keepdims=True)).astype(np.float32), device=dev)
    res = []
    for attach in (False, True):
        m = EndToEndGaze(B, T, dtype='bf16', device=dev, c3d_params=p3, grcn_params=ph)
        if attach:
            m.attach_process_group(dist)
        logits, probs = m.forward(video)
        loss = m.backward(video, logits, probs, gt)
        if attach:
            m.reducer.finish()
        torch.cuda.synchronize()
        grads = torch.cat([m.c3d.flat_grads, m.head.flat_grads]).double()
        for _ in range(2):
            loss2, gnorm = m.train_step(video, gt, 1e-4, max_grad_norm=10.0)
        torch.cuda.synchronize()
        res.append((float(loss), grads, float(loss2), float(gnorm), m.reducer.bytes_reduced if attach else 0))
    out['loss'] = [res[0][0], res[1][0]]
    # the filter gradients are summed with fp32 atomics (order varies from launch to launch): RMS level, not bits
    out['grad_rms_rel'] = float(((res[0][1] - res[1][1]) ** 2).mean().sqrt() / (res[0][1] ** 2).mean().sqrt())
    out['loss_after_2_steps'] = [res[0][2], res[1][2]]
    out['gnorm'] = [res[0][3], res[1][3]]
    out['bytes_reduced_per_step'] = res[1][4] // 3
    # bench.py's N > 1 leg (dist.dp_train_probe) through RCCL with the forced one-rank group, at config 4's shape
    out['probe'] = rdist.dp_train_probe(dist, dev, rank=0, batch=8, n_steps=35, steps=3, warmup=1)
    # ... at config 3's per-GPU shape: 256-workgroup persistent forward + BPTT launches with the three-bucket reducer live
    out['probe_b64'] = rdist.dp_train_probe(dist, dev, rank=0, batch=64, n_steps=16, steps=3, warmup=1)
    # ... and the config-5 leg (dist.dp_finetune_probe: nine + buckets through the side stream), at a small shape
    out['finetune_probe'] = rdist.dp_finetune_probe(dist, dev, rank=0, batch=1, n_steps=2, steps=2, warmup=1, model='cascade')
    out['finetune_probe_grcn'] = rdist.dp_finetune_probe(dist, dev, rank=0, batch=1, n_steps=2, steps=2, warmup=1, model='grcn')
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
