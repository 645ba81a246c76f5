"""GPU: the RCCL leg of dist.py on hardware, as far as one GPU allows (a one-rank `nccl` process group in a child
process; see tests/rccl_world1_child.py).  The N > 1 arithmetic is covered on CPU by tests/test_dist_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_one_rank_group_runs_the_reducer(gpu):
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rccl_world1_child.py')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, child], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['backend'] == 'nccl' and out['world'] == 1
    assert out['max'] == 1.25 and out['sum'] == 2.5 and out['avg_identity']
    assert out['loss'][0] == out['loss'][1], out                 # forward: deterministic
    assert out['grad_rms_rel'] < 1e-5, out
    # two optimizer steps later the runs have drifted apart by the atomics' summation order (Adam's first step is
    # lr * sign(g): noise-level gradients flip), so only the level is compared
    assert abs(out['gnorm'][0] - out['gnorm'][1]) < 5e-2 * abs(out['gnorm'][0]), out
    assert abs(out['loss_after_2_steps'][0] - out['loss_after_2_steps'][1]) < 1e-2 * abs(out['loss_after_2_steps'][0]), out
    # the head's flat buffer + the eight conv layers' buckets went through RCCL every step
    assert out['bytes_reduced_per_step'] > 100e6, out
    pr = out['probe']
    assert pr['backend'] == 'nccl' and pr['ranks_seen'] == 1 and pr['replicas_in_sync'], pr
    assert 11e6 < pr['allreduce_bytes_per_step'] < 13e6 and pr['ms_per_step'] > 0 and pr['grad_norm_last'] > 0, pr
    assert pr['allreduce_buckets_per_step'] == 3 and pr['convgru'] == 'persistent', pr
    # config 4's 8 clips per GPU leave 192 CUs to the collective: the first bucket leaves ahead of the BPTT launch
    assert pr['convgru_workgroups'] == 64 and pr['top_bucket_release'] == 'before the BPTT launch', pr
    assert pr['convgru_fallbacks'] == 0 and len(pr['per_rank_ms']) == 1 and pr['per_rank_ms'][0] <= pr['ms_per_step'] + 1e-3, pr
    # config 3's 64 clips per GPU: full-chip persistent launches (256 workgroups) with the reducer live through RCCL
    pb = out['probe_b64']
    assert pb['backend'] == 'nccl' and pb['replicas_in_sync'] and pb['convgru'] == 'persistent' and pb['convgru_fallbacks'] == 0, pb
    assert pb['convgru_workgroups'] == 256 and pb['top_bucket_release'] == 'behind the BPTT launch', pb
    assert 11e6 < pb['allreduce_bytes_per_step'] < 13e6 and pb['grad_norm_last'] > 0, pb
    # config 5's leg: the cascade's flat buffer + eight conv buckets; the gaze_grcn variant: three head + eight conv buckets
    fp, fg = out['finetune_probe'], out['finetune_probe_grcn']
    for q, buckets in ((fp, 9), (fg, 11)):
        assert q['backend'] == 'nccl' and q['ranks_seen'] == 1 and q['replicas_in_sync'] and q['finite'], q
        assert q['allreduce_buckets_per_step'] == buckets and q['allreduce_bytes_per_step'] == q['gradient_bytes'], q
        assert q['ms_per_step'] > 0 and q['ms_per_step_no_allreduce'] > 0, q
        assert q['convgru_fallbacks'] == 0 and len(q['per_rank_ms']) == 1, q
    assert fp['gradient_bytes'] > 300e6 and 120e6 < fg['gradient_bytes'] < 125e6, (fp, fg)


def test_two_ranks_on_half_batches_reproduce_the_full_batch_step(gpu):
    """World size 2 with the real engines (gloo, both ranks on this GPU): tests/dp2_gpu_child.py."""
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp2_gpu_child.py')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # a port nobody listens on right now
        sk.bind(('127.0.0.1', 0))
        port = str(sk.getsockname()[1])
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', port, child],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert abs(out['loss_dp_mean'] - out['loss_ref']) < 1e-5 * abs(out['loss_ref']), out
    assert out['grad_rms_rel'] < 1e-5 and out['grad_max_rel'] < 1e-4, out
    assert abs(out['gnorm_dp'] - out['gnorm_ref']) < 1e-4 * out['gnorm_ref'], out
    assert out['param_step_agree'] > 0.99, out
    assert out['bytes_reduced'] > 2 * 100e6, out                 # two steps' worth of buckets went through the group
    pr = out['probe']
    assert pr['world'] == 2 and pr['ranks_seen'] == 2 and pr['replicas_in_sync'], pr
    assert 11e6 < pr['allreduce_bytes_per_step'] < 13e6, pr
    assert pr['convgru'] == 'per-step launches' and pr['convgru_fallbacks'] == 0 and pr['convgru_workgroups'] == 0, pr
    assert len(pr['per_rank_ms']) == 2 and abs(max(pr['per_rank_ms']) - pr['ms_per_step']) < 1e-3, pr
    ft = out['finetune_probe']
    assert ft['world'] == 2 and ft['ranks_seen'] == 2 and ft['replicas_in_sync'] and ft['finite'], ft
    assert ft['allreduce_buckets_per_step'] == 11 and ft['allreduce_bytes_per_step'] == ft['gradient_bytes'], ft
    assert len(ft['per_rank_ms']) == 2 and ft['convgru_fallbacks'] == 0, ft


def test_probe_falls_back_and_says_so_when_a_persistent_launch_loses_a_member(gpu):
    """bench.py's data-parallel probe under the one failure it exists to report: the BPTT launch of the first timed step
    loses a workgroup (fault injection; on an 8-GPU node: a CU held by a collective for the whole deadline).  The probe
    re-runs on per-timestep launches from the initial weights and reports the fall-back instead of a silent slower number."""
    from recurrent_gaze_prediction_amd import dist as rdist
    pr = rdist.dp_train_probe(None, gpu, rank=0, batch=2, n_steps=3, steps=2, warmup=1, inject_fault='bptt')
    assert pr['convgru_fallbacks'] == 1 and pr['convgru'] == 'per-step launches (fallback after RGP_ETIMEOUT)', pr
    assert pr['convgru_workgroups'] == 0 and pr['replicas_in_sync'] and pr['grad_norm_last'] > 0, pr
    ok = rdist.dp_train_probe(None, gpu, rank=0, batch=2, n_steps=3, steps=2, warmup=1)
    assert ok['convgru_fallbacks'] == 0 and ok['convgru'] == 'persistent' and ok['convgru_workgroups'] == 16, ok
    # same seeds, same data: the fall-back run trained the same model (per-step and persistent kernels agree to bf16 rounding)
    assert abs(ok['grad_norm_last'] - pr['grad_norm_last']) < 5e-2 * ok['grad_norm_last'], (ok, pr)


def test_top_bucket_release_follows_the_co_residency_rule(gpu):
    """include/rgp.h: the first gradient bucket is released ahead of the persistent BPTT launch only when that launch leaves
    RGP_RCCL_CU_RESERVE (64) CUs free; the events are ordered accordingly on the stream."""
    from recurrent_gaze_prediction_amd import synthetic as syn
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    import torch
    n_cu = torch.cuda.get_device_properties(gpu).multi_processor_count
    for B, per_step in ((8, False), (24, False), (25, False), (64, False), (64, True)):
        eng = GrcnEngine(B, 2, dtype='bf16', save_for_backward=True, device=gpu, per_step=per_step)
        wg = 0 if per_step else 8 * ((B + (B + 31) // 32 - 1) // ((B + 31) // 32))
        assert eng.persistent_workgroups == wg, (B, per_step, eng.persistent_workgroups)
        assert eng.grads_top_early == (per_step or wg <= n_cu - 64), (B, per_step, n_cu)
    # the full-chip plan still delivers every bucket: gradients after reduce_buckets + finish on a side stream == flat_grads
    eng = GrcnEngine(64, 2, dtype='bf16', save_for_backward=True, device=gpu)
    eng.set_weights(syn.grcn_params(5, 2, gru_std=0.05, random_bn=True))
    x = torch.tensor(syn.c3d_features(6, 64, 2), device=gpu)
    gt = torch.rand(64, 2, 49, 49, device=gpu) + 1e-3
    gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
    z, p = eng.forward(x)
    eng.backward(z, p, gt)
    side = torch.cuda.Stream(gpu)
    copies = []
    with torch.cuda.stream(side):
        for bucket, ready in eng.grad_buckets():       # completion order (TOP, GRU, PROJ), not the flat buffer's order
            ready(side)
            copies.append((bucket, bucket.clone()))
    torch.cuda.current_stream(gpu).wait_stream(side)
    torch.cuda.synchronize()
    eng.status()
    assert sum(b.numel() for b, _ in copies) == eng.flat_grads.numel()
    for bucket, copy in copies:                         # what the side stream saw behind `ready` is the final gradient
        assert torch.equal(bucket, copy) and bool(torch.isfinite(copy).all()) and float(copy.abs().max()) > 0
