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
    # config 5's leg: the cascade's flat buffer + eight conv buckets; the gaze_grcn variant: three head + eight conv buckets
    fp, fg = out['finetune_probe'], out['finetune_probe_grcn']
    for q, buckets in ((fp, 9), (fg, 11)):
        assert q['backend'] == 'nccl' and q['ranks_seen'] == 1 and q['replicas_in_sync'] and q['finite'], q
        assert q['allreduce_buckets_per_step'] == buckets and q['allreduce_bytes_per_step'] == q['gradient_bytes'], q
        assert q['ms_per_step'] > 0 and q['ms_per_step_no_allreduce'] > 0, q
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
    assert pr['convgru'] == 'per-step launches', pr
    ft = out['finetune_probe']
    assert ft['world'] == 2 and ft['ranks_seen'] == 2 and ft['replicas_in_sync'] and ft['finite'], ft
    assert ft['allreduce_buckets_per_step'] == 11 and ft['allreduce_bytes_per_step'] == ft['gradient_bytes'], ft
