"""CPU, world_size 2, gloo: the multi-rank plumbing bench.py relies on (clip sharding,
barrier, max-over-ranks clock, mean all-reduce)."""
import os
import subprocess
import sys

import pytest

from recurrent_gaze_prediction_amd import dist as rdist


def _free_port():
    """A TCP port nobody listens on right now (a fixed rendezvous port fails when an earlier run's socket still lingers)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, json, torch
sys.path.insert(0, %r)
from recurrent_gaze_prediction_amd import dist as rdist
rank, local, world = rdist.env_world()
d = rdist.init(backend='gloo')
assert d is not None and d.get_world_size() == 2
lo, hi = rdist.shard_clips(7, rank, world)
rdist.barrier(d)
elapsed = rdist.max_over_ranks(d, 1.0 + rank)             # rank 1 is the slow one
total = rdist.sum_over_ranks(d, hi - lo)
g = [torch.full((3,), float(rank + 1)), torch.full((2, 2), float(10 * (rank + 1)))]
rdist.allreduce_mean_(d, g)
rdist.barrier(d)
if rank == 0:
    print(json.dumps({'elapsed': elapsed, 'total': total, 'lo': lo, 'hi': hi,
                      'g0': g[0].tolist(), 'g1': g[1].flatten().tolist()}))
d.destroy_process_group()
''' % ROOT


def test_shard_clips_is_a_balanced_partition():
    for n in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [rdist.shard_clips(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_control_plane(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    r = json.loads(line)
    assert r['elapsed'] == 2.0 and r['total'] == 7.0 and (r['lo'], r['hi']) == (0, 4)
    assert r['g0'] == [1.5] * 3 and r['g1'] == [15.0] * 4


def test_single_process_is_a_noop():
    assert rdist.max_over_ranks(None, 3.5) == 3.5
    assert rdist.shard_clips(5, 0, 1) == (0, 5)


# ---- data-parallel training step: two ranks on half batches == one rank on the full batch -----------------------------
DP_WORKER = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, %r)
from recurrent_gaze_prediction_amd import dist as rdist
from oracle import torch_ref

# Stub engine with the engines' contract (flat_params / flat_grads views, backward fills the flat gradient of the
# per-rank MEAN loss): a linear read-out  loss = sum_clips 0.5 ||x W - y||^2 / n_clips  -- like the gaze loss it is a
# sum over clips divided by the clip count (gaze_rnn.py:406-407), so the mean of the two half-batch gradients is the
# full-batch gradient, which is what the all-reduce before the clip must deliver (base.py:286-292 on the global batch).
class Stub(object):
    def __init__(self, W):
        self.flat_params = torch.tensor(W, dtype=torch.float64).reshape(-1).clone()
        self.flat_grads = torch.zeros_like(self.flat_params)
        self.shape = W.shape
    def backward(self, x, y):
        W = self.flat_params.view(self.shape)
        r = x @ W - y
        self.flat_grads.copy_((x.t() @ r / x.shape[0]).reshape(-1))

def train_step(eng, m, v, step, x, y, reducer):
    eng.backward(x, y)
    half = eng.flat_grads.numel() // 2           # two buckets, like the per-layer buckets of the conv stack
    # completion order, with `ready` callbacks (ignored off the GPU): GrcnEngine.grad_buckets()'s form
    reducer.reduce_buckets([(eng.flat_grads[half:], lambda stream: 1 / 0), (eng.flat_grads[:half], None)])
    reducer.finish()
    g, norm = torch_ref.clip_by_global_norm({'w': eng.flat_grads.clone()}, 0.05)
    p, m, v = torch_ref.adam_step_tf({'w': eng.flat_params}, g, m, v, step, 1e-2)
    eng.flat_params = p['w']
    return norm, m, v

rank, local, world = rdist.env_world()
d = rdist.init(backend='gloo') if world > 1 else None
rs = np.random.RandomState(3)
W0 = rs.randn(6, 4)
X = torch.tensor(rs.randn(8, 6)); Y = torch.tensor(rs.randn(8, 4))
lo, hi = rdist.shard_clips(8, rank, world)
eng = Stub(W0)
red = rdist.GradBucketReducer(d, 'cpu')
m, v = {'w': torch.zeros(24, dtype=torch.float64)}, {'w': torch.zeros(24, dtype=torch.float64)}
norms = []
for step in range(3):
    n, m, v = train_step(eng, m, v, step, X[lo:hi], Y[lo:hi], red)
    norms.append(n)
# the per-rank flip-augmentation seeds differ and are recorded (gaze_rnn.py:504-510 draws from one global RNG)
from recurrent_gaze_prediction_amd.models import gaze_rnn
seed = (0 * 1000003 + 7919 * rank + 12345) & 0x7fffffff
# a failure on ONE rank is seen by all of them (dp_train_probe exits every rank together)
agree = [rdist.all_ranks_ok(d, True), rdist.all_ranks_ok(d, rank != world - 1 or world == 1)]
# a persistent-launch time-out on ONE rank is counted on all of them (dp_train_probe: every rank then rebuilds its engine on
# per-step launches together); per_rank clocks come back in rank order
from recurrent_gaze_prediction_amd import _lib
class TimedOut(object):
    def __init__(self, fail): self.fail = fail
    def status(self):
        if self.fail:
            e = _lib.RgpError('lost member'); e.code = _lib.RGP_ETIMEOUT
            raise e
lost = [rdist.engine_timeouts_all_ranks(d, TimedOut(world > 1 and rank == world - 1), 'cpu'),
        rdist.engine_timeouts_all_ranks(d, TimedOut(False), 'cpu')]
clocks = rdist.gather_over_ranks(d, 10.0 + rank)
# (ONE write per rank: print() emits the text and the newline separately, and two ranks share this pipe)
sys.stdout.write(json.dumps({'rank': rank, 'world': world, 'W': eng.flat_params.tolist(), 'norms': norms, 'flip_seed': seed,
                             'bytes': red.bytes_reduced, 'buckets': red.buckets_reduced, 'agree': agree, 'lost': lost,
                             'clocks': clocks}) + '\n')
sys.stdout.flush()
if d is not None:
    d.barrier(); d.destroy_process_group()
''' % ROOT


def _run_dp(tmp_path, world, port):
    script = tmp_path / ('dp_worker_%d.py' % world)
    script.write_text(DP_WORKER)
    if world == 1:
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    else:
        out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
                              '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
                             capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    return [json.loads(l) for l in out.stdout.splitlines() if l.startswith('{')]


def test_two_rank_dp_step_reproduces_the_full_batch_step(tmp_path):
    import numpy as np
    one = _run_dp(tmp_path, 1, 0)[0]
    two = sorted(_run_dp(tmp_path, 2, _free_port()), key=lambda r: r['rank'])
    assert [r['world'] for r in two] == [2, 2]
    for r in two:        # every rank ends with the weights of the single-process run on all 8 clips
        assert np.allclose(r['W'], one['W'], rtol=0, atol=1e-12)
        assert np.allclose(r['norms'], one['norms'], rtol=1e-12)      # the clip saw the global-batch gradient
        assert r['bytes'] == 3 * 24 * 8 and r['buckets'] == 6         # both buckets, every step, reduced in place
        assert r['agree'] == [True, False]                            # the last rank's failure reaches rank 0 too
        assert r['lost'] == [1, 0] and r['clocks'] == [10.0, 11.0]   # rank 1's time-out is seen by rank 0; clocks in rank order
    assert one['agree'] == [True, True] and one['lost'] == [0, 0] and one['clocks'] == [10.0]
    assert one['norms'][0] > 0.05                                     # the clip was active
    assert two[0]['flip_seed'] != two[1]['flip_seed']


def test_spawn_ranks_starts_one_process_per_rank(tmp_path):
    """`python bench.py --gpus N` without a launcher goes through dist.spawn_ranks: N fresh ranks, rank 0's stdout
    relayed, the launcher's exit code returned."""
    script = tmp_path / 'echo_rank.py'
    # (one write() per line: print() writes the text and the newline separately, and two ranks share this pipe)
    script.write_text("import os, sys\nsys.stdout.write('RANK %s of %s args %s\\n' % (os.environ['RANK'], os.environ['WORLD_SIZE'], sys.argv[1:]))\nsys.stdout.flush()\n"
                      "sys.exit(3 if '--fail' in sys.argv and os.environ['RANK'] == '1' else 0)\n")
    code = ("import sys; sys.path.insert(0, %r)\nfrom recurrent_gaze_prediction_amd import dist as rdist\n"
            "sys.exit(rdist.spawn_ranks(2, %r, sys.argv[1:]))\n" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, '-c', code, '--steps', '5'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = sorted(l for l in out.stdout.splitlines() if l.startswith('RANK'))
    assert lines == ["RANK 0 of 2 args ['--steps', '5']", "RANK 1 of 2 args ['--steps', '5']"]
    bad = subprocess.run([sys.executable, '-c', code, '--fail'], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


def test_bench_refuses_a_world_size_that_is_not_gpus(tmp_path):
    """bench.py under a launcher with WORLD_SIZE != --gpus exits non-zero before touching the GPU."""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and 'WORLD_SIZE=1' in (out.stderr + out.stdout)
