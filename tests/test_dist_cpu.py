"""CPU, world_size 2, gloo: the multi-rank plumbing bench.py relies on (clip sharding,
barrier, max-over-ranks clock, mean all-reduce)."""
import os
import subprocess
import sys

import pytest

from recurrent_gaze_prediction_amd import dist as rdist

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
                          '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)],
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
