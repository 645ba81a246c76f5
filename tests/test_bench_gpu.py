"""GPU: the driver's command line -- `python bench.py` with its defaults, shortened to 2 timed steps -- prints ONE JSON
line that keeps the contract the driver parses (metric / value / unit / roofline / cpu_baseline) and is self-consistent."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_keeps_the_contract(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--cpu-seconds', '3'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                    # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d['unit'] == 'frames/s' and d['metric'].startswith('frames/sec') and d['higher_is_better'] is True
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'bf16' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    # value = frames of the timed region / its duration: 64 clips x 16 steps per step
    assert abs(d['value'] - 1024 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and rf['peak'] == 2500.0
    assert 0.3 < rf['frac'] < 1.0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
    # conv2a: the dominant kernel of the step (either variant of its patch kernel, csrc/rgp_c3d_plan.h conv2a_slab)
    assert rf['kernel'].startswith(('conv_patch_bf16_kernel<64,128,56,16', 'conv_patch_slab_bf16_kernel<64,128,56,16'))
    assert rf['launches'] == 2 and rf['avg_launch_ms'] < d['ms_per_step']
    # algorithmic FLOPs of conv2a per launch of 1024 windows: 2 x 16 x 56^2 x 27 x 64 x 128 x 1024
    assert abs(rf['algorithmic_gflop_per_launch'] - 2 * 16 * 56 * 56 * 27 * 64 * 128 * 1024 / 1e9) < 1.0
    # traffic beyond L2 per launch (committed PMC summary): at least the layer's input + output, below 2x of it
    # (null + traffic_stale once the kernel's sources differ from the ones the summary was taken with)
    assert rf['traffic_unit'] == 'GB per launch' and rf['traffic_stale'] in (True, False)
    assert (rf['traffic'] is None) if rf['traffic_stale'] else (8.0 < rf['traffic'] < 16.0), rf
    cb = d['cpu_baseline']
    assert cb['kind'] in ('port', 'reference') and cb['unit'] == 'frames/s' and cb['cores'] >= 1 and cb['value'] > 0 and cb['sample']
    # the stage timers of the library cover the step
    assert abs(sum(d['stage_ms_per_step'].values()) - d['ms_per_step']) < 0.05 * d['ms_per_step']


def _rehearse(extra_env, extra_args=(), timeout=900):
    """`bench.py --gpus 2 --rehearse`: the driver's N > 1 control flow (spawned ranks, barrier + MAX over ranks, per-rank
    times, the three data-parallel probes, ONE line from rank 0) with gloo and both ranks on this GPU.  Small headline shape:
    the numbers mean nothing."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rehearse', '--steps', '2', '--warmup', '1',
           '--batch', '4', '--n-steps', '4'] + list(extra_args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
    return r, lines


def test_two_rank_rehearsal_prints_one_line_with_per_rank_times_and_all_three_probes(gpu):
    r, lines = _rehearse({})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rehearsal'] is True and 'dp_probe_error' not in d
    assert len(d['per_rank_ms']) == 2 and abs(max(d['per_rank_ms']) - d['ms_per_step']) < 1e-3
    # whole-job value: both ranks' frames over the MAX of their times
    assert abs(d['value'] - 2 * 16 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    assert 'cpu_baseline' not in d                                            # rank 0 at N = 1 only
    for key in ('dp_train', 'dp_train_b64', 'dp_finetune'):
        p = d[key]
        assert p['ranks_seen'] == 2 and p['replicas_in_sync'] is True and len(p['per_rank_ms']) == 2, (key, p)
        assert p['convgru_fallbacks'] == 0, (key, p)
    assert d['dp_train_b64']['convgru'] == 'per-step launches'                        # the rehearsal's plans, said so in the line


@pytest.mark.parametrize('fault,rc', [('raise:1', 3), ('raise:0', 4), ('hang:1', 3)])
def test_a_failing_probe_does_not_cost_the_headline(gpu, fault, rc):
    """Whatever happens in the N > 1 probes -- a peer raises (rank 0 hears of it through the c10d store), rank 0 raises, a peer
    never arrives (deadline) -- rank 0 prints its one line, with the complete headline and 'dp_probe_error', and the job fails."""
    r, lines = _rehearse({'RGP_BENCH_INJECT_PROBE_FAULT': fault}, ('--probe-timeout', '25'), timeout=600)
    assert r.returncode != 0, r.stdout[-2000:]
    assert len(lines) == 1, (lines, r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['roofline']['frac'] > 0 and len(d['per_rank_ms']) == 2
    want = 'did not finish within 25 s' if fault.startswith('hang') else 'rank %s: RuntimeError: injected probe fault' % fault[-1]
    assert want in d['dp_probe_error'], d['dp_probe_error']
