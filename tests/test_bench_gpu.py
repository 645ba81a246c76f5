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
