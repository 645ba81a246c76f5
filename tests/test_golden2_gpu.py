"""GPU: the HIP paths of the later-added models against the committed golden fixtures
(tests/golden/make_golden.py new: float64 oracle outputs; inputs and weights regenerate from seeds)."""
import os

import numpy as np
import pytest
import torch

from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize('dtype,tol,gtol', [('f32', 2e-4, 2e-3), ('bf16', 6e-2, 2e-1)])
def test_cascade_against_golden(gpu, dtype, tol, gtol):
    from recurrent_gaze_prediction_amd.engine import CascadeEngine
    g = np.load(os.path.join(GOLD, 'cascade_small.npz'))
    B, T, seed = [int(v) for v in g['config']]
    p = syn.cascade_params(seed)
    rs = np.random.RandomState(seed + 7)
    frames = rs.rand(B, T, 98, 98, 3).astype(np.float32)
    c3d = syn.c3d_features(seed + 8, B, T)
    gt, _ = syn.gaze_maps(seed + 9, B, T)
    gt = (gt / gt.max()).astype(np.float32)
    eng = CascadeEngine(B, T, 98, dtype=dtype, device=gpu, save_for_backward=True)
    eng.set_weights(p)
    maps = eng.forward(torch.tensor(frames, device=gpu), torch.tensor(c3d, device=gpu))
    assert rel(maps.cpu().numpy(), g['maps']) < tol
    loss = 0.5 * float(((maps.cpu().double() - torch.tensor(gt).double()) ** 2).sum()) / (B * T)
    assert abs(loss - float(g['loss'])) < tol * abs(float(g['loss'])) * 4
    for name, key in (('frm_sal', 'sal'), ('rcn_outputs', 'bottom'), ('rcn_upsampled_outputs', 'up'), ('gaze_rcn_outputs', 'top')):
        got = float(eng.read_buffer(name).abs().double().sum())
        assert abs(got - float(g['abs_sum_' + key])) < 5 * tol * float(g['abs_sum_' + key]), name
    grads, _ = eng.backward(maps, torch.tensor(gt, device=gpu))
    for field, key in CascadeEngine.KEYS:
        want = float(g['gnorm_' + key.replace('/', '.')])
        got = float(grads[field].double().norm())
        assert abs(got - want) < gtol * want, (key, got, want)


@pytest.mark.parametrize('dtype,tol,gtol', [('f32', 1e-4, 1e-3), ('bf16', 3e-2, 6e-2)])
def test_fcgru_against_golden(gpu, dtype, tol, gtol):
    from recurrent_gaze_prediction_amd.engine import FcGruEngine
    g = np.load(os.path.join(GOLD, 'fcgru_small.npz'))
    B, T, GH, seed = [int(v) for v in g['config']]
    p = syn.fcgru_params(seed, GH, GH)
    x = syn.c3d_features(seed + 1, B, T)
    rs = np.random.RandomState(seed + 2)
    gt = rs.rand(B, T, GH, GH).astype(np.float32)
    gt /= gt.sum(axis=(2, 3), keepdims=True)
    eng = FcGruEngine(B, T, (GH, GH), dtype=dtype, device=gpu, save_for_backward=True)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    assert rel(logits.cpu().numpy(), g['logits']) < tol
    grads = eng.backward(logits, probs, torch.tensor(gt, device=gpu))
    for k in p:
        want = float(g['gnorm_' + k])
        assert abs(float(grads[k].double().norm()) - want) < gtol * want, k


def test_frontend_window_against_golden(gpu):
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    g = np.load(os.path.join(GOLD, 'frontend_window.npz'))
    rs = np.random.RandomState(int(g['config'][0]))
    frames = rs.randint(0, 256, size=(18, 240, 320, 3)).astype(np.uint8)
    mean = (rs.rand(3, 16, 128, 171) * 120).astype(np.float32)
    eng = C3DEngine(1, dtype='bf16', device=gpu)
    v = eng.frames_to_video(torch.tensor(frames, device=gpu), [1], torch.tensor(mean, device=gpu)).cpu().numpy()
    d = np.abs(v[0, ::3, ::7, ::5] - g['sample'])
    assert d.max() <= 1.0 and float((d > 0).mean()) < 1e-3
    assert abs(float(np.abs(v.astype(np.float64)).sum()) - float(g['checksum'])) < 1e-5 * float(g['checksum'])
