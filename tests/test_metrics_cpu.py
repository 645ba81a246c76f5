"""CPU: this package's evaluation_metrics against golden scores produced by the
REFERENCE's own evaluation_metrics.py (tests/golden/make_golden.py: metrics_case)."""
import os

import numpy as np
import pytest

from recurrent_gaze_prediction_amd import evaluation_metrics as em
from recurrent_gaze_prediction_amd import synthetic as syn

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metrics_ref.npz'))


@pytest.fixture(scope='module')
def maps():
    seed, n = [int(v) for v in GOLD['config']]
    gt, centres = syn.gaze_maps(seed, n, 1)
    fix = syn.fixation_maps(seed + 1, centres)[:, 0]
    gt = gt[:, 0]
    rs = np.random.RandomState(seed + 2)
    pred = (gt + 0.3 * rs.rand(*gt.shape) + 0.2 * np.roll(gt, 3, axis=2)).astype(np.float32)
    return pred, gt, fix


def test_sim_cc_match_reference(maps):
    pred, gt, fix = maps
    sim = [em.saliency_score_single('sim', p, g, f) for p, g, f in zip(pred, gt, fix)]
    cc = [em.saliency_score_single('cc', p, g, f) for p, g, f in zip(pred, gt, fix)]
    assert np.allclose(sim, GOLD['sim'], rtol=0, atol=1e-12)
    assert np.allclose(cc, GOLD['cc'], rtol=0, atol=1e-12)


def test_auc_judd_borji_match_reference_with_same_seed(maps):
    pred, gt, fix = maps
    for i, (p, g, f) in enumerate(zip(pred, gt, fix)):
        np.random.seed(1000 + i)
        assert abs(em.saliency_score_single('AUC_Judd', p, g, f) - GOLD['AUC_Judd'][i]) < 1e-12
        np.random.seed(2000 + i)
        assert abs(em.saliency_score_single('AUC_Borji', p, g, f) - GOLD['AUC_Borji'][i]) < 1e-12


@pytest.mark.parametrize('metric', ['sim', 'cc', 'AUC_Borji'])
def test_saliency_score_matches_reference(maps, metric):
    pred, gt, fix = maps
    np.random.seed(3000)
    assert abs(em.saliency_score(metric, list(pred), list(gt), list(fix)) - float(GOLD['score_' + metric])) < 1e-12


def test_inputs_are_not_mutated_and_extras_run(maps):
    pred, gt, fix = maps
    p0 = pred[0].copy()
    np.random.seed(1)
    em.saliency_score_single('AUC_Judd', pred[0], gt[0], fix[0])
    assert np.array_equal(p0, pred[0])
    np.random.seed(2)
    s = em.saliency_score('AUC_shuffled', list(pred), list(gt), list(fix))   # runs under py3 here
    assert 0.0 <= s <= 1.0
    assert em.nss(fix[0], gt[0]) > 0.5                       # gt blob is centred on its fixations
    assert np.isnan(em.AUC_Judd(np.zeros((49, 49)), pred[0]))
    assert em.AVAILABLE_METRICS == ('sim', 'cc', 'AUC_shuffled', 'AUC_Borji')


def test_resize_identity_and_sparse_onehot():
    a = np.random.RandomState(3).rand(49, 49)
    assert np.array_equal(em.resize(a, (49, 49)), a)
    assert em.resize(a, (98, 120)).shape == (98, 120)
    x = np.zeros((10, 10))
    x[9, 9] = 1
    x[0, 3] = 1
    r = em.resize_onehot_tensor_sparse(x, (49, 49))
    assert r[48, 48] and r[0, 16] and r.sum() == 2
