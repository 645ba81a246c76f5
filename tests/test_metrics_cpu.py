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


def test_auc_shuffled_matches_reference_with_same_seed(maps):
    """The reference's own AUC_shuffled (evaluation_metrics.py:167-204, run by make_golden.py with a list-returning
    ``map``): per frame, negatives = union of the other frames' fixations, seeded draw order identical."""
    pred, gt, fix = maps
    n = len(pred)
    for i, (p, g, f) in enumerate(zip(pred, gt, fix)):
        other = np.zeros(fix[0].shape)
        for j in range(n):
            if j != i:
                other += (fix[j] > 0).astype(int)
        np.random.seed(4000 + i)
        got = em.saliency_score_single('AUC_shuffled', p, g, f, other)
        assert abs(got - GOLD['AUC_shuffled'][i]) < 1e-12, (i, got, GOLD['AUC_shuffled'][i])
    with pytest.raises(ValueError):
        em.saliency_score_single('AUC_shuffled', pred[0], gt[0], fix[0])          # :262-263
    with pytest.raises(ValueError):
        em.AUC_shuffled(fix[0], pred[0], np.zeros((3, 3)))                         # :191-192


@pytest.mark.parametrize('metric', ['sim', 'cc', 'AUC_Borji', 'AUC_shuffled'])
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


def test_nss_known_answers():
    """NSS is not in the reference (north_star lists it): pinned by closed forms.  Two-valued map, value a on m of the
    N pixels and b elsewhere (a > b): mean = b + (a-b) m/N, std = (a-b) sqrt(p(1-p)) with p = m/N, so a fixation on the
    high region scores sqrt((1-p)/p), one on the low region -sqrt(p/(1-p)); a mix of k high and l low fixations the
    weighted mean.  Invariant to positive affine maps of the saliency (z-score)."""
    N = 49 * 49
    sal = np.full((49, 49), 0.2)
    sal[10:17, 20:27] = 0.9                                   # m = 49 pixels
    p = 49.0 / N
    hi, lo = np.sqrt((1 - p) / p), -np.sqrt(p / (1 - p))
    f = np.zeros((49, 49))
    f[12, 22] = 1
    assert abs(em.nss(f, sal) - hi) < 1e-12
    f2 = np.zeros((49, 49))
    f2[0, 0] = 1
    assert abs(em.nss(f2, sal) - lo) < 1e-12
    f3 = np.zeros((49, 49))
    f3[12, 22] = f3[13, 23] = f3[40, 40] = 1                  # 2 high, 1 low
    assert abs(em.nss(f3, sal) - (2 * hi + lo) / 3) < 1e-12
    assert abs(em.nss(f3, 7.5 * sal + 3.0) - (2 * hi + lo) / 3) < 1e-9
    assert abs(em.saliency_score_single('NSS', sal, sal, f3) - (2 * hi + lo) / 3) < 1e-9   # min-max normalised inside
    # every pixel fixated: the mean of a z-scored map is 0; a constant map has no contrast: 0; no fixation: nan
    assert abs(em.nss(np.ones((49, 49)), sal)) < 1e-12
    assert em.nss(f, np.full((49, 49), 0.3)) == 0.0
    assert np.isnan(em.nss(np.zeros((49, 49)), sal))
    # one-hot saliency at the fixation: z = (1 - 1/N) / sqrt((1/N)(1 - 1/N)) = sqrt(N - 1)
    one = np.zeros((49, 49))
    one[12, 22] = 1.0
    assert abs(em.nss(f, one) - np.sqrt(N - 1.0)) < 1e-9
    assert 'NSS' in em.EXTRA_METRICS


def test_resize_identity_and_sparse_onehot():
    a = np.random.RandomState(3).rand(49, 49)
    assert np.array_equal(em.resize(a, (49, 49)), a)
    assert em.resize(a, (98, 120)).shape == (98, 120)
    x = np.zeros((10, 10))
    x[9, 9] = 1
    x[0, 3] = 1
    r = em.resize_onehot_tensor_sparse(x, (49, 49))
    assert r[48, 48] and r[0, 16] and r.sum() == 2
