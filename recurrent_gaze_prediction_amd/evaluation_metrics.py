"""Saliency metrics with the reference's names, signatures and RNG consumption.

Host-side (numpy) mirror of /root/reference/evaluation_metrics.py:15-297 so that
``saliency_score(metric, pred_maps, gt_maps, fixation_maps)`` and
``AVAILABLE_METRICS`` can be used on this package's ``generate()`` output exactly
as the reference's ``GazePredictionGRU.evaluate`` does (gaze_rnn.py:653-674).
The scores are pinned against the reference file itself by
``tests/golden/metrics_ref.npz`` (see tests/test_metrics.py).

Differences, all deliberate:
* the ROC sweeps are vectorised (sort + searchsorted) instead of Python loops;
  thresholds, counts and the trapezoid rule are the same, so scores agree to
  rounding;
* ``AUC_shuffled`` works under Python 3 (the reference's ``map`` object breaks
  there, evaluation_metrics.py:200-201) with the same draw order: ``n_rep``
  successive ``permutation`` calls;
* ``resize`` is a local cubic-spline resize (scikit-image is not a dependency);
  at equal shapes it is the identity, which is the only case the 49x49 parity
  tests need.  For unequal shapes it is NOT pinned against scikit-image.
* ``nss`` (normalised scanpath saliency) is new -- the reference has no NSS
  (SURVEY.md section 0); standard definition, not reference-pinned.
"""
from functools import partial

import numpy as np
import numpy.random as random
import scipy.ndimage
import scipy.sparse

_trapz = getattr(np, "trapezoid", None) or np.trapz


def normalize_range(x):
    """evaluation_metrics.py:15-17."""
    x = np.asarray(x)
    return (x - np.min(x)) / (np.max(x) - np.min(x))


def resize(image, output_shape, order=3, mode='reflect'):
    """Spline resize with scikit-image's pixel-centre convention; identity at
    equal shapes (stands in for skimage.transform.resize, evaluation_metrics.py:8)."""
    image = np.asarray(image, dtype=np.float64)
    output_shape = tuple(int(s) for s in output_shape)
    if image.shape == output_shape:
        return image.copy()
    coords = [(np.arange(o) + 0.5) * (float(i) / o) - 0.5 for i, o in zip(image.shape, output_shape)]
    grid = np.meshgrid(*coords, indexing='ij')
    return scipy.ndimage.map_coordinates(image, grid, order=order, mode=mode)


def resize_onehot_tensor_sparse(x, target_shape):
    """evaluation_metrics.py:19-39: move each positive pixel to its rounded
    position on the target grid."""
    assert len(target_shape) == 2
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError('x.shape : %s' % (x.shape,))
    h1, w1 = x.shape
    h2, w2 = target_shape
    ret = np.zeros((h2, w2), dtype=bool)
    ys, xs = np.where(x > 0)
    yy = (np.round(ys * (h2 - 1.0) / (h1 - 1.0)) + 1e-9).astype(int)
    xx = (np.round(xs * (w2 - 1.0) / (w1 - 1.0)) + 1e-9).astype(int)
    ret[yy, xx] = True
    return ret


def _prep(fixation_map, saliency_map):
    saliency_map = np.array(saliency_map, dtype=np.float64)
    fixation_map = np.asarray(fixation_map) > 0.5
    if saliency_map.shape != fixation_map.shape:
        saliency_map = resize(saliency_map, fixation_map.shape, order=3, mode='nearest')
    return fixation_map, saliency_map


def AUC_Judd(fixation_map, saliency_map, jitter=True):
    """evaluation_metrics.py:42-98.  Thresholds = saliency at the fixations,
    descending; tp_k = k/n_fix, fp_k = (#(S>=thr_k) - k)/(n_pix - n_fix)."""
    if not np.any(np.asarray(fixation_map) > 0.5):
        print('no fixation to predict')
        return np.nan
    fix, sal = _prep(fixation_map, saliency_map)
    if jitter:
        sal = sal + random.rand(*sal.shape) * 1e-7     # same draw as the reference (:79)
    sal = normalize_range(sal)
    s = sal.ravel()
    s_fix = np.sort(s[fix.ravel()])[::-1]
    n_fix, n_pix = len(s_fix), len(s)
    s_sorted = np.sort(s)
    above = n_pix - np.searchsorted(s_sorted, s_fix, side='left')     # count of S >= thr
    k = np.arange(1, n_fix + 1)
    tp = np.concatenate([[0.0], k / float(n_fix), [1.0]])
    fp = np.concatenate([[0.0], (above - k) / float(n_pix - n_fix), [1.0]])
    return _trapz(tp, fp)


def AUC_Borji(fixation_map, saliency_map, n_rep=100, step_size=0.1, rand_sampler=None):
    """evaluation_metrics.py:101-164."""
    if not np.any(np.asarray(fixation_map) > 0.5):
        print('no fixation to predict')
        return np.nan
    fix, sal = _prep(fixation_map, saliency_map)
    sal = normalize_range(sal)
    s = sal.ravel()
    f = fix.ravel()
    s_fix = s[f]
    n_fix, n_pix = len(s_fix), len(s)
    if rand_sampler is None:
        r = random.randint(0, n_pix, [n_fix, n_rep])   # same draw as the reference (:148)
        s_rand = s[r]
    else:
        s_rand = rand_sampler(s, f, n_rep, n_fix)
    fix_sorted = np.sort(s_fix)
    auc = np.full(n_rep, np.nan)
    for rep in range(n_rep):
        col = s_rand[:, rep]
        top = max(s_fix.max(), col.max())
        thr = np.r_[0:top:step_size][::-1]
        col_sorted = np.sort(col)
        tp = np.concatenate([[0.0], (n_fix - np.searchsorted(fix_sorted, thr, side='left')) / float(n_fix), [1.0]])
        fp = np.concatenate([[0.0], (n_fix - np.searchsorted(col_sorted, thr, side='left')) / float(n_fix), [1.0]])
        auc[rep] = _trapz(tp, fp)
    return np.mean(auc)


def AUC_shuffled(fixation_map, saliency_map, other_map, n_rep=100, step_size=0.1):
    """evaluation_metrics.py:167-204: negatives drawn from fixations of other images."""
    other_map = np.asarray(other_map) > 0.5
    if other_map.shape != np.asarray(fixation_map).shape:
        raise ValueError('other_map.shape != fixation_map.shape')

    def sample_other(other, s, f, n_rep, n_fix):
        fixated = np.nonzero(other)[0]
        idx = np.stack([random.permutation(len(fixated))[:n_fix] for _ in range(n_rep)], axis=1)
        return s[fixated[idx]]

    return AUC_Borji(fixation_map, saliency_map, n_rep, step_size,
                     partial(sample_other, other_map.ravel()))


def similarity(gtsAnn, resAnn):
    """evaluation_metrics.py:207-218: histogram intersection of the two
    sum-normalised maps."""
    g = np.asarray(gtsAnn, dtype=np.float64)
    r = np.asarray(resAnn, dtype=np.float64)
    return np.minimum(g / g.sum(), r / r.sum()).sum()


def cc(gtsAnn, resAnn):
    """evaluation_metrics.py:221-236: Pearson correlation of the two maps."""
    g = np.asarray(gtsAnn, dtype=np.float64)
    r = np.asarray(resAnn, dtype=np.float64)
    g = g - g.mean()
    if g.max() > 0:
        g = g / g.std()
    r = r - r.mean()
    if r.max() > 0:
        r = r / r.std()
    return np.corrcoef(r.reshape(-1), g.reshape(-1))[0][1]


def nss(fixation_map, saliency_map):
    """Normalised scanpath saliency: mean of the z-scored saliency at the fixated
    pixels.  NEW (not in the reference)."""
    fix, sal = _prep(fixation_map, saliency_map)
    if not np.any(fix):
        return np.nan
    sd = sal.std()
    z = (sal - sal.mean()) / (sd if sd > 0 else 1.0)
    return float(z[fix].mean())


def saliency_score_single(metric, pred_map, gt_map, fixation_map, other_map_union=None):
    """evaluation_metrics.py:239-272."""
    if scipy.sparse.issparse(fixation_map):
        fixation_map = fixation_map.toarray()
    fixation_map = np.asarray(fixation_map)
    pred_map = normalize_range(pred_map)
    pred_map_orig = resize(pred_map, fixation_map.shape, order=3)
    gt_map_orig = resize(gt_map, fixation_map.shape, order=3)
    if metric == 'cc':
        return cc(gt_map_orig, pred_map_orig)
    if metric == 'sim':
        return similarity(gt_map_orig, pred_map_orig)
    if metric == 'AUC_Judd':
        return AUC_Judd(fixation_map, pred_map_orig)
    if metric == 'AUC_Borji':
        return AUC_Borji(fixation_map, pred_map_orig)
    if metric == 'AUC_shuffled':
        if other_map_union is None:
            raise ValueError('other_map_union required')
        return AUC_shuffled(fixation_map, pred_map_orig, other_map_union)
    if metric == 'NSS':
        return nss(fixation_map, pred_map_orig)
    raise ValueError(metric)


def saliency_score(metric, pred_maps, gt_maps, fixation_maps):
    """evaluation_metrics.py:275-295 (needs >= 10 fixation maps; the union of 10
    random ones is the AUC_shuffled negative set, drawn from the global RNG)."""
    assert len(gt_maps) == len(pred_maps) == len(fixation_maps)
    m = 10
    assert len(fixation_maps) >= m
    first = fixation_maps[0].toarray() if scipy.sparse.issparse(fixation_maps[0]) else np.asarray(fixation_maps[0])
    other_map_union = np.zeros(first.shape)
    for i in random.choice(range(len(fixation_maps)), m, replace=False):
        fm = fixation_maps[i].toarray() if scipy.sparse.issparse(fixation_maps[i]) else np.asarray(fixation_maps[i])
        other_map_union += (fm > 0).astype(int)
    scores = [saliency_score_single(metric, p, g, f, other_map_union)
              for g, p, f in zip(gt_maps, pred_maps, fixation_maps)]
    return np.mean(scores)


AVAILABLE_METRICS = ('sim', 'cc', 'AUC_shuffled', 'AUC_Borji',)   # evaluation_metrics.py:297
EXTRA_METRICS = ('AUC_Judd', 'NSS')
