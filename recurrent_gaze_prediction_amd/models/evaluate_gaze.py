"""Mirror of /root/reference/models/evaluate_gaze.py: per-frame scoring of a
model's generate() output, per-frame dumps and ``overall.txt``; plus the long-clip inference of
models/extract_map.py:148-229.  The two ``pdb.set_trace()`` calls of the reference (:100, :189)
are not reproduced."""
import logging
import os
from collections import OrderedDict, defaultdict

import numpy as np
import scipy.sparse

from ..evaluation_metrics import resize_onehot_tensor_sparse, saliency_score_single

log = logging.getLogger('rgp')
FRAME_METRICS = ('sim', 'cc', 'AUC_Borji', 'AUC_Judd', 'AUC_shuffled')       # evaluate_gaze.py:135


def _dense(m):
    return m.toarray() if scipy.sparse.issparse(m) else np.asarray(m)


def handle_frame(i, n_images, image, pred_gazemap, gt_gazemap, fixationmap, out_dir, fixationmaps_all, rng,
                 dump_images=True):
    """evaluate_gaze.py:116-158: union of 10 random other fixation maps, 5 metrics, dumps."""
    fixationmap = _dense(fixationmap)
    other_map_union = np.zeros(fixationmap.shape, np.uint8)
    for oth in rng.choice(range(len(fixationmaps_all)), 10, replace=False):
        other_map = _dense(fixationmaps_all[oth])
        if other_map.shape != fixationmap.shape:
            other_map = resize_onehot_tensor_sparse(other_map, fixationmap.shape)
        other_map_union += (other_map > 0).astype(np.uint8)
    scores = OrderedDict()
    for metric in FRAME_METRICS:
        scores[metric] = saliency_score_single(metric, pred_map=pred_gazemap, gt_map=gt_gazemap,
                                               fixation_map=fixationmap, other_map_union=other_map_union)
    if out_dir is not None:
        if dump_images:
            try:
                from PIL import Image

                def save(name, arr):
                    a = np.asarray(arr, np.float64)
                    a = (a - a.min()) / max(a.max() - a.min(), 1e-12)
                    Image.fromarray((a * 255).astype(np.uint8)).save(os.path.join(out_dir, name))
                save('%05d.frame.jpg' % i, image)
                save('%05d.gaze_pred.jpg' % i, pred_gazemap)
                save('%05d.gaze_gt.jpg' % i, gt_gazemap)
            except ImportError:
                pass
        with open(os.path.join(out_dir, '%05d.scores.txt' % i), 'w') as fp:
            fp.write('%d / %d\n' % (i, n_images))
            for k, v in scores.items():
                fp.write('%s : %.4f\n' % (k, v))
    return scores


def run_evaluation(model, data_sets, out_dir, num_frames=1000, seed=0, dump_images=False):
    """evaluate_gaze.py:172-227 -> {metric: mean}; writes <out_dir>/overall.txt in the reference's format."""
    assert out_dir is not None
    os.makedirs(out_dir, exist_ok=True)
    T = model.n_lstm_steps
    ret = model.generate(data_sets.valid, max_instances=int(np.divide(num_frames, T, dtype=float) + 1))
    pred, gt = np.asarray(ret['pred_gazemap_list']), np.asarray(ret['gt_gazemap_list'])
    images, fix = ret['images_list'], ret['fixationmap_list']
    n_images = len(pred)
    assert n_images == len(gt) == len(images) == len(fix)
    rng = np.random.RandomState(seed)
    state = np.random.get_state()
    np.random.seed(seed)                       # AUC_Judd / AUC_Borji draw from the global RNG (9-Q11)
    try:
        aggregated = defaultdict(list)
        for i in range(n_images):
            scores = handle_frame(i, n_images, images[i], pred[i], gt[i], fix[i], out_dir, fix, rng, dump_images)
            for metric, score in scores.items():
                aggregated[metric].append(score)
    finally:
        np.random.set_state(state)
    overall = OrderedDict()
    with open(os.path.join(out_dir, 'overall.txt'), 'w') as fp:
        for metric, score_list in aggregated.items():
            overall[metric] = float(np.mean(score_list))
            fp.write("Average %s : %.4f\n" % (metric, overall[metric]))
            fp.write(''.join('%.3f ' % s for s in score_list) + '\n')
    return overall


def predict_long_clip(model, c3d, frames=None, pool_to_7x7=False):
    """extract_map.py:148-229: a clip of any length through a fixed-T model.  c3d [N,1024,7,7]
    (or [N,512,2,7,7]) is cut into T-chunks, the tail zero-padded, B chunks per call; returns
    [N,49,49] (or [N,7,7] with a 7x7 average re-pool of each map)."""
    c3d = np.asarray(c3d, np.float32).reshape(len(c3d), 1024, 7, 7)
    n, T, B = len(c3d), model.n_lstm_steps, model.batch_size
    n_chunks = -(-n // T)
    padded = np.zeros((n_chunks * T, 1024, 7, 7), np.float32)
    padded[:n] = c3d
    chunks = padded.reshape(n_chunks, T, 1024, 7, 7)
    outs = []
    for i in range(0, n_chunks, B):
        batch = chunks[i:i + B]
        if len(batch) < B:
            batch = np.concatenate([batch, np.zeros((B - len(batch),) + batch.shape[1:], np.float32)])
        maps = model.predict(batch, frames).cpu().numpy()
        outs.append(maps[:min(B, n_chunks - i)])
    maps = np.concatenate(outs).reshape(n_chunks * T, model.gazemap_height, model.gazemap_width)[:n]
    if pool_to_7x7 and maps.shape[-1] == 49:
        maps = maps.reshape(n, 7, 7, 7, 7).mean(axis=(2, 4))
    return maps
