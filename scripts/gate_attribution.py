"""Where does the bf16 drift of the north-star metrics gate come from?  

The gate's workload (tests/test_golden_gpu.py::test_end_to_end_metrics_gate_T16: 3 clips x T = 16 of synthetic video ->
C3D -> gaze_grcn head -> softmax) is run with every combination of {oracle fp32 CPU, HIP f32, HIP bf16} conv stack and
{oracle fp32 CPU, HIP f32, HIP bf16} head, and each combination is scored like the gate (cases A, B, C) over several
fixation seeds.  Prints a JSON document (kept as profiles/r04_gate_attribution.json).

The oracle is used here as the CHECKER (test infrastructure), never as a product path."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import torch_ref                                   # noqa: E402
from recurrent_gaze_prediction_amd import evaluation_metrics as em  # noqa: E402
from recurrent_gaze_prediction_amd import synthetic as syn     # noqa: E402
from recurrent_gaze_prediction_amd.engine import C3DEngine, GrcnEngine  # noqa: E402
from test_golden_gpu import _fixations_following, _metric_scores  # noqa: E402

METRICS = ('cc', 'sim', 'AUC_Borji', 'AUC_Judd', 'AUC_shuffled', 'NSS')


def rel(a, r):
    a, r = np.asarray(a, np.float64), np.asarray(r, np.float64)
    return float(np.abs(a - r).max() / np.abs(r).max())


def rel_range(a, r):
    """per-frame max |dlogit| over the frame's logit range (what min-max normalised metrics see), worst frame and mean"""
    a, r = np.asarray(a, np.float64).reshape(-1, 2401), np.asarray(r, np.float64).reshape(-1, 2401)
    e = np.abs(a - r).max(1) / (r.max(1) - r.min(1))
    return float(e.max()), float(e.mean())


def main():
    dev = torch.device('cuda:0')
    B, T = 3, 16
    n = B * T
    seeds = [int(s) for s in os.environ.get('GATE_SEEDS', '68,168,268,368,468').split(',')]
    cp = syn.c3d_params(65, scale='he')
    video = syn.video_windows(66, n)
    torch.set_num_threads(16)
    with torch.no_grad():
        feat_ref = torch_ref.c3d_forward(torch.tensor(video), {k: torch.tensor(v) for k, v in cp.items()})
    feats = {'oracle': feat_ref.reshape(B, T, 1024, 7, 7).contiguous()}
    for dtype in ('f32', 'bf16'):
        c3d = C3DEngine(n, dtype=dtype, device=dev)
        c3d.set_weights(cp)
        f, _ = c3d.forward(torch.tensor(video, device=dev), want_features=True, want_rows=True)
        feats[dtype] = f.float().cpu().reshape(B, T, 1024, 7, 7).contiguous()
        del c3d
    doc = {'workload': 'B3 x T16, tests/test_golden_gpu.py gate', 'fixation_seeds': seeds,
           'feature_rel_err': {k: rel(v.numpy(), feat_ref.numpy().reshape(B, T, 1024, 7, 7)) for k, v in feats.items()},
           'cases': {}}
    gt, centres = syn.gaze_maps(67, B, T)
    for label, out_scale in (('A', 1.0), ('BC', 40.0)):
        hp = syn.grcn_params(61, T, gru_std=0.05, random_bn=True)
        hp['out_W'] = hp['out_W'] * out_scale
        hpt = {k: torch.tensor(v) for k, v in hp.items()}
        maps, logits = {}, {}
        for cname, f in feats.items():
            with torch.no_grad():
                lg = torch_ref.grcn_forward(f, hpt)
            logits[(cname, 'oracle')] = lg.numpy()
            maps[(cname, 'oracle')] = torch_ref.softmax_maps(lg).numpy().reshape(n, 49, 49)
            for hd in ('f32', 'bf16'):
                head = GrcnEngine(B, T, dtype=hd, device=dev)
                head.set_weights(hp)
                lg2, pr = head.forward(f.to(dev))
                head.status()
                logits[(cname, hd)] = lg2.cpu().numpy()
                maps[(cname, hd)] = pr.cpu().numpy().reshape(n, 49, 49)
                del head
        ref = maps[('oracle', 'oracle')]
        ref_lg = logits[('oracle', 'oracle')]
        names = ['A'] if label == 'A' else ['B', 'C']
        for name in names:
            rows = {}
            for key, m in maps.items():
                deltas = {k: [] for k in METRICS}
                base = {k: [] for k in METRICS}
                for s in seeds:
                    if name == 'B':
                        g_, f_ = _fixations_following(ref, s + 1)
                    else:
                        g_, f_ = gt, syn.fixation_maps(s, centres)
                    s_ref, s_got = _metric_scores(ref, g_, f_, n), _metric_scores(m, g_, f_, n)
                    for k in METRICS:
                        deltas[k].append(s_got[k] - s_ref[k])
                        base[k].append(s_ref[k])
                wl, ml = rel_range(logits[key], ref_lg)
                rows['c3d=%s,head=%s' % key] = {
                    'logit_err_over_range_worst_frame': wl, 'logit_err_over_range_mean': ml,
                    'max_abs_delta': {k: float(np.abs(deltas[k]).max()) for k in METRICS},
                    'mean_delta': {k: float(np.mean(deltas[k])) for k in METRICS},
                    'oracle_score_mean': {k: float(np.mean(base[k])) for k in METRICS}}
            doc['cases'][name] = rows
    print(json.dumps(doc, indent=1))


if __name__ == '__main__':
    main()
