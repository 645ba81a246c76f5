"""GPU parity at the BENCHMARK'S OWN SIZE with distinct data: BASELINE config 3's workload (64 clips x T = 16 = 1024
C3D windows, every window different) through the default bf16 plan, against the fp32 CPU oracle of the same chain
computed for ALL 1024 windows (about a minute on the box's 16 host cores).

* every window's conv5b features against ``torch_ref.c3d_forward`` (a tile walk that swapped, dropped or duplicated a
  window -- the patch kernels order their tiles in chunks of 16 / 4 windows -- would show in that window's row);
* every layer of the run against the library's second kernel family (``RGP_C3D_KERNELS_IGEMM``) on the same 1024 windows,
  and against the oracle's own activations for three scattered windows (0, 511, 1023);
* the north-star acceptance gate on its own workload: cc, sim, AUC_Borji, AUC_shuffled (the reference's
  AVAILABLE_METRICS, evaluation_metrics.py:297), AUC_Judd and NSS of the 1024 maps, HIP bf16 against the oracle.

VERDICT r03 "weak" 3 (bench-scale correctness rested on 8 distinct windows) and "next" 1c / 2."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn
from test_golden_gpu import _fixations_following, _metric_scores

pytestmark = pytest.mark.gpu

B, T = 64, 16
N = B * T


def _rel(a, r):
    return float((a - r).abs().max() / r.abs().max().clamp_min(1e-30))


@pytest.fixture(scope='module')
def workload(gpu):
    """1024 distinct windows (device generator, as bench.py draws them), both plans' runs and the oracle's features."""
    from recurrent_gaze_prediction_amd.engine import C3DEngine, C3D_LAYER_NAMES
    g = torch.Generator(device=gpu)
    g.manual_seed(20261004)
    video = torch.rand(N, 16, 112, 112, 3, device=gpu, generator=g) - 0.5
    cp = syn.c3d_params(65, scale='he')
    eng = C3DEngine(N, dtype='bf16', device=gpu)
    eng.set_weights(cp)
    feats, rows = eng.forward(video, want_features=True, want_rows=True)
    feats, rows = feats.clone(), rows.clone()
    torch.cuda.synchronize()
    # the oracle: all 1024 windows, one clip (16 windows) at a time; three windows keep every layer's activations
    cpt = {k: torch.tensor(v) for k, v in cp.items()}
    keep = {0: None, 511: None, 1023: None}
    ref = torch.empty(N, 1024, 7, 7)
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        with torch.no_grad():
            for c in range(B):
                v = video[c * T:(c + 1) * T].cpu()
                ref[c * T:(c + 1) * T] = torch_ref.c3d_forward(v, cpt)
                for w in keep:
                    if c * T <= w < (c + 1) * T:
                        _, acts = torch_ref.c3d_forward(v[w - c * T:w - c * T + 1], cpt, want_all=True)
                        keep[w] = {k: a[0].permute(1, 2, 3, 0).contiguous() for k, a in acts.items()}   # CDHW -> DHWC
    finally:
        torch.set_num_threads(old)
    return {'video': video, 'cp': cp, 'eng': eng, 'feats': feats, 'rows': rows, 'ref': ref, 'keep': keep,
            'names': C3D_LAYER_NAMES}


def test_every_window_against_the_oracle(gpu, workload):
    """conv5b features of all 1024 distinct windows: per-window max-abs error over the window's own max-abs."""
    got, ref = workload['feats'].cpu(), workload['ref']
    err = (got - ref).reshape(N, -1).abs().amax(1) / ref.reshape(N, -1).abs().amax(1)
    worst = int(err.argmax())
    assert float(err.max()) < 3e-2, 'window %d: rel err %.3e (median %.3e)' % (worst, float(err.max()), float(err.median()))
    assert float(err.median()) < 1.5e-2
    # a permutation of windows could pass a per-window NORM check (uniform-noise windows give nearly parallel feature
    # vectors: cosine 1.000 between ANY two of them), so identify every window: with the mean over windows removed, the
    # oracle window nearest to HIP window i must be window i itself
    g, r = got.reshape(N, -1).to(gpu, torch.float64), ref.reshape(N, -1).to(gpu, torch.float64)
    mu = r.mean(0, keepdim=True)
    g, r = g - mu, r - mu
    d2 = (g * g).sum(1)[:, None] + (r * r).sum(1)[None, :] - 2.0 * g @ r.t()
    nearest = d2.argmin(1).cpu()
    own = d2.diagonal().clone()
    d2.fill_diagonal_(float('inf'))
    margin = float((d2.amin(1) / own.clamp_min(1e-300)).min())
    print('window identification: %d of %d HIP windows are nearest to their own oracle window; the next-nearest oracle '
          'window is >= %.1f x as far (squared distance)' % (int((nearest == torch.arange(N)).sum()), N, margin))
    assert torch.equal(nearest, torch.arange(N)), 'windows matched to another window: %s' % (nearest != torch.arange(N)).nonzero().flatten()[:16].tolist()
    assert margin > 4.0, margin
    # rows [n*49, d*512 + c] hold the same numbers as the feature blob [n, c*2 + d, 7, 7] (models/gaze_rnn.py:494-497)
    rows = workload['rows'].float().reshape(N, 7, 7, 2, 512).permute(0, 4, 3, 1, 2).reshape(N, 1024, 7, 7)
    assert torch.equal(rows.cpu(), got.to(torch.bfloat16).float()) or _rel(rows.cpu(), got) < 4e-3


def test_every_layer_against_the_second_kernel_family_and_three_oracle_windows(gpu, workload):
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    eng, names = workload['eng'], workload['names']
    other = C3DEngine(N, dtype='bf16', device=gpu, kernels='igemm')
    other.set_weights(workload['cp'])
    assert other.layer_kernel_name(1, N).startswith('igemm_wide_kernel') and eng.layer_kernel_name(1, N).startswith('conv_patch')
    rows_b = other.forward(workload['video'], want_features=False, want_rows=True)[1]
    report = {}
    for i, name in enumerate(names):
        if i == 7:
            a, b = workload['rows'].float(), rows_b.float()
        else:
            a, b = eng.read_layer(i, N), other.read_layer(i, N)
        per = a.numel() // N
        a, b = a.reshape(N, per), b.reshape(N, per)
        e = (a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-30)          # per window
        report[name] = float(e.max())
        assert float(e.max()) < 3e-2, '%s: window %d differs between the kernel families by %.3e' % (name, int(e.argmax()), float(e.max()))
        assert float(b.abs().amax(1).min()) > 0, name                          # no window left unwritten
        if i < 7:
            for w, acts in workload['keep'].items():
                ref = acts[name].reshape(-1).to(gpu)
                ew = float((a[w] - ref).abs().max() / ref.abs().max())
                assert ew < 3e-2, '%s window %d vs oracle: %.3e' % (name, w, ew)
        del a, b
    print('patch vs igemm family, worst window per layer:', {k: '%.2e' % v for k, v in report.items()})


@pytest.mark.parametrize('seeds', [(61, 67, 68, 69), (161, 167, 168, 169)], ids=['draw0', 'draw1'])
def test_metrics_gate_on_the_config3_workload(gpu, workload, seeds):
    """north_star: "AUC/CC within +-1e-3 of reference" on gaze_grcn 16-frame clips.  1024 maps of the default bf16 plan
    (patch kernels, persistent ConvGRU, bf16 head) against the fp32 oracle chain, scored as models/evaluate_gaze.py
    scores them.  Cases as in test_end_to_end_metrics_gate_T16: A random-init head; B peaked maps on fixations that
    follow the oracle's maps (a trained model's regime); C peaked maps on independent gaze data.

    cc, sim, AUC_Judd: within 1e-3 everywhere.  AUC_Borji / AUC_shuffled sweep their threshold in steps of 0.1 of the
    map's range (evaluation_metrics.py:101-164): each is a step function of the map, a fixation whose saliency sits
    near a multiple of 0.1 flips one ROC step under ANY perturbation, and bf16 perturbs the logits by 5e-3 of their range
    (profiles/r04_gate_attribution.json: half from the bf16 conv features, half from the bf16 head, neither alone under
    1e-3 at 48 frames).  The flips are zero-mean, so their average shrinks with the number of frames scored: at this
    workload's 1024 frames the drift is inside +-1e-3 (asserted); NSS (unbounded) within 1e-3 * max(1, |NSS|).

    Two independent draws of everything downstream of the conv features (head weights, gaze maps, fixations, the
    fixations that follow the maps), so the +-1e-3 statement does not rest on one sample; the
    1024 windows and the conv weights are the module fixture's."""
    s_head, s_gaze, s_fix, s_follow = seeds
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    ref_feat = workload['ref'].reshape(B, T, 1024, 7, 7)
    gt, centres = syn.gaze_maps(s_gaze, B, T)
    fix = syn.fixation_maps(s_fix, centres)
    report, bad = {}, {}
    for label, out_scale in (('A', 1.0), ('BC', 40.0)):
        hp = syn.grcn_params(s_head, T, gru_std=0.05, random_bn=True)
        hp['out_W'] = hp['out_W'] * out_scale
        with torch.no_grad():
            ref = torch_ref.softmax_maps(torch_ref.grcn_forward(ref_feat, {k: torch.tensor(v) for k, v in hp.items()}))
        ref = ref.numpy().reshape(N, 49, 49)
        head = GrcnEngine(B, T, dtype='bf16', device=gpu)
        head.set_weights(hp)
        _, probs = head.forward_rows(workload['rows'])
        head.status()
        got = probs.cpu().numpy().reshape(N, 49, 49)
        assert np.isfinite(got).all()
        cases = {'A': (gt, fix)} if label == 'A' else {'B': _fixations_following(ref, s_follow), 'C': (gt, fix)}
        for name, (g_, f_) in cases.items():
            s_ref, s_got = _metric_scores(ref, g_, f_, N), _metric_scores(got, g_, f_, N)
            for metric in s_ref:
                d = s_got[metric] - s_ref[metric]
                report[(name, metric)] = (round(s_ref[metric], 5), round(d, 6))
                tol = 1e-3 * max(1.0, abs(s_ref[metric])) if metric == 'NSS' else 1e-3
                if not abs(d) < tol:
                    bad[(name, metric)] = report[(name, metric)]
    print('config-3 metrics gate, 1024 frames, seeds %s (oracle score, HIP - oracle):' % (seeds,), report)
    assert not bad, (bad, report)
    assert report[('B', 'AUC_Judd')][0] > 0.8 and report[('B', 'AUC_Borji')][0] > 0.65, report
