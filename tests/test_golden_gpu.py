"""GPU parity against the committed golden fixtures, every intermediate of the head, and
size-independent properties at the benchmark's full size (B=64, T=16)."""
import os

import numpy as np
import pytest
import torch

from oracle import grcn, torch_ref
from recurrent_gaze_prediction_amd import evaluation_metrics as em
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = {'f32': 2e-5, 'bf16': 2e-2}


def rel_err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def regen(gold):
    B, T, P, S, seed = [int(v) for v in gold['config']]
    p = syn.grcn_params(seed, T, P, S, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(seed + 1, B, T)
    gt, centres = syn.gaze_maps(seed + 2, B, T)
    return (B, T, P, S), p, x, gt, centres


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('name', ['grcn_small.npz', 'grcn_refdims.npz'])
def test_logits_probs_loss_against_golden(gpu, name, dtype):
    from recurrent_gaze_prediction_amd.engine import GrcnEngine, softmax_xent
    gold = np.load(os.path.join(GOLD, name))
    (B, T, P, S), p, x, gt, _ = regen(gold)
    eng = GrcnEngine(B, T, P, S, dtype=dtype, device=gpu)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    assert rel_err(logits.cpu().numpy(), gold['logits']) < TOL[dtype]
    assert rel_err(probs.cpu().numpy(), gold['probs']) < TOL[dtype]
    g = torch.tensor(grcn.normalize_probability_map(gt).astype(np.float32), device=gpu)
    _, frame_loss, loss = softmax_xent(logits, g)
    # loss ~ ln 2401 + O(logit spread): compare the part that depends on the logits
    assert abs(loss.item() - float(gold['loss'])) < (2e-5 if dtype == 'f32' else 2e-3)
    assert abs(frame_loss.sum().item() / (B * T) - loss.item()) < 1e-5
    assert rel_err(eng.read_buffer('rcn_outputs').cpu().numpy().reshape(B, T, 7, 7, S)[:, -1], gold['h_last']) < 3 * TOL[dtype]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_every_intermediate_against_oracle(gpu, dtype):
    """projection, hoisted W*x, gates, states, BN, both hidden deconvs, logits."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T, P, S = 2, 3, 64, 64
    p = syn.grcn_params(51, T, P, S, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(52, B, T)
    ref_logits, it = grcn.forward(x, p, want_intermediates=True)
    # (unfolded_head: the plan that HAS the intermediate maps d1 / d2 -- the three-stage head; the folded one is checked against it
    # and against the oracle in tests/test_grcn_gpu.py)
    eng = GrcnEngine(B, T, P, S, dtype=dtype, save_for_backward=True, device=gpu, unfolded_head=True)
    eng.set_weights(p)
    logits, _ = eng.forward(torch.tensor(x, device=gpu))
    tol = TOL[dtype]
    get = lambda n: eng.read_buffer(n).cpu().numpy()
    assert rel_err(get('c3d_embedded').reshape(B, T, 7, 7, P), it['c3d_embedded']) < tol
    from oracle import np_ops
    emb = it['c3d_embedded'].reshape(B * T, 7, 7, P)
    xpre_ref = np.concatenate([np_ops.conv2d_same(emb, p[k]) for k in ('GRU_Conv_Wz', 'GRU_Conv_Wr', 'GRU_Conv_W')], -1)
    assert rel_err(get('xpre').reshape(B * T, 7, 7, 3 * S), xpre_ref) < tol
    for key in ('u', 'r', 'c'):
        ref = np.stack([g[key] for g in it['gates']], 0)                       # [T,B,7,7,S]
        assert rel_err(get(key).reshape(T, B, 7, 7, S), ref) < 3 * tol, key
    assert rel_err(get('rcn_outputs').reshape(B, T, 7, 7, S), it['rcn_outputs']) < 3 * tol
    for key, shape in (('bn', (B, T, 7, 7, S)), ('d1', (B, T, 23, 23, 64)), ('d2', (B, T, 49, 49, 32))):
        ref = np.stack([h[key] for h in it['head']], 1)
        assert rel_err(get(key).reshape(shape), ref) < 3 * tol, key
    assert rel_err(logits.cpu().numpy(), ref_logits) < tol


def test_saliency_metrics_within_1e3_of_oracle(gpu):
    """north star: AUC/CC (and sim) of the bf16 path within +-1e-3 of the fp32 oracle's."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T = 4, 3
    p = syn.grcn_params(61, T, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(62, B, T)
    gt, centres = syn.gaze_maps(63, B, T)
    fix = syn.fixation_maps(64, centres)
    ref = torch_ref.softmax_maps(torch_ref.grcn_forward(torch.tensor(x), {k: torch.tensor(v) for k, v in p.items()})).numpy()
    eng = GrcnEngine(B, T, dtype='bf16', device=gpu)
    eng.set_weights(p)
    _, probs = eng.forward(torch.tensor(x, device=gpu))
    got = probs.cpu().numpy()
    flat = lambda a: list(a.reshape(B * T, 49, 49))
    for metric in ('cc', 'sim', 'AUC_Borji', 'AUC_Judd'):
        scores = []
        for maps in (ref, got):
            np.random.seed(7)
            if metric == 'AUC_Judd':
                scores.append(np.mean([em.saliency_score_single(metric, m, g, f) for m, g, f in zip(flat(maps), flat(gt), flat(fix))]))
            else:
                scores.append(em.saliency_score(metric, flat(maps), flat(gt), flat(fix)))
        assert abs(scores[0] - scores[1]) < 1e-3, (metric, scores)


@pytest.fixture(scope='module')
def full_engine(gpu):
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    eng = GrcnEngine(64, 16, dtype='bf16', device=gpu)
    eng.set_weights(syn.grcn_params(71, 16, gru_std=0.05, random_bn=True))
    return eng


def test_full_size_properties(gpu, full_engine):
    """B=64, T=16 (the benchmark size), properties that need no oracle run:
    maps are distributions; clips are independent (permuting clips permutes outputs
    bit-exactly); the recurrence is causal (frames < t ignore a change at frame t);
    repeated calls are bit-identical."""
    eng = full_engine
    g = torch.Generator(device=gpu)
    g.manual_seed(5)
    x = torch.relu(torch.randn(64, 16, 1024, 7, 7, device=gpu, generator=g))
    logits, probs = eng.forward(x)
    logits, probs = logits.clone(), probs.clone()
    assert torch.isfinite(logits).all()
    assert torch.allclose(probs.reshape(64 * 16, -1).sum(-1), torch.ones(64 * 16, device=gpu), atol=1e-5)
    l2, _ = eng.forward(x)
    assert torch.equal(l2, logits)
    perm = torch.randperm(64, device=gpu, generator=g)
    lp, _ = eng.forward(x[perm].contiguous())
    assert torch.equal(lp, logits[perm])
    x2 = x.clone()
    x2[:, 9] = torch.relu(torch.randn(64, 1024, 7, 7, device=gpu, generator=g))
    lc, _ = eng.forward(x2)
    assert torch.equal(lc[:, :9], logits[:, :9])
    assert not torch.equal(lc[:, 9:], logits[:, 9:])


def test_full_size_spot_check_against_oracle(gpu, full_engine):
    """Two of the 64 clips of the full-size batch recomputed by the oracle."""
    eng = full_engine
    g = torch.Generator(device=gpu)
    g.manual_seed(6)
    x = torch.relu(torch.randn(64, 16, 1024, 7, 7, device=gpu, generator=g))
    logits, _ = eng.forward(x)
    p = {k: torch.tensor(v) for k, v in syn.grcn_params(71, 16, gru_std=0.05, random_bn=True).items()}
    for b in (0, 63):
        ref = torch_ref.grcn_forward(x[b:b + 1].cpu(), p).numpy()
        assert rel_err(logits[b:b + 1].cpu().numpy(), ref) < 2e-2


def test_c3d_golden_features_and_e2e_rows_path(gpu):
    """C3D features vs the float64 golden window; head fed by rows == head fed by features."""
    from recurrent_gaze_prediction_amd.engine import C3DEngine, GrcnEngine
    gold = np.load(os.path.join(GOLD, 'c3d_one_window.npz'))
    seed = int(gold['config'][0])
    cp = syn.c3d_params(seed, scale='he')
    v = torch.tensor(syn.video_windows(seed + 1, 1), device=gpu)
    for dtype, tol in (('f32', 1e-4), ('bf16', 3e-2)):
        c3d = C3DEngine(1, dtype=dtype, device=gpu)
        c3d.set_weights(cp)
        feats, rows = c3d.forward(v, want_features=True, want_rows=True)
        assert rel_err(feats.cpu().numpy(), gold['features']) < tol
        head = GrcnEngine(1, 1, dtype=dtype, device=gpu)
        head.set_weights(syn.grcn_params(81, 1, random_bn=True))
        la, _ = head.forward_rows(rows)
        lb, _ = head.forward(feats.reshape(1, 1, 1024, 7, 7).contiguous())
        assert rel_err(la.cpu().numpy(), lb.cpu().numpy()) < (1e-5 if dtype == 'f32' else 1e-2)


def test_softmax_xent_kernel_known_answers(gpu):
    from recurrent_gaze_prediction_amd.engine import softmax_xent
    z = torch.full((3, 2, 49, 49), 0.07, device=gpu)
    g = torch.rand(3, 2, 49, 49, device=gpu)
    g = g / g.sum((-1, -2), keepdim=True)
    probs, fl, loss = softmax_xent(z, g.contiguous())
    assert torch.allclose(probs, torch.full_like(probs, 1.0 / 2401), atol=1e-9)
    assert abs(loss.item() - np.log(2401.0)) < 1e-5
    z7 = torch.randn(5, 7, 7, device=gpu)                       # 7x7 maps (gaze_rnn77 style)
    p7, _, _ = softmax_xent(z7)
    assert torch.allclose(p7, torch.softmax(z7.reshape(5, -1), -1).reshape(5, 7, 7), atol=1e-6)


def _metric_scores(maps, gt, fix, n):
    flat = lambda a: list(np.asarray(a).reshape(n, 49, 49))
    out = {}
    for metric in ('cc', 'sim', 'AUC_Borji', 'AUC_shuffled', 'AUC_Judd', 'NSS'):      # AVAILABLE_METRICS + EXTRA_METRICS
        np.random.seed(7)
        if metric in ('AUC_Judd', 'NSS'):
            out[metric] = float(np.mean([em.saliency_score_single(metric, m, g, f) for m, g, f in zip(flat(maps), flat(gt), flat(fix))]))
        else:
            out[metric] = float(em.saliency_score(metric, flat(maps), flat(gt), flat(fix)))
    return out


def _fixations_following(maps, seed, n_fix=6, sigma=2.0, sharpen=2.0):
    """Fixation maps whose points are drawn from the given probability maps (raised to `sharpen`: observers look at the
    peaks), and the blurred ground-truth maps around them: what the data looks like to a TRAINED model (its maps and
    the fixations agree; AUC well above chance)."""
    rs = np.random.RandomState(seed)
    n = maps.shape[0]
    fix = np.zeros((n, 49, 49), np.float32)
    gt = np.zeros((n, 49, 49), np.float32)
    yy, xx = np.mgrid[0:49, 0:49]
    for i in range(n):
        p = np.asarray(maps[i], np.float64).reshape(-1) ** sharpen
        idx = rs.choice(2401, size=n_fix, replace=False, p=p / p.sum())
        for j in idx:
            y, x = divmod(int(j), 49)
            fix[i, y, x] = 1.0
            gt[i] += np.exp(-((yy - y) ** 2 + (xx - x) ** 2) / (2 * sigma ** 2))
    return gt, fix


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
def test_end_to_end_metrics_gate_T16(gpu, dtype):
    """The north-star acceptance gate on its own workload: 3 clips x T = 16 of synthetic video -> C3D conv1a..5b ->
    rows -> gaze_grcn head (bf16: patch kernels + persistent ConvGRU) -> per-frame softmax, against the fp32 CPU oracle
    of the SAME chain (torch_ref.c3d_forward -> grcn_forward): cc, sim, AUC_Borji, AUC_shuffled, AUC_Judd, NSS of the
    48 maps (evaluation_metrics.py:239-295) for
      A  the random-init head (nearly flat maps) on gaze data independent of them (AUCs at chance);
      B  peaked maps (output layer scaled to a logit range of ~12) on fixations that FOLLOW the oracle's maps -- the
         regime of a trained model, AUC well above chance;
      C  the peaked maps on independent gaze data.
    f32 plans: every metric within +-1e-3 in every case.  bf16 plans: cc, sim, AUC_Judd within +-1e-3; AUC_Borji and
    AUC_shuffled sweep thresholds in steps of 0.1 of the map's range, i.e. are step functions of the map, and bf16 moves
    the logits by 5e-3 of their range (half from the bf16 conv features, half from the bf16 head:
    profiles/r04_gate_attribution.json, scripts/gate_attribution.py) -- over 5 fixation seeds x 3 cases the measured
    drift of these 48-frame scores is <= 3.8e-3 (zero-mean flips of single ROC steps), NSS <= 2.8e-3 of its value: the
    bounds below are those measurements with a 1.6x margin, not a blanket 1e-2.  The flips average out with the number of
    frames: tests/test_gate_full_gpu.py holds +-1e-3 for every metric at the benchmark's 1024 frames."""
    from recurrent_gaze_prediction_amd.engine import C3DEngine, GrcnEngine
    B, T = 3, 16
    n = B * T
    cp = syn.c3d_params(65, scale='he')
    video = syn.video_windows(66, n)
    gt, centres = syn.gaze_maps(67, B, T)
    fix = syn.fixation_maps(68, centres)
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        with torch.no_grad():
            feat_ref = torch_ref.c3d_forward(torch.tensor(video), {k: torch.tensor(v) for k, v in cp.items()})
    finally:
        torch.set_num_threads(old)
    c3d = C3DEngine(n, dtype=dtype, device=gpu)
    c3d.set_weights(cp)
    feats, rows = c3d.forward(torch.tensor(video, device=gpu), want_features=True, want_rows=True)
    e_feat = rel_err(feats.cpu().numpy(), feat_ref.numpy())
    assert e_feat < (3e-2 if dtype == 'bf16' else 1e-4), e_feat
    report, bad = {}, {}
    for label, out_scale in (('A', 1.0), ('BC', 40.0)):
        hp = syn.grcn_params(61, T, gru_std=0.05, random_bn=True)
        hp['out_W'] = hp['out_W'] * out_scale
        with torch.no_grad():
            ref_logits = torch_ref.grcn_forward(feat_ref.reshape(B, T, 1024, 7, 7), {k: torch.tensor(v) for k, v in hp.items()})
            ref = torch_ref.softmax_maps(ref_logits).numpy().reshape(n, 49, 49)
        head = GrcnEngine(B, T, dtype=dtype, device=gpu)
        head.set_weights(hp)
        _, probs = head.forward_rows(rows)
        head.status()
        got = probs.cpu().numpy().reshape(n, 49, 49)
        assert np.isfinite(got).all()
        step_tol = 6e-3 if dtype == 'bf16' else 1e-3          # threshold-sweep AUCs and NSS at 48 frames (docstring)
        cases = {'A': (gt, fix)} if label == 'A' else {'B': _fixations_following(ref, 69), 'C': (gt, fix)}
        for name, (g_, f_) in cases.items():
            s_ref, s_got = _metric_scores(ref, g_, f_, n), _metric_scores(got, g_, f_, n)
            for metric in s_ref:
                report[(name, metric)] = (round(s_ref[metric], 5), round(s_got[metric] - s_ref[metric], 6))
                tol = {'AUC_Borji': step_tol, 'AUC_shuffled': step_tol,
                       'NSS': step_tol * max(1.0, abs(s_ref[metric]))}.get(metric, 1e-3)
                if not abs(s_ref[metric] - s_got[metric]) < tol:
                    bad[(name, metric)] = report[(name, metric)] + (tol,)
    assert not bad, ('(oracle score, HIP - oracle, bound)', bad, report)
    assert report[('B', 'AUC_Judd')][0] > 0.8 and report[('B', 'AUC_Borji')][0] > 0.7, report      # B really is the trained-like regime
    assert abs(report[('C', 'AUC_Borji')][0] - 0.5) < 0.1 and abs(report[('A', 'AUC_Borji')][0] - 0.5) < 0.1, report   # A, C sit at chance
    print('metrics gate %s: %s' % (dtype, report))
