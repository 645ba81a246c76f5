"""GPU parity: the two-level cascade (BASELINE config 5, gaze_grcn_cascade.py) against the torch-CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu
# max-abs error relative to the reference's max-abs, per stage
TOL = {'f32': dict(frm_sal=5e-5, rcn_outputs=5e-5, rcn_upsampled_outputs=5e-5, gaze_rcn_outputs=1e-4, maps=2e-4),
       'bf16': dict(frm_sal=3e-2, rcn_outputs=6e-2, rcn_upsampled_outputs=6e-2, gaze_rcn_outputs=6e-2, maps=6e-2)}


def rel_err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def to_t(p, dtype=torch.float64):
    return {k: to_t(v, dtype) if isinstance(v, dict) else torch.tensor(v, dtype=dtype) for k, v in p.items()}


def run_case(gpu, dtype, B, T, seed):
    from recurrent_gaze_prediction_amd.engine import CascadeEngine
    p = syn.cascade_params(seed)
    rs = np.random.RandomState(seed + 7)
    frames = rs.rand(B, T, 98, 98, 3).astype(np.float32)
    c3d = syn.c3d_features(seed + 8, B, T)
    ref, mid = torch_ref.cascade_forward(torch.tensor(frames, dtype=torch.float64), torch.tensor(c3d, dtype=torch.float64),
                                         to_t(p), want_all=True)
    eng = CascadeEngine(B, T, 98, dtype=dtype, device=gpu)
    eng.set_weights(p)
    maps = eng.forward(torch.tensor(frames, device=gpu), torch.tensor(c3d, device=gpu))
    got = {'frm_sal': mid['sal'], 'rcn_outputs': mid['bottom'], 'rcn_upsampled_outputs': mid['up'],
           'gaze_rcn_outputs': mid['top']}
    errs = {k: rel_err(eng.read_buffer(k).cpu().numpy(), v.numpy()) for k, v in got.items()}
    errs['maps'] = rel_err(maps.cpu().numpy(), ref.numpy())
    return errs, ref.numpy(), mid


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_cascade_matches_oracle(gpu, dtype):
    errs, ref, mid = run_case(gpu, dtype, 2, 3, 301)
    assert float((ref > 0).mean()) > 0.2 and np.abs(mid['top'].numpy()).max() > 0.05     # not degenerate
    for k, tol in TOL[dtype].items():
        assert errs[k] < tol, (k, errs)


def test_cascade_single_frame_and_second_call(gpu):
    """B = T = 1 (ragged smallest case), and a second forward on the same plan gives the same maps
    (the recurrent state is re-zeroed per call, gaze_grcn_cascade.py:293,357)."""
    from recurrent_gaze_prediction_amd.engine import CascadeEngine
    errs, _, _ = run_case(gpu, 'f32', 1, 1, 311)
    assert errs['maps'] < TOL['f32']['maps'], errs
    p = syn.cascade_params(5)
    eng = CascadeEngine(1, 2, 98, dtype='bf16', device=gpu)
    eng.set_weights(p)
    frames = torch.rand(1, 2, 98, 98, 3, device=gpu)
    c3d = torch.tensor(syn.c3d_features(6, 1, 2), device=gpu)
    a = eng.forward(frames, c3d).clone()
    b = eng.forward(frames, c3d)
    assert torch.equal(a, b)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_cascade_backward_matches_autograd(gpu, dtype):
    """rgp_cascade_backward: gradients of the l2 loss w.r.t. all 19 trainable arrays and w.r.t. the conv5b rows,
    against torch autograd through the float64 CPU restatement (ShallowNet frozen, base.py:264-265)."""
    from recurrent_gaze_prediction_amd.engine import CascadeEngine
    B, T = 2, 3
    p = syn.cascade_params(321)
    rs = np.random.RandomState(322)
    frames = rs.rand(B, T, 98, 98, 3).astype(np.float32)
    c3d = syn.c3d_features(323, B, T)
    gt, _ = syn.gaze_maps(324, B, T)
    gt = (gt / gt.max()).astype(np.float32)                      # same scale as the maps
    tp = to_t(p)
    keys = [k for k in tp if k != 'ShallowNet']
    for k in keys:
        tp[k].requires_grad_(True)
    x = torch.tensor(c3d, dtype=torch.float64, requires_grad=True)
    maps_ref = torch_ref.cascade_forward(torch.tensor(frames, dtype=torch.float64), x, tp)
    loss = torch_ref.gaze_loss(maps_ref, torch.tensor(gt, dtype=torch.float64), 'l2')
    loss.backward()
    eng = CascadeEngine(B, T, 98, dtype=dtype, device=gpu, save_for_backward=True)
    eng.set_weights(p)
    maps = eng.forward(torch.tensor(frames, device=gpu), torch.tensor(c3d, device=gpu))
    assert rel_err(maps.cpu().numpy(), maps_ref.detach().numpy()) < TOL[dtype]['maps']
    grads, d_rows = eng.backward(maps, torch.tensor(gt, device=gpu), want_d_rows=True)
    # bf16: a rounded forward flips the ReLU / maxout gate of a few near-zero units, which moves single entries of
    # the FC gradients by O(1) of their size; the RMS bound is the meaningful one there
    tol_max, tol_rms = (1e-3, 3e-4) if dtype == 'f32' else (3e-1, 6e-2)
    errs = {}
    for field, key in CascadeEngine.KEYS:
        ref = tp[key].grad.numpy()
        got = grads[field].cpu().numpy().astype(np.float64)
        assert np.abs(ref).max() > 0, key
        errs[key] = (rel_err(got, ref), float(np.sqrt(((got - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())))
    bad = {k: e for k, e in errs.items() if not (e[0] < tol_max and e[1] < tol_rms)}
    assert not bad, bad
    ref_rows = x.grad.reshape(B * T, 512, 2, 49).permute(0, 3, 2, 1).reshape(B * T * 49, 1024).numpy()
    assert rel_err(d_rows.cpu().numpy(), ref_rows) < (1e-3 if dtype == 'f32' else 1e-1)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_cascade_at_its_configured_length_T35(gpu, dtype):
    """BASELINE config 5 is "35-frame clips": B = 2, T = 35 -- the 256-channel bottom cell and the 49x49 top cell
    (5x5, 3 units) through 35 recurrent steps, forward per stage and the BPTT of both cells against float64 autograd
    (gaze_grcn_cascade.py:289-313, 346-382).  f32 <= 1e-3; bf16: stage tolerances of the T = 3 test, gradients <= 6e-2 RMS."""
    from recurrent_gaze_prediction_amd.engine import CascadeEngine
    B, T = 2, 35
    p = syn.cascade_params(331)
    rs = np.random.RandomState(332)
    frames = rs.rand(B, T, 98, 98, 3).astype(np.float32)
    c3d = syn.c3d_features(333, B, T)
    gt, _ = syn.gaze_maps(334, B, T)
    gt = (gt / gt.max()).astype(np.float32)
    tp = to_t(p)
    keys = [k for k in tp if k != 'ShallowNet']
    for k in keys:
        tp[k].requires_grad_(True)
    x = torch.tensor(c3d, dtype=torch.float64, requires_grad=True)
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        maps_ref, mid = torch_ref.cascade_forward(torch.tensor(frames, dtype=torch.float64), x, tp, want_all=True)
        torch_ref.gaze_loss(maps_ref, torch.tensor(gt, dtype=torch.float64), 'l2').backward()
    finally:
        torch.set_num_threads(old)
    eng = CascadeEngine(B, T, 98, dtype=dtype, device=gpu, save_for_backward=True)
    eng.set_weights(p)
    maps = eng.forward(torch.tensor(frames, device=gpu), torch.tensor(c3d, device=gpu))
    stages = {'frm_sal': mid['sal'], 'rcn_outputs': mid['bottom'], 'rcn_upsampled_outputs': mid['up'], 'gaze_rcn_outputs': mid['top']}
    errs = {k: rel_err(eng.read_buffer(k).cpu().numpy(), v.detach().numpy()) for k, v in stages.items()}
    errs['maps'] = rel_err(maps.cpu().numpy(), maps_ref.detach().numpy())
    # the last step is the hard one for bf16 (35 steps of compounding rounding)
    errs['last_top_state'] = rel_err(eng.read_buffer('gaze_rcn_outputs').cpu().numpy().reshape(B, T, 49, 49, 3)[:, -1],
                                     mid['top'].detach().numpy().reshape(B, T, 49, 49, 3)[:, -1])
    tol = dict(TOL[dtype], last_top_state=TOL[dtype]['gaze_rcn_outputs'])
    bad = {k: e for k, e in errs.items() if not e < (3 * tol[k] if dtype == 'f32' else tol[k])}
    assert not bad, (bad, errs)
    grads, d_rows = eng.backward(maps, torch.tensor(gt, device=gpu), want_d_rows=True)
    tol_max, tol_rms = (3e-3, 1e-3) if dtype == 'f32' else (3e-1, 6e-2)
    gerrs = {}
    for field, key in CascadeEngine.KEYS:
        ref = tp[key].grad.numpy()
        got = grads[field].cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all() and np.abs(ref).max() > 0, key
        gerrs[key] = (rel_err(got, ref), float(np.sqrt(((got - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())))
    bad = {k: e for k, e in gerrs.items() if not (e[0] < tol_max and e[1] < tol_rms)}
    assert not bad, (bad, gerrs)
    ref_rows = x.grad.reshape(B * T, 512, 2, 49).permute(0, 3, 2, 1).reshape(B * T * 49, 1024).numpy()
    assert rel_err(d_rows.cpu().numpy(), ref_rows) < (3e-3 if dtype == 'f32' else 1e-1)


def test_cascade_training_through_the_model_api(gpu, tmp_path):
    """single_step(train_mode=True) on the cascade class: l2 loss, backward, clipped TF-Adam; the loss on a fixed
    validation batch goes down and the state dict carries the updated variables (ShallowNet unchanged)."""
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_grcn_cascade import GazePredictionGRCN, GRUModelConfig
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir, cfg.init_seed = 2, 2, 'bf16', str(tmp_path), 3
    cfg.initial_learning_rate = 1e-3
    ds = type('DS', (), {})()
    ds.train = ds.valid = syn.SyntheticDataSet(8, 2, seed=5)
    model = GazePredictionGRCN(Session(gpu), ds, cfg)
    before = model.state_dict()
    model.single_step(train_mode=False, dataset=syn.SyntheticDataSet(8, 2, seed=5))
    loss0 = model.loss
    np.random.seed(4)
    for i in range(5):
        assert model.single_step(train_mode=True) == i + 1
    assert float(model.grad_norm.item()) > 0
    model.single_step(train_mode=False, dataset=syn.SyntheticDataSet(8, 2, seed=5))
    assert np.isfinite(model.loss) and model.loss < loss0, (loss0, model.loss)
    after = model.state_dict()
    assert not np.array_equal(after['LastProjection/fc2_w'], before['LastProjection/fc2_w'])
    assert not np.array_equal(after['RCNBottom/GRU_Conv_U'], before['RCNBottom/GRU_Conv_U'])
    assert np.array_equal(after['ShallowNet/fc1_w'], before['ShallowNet/fc1_w'])


def test_cascade_model_class(gpu, tmp_path):
    """models.gaze_grcn_cascade.GazePredictionGRCN: predict / l2 loss / checkpoint round trip."""
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_grcn_cascade import GazePredictionGRCN, GRUModelConfig
    cfg = GRUModelConfig()
    assert cfg.loss_type == 'l2'
    cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir, cfg.init_seed = 2, 2, 'f32', str(tmp_path), 3
    model = GazePredictionGRCN(Session(gpu), None, cfg)
    rs = np.random.RandomState(9)
    frames = rs.rand(2, 2, 98, 98, 3).astype(np.float32)
    c3d = syn.c3d_features(10, 2, 2)
    gt, _ = syn.gaze_maps(11, 2, 2)
    out = model.predict(c3d, frames).cpu().numpy()
    ref = torch_ref.cascade_forward(torch.tensor(frames, dtype=torch.float64), torch.tensor(c3d, dtype=torch.float64),
                                    to_t(model.variables))
    assert rel_err(out, ref.numpy()) < TOL['f32']['maps']
    want = float(torch_ref.gaze_loss(ref, torch.tensor(gt, dtype=torch.float64), 'l2'))
    assert abs(model.compute_loss(gt) - want) < 1e-4 * max(1.0, abs(want))
    state = model.state_dict()
    assert 'RCNGaze/GRU_Conv_Wz' in state and 'ShallowNet/fc1_w' in state and 'Upsampling/weight' in state
    state['LastProjection/fc2_b'] = state['LastProjection/fc2_b'] + 0.25
    model.load_state_dict(state)
    out2 = model.predict(c3d, frames).cpu().numpy()
    assert np.abs(out2 - out).max() > 0.1


@pytest.mark.parametrize('dtype,save', [('bf16', False), ('bf16', True), ('f32', False)])
def test_three_chain_forward_is_bit_identical_to_the_one_chain_form(gpu, dtype, save):
    """An eager forward runs the plan's three chains (bottom cell / per-step upsampling + input convolution / top cell, one time
    step apart on three streams); the same call captured into a HIP graph takes the one-chain form with the hoisted
    convolutions (no per-step events inside a capture).  Same GEMMs over a different partition of the rows: torch.equal --
    an event or a slice offset wrong in either form shows up here at every stage."""
    from recurrent_gaze_prediction_amd.engine import CascadeEngine
    B, T = 3, 6
    eng = CascadeEngine(B, T, 98, dtype=dtype, device=gpu, save_for_backward=save)
    eng.set_weights(syn.cascade_params(41))
    g = torch.Generator(device=gpu); g.manual_seed(5)
    frames = torch.rand(B, T, 98, 98, 3, device=gpu, generator=g)
    c3d = torch.tensor(syn.c3d_features(42, B, T), device=gpu)
    eager = eng.forward(frames, c3d)
    stages = {k: eng.read_buffer(k).clone() for k in eng.BUFFERS}
    assert bool(torch.isfinite(eager).all()) and float(eager.abs().max()) > 0
    s = torch.cuda.Stream(device=gpu)
    s.wait_stream(torch.cuda.current_stream(gpu))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        eng.forward(frames, c3d)                               # (eager on the capture stream first)
    torch.cuda.synchronize(gpu)
    with torch.cuda.graph(graph, stream=s):
        captured = eng.forward(frames, c3d)
    captured.zero_()
    graph.replay()
    torch.cuda.synchronize(gpu)
    assert torch.equal(captured, eager)
    for k, v in stages.items():
        assert torch.equal(eng.read_buffer(k), v), k
    # ... and the eager form again afterwards (its events and streams were not disturbed by the capture)
    assert torch.equal(eng.forward(frames, c3d), eager)


def test_three_chain_passes_are_repeatable(gpu):
    """The plan's streams and per-step events under repetition: 25 training passes on the same inputs give the same maps bit for
    bit and the same gradients up to the float-atomics' summation order (a missing dependency between the chains would show as
    a pass that read a half-written frame)."""
    from recurrent_gaze_prediction_amd.engine import CascadeEngine
    B, T = 4, 9
    eng = CascadeEngine(B, T, 98, dtype='bf16', device=gpu, save_for_backward=True)
    eng.set_weights(syn.cascade_params(43))
    g = torch.Generator(device=gpu); g.manual_seed(6)
    frames = torch.rand(B, T, 98, 98, 3, device=gpu, generator=g)
    c3d = torch.tensor(syn.c3d_features(44, B, T), device=gpu)
    gt = torch.rand(B, T, 49, 49, device=gpu, generator=g)
    maps0 = eng.forward(frames, c3d).clone()
    _, d0 = eng.backward(maps0, gt, want_d_rows=True)
    g0, d0 = eng.flat_grads.clone(), d0.clone()
    scale_g, scale_d = float(g0.abs().max()), float(d0.abs().max())
    assert scale_g > 0 and scale_d > 0
    for _ in range(25):
        maps = eng.forward(frames, c3d)
        assert torch.equal(maps, maps0)
        _, d = eng.backward(maps, gt, want_d_rows=True)
        assert float((eng.flat_grads - g0).abs().max()) < 2e-4 * scale_g
        assert float((d - d0).abs().max()) < 2e-3 * scale_d          # (bf16 rows behind fp32 atomics)
