"""GPU parity: the fc-GRU gaze model (BASELINE config 2) against the torch-CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TOL = {'f32': 5e-5, 'bf16': 3e-2}


def rel_err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('B,T,GH', [(3, 4, 49), (2, 3, 7)])
def test_fcgru_forward_matches_oracle(gpu, dtype, B, T, GH):
    from recurrent_gaze_prediction_amd.engine import FcGruEngine
    p = syn.fcgru_params(131, GH, GH)
    x = syn.c3d_features(132, B, T)
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    ref = torch_ref.fcgru_forward(torch.tensor(x, dtype=torch.float64), pt, GH, GH).numpy()
    eng = FcGruEngine(B, T, (GH, GH), dtype=dtype, device=gpu)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    assert rel_err(logits.cpu().numpy(), ref) < TOL[dtype]
    ref_p = torch.softmax(torch.tensor(ref).reshape(B, T, -1), -1).reshape(ref.shape).numpy()
    assert rel_err(probs.cpu().numpy(), ref_p) < TOL[dtype]


def test_gaze_rnn_model_class_runs_config2(gpu, tmp_path):
    """models.gaze_rnn.GazePredictionGRU: fc-GRU over conv5b features, 16-step clips, fp32 (config 2)."""
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_rnn import GazePredictionGRU, GRUModelConfig
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir = 2, 16, 'f32', str(tmp_path)
    ds = type('DS', (), {})()
    ds.train = ds.valid = syn.SyntheticDataSet(10, 16, seed=9)
    model = GazePredictionGRU(Session(gpu), ds, cfg, gazemap_height=7, gazemap_width=7)
    ret = model.generate(ds.valid, max_instances=4)
    assert ret['pred_gazemap_list'].shape == (4 * 16, 7, 7)
    assert np.allclose(ret['pred_gazemap_list'].reshape(64, -1).sum(-1), 1.0, atol=1e-5)
    ds2 = syn.SyntheticDataSet(10, 16, seed=9)
    _, _, _, c3d, _, _ = ds2.next_batch(2)
    pt = {k: torch.tensor(v) for k, v in model.variables.items()}
    ref = torch.softmax(torch_ref.fcgru_forward(torch.tensor(c3d.reshape(2, 16, 1024, 7, 7)), pt, 7, 7).reshape(2, 16, -1), -1)
    assert rel_err(ret['pred_gazemap_list'][:32].reshape(2, 16, 49), ref.numpy()) < 1e-4


@pytest.mark.parametrize('dtype,loss_type', [('f32', 'xentropy'), ('f32', 'l2'), ('bf16', 'xentropy')])
def test_fcgru_backward_matches_autograd(gpu, dtype, loss_type):
    """rgp_fcgru_backward: gradients of the reference loss w.r.t. all 8 variables vs float64 autograd."""
    from recurrent_gaze_prediction_amd.engine import FcGruEngine
    B, T, GH = 3, 4, 7
    p = syn.fcgru_params(141, GH, GH)
    p['gates_bias'] = p['gates_bias'] + np.linspace(-0.3, 0.3, p['gates_bias'].size).astype(np.float32)
    x = syn.c3d_features(142, B, T)
    rs = np.random.RandomState(143)
    gt = rs.rand(B, T, GH, GH).astype(np.float32)
    gt /= gt.sum(axis=(2, 3), keepdims=True)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    logits_ref = torch_ref.fcgru_forward(torch.tensor(x, dtype=torch.float64), pt, GH, GH)
    torch_ref.gaze_loss(logits_ref, torch.tensor(gt, dtype=torch.float64), loss_type).backward()
    eng = FcGruEngine(B, T, (GH, GH), dtype=dtype, device=gpu, save_for_backward=True)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    grads = eng.backward(logits, probs, torch.tensor(gt, device=gpu), loss_type)
    tol = 5e-4 if dtype == 'f32' else 5e-2
    errs = {k: rel_err(grads[k].cpu().numpy(), pt[k].grad.numpy()) for k in p}
    assert all(np.abs(pt[k].grad.numpy()).max() > 0 for k in p)
    assert max(errs.values()) < tol, errs


def test_config2_training_steps_through_the_model_api(gpu, tmp_path):
    """GazePredictionGRU.single_step(train_mode=True): fc-GRU training (config 2), loss falls on a fixed batch."""
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_rnn import GazePredictionGRU, GRUModelConfig
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir = 2, 4, 'bf16', str(tmp_path)
    cfg.initial_learning_rate = 1e-3
    ds = type('DS', (), {})()
    ds.train = ds.valid = syn.SyntheticDataSet(8, 4, seed=9)
    model = GazePredictionGRU(Session(gpu), ds, cfg)
    before = model.state_dict()
    model.single_step(train_mode=False, dataset=syn.SyntheticDataSet(8, 4, seed=9))
    loss0 = model.loss
    np.random.seed(1)
    for i in range(5):
        assert model.single_step(train_mode=True) == i + 1
    model.single_step(train_mode=False, dataset=syn.SyntheticDataSet(8, 4, seed=9))
    assert np.isfinite(model.loss) and model.loss < loss0, (loss0, model.loss)
    after = model.state_dict()
    assert not np.array_equal(after['gates_kernel'], before['gates_kernel'])


def test_fcgru_at_config2_shape_matches_oracle_forward_and_backward(gpu):
    """BASELINE config 2's own shape: 64 clips x T = 16, 7x7 maps, fp32 -- the recurrence GEMMs then have 64 live rows
    (igemm_skinny_kernel's full row tile; the small cases above run 2 - 3).  Forward against the float64 oracle, the
    gradients of all 8 variables against float64 autograd of the same loss (models/gaze_rnn.py:211-360)."""
    from recurrent_gaze_prediction_amd.engine import FcGruEngine
    B, T, GH = 64, 16, 7
    p = syn.fcgru_params(151, GH, GH)
    p['gates_bias'] = p['gates_bias'] + np.linspace(-0.3, 0.3, p['gates_bias'].size).astype(np.float32)
    x = syn.c3d_features(152, B, T)
    rs = np.random.RandomState(153)
    gt = rs.rand(B, T, GH, GH).astype(np.float32)
    gt /= gt.sum(axis=(2, 3), keepdims=True)
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
        logits_ref = torch_ref.fcgru_forward(torch.tensor(x, dtype=torch.float64), pt, GH, GH)
        torch_ref.gaze_loss(logits_ref, torch.tensor(gt, dtype=torch.float64), 'xentropy').backward()
    finally:
        torch.set_num_threads(old)
    eng = FcGruEngine(B, T, (GH, GH), dtype='f32', device=gpu, save_for_backward=True)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    ref = logits_ref.detach().numpy()
    # every clip on its own scale: a row tile that dropped or repeated a clip would show in that clip
    per_clip = np.abs(logits.cpu().numpy().astype(np.float64) - ref).reshape(B, -1).max(1) / np.abs(ref).reshape(B, -1).max(1)
    assert per_clip.max() < TOL['f32'], (int(per_clip.argmax()), float(per_clip.max()))
    grads = eng.backward(logits, probs, torch.tensor(gt, device=gpu), 'xentropy')
    errs = {k: rel_err(grads[k].cpu().numpy(), pt[k].grad.numpy()) for k in p}
    assert all(np.abs(pt[k].grad.numpy()).max() > 0 for k in p)
    assert max(errs.values()) < 5e-4, errs
