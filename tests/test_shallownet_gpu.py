"""GPU parity: the frame-wise ShallowNet (BASELINE config 1) against the torch-CPU oracle."""
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
@pytest.mark.parametrize('hw', [98, 112])
def test_shallownet_matches_oracle(gpu, dtype, hw):
    from recurrent_gaze_prediction_amd.engine import ShallowNetEngine
    p = syn.shallownet_params(141, hw)
    p = dict(p, conv1_b=np.linspace(-0.1, 0.1, 32).astype(np.float32), fc1_b=np.linspace(-0.05, 0.05, 4802).astype(np.float32),
             fc2_b=np.linspace(0.05, -0.05, 4802).astype(np.float32))
    rs = np.random.RandomState(142)
    frames = rs.rand(2, hw, hw, 3).astype(np.float32)          # configs[0]: batch = 2 frames
    ref = torch_ref.shallownet_forward(torch.tensor(frames, dtype=torch.float64),
                                       {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}).numpy()
    eng = ShallowNetEngine(4, hw, dtype=dtype, device=gpu)
    eng.set_weights(p)
    sal, sal7 = eng.forward(torch.tensor(frames, device=gpu), want_7x7=True)
    assert rel_err(sal.cpu().numpy(), ref) < TOL[dtype]
    ref7 = ref.reshape(2, 7, 7, 7, 7).mean(axis=(2, 4))          # 7x7 avg-pool, stride 7 (gaze_rnn.py:262-269)
    assert rel_err(sal7.cpu().numpy(), ref7) < TOL[dtype]
    assert float((ref > 0).mean()) > 0.2                          # not a degenerate all-zero map


def test_framewise_model_class_config1(gpu, tmp_path):
    """models.gaze_framewise_shallownet.FramewiseShallowNet: 112x112 frames -> 7x7 maps, batch 2 (configs[0])."""
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_framewise_shallownet import FramewiseShallowNet, GRUModelConfig
    cfg = GRUModelConfig()
    assert (cfg.n_lstm_steps, cfg.batch_size, cfg.loss_type) == (35, 5, 'l2')      # gaze_framewise_shallownet.py:43-57
    cfg.batch_size, cfg.n_lstm_steps, cfg.image_hw, cfg.train_dir = 2, 1, 112, str(tmp_path)
    model = FramewiseShallowNet(Session(gpu), None, cfg, gazemap_height=7, gazemap_width=7)
    frames = np.random.RandomState(5).rand(2, 1, 112, 112, 3).astype(np.float32)
    out = model.predict(np.zeros((2, 1, 1024, 7, 7), np.float32), frames).cpu().numpy()
    assert out.shape == (2, 1, 7, 7)
    ref = torch_ref.shallownet_forward(torch.tensor(frames.reshape(2, 112, 112, 3)),
                                       {k: torch.tensor(v) for k, v in model.variables.items()}).numpy()
    assert rel_err(out.reshape(2, 7, 7), ref.reshape(2, 7, 7, 7, 7).mean(axis=(2, 4))) < 1e-4


@pytest.mark.parametrize('dtype,hw', [('f32', 98), ('f32', 112), ('bf16', 98)])
def test_shallownet_backward_matches_autograd(gpu, dtype, hw):
    """rgp_shallownet_backward: gradients of <saliency, g> w.r.t. all ten variables vs float64 autograd.  The
    net is piecewise linear (ReLU, overlapping max-pools, maxout): an occasional gate resolved differently
    moves single entries, so the bound is on the RMS error plus a looser max."""
    from recurrent_gaze_prediction_amd.engine import ShallowNetEngine
    n = 3
    p = syn.shallownet_params(151, hw)
    p = dict(p, conv1_b=np.linspace(-0.1, 0.1, 32).astype(np.float32), conv2_b=np.linspace(0.05, -0.05, 64).astype(np.float32),
             fc1_b=np.linspace(-0.05, 0.05, 4802).astype(np.float32), fc2_b=np.linspace(0.05, -0.05, 4802).astype(np.float32))
    rs = np.random.RandomState(152)
    frames = rs.rand(n, hw, hw, 3).astype(np.float32)
    g = rs.randn(n, 49, 49).astype(np.float32)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    sal_ref = torch_ref.shallownet_forward(torch.tensor(frames, dtype=torch.float64), pt)
    (sal_ref * torch.tensor(g, dtype=torch.float64)).sum().backward()
    eng = ShallowNetEngine(4, hw, dtype=dtype, device=gpu, save_for_backward=True)
    eng.set_weights(p)
    sal, _ = eng.forward(torch.tensor(frames, device=gpu))
    assert rel_err(sal.cpu().numpy(), sal_ref.detach().numpy()) < TOL[dtype]
    grads = eng.backward(torch.tensor(g, device=gpu))
    # bf16: with ~3 significant digits the first maximum of a 3x3 window (9 near-equal candidates) often is a
    # neighbour of the fp64 one, which re-routes that window's gradient; the kernels' logic is pinned by the fp32
    # runs, the bf16 run must stay well aligned with the reference gradient (cosine) and of the same size
    bad = {}
    for k in p:
        ref = pt[k].grad.numpy()
        got = grads[k].cpu().numpy().astype(np.float64)
        assert np.abs(ref).max() > 0, k
        rms = float(np.sqrt(((got - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))
        cos = float((got * ref).sum() / np.sqrt((got ** 2).sum() * (ref ** 2).sum()))
        ok = (rms < 1e-3 and rel_err(got, ref) < 2e-2) if dtype == 'f32' else (cos > 0.95 and rms < 0.35)
        if not ok:
            bad[k] = (rms, cos)
    assert not bad, bad


@pytest.mark.parametrize('hw', [98, 112])
def test_conv1_frame_kernel_against_the_general_tile_and_the_oracle(gpu, hw):
    """From 48 frames up conv1 runs on shallow_conv1_bf16_kernel (a workgroup per frame); below, on the general
    implicit-GEMM tile.  The same two frames through both: equal maps up to bf16 rounding, both inside the oracle
    tolerance; and the arg-max codes the frame kernel records route the same gradient (all ten variables)."""
    from recurrent_gaze_prediction_amd.engine import ShallowNetEngine
    p = syn.shallownet_params(161, hw)
    p = dict(p, conv1_b=np.linspace(-0.1, 0.1, 32).astype(np.float32))
    rs = np.random.RandomState(162)
    frames = rs.rand(48, hw, hw, 3).astype(np.float32)
    g = np.zeros((48, 49, 49), np.float32)
    g[:2] = rs.randn(2, 49, 49)
    big = ShallowNetEngine(48, hw, dtype='bf16', device=gpu, save_for_backward=True)
    small = ShallowNetEngine(4, hw, dtype='bf16', device=gpu, save_for_backward=True)
    big.set_weights(p)
    small.set_weights(p)
    sal_big, _ = big.forward(torch.tensor(frames, device=gpu))
    sal_small, _ = small.forward(torch.tensor(frames[:2], device=gpu))
    ref = torch_ref.shallownet_forward(torch.tensor(frames[:2], dtype=torch.float64),
                                       {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}).numpy()
    assert rel_err(sal_big[:2].cpu().numpy(), ref) < TOL['bf16'] and rel_err(sal_small.cpu().numpy(), ref) < TOL['bf16']
    assert rel_err(sal_big[:2].cpu().numpy(), sal_small.cpu().numpy()) < 1e-2
    assert float((sal_big[2:] > 0).float().mean()) > 0.2           # the other 46 frames are not degenerate either
    gb = big.backward(torch.tensor(g, device=gpu))
    gs = small.backward(torch.tensor(g[:2], device=gpu))
    for k in p:
        a, b = gb[k].double().flatten(), gs[k].double().flatten()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos > 0.995 and abs(float(a.norm() / b.norm()) - 1) < 2e-2, (k, cos)


def test_framewise_training_through_the_model_api(gpu, tmp_path):
    """FramewiseShallowNet.single_step(train_mode=True): l2 loss on 49x49 maps, all ten variables move."""
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_framewise_shallownet import FramewiseShallowNet, GRUModelConfig
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.train_dir, cfg.compute_dtype = 2, 2, str(tmp_path), 'bf16'
    cfg.initial_learning_rate = 1e-4
    ds = type('DS', (), {})()
    ds.train = ds.valid = syn.SyntheticDataSet(8, 2, seed=7)
    model = FramewiseShallowNet(Session(gpu), ds, cfg)
    before = model.state_dict()
    model.single_step(train_mode=False, dataset=syn.SyntheticDataSet(8, 2, seed=7))
    loss0 = model.loss
    np.random.seed(2)
    for i in range(5):
        assert model.single_step(train_mode=True) == i + 1
    model.single_step(train_mode=False, dataset=syn.SyntheticDataSet(8, 2, seed=7))
    assert np.isfinite(model.loss) and model.loss < loss0, (loss0, model.loss)
    after = model.state_dict()
    assert all(not np.array_equal(after[k], before[k]) for k in ('conv1_w', 'conv3_w', 'fc1_w', 'fc2_b'))
