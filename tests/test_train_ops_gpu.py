"""GPU parity of the training ops added around the models: the dropout op (device Philox mask, forward and
backward of the fc-GRU and cascade sites fed the SAME mask as the oracle), the l2 loss kernel, the RMSProp /
momentum optimizers of base.py:268-273, and checkpoint resume with optimizer slots."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def rel_err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def to_t(p, dtype=torch.float64):
    return {k: to_t(v, dtype) if isinstance(v, dict) else torch.tensor(v, dtype=dtype) for k, v in p.items()}


@pytest.mark.parametrize('n,keep,seed,offset', [(1, 0.5, 0, 0), (7, 0.5, 3, 0), (4096, 0.5, 1234, 0),
                                                 (100003, 0.8, (1 << 40) + 17, 999), (50176, 0.25, 5, 1 << 33)])
def test_device_philox_mask_is_bit_exact(gpu, n, keep, seed, offset):
    """rgp_dropout_mask vs the oracle's Philox-4x32-10 (pinned to the published Random123 vectors in
    tests/test_oracle_cpu.py): every byte equal, for ragged n, 64-bit seeds and offsets."""
    from recurrent_gaze_prediction_amd import _lib
    lib = _lib.load()
    mask = torch.full((n + 5,), 7, dtype=torch.uint8, device=gpu)
    _lib.check(lib.rgp_dropout_mask(ctypes.c_void_p(mask.data_ptr()), n, keep, seed, offset,
                                    ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)))
    got = mask.cpu().numpy()
    assert np.array_equal(got[:n], torch_ref.dropout_mask(n, keep, seed, offset))
    assert (got[n:] == 7).all()                                   # nothing written past n


def test_dropout_site_advances_its_counter(gpu):
    """Two consecutive draws differ, equal the oracle at offsets 0 and ceil(n/4), and keep ~ keep_prob."""
    from recurrent_gaze_prediction_amd.engine import FcGruEngine
    eng = FcGruEngine(2, 3, (7, 7), dtype='f32', device=gpu)
    eng.dropout.configure(0.5, seed=42)
    n = eng.dropout.n
    eng.dropout.draw()
    a = eng.dropout.mask.cpu().numpy().copy()
    eng.dropout.draw()
    b = eng.dropout.mask.cpu().numpy().copy()
    assert np.array_equal(a, torch_ref.dropout_mask(n, 0.5, 42, 0))
    assert np.array_equal(b, torch_ref.dropout_mask(n, 0.5, 42, (n + 3) // 4))
    assert not np.array_equal(a, b) and abs(a.mean() - 0.5) < 0.03


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_fcgru_dropout_forward_and_backward_match_oracle(gpu, dtype):
    """keep 0.5 on c3d_embedded (gaze_rnn.py:302-303,529): logits and all 8 gradients vs float64 autograd through
    the oracle fed the same mask; with the site off the engine reproduces the inference forward."""
    from recurrent_gaze_prediction_amd.engine import FcGruEngine
    B, T, GH = 3, 4, 7
    p = syn.fcgru_params(151, GH, GH)
    x = syn.c3d_features(152, B, T)
    rs = np.random.RandomState(153)
    gt = rs.rand(B, T, GH, GH).astype(np.float32)
    gt /= gt.sum(axis=(2, 3), keepdims=True)
    mask = torch_ref.dropout_mask(B * T * 49 * 32, 0.5, 77).reshape(B * T * 49, 32)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    ref = torch_ref.fcgru_forward(torch.tensor(x, dtype=torch.float64), pt, GH, GH, keep_prob=0.5, drop_mask=mask)
    torch_ref.gaze_loss(ref, torch.tensor(gt, dtype=torch.float64), 'xentropy').backward()
    eng = FcGruEngine(B, T, (GH, GH), dtype=dtype, device=gpu, save_for_backward=True)
    eng.set_weights(p)
    xd = torch.tensor(x, device=gpu)
    plain = eng.forward(xd)[0].clone()
    eng.dropout.use(mask, 0.5)
    logits, probs = eng.forward(xd, train='keep')
    tol = 5e-5 if dtype == 'f32' else 3e-2
    assert rel_err(logits.cpu().numpy(), ref.detach().numpy()) < tol
    assert rel_err(plain.cpu().numpy(), ref.detach().numpy()) > 10 * tol        # the mask really acts
    grads = eng.backward(logits, probs, torch.tensor(gt, device=gpu), 'xentropy')
    gtol = 5e-4 if dtype == 'f32' else 5e-2
    errs = {k: rel_err(grads[k].cpu().numpy(), pt[k].grad.numpy()) for k in p}
    assert max(errs.values()) < gtol, errs
    again = eng.forward(xd)[0]                                                   # inference: site off again
    assert torch.equal(again, plain)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_cascade_fc1_dropout_matches_oracle(gpu, dtype):
    """keep 0.5 on relu(fc1) before the maxout (gaze_grcn_cascade.py:401-402): maps and the FC / top-cell gradients
    against autograd through the oracle with the same mask."""
    from recurrent_gaze_prediction_amd.engine import CascadeEngine
    B, T = 2, 2
    p = syn.cascade_params(331)
    rs = np.random.RandomState(332)
    frames = rs.rand(B, T, 98, 98, 3).astype(np.float32)
    c3d = syn.c3d_features(333, B, T)
    gt, _ = syn.gaze_maps(334, B, T)
    gt = (gt / gt.max()).astype(np.float32)
    mask = torch_ref.dropout_mask(B * T * 4802, 0.5, 99).reshape(B, T, 4802)
    tp = to_t(p)
    keys = [k for k in tp if k != 'ShallowNet']
    for k in keys:
        tp[k].requires_grad_(True)
    ref = torch_ref.cascade_forward(torch.tensor(frames, dtype=torch.float64), torch.tensor(c3d, dtype=torch.float64), tp,
                                    keep_prob=0.5, drop_mask=mask)
    torch_ref.gaze_loss(ref, torch.tensor(gt, dtype=torch.float64), 'l2').backward()
    eng = CascadeEngine(B, T, 98, dtype=dtype, device=gpu, save_for_backward=True)
    eng.set_weights(p)
    fd, cd = torch.tensor(frames, device=gpu), torch.tensor(c3d, device=gpu)
    plain = eng.forward(fd, cd).clone()
    eng.dropout.use(mask, 0.5)
    maps = eng.forward(fd, cd, train='keep')
    tol = 2e-4 if dtype == 'f32' else 6e-2
    assert rel_err(maps.cpu().numpy(), ref.detach().numpy()) < tol
    assert rel_err(plain.cpu().numpy(), ref.detach().numpy()) > 10 * tol
    grads, _ = eng.backward(maps, torch.tensor(gt, device=gpu))
    # bf16: a rounded forward flips single ReLU / maxout / (here also x2-scaled) dropout-survivor gates: single entries
    # move by O(1) of their size, the RMS bound is the meaningful one (as in tests/test_cascade_gpu.py)
    tol_max, tol_rms = (1e-3, 3e-4) if dtype == 'f32' else (8e-1, 6e-2)
    bad = {}
    for field, key in CascadeEngine.KEYS:
        r = tp[key].grad.numpy()
        g = grads[field].cpu().numpy().astype(np.float64)
        e = (rel_err(g, r), float(np.sqrt(((g - r) ** 2).mean()) / np.sqrt((r ** 2).mean())))
        if not (e[0] < tol_max and e[1] < tol_rms):
            bad[key] = e
    assert not bad, bad


def test_l2_loss_kernel(gpu):
    from recurrent_gaze_prediction_amd.engine import l2_loss
    rs = np.random.RandomState(5)
    a, b = rs.randn(3, 5, 49, 49).astype(np.float32), rs.rand(3, 5, 49, 49).astype(np.float32)
    got = float(l2_loss(torch.tensor(a, device=gpu), torch.tensor(b, device=gpu), 15).item())
    want = float(torch_ref.gaze_loss(torch.tensor(a, dtype=torch.float64), torch.tensor(b, dtype=torch.float64), 'l2'))
    assert abs(got - want) < 1e-5 * abs(want)


class _FlatEngine(object):
    """Minimal flat_params / flat_grads owner for the optimizer kernels (no plan behind it)."""

    def __init__(self, params, device):
        from recurrent_gaze_prediction_amd import _lib
        self.lib, self.device = _lib.load(), device
        self.flat_params = torch.tensor(params, device=device)
        self.flat_grads = torch.zeros_like(self.flat_params)

    def repack(self):
        pass


@pytest.mark.parametrize('method', ['adam', 'rmsprop', 'sgd'])
def test_optimizers_match_tf_semantics(gpu, method):
    """Three steps of clip_by_global_norm + the optimizer on two buffers (global norm over both) vs the oracle's
    float64 restatement of the TF update rules (base.py:268-297)."""
    from recurrent_gaze_prediction_amd.engine import clip_step_multi
    rs = np.random.RandomState(7)
    p0 = {'a': rs.randn(1000).astype(np.float32), 'b': rs.randn(37).astype(np.float32)}
    engs = [_FlatEngine(p0['a'], gpu), _FlatEngine(p0['b'], gpu)]
    ref = {k: torch.tensor(v, dtype=torch.float64) for k, v in p0.items()}
    s1 = {k: torch.zeros_like(v) for k, v in ref.items()}
    s2 = {k: (torch.ones_like(v) if method == 'rmsprop' else torch.zeros_like(v)) for k, v in ref.items()}
    for step in range(3):
        g = {'a': (rs.randn(1000) * 3).astype(np.float32), 'b': (rs.randn(37) * 3).astype(np.float32)}
        engs[0].flat_grads.copy_(torch.tensor(g['a']))
        engs[1].flat_grads.copy_(torch.tensor(g['b']))
        gn = clip_step_multi(engs, step, 1e-2, 10.0, method)
        gc, norm = torch_ref.clip_by_global_norm({k: torch.tensor(v, dtype=torch.float64) for k, v in g.items()}, 10.0)
        assert norm > 10.0 and abs(float(gn.item()) - norm) < 1e-4 * norm
        if method == 'adam':
            ref, s1, s2 = torch_ref.adam_step_tf(ref, gc, s1, s2, step, 1e-2)
        elif method == 'rmsprop':
            ref, s2, s1 = torch_ref.rmsprop_step_tf(ref, gc, s2, s1, 1e-2)
        else:
            ref, s1 = torch_ref.momentum_step_tf(ref, gc, s1, 1e-2)
    assert rel_err(engs[0].flat_params.cpu().numpy(), ref['a'].numpy()) < 2e-6
    assert rel_err(engs[1].flat_params.cpu().numpy(), ref['b'].numpy()) < 2e-6


def test_invalid_optimizer_is_refused(gpu, tmp_path):
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_grcn import GazePredictionGRCN, GRUModelConfig
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.train_dir, cfg.optimization_method = 2, 2, str(tmp_path), 'adagrad'
    with pytest.raises(ValueError):
        GazePredictionGRCN(Session(gpu), None, cfg)


@pytest.mark.parametrize('method', ['adam', 'rmsprop'])
def test_checkpoint_resume_equals_uninterrupted_training(gpu, tmp_path, method):
    """save -> load into a fresh model -> step  ==  step on the uninterrupted model (optimizer slots, global step,
    learning-rate scale and the augmentation stream all ride in the checkpoint, base.py:236-251)."""
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_grcn import GazePredictionGRCN, GRUModelConfig

    def make(d):
        cfg = GRUModelConfig()
        cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir = 2, 3, 'f32', str(d)
        cfg.initial_learning_rate, cfg.optimization_method, cfg.init_seed = 1e-3, method, 4
        ds = type('DS', (), {})()
        ds.train = ds.valid = syn.SyntheticDataSet(8, 3, seed=21)
        return GazePredictionGRCN(Session(gpu), ds, cfg)

    a = make(tmp_path / 'a')
    for _ in range(3):
        a.single_step(train_mode=True)
    a.decay_learning_rate(0.5)
    path = a.save_model_checkpoint(a.train_dir)
    ck0 = a.state_dict()
    b = make(tmp_path / 'b')
    for _ in range(3):                                   # same position in the data stream
        b.data_sets.train.next_batch(2)
    b.load_model_from_checkpoint_file(path)
    assert b.current_step == 3 and b.current_learning_rate == a.current_learning_rate
    a.single_step(train_mode=True)
    b.single_step(train_mode=True)
    sa, sb = a.state_dict(), b.state_dict()
    step_size = max(np.abs(sa[k] - ck0[k]).max() for k in sa)           # how far the 4th step moved the variables
    for k in sa:      # (the filter gradients are summed with float atomics: equal to rounding, not bit for bit)
        assert np.abs(sa[k] - sb[k]).max() < 2e-3 * step_size, k
    # and the slots matter: a resume WITHOUT them takes a visibly different step
    c = make(tmp_path / 'c')
    for _ in range(3):
        c.data_sets.train.next_batch(2)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    c.load_state_dict(ck['variables'])
    c._global_step = 3
    c.flip_rng.set_state(ck['flip_rng_state'])
    c._learning_rate_scale = ck['learning_rate_scale']
    c.single_step(train_mode=True)
    sc = c.state_dict()
    assert max(np.abs(sc[k] - sa[k]).max() for k in sa) > 0.2 * step_size
