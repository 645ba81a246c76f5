"""GPU: an exported TF-name checkpoint dict, imported through checkpoint.py and loaded into each
model class, reproduces the committed golden outputs of that graph."""
import os

import numpy as np
import pytest
import torch

from recurrent_gaze_prediction_amd import checkpoint
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


def _scramble(state):
    """A state dict the model must NOT keep: every array replaced by noise."""
    rs = np.random.RandomState(0)
    return {k: rs.randn(*np.shape(v)).astype(np.float32) * 0.01 for k, v in state.items()}


def test_gaze_grcn_import_reproduces_golden_logits(gpu, tmp_path):
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_grcn import GazePredictionGRCN, GRUModelConfig
    g = np.load(os.path.join(GOLD, 'grcn_refdims.npz'))
    B, T, P, S, seed = [int(v) for v in g['config']]
    assert (P, S) == (512, 128)
    want = syn.grcn_params(seed, T, P, S, gru_std=0.05, random_bn=True)
    tf_vars = {k + ':0': v for k, v in checkpoint.export_tf_variables(want).items()}
    np.savez(tmp_path / 'export.npz', **tf_vars)
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir = B, T, 'f32', str(tmp_path)
    model = GazePredictionGRCN(Session(gpu), None, cfg)
    model.load_state_dict(checkpoint.import_model_variables('gaze_grcn', checkpoint.load_tf_export(str(tmp_path / 'export.npz')), T))
    probs = model.predict(syn.c3d_features(seed + 1, B, T)).cpu().numpy()
    assert rel(model.predicted_gazemaps_logit.cpu().numpy(), g['logits']) < 2e-5
    assert rel(probs, g['probs']) < 2e-5


def test_fcgru_import_reproduces_golden_logits(gpu, tmp_path):
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_rnn import GazePredictionGRU, GRUModelConfig
    g = np.load(os.path.join(GOLD, 'fcgru_small.npz'))
    B, T, GH, seed = [int(v) for v in g['config']]
    want = syn.fcgru_params(seed, GH, GH)
    tf_vars = checkpoint.export_model_variables('gaze_rnn', want)
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir, cfg.init_seed = B, T, 'f32', str(tmp_path), 99
    model = GazePredictionGRU(Session(gpu), None, cfg, gazemap_height=GH, gazemap_width=GH)
    model.predict(syn.c3d_features(seed + 1, B, T))
    assert rel(model.predicted_gazemaps_logit.cpu().numpy(), g['logits']) > 1e-2       # its own init is another net
    model.load_state_dict(checkpoint.import_model_variables('gaze_rnn', tf_vars))
    model.predict(syn.c3d_features(seed + 1, B, T))
    assert rel(model.predicted_gazemaps_logit.cpu().numpy(), g['logits']) < 1e-4


def test_cascade_import_and_pretrained_shallownet(gpu, tmp_path):
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_grcn_cascade import GazePredictionGRCN, GRUModelConfig
    g = np.load(os.path.join(GOLD, 'cascade_small.npz'))
    B, T, seed = [int(v) for v in g['config']]
    p = syn.cascade_params(seed)
    flat = {k: v for k, v in p.items() if k != 'ShallowNet'}
    flat.update({'ShallowNet/' + k: v for k, v in p['ShallowNet'].items()})
    tf_vars = checkpoint.export_model_variables('gaze_grcn_cascade', flat)
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir, cfg.init_seed = B, T, 'f32', str(tmp_path), 77
    model = GazePredictionGRCN(Session(gpu), None, cfg)
    rs = np.random.RandomState(seed + 7)
    frames = rs.rand(B, T, 98, 98, 3).astype(np.float32)
    c3d = syn.c3d_features(seed + 8, B, T)
    # (i) everything but the ShallowNet from the cascade export, the ShallowNet scrambled
    state = checkpoint.import_model_variables('gaze_grcn_cascade', tf_vars)
    state.update({k: v for k, v in _scramble(state).items() if k.startswith('ShallowNet/')})
    model.load_state_dict(state)
    assert rel(model.predict(c3d, frames).cpu().numpy(), g['maps']) > 1e-3
    # (ii) initialize_pretrained_shallownet (gaze_rnn.py:412-433) from a separate ShallowNet-only export
    np.savez(tmp_path / 'shallownet.npz', **{k + ':0': v for k, v in checkpoint.export_shallownet_variables(p['ShallowNet']).items()})
    model.initialize_pretrained_shallownet(str(tmp_path / 'shallownet.npz'))
    assert rel(model.predict(c3d, frames).cpu().numpy(), g['maps']) < 2e-4


def test_framewise_shallownet_import(gpu, tmp_path):
    from oracle import torch_ref
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_framewise_shallownet import FramewiseShallowNet, GRUModelConfig
    sp = syn.shallownet_params(41)
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.compute_dtype, cfg.train_dir, cfg.init_seed = 2, 1, 'f32', str(tmp_path), 5
    model = FramewiseShallowNet(Session(gpu), None, cfg)
    np.savez(tmp_path / 's.npz', **checkpoint.export_shallownet_variables(sp))
    model.initialize_pretrained_shallownet(str(tmp_path / 's.npz'))
    frames = np.random.RandomState(42).rand(2, 1, 98, 98, 3).astype(np.float32)
    out = model.predict(np.zeros((2, 1, 1024, 7, 7), np.float32), frames).cpu().numpy()
    ref = torch_ref.shallownet_forward(torch.tensor(frames.reshape(2, 98, 98, 3), dtype=torch.float64),
                                       {k: torch.tensor(v, dtype=torch.float64) for k, v in sp.items()}).numpy()
    assert rel(out.reshape(2, 49, 49), ref) < 5e-5
