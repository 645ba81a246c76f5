"""CPU: host logic of the reference-named API (config round trip, label normalisation, dataset contract)."""
import io

import numpy as np

from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.models import model_util
from recurrent_gaze_prediction_amd.models.base import BaseModelConfig


def test_config_json_roundtrip_like_reference_self_test():
    """base.py:408-424 (_self_test): dump to a stream, load back, same fields."""
    c = BaseModelConfig()
    c.train_tag = 'x'
    s = io.StringIO()
    c.dump(s)
    c2 = BaseModelConfig.load(io.StringIO(s.getvalue()))
    assert vars(c2) == vars(c) and c2.max_grad_norm == 10.0 and c2.learning_rate_decay == 0.8
    assert repr(c).startswith('ModelConfig{')


def test_gru_model_config_defaults():
    from recurrent_gaze_prediction_amd.models.gaze_rnn import GRUModelConfig
    c = GRUModelConfig()
    assert (c.n_lstm_steps, c.batch_size, c.loss_type, c.use_flip_batch) == (42, 7, 'xentropy', True)   # gaze_rnn.py:50-61


def test_normalize_probability_map_and_map():
    t = np.random.RandomState(0).rand(2, 3, 49, 49).astype(np.float32)
    p = model_util.normalize_probability_map(t)
    assert np.allclose(p.reshape(6, -1).sum(-1), 1.0, atol=1e-5) and p is not t
    m = model_util.normalize_map(t[0])
    assert np.allclose(m.reshape(3, -1).min(-1), 0) and np.allclose(m.reshape(3, -1).max(-1), 1)
    z = np.zeros((1, 49, 49), np.float32)
    with np.errstate(invalid='ignore', divide='ignore'):
        assert np.isnan(model_util.normalize_probability_map(z)).all()        # no epsilon (9-Q8)


def test_synthetic_dataset_contract():
    ds = syn.SyntheticDataSet(5, 4, seed=1)
    assert len(ds) == 5
    images, maps, fix, c3d, pupils, names = ds.next_batch(2)
    assert images.shape == (2, 4, 98, 98, 3) and images.dtype == np.float32
    assert maps.shape == (2, 4, 49, 49) and fix.shape == (2, 4, 49, 49) and c3d.shape == (2, 4, 512, 2, 7, 7)
    assert pupils.shape == (2, 4) and len(names) == 2 and (maps.reshape(8, -1).sum(-1) > 0).all()
    assert (c3d >= 0).all() and 0.3 < (c3d == 0).mean() < 0.7             # post-ReLU features
