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


class _Site(object):
    def __init__(self, seed):
        self.seed, self.draws, self.keep_prob = seed, 0, 0.5


def _stub_model(tmp_path, n_sites=1):
    """ModelBase with host-only state: exercises the checkpoint logic without a device engine."""
    from recurrent_gaze_prediction_amd.models.base import ModelBase
    from recurrent_gaze_prediction_amd.models.gaze_rnn import GRUModelConfig, dropout_seed

    class Stub(ModelBase):
        def __init__(self, config):
            self.config = config
            ModelBase.__init__(self, config)
            self.vars = {'w': np.arange(4, dtype=np.float32)}
            self.sites = [_Site(dropout_seed(config, 17 + i)) for i in range(n_sites)]

        def state_dict(self):
            return dict(self.vars)

        def load_state_dict(self, state):
            self.vars = dict(state)

        def _dropout_sites(self):
            return self.sites

        def _optimizer_engines(self):
            return []

    c = GRUModelConfig()
    c.train_dir = str(tmp_path)
    return Stub(c)


def test_checkpoint_resume_keeps_dropout_streams_distinct_per_rank(tmp_path, monkeypatch):
    """every rank loads the ONE saved file; the Philox key must move to the loading rank's stream
    (models/gaze_rnn.py dropout_seed mixes the rank in) while the draw counter continues."""
    import pytest
    monkeypatch.setenv('RANK', '0')
    m0 = _stub_model(tmp_path / 'a')
    m0.sites[0].draws = 5
    path = m0.save_model_checkpoint(m0.train_dir)
    keys = {}
    for rank in (0, 1, 3):
        monkeypatch.setenv('RANK', str(rank))
        m = _stub_model(tmp_path / ('r%d' % rank))
        fresh = m.sites[0].seed                              # what configure() derives on this rank
        m.sites[0].seed = 0
        m.load_model_from_checkpoint_file(path)
        assert m.sites[0].seed == fresh and m.sites[0].draws == 5
        keys[rank] = m.sites[0].seed
    assert len(set(keys.values())) == 3
    monkeypatch.setenv('RANK', '0')
    two = _stub_model(tmp_path / 'two', n_sites=2)
    with pytest.raises(ValueError, match='dropout site'):
        two.load_model_from_checkpoint_file(path)


def test_checkpoint_without_dropout_state_still_loads(tmp_path, monkeypatch):
    """a checkpoint written before the 'dropout' key existed loads into a model WITH dropout sites -- the
    sites keep this rank's freshly derived key and start at draw 0; only a PRESENT key of the wrong length is an error."""
    import torch
    monkeypatch.setenv('RANK', '0')
    m0 = _stub_model(tmp_path / 'a')
    path = m0.save_model_checkpoint(m0.train_dir)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    del ck['dropout']
    old = str(tmp_path / 'old.pt')
    torch.save(ck, old)
    m = _stub_model(tmp_path / 'b')
    fresh = m.sites[0].seed
    m.sites[0].draws = 0
    m.load_model_from_checkpoint_file(old)
    assert m.sites[0].seed == fresh and m.sites[0].draws == 0
    assert np.array_equal(m.vars['w'], np.arange(4, dtype=np.float32))
