"""GPU: the reference-named model classes drive the HIP path (generate / evaluate / checkpoints)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def make_model(gpu, tmp_path, T=3, B=2, loss_type='xentropy', dtype='f32'):
    from recurrent_gaze_prediction_amd.models.base import Session
    from recurrent_gaze_prediction_amd.models.gaze_grcn import CONSTANTS, GazePredictionGRCN, GRUModelConfig
    assert CONSTANTS.gazemap_height == 49
    cfg = GRUModelConfig()
    cfg.batch_size, cfg.n_lstm_steps, cfg.loss_type, cfg.compute_dtype = B, T, loss_type, dtype
    cfg.train_dir = str(tmp_path)
    ds = type('DS', (), {})()
    ds.train = ds.valid = syn.SyntheticDataSet(12, T, seed=5)
    return GazePredictionGRCN(Session(gpu), ds, cfg), ds


def test_generate_evaluate_and_checkpoint_roundtrip(gpu, tmp_path):
    model, ds = make_model(gpu, tmp_path)
    w = syn.grcn_params(91, 3, gru_std=0.05, random_bn=True)
    model.load_state_dict(w)
    ret = model.generate(ds.valid, max_instances=12)
    assert set(ret) == {'pred_gazemap_list', 'gt_gazemap_list', 'images_list', 'fixationmap_list',
                        'clipname_list', 'c3d_list'}
    n = 12 * 3
    assert ret['pred_gazemap_list'].shape == (n, 49, 49) and ret['c3d_list'].shape == (n, 1024, 7, 7)
    assert np.allclose(ret['pred_gazemap_list'].reshape(n, -1).sum(-1), 1.0, atol=1e-5)   # xentropy -> softmax maps
    # oracle on the first batch
    ds2 = syn.SyntheticDataSet(12, 3, seed=5)
    _, maps, _, c3d, _, _ = ds2.next_batch(2)
    x = torch.tensor(c3d.reshape(2, 3, 1024, 7, 7))
    ref = torch_ref.softmax_maps(torch_ref.grcn_forward(x, {k: torch.tensor(v) for k, v in w.items()})).numpy()
    assert np.abs(ret['pred_gazemap_list'][:6] - ref.reshape(6, 49, 49)).max() < 2e-5 * ref.max()
    np.random.seed(0)
    _, scores = model.generate_and_evaluate(ds.valid, max_instances=12)          # no TypeError (9-Q6)
    assert set(scores) == {'sim', 'cc', 'AUC_shuffled', 'AUC_Borji'} and all(np.isfinite(list(scores.values())))
    # validation step computes the reference's loss
    step = model.single_step(train_mode=False, dataset=syn.SyntheticDataSet(12, 3, seed=5))
    from oracle import grcn
    logits = torch_ref.grcn_forward(x, {k: torch.tensor(v) for k, v in w.items()}).numpy()
    assert step == 0 and abs(model.loss - grcn.loss(logits, grcn.normalize_probability_map(maps))) < 1e-4
    # checkpoint round trip
    path = model.save_model_checkpoint(model.train_dir)
    model2, _ = make_model(gpu, tmp_path / 'b')
    model2.load_model_from_checkpoint_file(path)
    a = model.predict(c3d).cpu().numpy()
    b = model2.predict(c3d).cpu().numpy()
    assert np.array_equal(a, b)


def test_training_steps_through_the_model_api(gpu, tmp_path):
    """single_step(train_mode=True): flip augmentation, backward, clipped TF-Adam, lr schedule, global_step;
    the loss on a fixed validation batch goes down and checkpoints carry the updated variables."""
    model, ds = make_model(gpu, tmp_path, T=3, B=4, dtype='bf16')
    model.load_state_dict(syn.grcn_params(95, 3, gru_std=0.05, random_bn=True))
    model.config.initial_learning_rate = model.initial_learning_rate = 1e-3
    val = syn.SyntheticDataSet(12, 3, seed=5)
    model.single_step(train_mode=False, dataset=val)
    loss0 = model.loss
    np.random.seed(3)
    for i in range(6):
        assert model.single_step(train_mode=True) == i + 1
    assert model.current_step == 6 and float(model.grad_norm.item()) > 0
    val = syn.SyntheticDataSet(12, 3, seed=5)
    model.single_step(train_mode=False, dataset=val)
    assert model.loss < loss0
    sd = model.state_dict()
    assert not np.array_equal(sd['out_W'], syn.grcn_params(95, 3, gru_std=0.05, random_bn=True)['out_W'])


def test_l2_loss_type_returns_raw_maps_and_lr_schedule(gpu, tmp_path):
    model, ds = make_model(gpu, tmp_path, loss_type='l2', dtype='bf16')
    model.load_state_dict(syn.grcn_params(92, 3, gru_std=0.05))
    _, _, _, c3d, _, _ = ds.valid.next_batch(2)
    out = model.predict(c3d).cpu().numpy()
    assert not np.allclose(out.reshape(6, -1).sum(-1), 1.0)            # raw logits, not softmax (9-Q5)
    assert model.learning_rate_at(0) == model.initial_learning_rate
    assert abs(model.learning_rate_at(1000) - model.initial_learning_rate * 0.8 ** 2) < 1e-12
    bad = syn.grcn_params(93, 5)
    with pytest.raises(AssertionError):
        model.load_state_dict(bad)                                      # T=5 BN layers into a T=3 model (9-Q1)
