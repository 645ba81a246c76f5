"""GPU: HIP-graph replay of the head's training step (device-side step counter and lr schedule)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _setup(gpu, dtype, B=2, T=3):
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    p = syn.grcn_params(201, T, gru_std=0.05, random_bn=True)
    x = torch.tensor(syn.c3d_features(202, B, T), device=gpu)
    gt, _ = syn.gaze_maps(203, B, T)
    gt = torch.tensor((gt / gt.sum(axis=(2, 3), keepdims=True)).astype(np.float32), device=gpu)
    eng = GrcnEngine(B, T, dtype=dtype, save_for_backward=True, device=gpu)
    eng.set_weights(p)
    return eng, x, gt


def test_graphed_steps_equal_eager_steps(gpu):
    """Four steps with a schedule that decays every 2 steps: the replayed graphs (device-side counter, lr and
    bias correction) land on the same parameters as the eager path with the host-side schedule."""
    from recurrent_gaze_prediction_amd.graph import GraphedHeadTrainStep
    lr0, decay, every = 1e-3, 0.5, 2
    eager, x, gt = _setup(gpu, 'f32')
    for k in range(4):
        logits, probs = eager.forward(x)
        eager.backward(logits, probs, gt)
        eager.adam_step(k, torch_ref.learning_rate(lr0, decay, k, every), max_grad_norm=10.0)
    graphed, x2, gt2 = _setup(gpu, 'f32')
    gs = GraphedHeadTrainStep(graphed, x2, gt2, lr0, decay, every, 10.0)
    assert gs.g1 is not None and gs.g2 is not None
    for _ in range(4):
        gs.step()
    assert gs.global_step == 4
    # out_b (the last element) is excluded: its gradient is exactly 0 in theory (softmax is shift invariant), so
    # Adam steps on round-off noise there and the two runs differ by O(lr)
    a, b = eager.flat_params.cpu().numpy()[:-1], graphed.flat_params.cpu().numpy()[:-1]
    assert np.abs(a - b).max() < 2e-5 * np.abs(a).max(), float(np.abs(a - b).max())
    from recurrent_gaze_prediction_amd.engine import GRCN_PARAM_TO_FIELD
    p0 = syn.grcn_params(201, 3, gru_std=0.05, random_bn=True)
    init = np.concatenate([np.asarray(p0[k], np.float32).ravel() for k in GRCN_PARAM_TO_FIELD])[:-1]
    assert np.abs(a - init).max() > 1e-4


def test_graph_replay_sees_new_inputs(gpu):
    """The graph reads the static buffers at replay time: new data copied into them changes the outcome."""
    from recurrent_gaze_prediction_amd.graph import GraphedHeadTrainStep
    eng, x, gt = _setup(gpu, 'bf16')
    gs = GraphedHeadTrainStep(eng, x, gt, 1e-4)
    gs.step()
    l1 = gs.logits.clone()
    x.copy_(torch.tensor(syn.c3d_features(999, 2, 3), device=gpu))
    gs.step()
    assert float((gs.logits - l1).abs().max()) > 1e-3 and bool(torch.isfinite(gs.logits).all())


def test_wait_grads_refuses_stale_events_after_a_captured_backward(gpu):
    """a backward captured into a graph records no gradient events; the ones of an earlier eager backward say
    nothing about a replay, so rgp_grcn_wait_grads must answer RGP_ESTATE until the next eager backward."""
    from recurrent_gaze_prediction_amd import _lib
    from recurrent_gaze_prediction_amd.graph import GraphedHeadTrainStep
    eng, x, gt = _setup(gpu, 'bf16')
    z, p = eng.forward(x)
    eng.backward(z, p, gt)
    side = torch.cuda.Stream(gpu)
    for _, ready in eng.grad_buckets():
        ready(side)                                   # eager backward: events exist
    gs = GraphedHeadTrainStep(eng, x, gt, 1e-4)       # warm-up (eager, on a side stream) + capture
    gs.step()
    with pytest.raises(_lib.RgpError) as ei:
        eng.grad_buckets()[0][1](side)
    assert ei.value.code == -4                        # RGP_ESTATE
    z, p = eng.forward(x)
    eng.backward(z, p, gt)
    eng.grad_buckets()[0][1](side)                    # valid again
    torch.cuda.synchronize()
