"""GPU parity of the backward pass + optimizer against the float64 autograd oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import grcn, torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# per-tensor relative Frobenius error ||g - g_ref|| / ||g_ref||
TOL = {'f32': 2e-4, 'bf16': 3e-2}


def fro_err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30)


def case(seed, B, T, P, S):
    p = syn.grcn_params(seed, T, P, S, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(seed + 1, B, T)
    gt, _ = syn.gaze_maps(seed + 2, B, T)
    return p, x, grcn.normalize_probability_map(gt).astype(np.float32)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('B,T,P,S', [(2, 3, 64, 64), (1, 2, 512, 128)])
@pytest.mark.parametrize('loss_type', ['xentropy', 'l2'])
@pytest.mark.parametrize('head', ['folded', 'three-stage'])
def test_gradients_match_autograd(gpu, dtype, B, T, P, S, loss_type, head):
    """Both implementations of the saliency head's backward -- the chain rule through the folded 19x19 filter
    (csrc/head_fold.hip.h, the default) and the three transposed convolutions one by one (RGP_GRCN_UNFOLDED_HEAD) -- against
    float64 autograd of the reference op sequence."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    p, x, g = case(101, B, T, P, S)
    _, _, ref = torch_ref.grcn_loss_and_grads(x, g, p, loss_type=loss_type)
    eng = GrcnEngine(B, T, P, S, dtype=dtype, save_for_backward=True, device=gpu, unfolded_head=head == 'three-stage')
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    grads = eng.backward(logits, probs, torch.tensor(g, device=gpu), loss_type)
    torch.cuda.synchronize()
    for k in ref:
        if k == 'out_b' and loss_type == 'xentropy':
            # d loss / d out_b = sum_j (p_j sum(g) - g_j) / (BT) = 0 exactly: only round-off remains
            assert abs(grads[k].item()) < 1e-6 and abs(ref[k].item()) < 1e-12
            continue
        e = fro_err(grads[k].cpu().numpy(), ref[k].numpy())
        assert e < TOL[dtype], '%s: rel err %.3e' % (k, e)


@pytest.mark.parametrize('head', ['folded', 'three-stage'])
@pytest.mark.parametrize('dtype,tol', [('f32', 1e-3), ('bf16', 3e-2)])
def test_bptt_through_35_steps_matches_autograd(gpu, dtype, tol, head):
    """BASELINE config 4's clip length (the reference's older default, model_gru_rcn.py:188): B=2, T=35 at the
    reference widths against float64 autograd -- the error growth of bf16 operands through a 35-step backward."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T, P, S = 2, 35, 512, 128
    p, x, g = case(211, B, T, P, S)
    _, _, ref = torch_ref.grcn_loss_and_grads(x, g, p, loss_type='xentropy')
    eng = GrcnEngine(B, T, P, S, dtype=dtype, save_for_backward=True, device=gpu, unfolded_head=head == 'three-stage')
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    grads = eng.backward(logits, probs, torch.tensor(g, device=gpu), 'xentropy')
    errs = {k: fro_err(grads[k].cpu().numpy(), ref[k].numpy()) for k in ref if k != 'out_b'}
    assert max(errs.values()) < tol, errs
    # the recurrent filters see all 35 steps: they must carry real signal, not round-off
    assert float(np.linalg.norm(ref['GRU_Conv_U'].numpy())) > 1e-6


@pytest.mark.parametrize('B,T', [(3, 4), (33, 3), (8, 35)])
def test_persistent_bptt_group_shapes(gpu, B, T):
    """The persistent BPTT kernel with a group count that is not a multiple of 8 (B = 3), with two clips per group
    and a ragged last group (B = 33), and at BASELINE config 4's per-GPU shape (8 clips x T = 35: 64 workgroups through 35
    steps), against float64 autograd."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    P, S = 512, 128
    p, x, g = case(300 + B, B, T, P, S)
    _, _, ref = torch_ref.grcn_loss_and_grads(x, g, p, loss_type='xentropy')
    eng = GrcnEngine(B, T, P, S, dtype='bf16', save_for_backward=True, device=gpu)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    grads = eng.backward(logits, probs, torch.tensor(g, device=gpu), 'xentropy')
    errs = {k: fro_err(grads[k].cpu().numpy(), ref[k].numpy()) for k in ref if k != 'out_b'}
    assert max(errs.values()) < TOL['bf16'], errs


def test_gradients_match_golden_fixture(gpu):
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    gold = np.load(os.path.join(GOLD, 'grcn_grads_small.npz'))
    B, T, P, S, seed = [int(v) for v in gold['config']]
    p, x, g = case(seed, B, T, P, S)
    eng = GrcnEngine(B, T, P, S, dtype='f32', save_for_backward=True, device=gpu)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    grads = eng.backward(logits, probs, torch.tensor(g, device=gpu))
    for k in grads:
        n = float(grads[k].double().norm().item())
        if k == 'out_b':
            assert n < 1e-6
            continue
        assert abs(n - float(gold['gnorm_' + k])) < 2e-4 * max(float(gold['gnorm_' + k]), 1e-12), k
        if 'grad_' + k in gold:
            assert fro_err(grads[k].cpu().numpy(), gold['grad_' + k]) < 2e-4, k
    # one clipped TF-Adam step against the fixture (lr 1e-4, clip 10, step 0)
    norm = eng.adam_step(0, 1e-4, max_grad_norm=10.0)
    assert abs(norm.item() - float(gold['global_norm'])) < 2e-4 * float(gold['global_norm'])
    for k in grads:
        # out_b is skipped: its gradient is pure round-off (exactly 0 in theory), and Adam's
        # m/(sqrt(v)+eps) turns round-off of either sign into an O(lr) step
        if 'adam1_' + k in gold and k != 'out_b':
            assert np.abs(eng.weights[k].cpu().numpy() - gold['adam1_' + k]).max() < 2e-6, k


def test_clip_scales_when_norm_exceeds_threshold(gpu):
    """clip_by_global_norm: with a tiny threshold the update equals Adam on g*clip/norm; Adam's first
    step is sign(g)*lr_t*|g|/(|g|+eps) so a clipped and an unclipped first step differ only through eps."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    p, x, g = case(7, 1, 2, 64, 64)
    a = GrcnEngine(1, 2, 64, 64, dtype='f32', save_for_backward=True, device=gpu)
    a.set_weights(p)
    lg, pr = a.forward(torch.tensor(x, device=gpu))
    a.backward(lg, pr, torch.tensor(g, device=gpu))
    gflat = a.flat_grads.clone()
    w0 = a.flat_params.clone()
    norm = a.adam_step(0, 1e-3, max_grad_norm=1e-3).item()
    assert abs(norm - gflat.double().norm().item()) < 1e-5 * norm
    scale = 1e-3 / max(norm, 1e-3)
    gi = gflat.double() * scale
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    expect = w0.double() - lr_t * (0.1 * gi) / (torch.sqrt(0.001 * gi * gi) + 1e-8)
    assert (a.flat_params.double() - expect).abs().max().item() < 1e-7


def test_training_reduces_loss(gpu):
    """A few TF-Adam steps on one synthetic batch: the reference loss goes down (bf16 path)."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine, softmax_xent
    B, T = 4, 3
    p, x, g = case(55, B, T, 512, 128)
    eng = GrcnEngine(B, T, dtype='bf16', save_for_backward=True, device=gpu)
    eng.set_weights(p)
    xd, gd = torch.tensor(x, device=gpu), torch.tensor(g, device=gpu)
    losses = []
    for step in range(8):
        logits, probs = eng.forward(xd)
        losses.append(softmax_xent(logits, gd, want_probs=False)[2].item())
        eng.backward(logits, probs, gd)
        eng.adam_step(step, 1e-3)
    assert losses[-1] < losses[0] - 1e-3, losses
