"""GPU parity: the HIP gaze_grcn path (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu

# tolerances: max-abs error relative to the max-abs of the oracle tensor.
# f32 path = exact fp32 FMA chains (v_mfma_f32_16x16x4_f32) vs a float64 oracle;
# bf16 path = bf16 MFMA operands with fp32 accumulate/state (<= 2e-2 on logits).
TOL = {'f32': 2e-5, 'bf16': 2e-2}
# hidden states at gru_std=0.05 sit in the saturated part of tanh/sigmoid (pre-activation
# std ~5), where one bf16 ulp of the operands moves single elements by a few 1e-2: bound
# the worst element looser and the RMS tighter.
TOL_H_MAX = {'f32': 5e-5, 'bf16': 6e-2}
TOL_H_RMS = {'f32': 1e-5, 'bf16': 1e-2}


def rel_err(a, ref):
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def oracle_forward(x, p):
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    logits, hs, emb = torch_ref.grcn_forward(torch.tensor(x, dtype=torch.float64), pt, want_hidden=True)
    return logits.numpy(), hs.numpy(), emb.numpy()


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('B,T,P,S', [(2, 3, 512, 128), (3, 2, 64, 64)])
def test_grcn_forward_matches_oracle(gpu, dtype, B, T, P, S):
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    p = syn.grcn_params(11, T, P, S, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(12, B, T)
    ref_logits, ref_h, ref_emb = oracle_forward(x, p)
    eng = GrcnEngine(B, T, P, S, dtype=dtype, device=gpu)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    torch.cuda.synchronize()
    tol = TOL[dtype]
    emb = eng.read_buffer('c3d_embedded').cpu().numpy().reshape(ref_emb.shape)
    assert rel_err(emb, ref_emb) < tol, 'projection'
    h = eng.read_buffer('rcn_outputs').cpu().numpy().reshape(ref_h.shape)
    assert rel_err(h, ref_h) < TOL_H_MAX[dtype], 'ConvGRU states (max)'
    assert np.sqrt(((h - ref_h) ** 2).mean()) / np.sqrt((ref_h ** 2).mean()) < TOL_H_RMS[dtype], 'ConvGRU states (rms)'
    assert rel_err(logits.cpu().numpy(), ref_logits) < tol, 'logits'
    ref_probs = torch_ref.softmax_maps(torch.tensor(ref_logits)).numpy()
    assert rel_err(probs.cpu().numpy(), ref_probs) < tol, 'softmax maps'
    assert np.allclose(probs.cpu().numpy().reshape(B, T, -1).sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_long_recurrence_reference_default_T42(gpu, dtype):
    """T = 42 is the reference's default n_lstm_steps (gaze_rnn.py:50): rounding compounds over the
    recurrence, so the last frames are the hard case for bf16 operands."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T = 2, 42
    p = syn.grcn_params(21, T, gru_std=0.02, random_bn=True)
    x = syn.c3d_features(22, B, T)
    ref_logits, ref_h, _ = oracle_forward(x, p)
    eng = GrcnEngine(B, T, dtype=dtype, device=gpu)
    eng.set_weights(p)
    logits, _ = eng.forward(torch.tensor(x, device=gpu))
    got = logits.cpu().numpy()
    assert rel_err(got, ref_logits) < TOL[dtype]
    assert rel_err(got[:, -1], ref_logits[:, -1]) < TOL[dtype], 'last timestep'
    h = eng.read_buffer('rcn_outputs').cpu().numpy().reshape(ref_h.shape)
    assert np.sqrt(((h[:, -1] - ref_h[:, -1]) ** 2).mean()) / np.sqrt((ref_h[:, -1] ** 2).mean()) < TOL_H_RMS[dtype]


def test_single_clip_single_step_edge(gpu):
    """B = 1, T = 1: one M-tile with 79 clamped rows, one timestep (h_0 = 0 path only)."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    p = syn.grcn_params(23, 1, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(24, 1, 1)
    ref_logits, _, _ = oracle_forward(x, p)
    eng = GrcnEngine(1, 1, dtype='f32', device=gpu)
    eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device=gpu))
    assert rel_err(logits.cpu().numpy(), ref_logits) < TOL['f32']
    assert abs(probs.sum().item() - 1.0) < 1e-5


@pytest.mark.parametrize('B,T', [(3, 5), (33, 4), (64, 3), (1, 2)])
def test_persistent_sequence_kernel_group_shapes(gpu, B, T):
    """The persistent ConvGRU kernel (bf16, convgru_seq.hip.h) deals clips to groups of 8 workgroups: one clip per group
    with a group count that is not a multiple of 8 (B = 3, plain group numbering), two clips per group with a ragged
    last group (B = 33: 17 groups, the last one holds one clip), the full chip (B = 64: 32 groups) and a single group.
    States of every step and logits against the float64 oracle, plus the batch-normalised head input."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    p = syn.grcn_params(31 + B, T, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(32 + B, B, T)
    ref_logits, ref_h, _ = oracle_forward(x, p)
    eng = GrcnEngine(B, T, dtype='bf16', device=gpu)
    eng.set_weights(p)
    logits, _ = eng.forward(torch.tensor(x, device=gpu))
    h = eng.read_buffer('rcn_outputs').cpu().numpy().reshape(ref_h.shape)
    assert np.isfinite(h).all()
    assert rel_err(h, ref_h) < TOL_H_MAX['bf16']
    for t in range(T):          # every step, every clip (a clip dealt to the wrong group would be O(1) off)
        for b in (0, B // 2, B - 1):
            assert rel_err(h[b, t], ref_h[b, t]) < TOL_H_MAX['bf16'], (b, t)
    assert rel_err(logits.cpu().numpy(), ref_logits) < TOL['bf16']
    # a second call on the same plan gives the same result (phase counters are re-zeroed per launch)
    logits2, _ = eng.forward(torch.tensor(x, device=gpu))
    assert torch.equal(logits, logits2)


@pytest.mark.parametrize('B,T', [(2, 6), (33, 3)])
def test_per_step_recurrence_agrees_with_persistent_kernels(gpu, B, T):
    """RGP_GRCN_PER_STEP: the same bf16 plan with the recurrence (and its BPTT) as per-timestep launches -- the path
    f32 plans and other cell widths always take -- against the oracle AND against the persistent kernels on the same
    inputs: states, logits and every gradient (they differ only by the K-split summation order feeding bf16 rounding)."""
    from oracle import grcn
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    p = syn.grcn_params(71 + B, T, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(72 + B, B, T)
    gt, _ = syn.gaze_maps(73 + B, B, T)
    g = torch.tensor(grcn.normalize_probability_map(gt).astype(np.float32), device=gpu)
    ref_logits, ref_h, _ = oracle_forward(x, p)
    out = {}
    for per_step in (False, True):
        eng = GrcnEngine(B, T, dtype='bf16', device=gpu, save_for_backward=True, per_step=per_step)
        eng.set_weights(p)
        logits, probs = eng.forward(torch.tensor(x, device=gpu))
        grads = {k: v.cpu().numpy().copy() for k, v in eng.backward(logits, probs, g).items()}
        h = eng.read_buffer('rcn_outputs').cpu().numpy().reshape(ref_h.shape)
        assert rel_err(h, ref_h) < TOL_H_MAX['bf16'] and rel_err(logits.cpu().numpy(), ref_logits) < TOL['bf16'], per_step
        out[per_step] = (h, logits.cpu().numpy(), grads)
        if per_step:
            from recurrent_gaze_prediction_amd._lib import RgpError
            with pytest.raises(RgpError):
                eng.inject_fault('seq')                    # no persistent kernel on this plan
    (h0, l0, g0), (h1, l1, g1) = out[False], out[True]
    assert rel_err(h0, h1) < 2e-2 and rel_err(l0, l1) < 1e-2
    for k in g0:
        if k == 'out_b':
            continue
        e = np.linalg.norm(g0[k] - g1[k]) / max(np.linalg.norm(g1[k]), 1e-30)
        assert e < 2e-2, (k, e)


def test_lost_group_member_is_loud(gpu):
    """A persistent ConvGRU launch whose group never completes (fault injection: one of the 8 workgroups of group 0
    leaves at once -- what a second resident launch on the device would cause) must not return finite maps: after the
    ~1 s time-out every logit / map of the group's clips is NaN, the plan reports RGP_ETIMEOUT (status() and the next
    call), the other groups' clips are intact, and the plan works again afterwards.  Same for the BPTT kernel."""
    from oracle import grcn
    from recurrent_gaze_prediction_amd import _lib
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T = 3, 4                                            # one clip per group: group 0 = clip 0
    p = syn.grcn_params(81, T, gru_std=0.05, random_bn=True)
    x = torch.tensor(syn.c3d_features(82, B, T), device=gpu)
    gt, _ = syn.gaze_maps(83, B, T)
    g = torch.tensor(grcn.normalize_probability_map(gt).astype(np.float32), device=gpu)
    eng = GrcnEngine(B, T, dtype='bf16', device=gpu, save_for_backward=True)
    eng.set_weights(p)
    good_logits, good_probs = [t.clone() for t in eng.forward(x)]
    eng.status()                                           # clean
    eng.inject_fault('seq')
    logits, probs = eng.forward(x)
    with pytest.raises(_lib.RgpError, match='lost a group member'):
        eng.status()
    assert torch.isnan(logits[0]).all() and torch.isnan(probs[0]).all()
    assert torch.equal(logits[1:], good_logits[1:])        # the other groups never noticed
    eng.status()                                           # reported once, then clear
    # the error also surfaces on the next call of a caller that never asks
    eng.inject_fault('seq')
    eng.forward(x)
    torch.cuda.synchronize()
    with pytest.raises(_lib.RgpError, match='lost a group member'):
        eng.forward(x)
    logits2, _ = eng.forward(x)
    assert torch.equal(logits2, good_logits)               # the plan is usable again, bit for bit
    # BPTT
    good = {k: v.clone() for k, v in eng.backward(good_logits, good_probs, g).items()}
    eng.forward(x)
    eng.inject_fault('bptt')
    grads = eng.backward(good_logits, good_probs, g)
    with pytest.raises(_lib.RgpError, match='lost a group member'):
        eng.status()
    assert not torch.isfinite(grads['GRU_Conv_Wz']).all() and not torch.isfinite(grads['proj_c3d_W']).all()
    eng.forward(x)
    again = eng.backward(good_logits, good_probs, g)
    for k in good:
        assert torch.allclose(again[k], good[k], rtol=1e-4, atol=1e-6 * float(good[k].abs().max())), k


def test_persistent_launches_from_two_streams_are_serialised(gpu):
    """Two plans driven from two streams of one process: the library orders their persistent launches (one in flight
    per device), so neither times out and both give their single-stream results."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T = 64, 16                                          # 32 groups: a launch fills all 256 CUs
    p = syn.grcn_params(91, T, gru_std=0.05, random_bn=True)
    xs = [torch.tensor(syn.c3d_features(92 + i, B, T), device=gpu) for i in range(2)]
    engs = [GrcnEngine(B, T, dtype='bf16', device=gpu) for _ in range(2)]
    for e in engs:
        e.set_weights(p)
    ref = [e.forward(x)[0].clone() for e, x in zip(engs, xs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu) for _ in range(2)]
    outs = [None, None]
    for rep in range(3):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs[i] = engs[i].forward(xs[i])[0]
    torch.cuda.synchronize()
    for i in range(2):
        engs[i].status()
        assert torch.equal(outs[i], ref[i]), i


def test_persistent_launches_from_two_host_threads_are_serialised(gpu):
    """ctypes releases the GIL inside a library call, so two Python threads (each on its own stream and
    plan) can be inside rgp_grcn_forward together.  The wait for the previous persistent launch, the launch and its
    record are one critical section (csrc/rgp_host.h PersistentLaunch): neither plan times out, both give their
    single-thread results."""
    import threading
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T = 64, 16
    p = syn.grcn_params(93, T, gru_std=0.05, random_bn=True)
    xs = [torch.tensor(syn.c3d_features(94 + i, B, T), device=gpu) for i in range(2)]
    engs = [GrcnEngine(B, T, dtype='bf16', device=gpu) for _ in range(2)]
    for e in engs:
        e.set_weights(p)
    ref = [e.forward(x)[0].clone() for e, x in zip(engs, xs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu) for _ in range(2)]
    outs, errs = [None, None], []
    gate = threading.Barrier(2)

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                gate.wait()
                for _ in range(20):
                    outs[i] = engs[i].forward(xs[i])[0]
                streams[i].synchronize()
        except Exception as exc:                              # surfaced below: a thread's exception is otherwise lost
            errs.append(exc)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(2):
        engs[i].status()
        assert torch.equal(outs[i], ref[i]), i


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_folded_head_equals_the_three_stage_head_and_the_oracle(gpu, dtype):
    """Inference plans run deconv1 . deconv2 . deconv3 . out_W (gaze_grcn.py:326-361: no bias, no non-linearity between
    them) as ONE GEMM with a filter folded when the weights are set (csrc/head_fold.hip.h).  Checked against (i) the
    float64 oracle of the reference op sequence, (ii) the library's three-stage head (RGP_GRCN_UNFOLDED_HEAD) on the same
    plan inputs, at the reference widths, (iii) again after the weights change (the fold follows set_weights), and
    (iv) at the benchmark's 1024 frames, where the GEMM picks another tile."""
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T = 3, 2
    x = syn.c3d_features(202, B, T)
    xd = torch.tensor(x, device=gpu)
    folded = GrcnEngine(B, T, dtype=dtype, device=gpu)
    staged = GrcnEngine(B, T, dtype=dtype, device=gpu, unfolded_head=True)
    assert folded.read_buffer_elems('d1') == 0 and staged.read_buffer_elems('d1') == B * T * 529 * 64
    for seed in (201, 211):
        p = syn.grcn_params(seed, T, gru_std=0.05, random_bn=True)
        ref = torch_ref.grcn_forward(torch.tensor(x, dtype=torch.float64),
                                     {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}).numpy()
        folded.set_weights(p)
        staged.set_weights(p)
        lf, pf = folded.forward(xd)
        ls, ps = staged.forward(xd)
        ef, es = rel_err(lf.cpu().numpy(), ref), rel_err(ls.cpu().numpy(), ref)
        if dtype == 'f32':
            assert ef < 2e-5 and es < 2e-5, (ef, es)
            assert rel_err(lf.cpu().numpy(), ls.cpu().numpy()) < 2e-5
        else:
            # one bf16 rounding of the folded filter instead of two rounded intermediate maps: no less accurate
            assert ef < 2e-2 and ef < 1.25 * es + 1e-3, (ef, es)
        assert torch.allclose(pf.sum((-1, -2)), torch.ones(B, T, device=gpu), atol=1e-4)
    # an odd frame count (35 frames = 1715 GEMM rows: partial row tiles) ...
    odd = GrcnEngine(5, 7, dtype=dtype, device=gpu)
    odd_s = GrcnEngine(5, 7, dtype=dtype, device=gpu, unfolded_head=True)
    p = syn.grcn_params(231, 7, gru_std=0.05, random_bn=True)
    odd.set_weights(p)
    odd_s.set_weights(p)
    xo = torch.tensor(syn.c3d_features(232, 5, 7), device=gpu)
    e = rel_err(odd.forward(xo)[0].cpu().numpy(), odd_s.forward(xo)[0].cpu().numpy())
    assert e < (2e-5 if dtype == 'f32' else 2e-2), e
    # ... and the benchmark's 1024 frames (other GEMM tiles; the fold's clipping at the map edge is the same code for every frame)
    big = GrcnEngine(64, 16, dtype=dtype, device=gpu)
    big_s = GrcnEngine(64, 16, dtype=dtype, device=gpu, unfolded_head=True)
    p = syn.grcn_params(221, 16, gru_std=0.05, random_bn=True)
    big.set_weights(p)
    big_s.set_weights(p)
    xb = torch.tensor(syn.c3d_features(222, 64, 16), device=gpu)
    lb, _ = big.forward(xb)
    lbs, _ = big_s.forward(xb)
    big.status()
    big_s.status()
    e = rel_err(lb.cpu().numpy(), lbs.cpu().numpy())
    assert e < (2e-5 if dtype == 'f32' else 2e-2), e
