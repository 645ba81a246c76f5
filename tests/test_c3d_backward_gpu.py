"""GPU parity: backward of the C3D conv stack (rgp_c3d_backward: wgrad with transposing LDS reads, dgrad,
arg-max unpooling) against torch's CPU gradient operators (oracle side).

The stack is piecewise linear (ReLU gates, max-pool routing).  A bf16 forward flips the gates of near-zero
activations, and even the fp32 path resolves an occasional near-tie of a pooling window differently from a
CPU run, so an end-to-end comparison with autograd measures those flips, not the kernels.  The parity
test therefore checks every operator LOCALLY on the operands the kernels really consumed (read back from
the device): filter and bias gradients, the input gradient through the ReLU gate, and the pooling
routing (one member per window, a maximal one, carrying the gated pooled gradient).  The end-to-end
autograd comparison is kept as an RMS-level check.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu
NAMES = ('conv1a', 'conv2a', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'conv5a', 'conv5b')
POOL = {name: pool for name, _, _, pool in torch_ref.C3D_LAYERS}
# local operator checks: max-abs error relative to the reference's max-abs
TOL_LOCAL = {'f32': 1e-4, 'bf16': 1.5e-2}
# end-to-end vs autograd on the fp32 CPU graph: RMS error relative to RMS (see module docstring)
TOL_E2E_RMS = {'f32': 3e-2, 'bf16': 0.5}


def rel(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


def rms_rel(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.sqrt(((np.asarray(a, np.float64) - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-30))


@pytest.fixture(scope='module')
def case():
    p = syn.c3d_params(31)
    rs = np.random.RandomState(32)
    video = (rs.rand(2, 16, 112, 112, 3).astype(np.float32) - 0.5) * 2
    g = rs.randn(2, 1024, 7, 7).astype(np.float32)
    return p, video, g


@pytest.fixture(scope='module')
def autograd(case):
    p, video, g = case
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    feat = torch_ref.c3d_forward(torch.tensor(video), tp)
    (feat * torch.tensor(g)).sum().backward()
    torch.set_num_threads(old)
    return feat.detach().numpy(), {k: v.grad.numpy() for k, v in tp.items()}


def ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize('dtype,kernels', [('f32', 'patch'), ('bf16', 'patch'), ('bf16', 'igemm')])
def test_c3d_backward_operators(gpu, case, autograd, dtype, kernels):
    """kernels = 'igemm': the second kernel family (RGP_C3D_KERNELS_IGEMM) -- wgrad_kernel and the implicit-GEMM input
    gradients on conv2a..conv4b instead of the patch kernels -- under the same operator-local checks."""
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    p, video, g = case
    feat_ref, grads_ref = autograd
    n = video.shape[0]
    tol = TOL_LOCAL[dtype]
    rnd = (lambda t: t.bfloat16().float()) if dtype == 'bf16' else (lambda t: t)     # operand rounding of the kernels
    eng = C3DEngine(n, dtype=dtype, device=gpu, save_for_backward=True, kernels=kernels)
    eng.set_weights(p)
    feat, _ = eng.forward(torch.tensor(video, device=gpu))
    assert rel(feat.cpu().numpy(), feat_ref) < (2e-5 if dtype == 'f32' else 3e-2)
    eng.backward(d_features=torch.tensor(g, device=gpu))
    grads = {k: v.cpu() for k, v in eng.grad_views().items()}
    # operands as the device holds them
    acts = [rnd(torch.tensor(video))] + [eng.read_layer(i, n).cpu().reshape(
        (n,) + tuple(int(v) for v in (torch_ref_out_shape(i)))) for i in range(7)]
    dys = [eng.read_grad_image(i, n).cpu() for i in range(8)]
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        # conv5b: external gradient gated by the forward output
        f5 = feat.cpu().reshape(n, 512, 2, 7, 7).permute(0, 2, 3, 4, 1)
        g5 = torch.tensor(g).reshape(n, 512, 2, 7, 7).permute(0, 2, 3, 4, 1)
        assert rel(dys[7].numpy(), (rnd(g5) * (f5 > 0)).numpy()) < 1e-6
        for i in range(7, -1, -1):
            name = NAMES[i]
            x, dy = ncdhw(acts[i]), ncdhw(dys[i])
            w = rnd(torch.tensor(p[name + '_w'])).permute(4, 3, 0, 1, 2).contiguous()          # [Co,Ci,kd,kh,kw]
            dw = torch.nn.grad.conv3d_weight(x, w.shape, dy, padding=1).permute(2, 3, 4, 1, 0)
            assert rel(grads[name + '_w'].numpy(), dw.numpy()) < tol, ('wgrad', name)
            assert rel(grads[name + '_b'].numpy(), dy.sum(dim=(0, 2, 3, 4)).numpy()) < tol, ('bias', name)
            if i == 0:
                break
            dx = torch.nn.grad.conv3d_input(x.shape, w, dy, padding=1).permute(0, 2, 3, 4, 1)   # d loss / d act[i]
            gated = dx * (acts[i] > 0)
            lo = NAMES[i - 1]
            if POOL[lo] is None:
                assert rel(dys[i - 1].numpy(), gated.numpy()) < tol, ('dgrad', name)
                continue
            pd, ph = POOL[lo]
            d = dys[i - 1]
            nn_, D, H, W, C = d.shape
            win = d.reshape(nn_, D // pd, pd, H // ph, ph, W // ph, ph, C).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(
                nn_, D // pd, H // ph, W // ph, C, pd * ph * ph)
            assert int(((win != 0).sum(-1) > 1).sum()) == 0, ('unpool: more than one member per window', lo)
            assert rel(win.sum(-1).numpy(), gated.numpy()) < tol, ('dgrad+unpool', name)
            # the routed member attains the window maximum of the conv output (ties within rounding)
            wl = rnd(torch.tensor(p[lo + '_w'])).permute(4, 3, 0, 1, 2).contiguous()
            z = F.conv3d(ncdhw(acts[i - 1]), wl, torch.tensor(p[lo + '_b']), padding=1).permute(0, 2, 3, 4, 1)
            zw = z.reshape(nn_, D // pd, pd, H // ph, ph, W // ph, ph, C).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(win.shape)
            chosen = (win != 0)
            slack = zw.max(-1, keepdim=True).values - zw
            worst = float((slack * chosen).max() / zw.abs().max())
            assert worst < (1e-5 if dtype == 'f32' else 2e-2), ('unpool routing', lo, worst)
    finally:
        torch.set_num_threads(old)
    # end-to-end against autograd of the fp32 CPU graph (RMS level, see module docstring)
    errs = {k: rms_rel(grads[k].numpy(), grads_ref[k]) for k in grads_ref}
    assert max(errs.values()) < TOL_E2E_RMS[dtype], errs
    # d_rows entry (layout of the rows buffer: [n*49][d*512+c]) gives the same gradients
    rows = torch.tensor(g, device=gpu).reshape(n, 512, 2, 49).permute(0, 3, 2, 1).reshape(n * 49, 1024).contiguous()
    eng.forward(torch.tensor(video, device=gpu))
    eng.backward(d_rows=rows)
    again = {k: v.cpu().numpy() for k, v in eng.grad_views().items()}
    assert all(rel(again[k], grads[k].numpy()) < 1e-4 for k in grads)      # fp32 atomics: order-dependent rounding only


def test_patch_filter_gradients_odd_window_count(gpu):
    """wgrad_patch.hip.h (conv2a ... conv4b; bf16): 5 windows give column ranges of unequal length per block
    (35 / 70 columns over 8 / 64 ranges), empty ranges, and several seamless column changes per block; on the 14 x 14
    layers (conv4a, conv4b: a column = a PAIR of windows) 3 columns of which the last holds a single window, i.e. the
    missing-window path and an odd number of groups per block."""
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    n = 5
    p = syn.c3d_params(33)
    rs = np.random.RandomState(34)
    video = (rs.rand(n, 16, 112, 112, 3).astype(np.float32) - 0.5) * 2
    g = rs.randn(n, 1024, 7, 7).astype(np.float32)
    eng = C3DEngine(n, dtype='bf16', device=gpu, save_for_backward=True)
    eng.set_weights(p)
    eng.forward(torch.tensor(video, device=gpu))
    eng.backward(d_features=torch.tensor(g, device=gpu))
    grads = {k: v.cpu() for k, v in eng.grad_views().items()}
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        for i in (1, 2, 3, 4, 5):
            name = NAMES[i]
            x = ncdhw(eng.read_layer(i - 1, n).cpu().reshape((n,) + tuple(int(v) for v in torch_ref_out_shape(i - 1))))
            dy = ncdhw(eng.read_grad_image(i, n).cpu())
            shape = tuple(torch.tensor(p[name + '_w']).permute(4, 3, 0, 1, 2).shape)
            dw = torch.nn.grad.conv3d_weight(x, shape, dy, padding=1).permute(2, 3, 4, 1, 0)
            assert np.abs(dw.numpy()).max() > 0
            assert rel(grads[name + '_w'].numpy(), dw.numpy()) < TOL_LOCAL['bf16'], ('wgrad', name)
    finally:
        torch.set_num_threads(old)


def test_filter_gradients_at_bench_scale(gpu):
    """wgrad_patch.hip.h at a window count where its column-range partition, XCD placement and 8-slot plane ring do what
    they do in the fine-tune benchmark: 67 windows (odd), replicas of 2 distinct windows with the same upstream gradient.
    Windows are independent, so dW = 34 dW(w0) + 33 dW(w1) with each term from torch.nn.grad.conv3d_weight on the
    operands the device holds for windows 0 and 1 -- every block of the launch works on real, non-zero data.  conv2a ...
    conv4b (patch kernels; conv4a / conv4b: 34 window pairs, the last one half empty) and conv5a / conv5b (wgrad_kernel);
    then the same check for the second
    kernel family (wgrad_kernel on every layer), each against the operands ITS OWN forward / backward chain left on the
    device (the two chains differ by ReLU-gate and pooling-route flips, so their gradients are not compared with each other)."""
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    n = 67
    p = syn.c3d_params(35)
    rs = np.random.RandomState(36)
    v2 = (rs.rand(2, 16, 112, 112, 3).astype(np.float32) - 0.5) * 2
    g2 = rs.randn(2, 1024, 7, 7).astype(np.float32)
    idx = np.arange(n) % 2
    video = torch.tensor(v2[idx], device=gpu)
    g = torch.tensor(g2[idx], device=gpu)
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        for kernels in ('patch', 'igemm'):
            eng = C3DEngine(n, dtype='bf16', device=gpu, save_for_backward=True, kernels=kernels)
            eng.set_weights(p)
            eng.forward(video)
            eng.backward(d_features=g)
            got = {k: v.cpu().double() for k, v in eng.grad_views().items() if k.endswith('_w')}
            xs = {i: ncdhw(eng.read_layer(i - 1, 2).cpu().reshape((2,) + tuple(int(v) for v in torch_ref_out_shape(i - 1)))) for i in range(1, 8)}
            dys = {i: ncdhw(eng.read_grad_image(i, 2).cpu()) for i in range(1, 8)}
            last = ncdhw(eng.read_grad_image(3, n).cpu()[n - 1:])        # the replicas really are replicas
            assert torch.equal(last, dys[3][(n - 1) % 2:(n - 1) % 2 + 1])
            del eng
            for i in range(1, 8):
                name = NAMES[i]
                shape = tuple(torch.tensor(p[name + '_w']).permute(4, 3, 0, 1, 2).shape)
                dw = sum(w * torch.nn.grad.conv3d_weight(xs[i][k:k + 1], shape, dys[i][k:k + 1], padding=1)
                         for k, w in ((0, 34.0), (1, 33.0))).permute(2, 3, 4, 1, 0)
                assert float(dw.abs().max()) > 0
                assert rel(got[name + '_w'].numpy(), dw.numpy()) < TOL_LOCAL['bf16'], ('wgrad', kernels, name)
    finally:
        torch.set_num_threads(old)


def test_backward_composition_under_the_devices_own_decisions(gpu, case):
    """The COMPOSITION of the bf16 conv-stack backward (layer order, gating, routing, what feeds what) at a bound that means
    something: the end-to-end comparison with autograd above needs an RMS tolerance of 0.5 because a bf16 forward flips ReLU
    gates and pooling routes, after which the two graphs are different piecewise-linear maps.  Here the reference chain is
    given the DEVICE's decisions -- gates = its activations > 0, routes = the member of each pooling window that carries its
    gradient (torch's arg-max of the recomputed conv output where the device's window is all zero) -- and otherwise runs on
    its own: the gradient it propagates from layer to layer is its own fp32 one (never re-seeded from the device, unlike the
    operator-local test), with torch.nn.grad operators.  What is left is the rounding of bf16 operands and of the bf16
    gradient images through eight layers: every filter and bias gradient within 1e-2 (relative Frobenius error)."""
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    p, video, g = case
    n = video.shape[0]
    rnd = lambda t: t.bfloat16().float()
    eng = C3DEngine(n, dtype='bf16', device=gpu, save_for_backward=True)
    eng.set_weights(p)
    feat, _ = eng.forward(torch.tensor(video, device=gpu))
    eng.backward(d_features=torch.tensor(g, device=gpu))
    grads = {k: v.cpu() for k, v in eng.grad_views().items()}
    acts = [rnd(torch.tensor(video))] + [eng.read_layer(i, n).cpu().reshape((n,) + tuple(int(v) for v in torch_ref_out_shape(i))) for i in range(7)]
    dys_dev = [eng.read_grad_image(i, n).cpu() for i in range(8)]
    old = torch.get_num_threads()
    torch.set_num_threads(16)
    errs = {}
    try:
        f5 = feat.cpu().reshape(n, 512, 2, 7, 7).permute(0, 2, 3, 4, 1)
        g_ref = torch.tensor(g).reshape(n, 512, 2, 7, 7).permute(0, 2, 3, 4, 1) * (f5 > 0)          # NDHWC, conv5b's gate
        for i in range(7, -1, -1):
            name = NAMES[i]
            x, dy = ncdhw(acts[i]), ncdhw(g_ref)
            w = rnd(torch.tensor(p[name + '_w'])).permute(4, 3, 0, 1, 2).contiguous()
            dw = torch.nn.grad.conv3d_weight(x, w.shape, dy, padding=1).permute(2, 3, 4, 1, 0)
            fro = lambda a, r: float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30))
            errs[name + '_w'] = fro(grads[name + '_w'], dw)
            errs[name + '_b'] = fro(grads[name + '_b'], dy.sum(dim=(0, 2, 3, 4)))
            if i == 0:
                break
            dx = torch.nn.grad.conv3d_input(x.shape, w, dy, padding=1).permute(0, 2, 3, 4, 1) * (acts[i] > 0)
            lo = NAMES[i - 1]
            if POOL[lo] is None:
                g_ref = dx
                continue
            pd, ph = POOL[lo]
            d = dys_dev[i - 1]
            nn_, D, H, W, C = d.shape
            to_win = lambda t: t.reshape(nn_, D // pd, pd, H // ph, ph, W // ph, ph, C).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(
                nn_, D // pd, H // ph, W // ph, C, pd * ph * ph)
            route = to_win(d) != 0                                                    # the device's routed member
            none = ~route.any(-1, keepdim=True)
            wl = rnd(torch.tensor(p[lo + '_w'])).permute(4, 3, 0, 1, 2).contiguous()
            z = F.conv3d(ncdhw(acts[i - 1]), wl, torch.tensor(p[lo + '_b']), padding=1).permute(0, 2, 3, 4, 1)
            zw = to_win(z)
            fallback = torch.zeros_like(route).scatter_(-1, zw.argmax(-1, keepdim=True), True)
            route = torch.where(none, fallback, route)
            assert int((route.sum(-1) != 1).sum()) == 0
            win = route * dx.unsqueeze(-1)                                            # [n, Dp, Hp, Wp, C, members]
            g_ref = win.reshape(nn_, D // pd, H // ph, W // ph, C, pd, ph, ph).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(nn_, D, H, W, C)
    finally:
        torch.set_num_threads(old)
    print('composition under device decisions, relative Frobenius error per gradient:', {k: '%.2e' % v for k, v in errs.items()})
    assert max(errs.values()) < 1e-2, errs          # measured: 1.7e-3 (conv5b) ... 4.7e-3 (conv1a), growing with depth


def torch_ref_out_shape(i):
    """NDHWC extent of layer i's pooled output."""
    d, h = {0: (16, 56), 1: (8, 28), 2: (8, 28), 3: (4, 14), 4: (4, 14), 5: (2, 7), 6: (2, 7)}[i]
    return d, h, h, torch_ref.C3D_LAYERS[i][2]


def test_training_forward_equals_inference_forward(gpu, case):
    """Recording the arg-max must not change the forward output (bit-exact, bf16 path incl. the conv1a kernel)."""
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    p, video, _ = case
    v = torch.tensor(video[:1], device=gpu)
    a = C3DEngine(1, dtype='bf16', device=gpu)
    a.set_weights(p)
    b = C3DEngine(1, dtype='bf16', device=gpu, save_for_backward=True)
    b.set_weights(p)
    fa, fb = a.forward(v)[0], b.forward(v)[0]
    assert torch.equal(fa, fb)


def test_backward_requires_training_plan_and_matching_windows(gpu, case):
    from recurrent_gaze_prediction_amd._lib import RgpError
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    p, video, g = case
    eng = C3DEngine(2, dtype='bf16', device=gpu, save_for_backward=True)
    eng.set_weights(p)
    eng.forward(torch.tensor(video[:1], device=gpu))
    with pytest.raises(RgpError):
        eng.backward(d_features=torch.tensor(g, device=gpu))          # 2 windows of gradient, 1 forwarded
    inf = C3DEngine(1, dtype='bf16', device=gpu)
    inf.set_weights(p)
    with pytest.raises(AssertionError):
        inf.backward(d_features=torch.tensor(g[:1], device=gpu))
