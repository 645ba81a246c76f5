"""Dev: soak of the cascade's multi-stream passes at config 5's shape (16 x 35): N training passes on the same inputs -- maps
bit-identical, gradients within the atomics' noise, then the head-only grcn training step (B 8 x T 35, B 64 x T 16) the same way."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import CascadeEngine, GrcnEngine
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator(device=dev); g.manual_seed(1)
B, T = 16, 35
eng = CascadeEngine(B, T, 98, dtype='bf16', device=dev, save_for_backward=True)
eng.set_weights(syn.cascade_params(0))
frames = torch.rand(B, T, 98, 98, 3, device=dev, generator=g)
c3d = torch.tensor(syn.c3d_features(1, B, T), device=dev)
gt = torch.rand(B, T, 49, 49, device=dev, generator=g)
m0 = eng.forward(frames, c3d).clone(); _, d0 = eng.backward(m0, gt, want_d_rows=True); g0 = eng.flat_grads.clone(); d0 = d0.clone()
worst_g = worst_d = 0.0
t0 = time.time()
for i in range(N):
    m = eng.forward(frames, c3d)
    assert torch.equal(m, m0), 'maps differ at pass %d' % i
    _, d = eng.backward(m, gt, want_d_rows=True)
    worst_g = max(worst_g, float((eng.flat_grads - g0).abs().max() / g0.abs().max()))
    worst_d = max(worst_d, float((d - d0).abs().max() / d0.abs().max()))
print('cascade 16 x 35: %d passes, maps identical, worst gradient deviation %.2e (flat) %.2e (d_rows), %.1f s' % (N, worst_g, worst_d, time.time() - t0))
for B, T in ((8, 35), (64, 16)):
    h = GrcnEngine(B, T, dtype='bf16', save_for_backward=True, device=dev)
    h.set_weights(syn.grcn_params(1, T))
    x = torch.relu(torch.randn(B, T, 1024, 7, 7, device=dev, generator=g))
    gt = torch.rand(B, T, 49, 49, device=dev, generator=g) + 1e-3
    gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
    lg, pr = h.forward(x); lg0 = lg.clone(); h.backward(lg, pr, gt); g0 = h.flat_grads.clone()
    worst = 0.0
    for i in range(N):
        lg, pr = h.forward(x)
        assert torch.equal(lg, lg0), 'logits differ at pass %d' % i
        h.backward(lg, pr, gt)
        worst = max(worst, float((h.flat_grads - g0).abs().max() / g0.abs().max()))
    print('grcn head %d x %d: %d passes, logits identical, worst gradient deviation %.2e' % (B, T, N, worst))
