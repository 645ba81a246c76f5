"""Dev (make DEV=1 build of rgp_c3d_bwd.o): filter gradients of conv2a / conv3a / conv3b at the benchmark's size (256
windows) from wgrad_patch.hip.h against wgrad_kernel on the same operands (RGP_WGPATCH = 7 / 0 in one process)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda:0')
eng = C3DEngine(n, dtype='bf16', device=dev, save_for_backward=True)
eng.set_weights(syn.c3d_params(5))
g = torch.Generator(device=dev); g.manual_seed(7)
video = torch.rand(n, 16, 112, 112, 3, device=dev, generator=g) - 0.5
d_feat = torch.randn(n, 1024, 7, 7, device=dev, generator=g)
eng.forward(video)
res = {}
for mask in ('0', '7'):
    os.environ['RGP_WGPATCH'] = mask
    eng.backward(d_features=d_feat)
    torch.cuda.synchronize()
    res[mask] = {k: v.double().clone() for k, v in eng.grad_views().items() if k.endswith('_w')}
for k in ('conv2a_w', 'conv3a_w', 'conv3b_w', 'conv4a_w'):
    a, b = res['0'][k], res['7'][k]
    print('%-9s max|ref| %.3e  max-abs diff / max|ref| %.2e  rms rel %.2e' % (
        k, float(a.abs().max()), float((a - b).abs().max() / a.abs().max()), float(((a - b) ** 2).mean().sqrt() / (a ** 2).mean().sqrt())))
