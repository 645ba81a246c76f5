"""Dev: time the C3D backward's wgrad launches per layer (use with RGP_WG_ABLATE)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine
n = 256
eng = C3DEngine(n, dtype='bf16', save_for_backward=True)
eng.set_weights(syn.c3d_params(1))
v = torch.rand(n, 16, 112, 112, 3, device='cuda') - 0.5
g = torch.randn(n, 1024, 7, 7, device='cuda')
eng.forward(v, want_features=False)
for _ in range(2):
    eng.backward(d_features=g)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    eng.backward(d_features=g)
torch.cuda.synchronize()
print('ABLATE=%s backward %.2f ms' % (os.environ.get('RGP_WG_ABLATE', '0'), (time.time() - t0) / 3 * 1e3), flush=True)
