"""Dev: phase stamps of the staggered conv kernel for one C3D layer (RGP_ABLATE=32 RGP_STAMP=<layer>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine
n = 256
eng = C3DEngine(n, dtype='bf16')
eng.set_weights(syn.c3d_params(1))
v = torch.rand(n, 16, 112, 112, 3, device='cuda') - 0.5
for _ in range(int(os.environ.get("RGP_STAMP_ITERS", "2"))):
    eng.forward(v, want_features=False)
torch.cuda.synchronize()
