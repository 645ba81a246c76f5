"""Dev: C3D forward over 1024 windows in one launch chain vs four chains of 256 -- must be bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = syn.c3d_params(1)
v = torch.rand(N, 16, 112, 112, 3, device='cuda') - 0.5
big = C3DEngine(N, dtype='bf16'); big.set_weights(w)
fb = big.forward(v)[0].clone()
del big
small = C3DEngine(256, dtype='bf16'); small.set_weights(w)
fs = torch.cat([small.forward(v[i:i + 256])[0].clone() for i in range(0, N, 256)])
print('chunk', N, 'vs 256: identical =', bool(torch.equal(fb, fs)), 'max abs', float((fb - fs).abs().max()), flush=True)
