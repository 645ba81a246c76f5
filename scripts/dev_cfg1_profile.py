"""Dev: config 1 (FramewiseShallowNet's ShallowNet, 112 x 112 frames) forward / forward + backward loops for rocprofv3."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import ShallowNetEngine

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
g = torch.Generator(device=dev); g.manual_seed(0)
eng = ShallowNetEngine(n, 112, dtype='bf16', device=dev, save_for_backward=True)
eng.set_weights(syn.shallownet_params(1, 112))
fr = torch.rand(n, 112, 112, 3, device=dev, generator=g)
d = torch.rand(n, 49, 49, device=dev, generator=g)
for _ in range(20):
    eng.forward(fr, want_7x7=(mode == 'fwd'))
    if mode != 'fwd':
        eng.backward(d)
torch.cuda.synchronize()
print('done')
