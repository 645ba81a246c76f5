"""Dev: config 2 (fc-GRU over C3D features, f32, B=64 x T=16) forward and training step, for rocprofv3 --stats."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import FcGruEngine

dev = torch.device('cuda:0')
B, T = 64, 16
g = torch.Generator(device=dev); g.manual_seed(0)
eng = FcGruEngine(B, T, (7, 7), dtype='f32', device=dev, save_for_backward=True)
eng.set_weights(syn.fcgru_params(2, 7, 7))
x = torch.relu(torch.randn(B, T, 1024, 7, 7, device=dev, generator=g))
gt7 = torch.rand(B, T, 7, 7, device=dev, generator=g)
gt7 = (gt7 / gt7.sum((-1, -2), keepdim=True)).contiguous()
mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
for _ in range(20):
    lg, pr = eng.forward(x)
    if mode == 'train':
        eng.backward(lg, pr, gt7)
        eng.adam_step(0, 1e-4)
torch.cuda.synchronize()
