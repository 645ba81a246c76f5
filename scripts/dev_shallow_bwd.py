import sys, torch
sys.path.insert(0, '.')
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import ShallowNetEngine
dev = torch.device('cuda:0')
n = 512
eng = ShallowNetEngine(n, 112, dtype='bf16', device=dev, save_for_backward=True)
eng.set_weights(syn.shallownet_params(1, 112))
fr = torch.rand(n, 112, 112, 3, device=dev)
d = torch.rand(n, 49, 49, device=dev)
for _ in range(3):
    eng.forward(fr); eng.backward(d)
torch.cuda.synchronize()
