"""Dev: config 2 (fc-GRU, f32, B=64 x T=16) forward / training-step time, un-profiled (median of 20)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import FcGruEngine
dev = torch.device('cuda:0')
B, T = 64, 16
g = torch.Generator(device=dev); g.manual_seed(0)
eng = FcGruEngine(B, T, (7, 7), dtype='f32', device=dev, save_for_backward=True)
eng.set_weights(syn.fcgru_params(2, 7, 7))
x = torch.relu(torch.randn(B, T, 1024, 7, 7, device=dev, generator=g))
gt7 = torch.rand(B, T, 7, 7, device=dev, generator=g); gt7 = (gt7 / gt7.sum((-1, -2), keepdim=True)).contiguous()
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
def step():
    lg, pr = eng.forward(x); eng.backward(lg, pr, gt7); eng.adam_step(0, 1e-4)
print('cfg2 fwd %.3f ms  train step %.3f ms' % (timed(lambda: eng.forward(x)), timed(step)))
