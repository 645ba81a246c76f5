"""Dev: config 1 (frame-wise ShallowNet, 512 frames of 112 x 112, bf16) forward / training-step time, un-profiled (median of 20)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import ShallowNetEngine
dev = torch.device('cuda:0')
n = 512
g = torch.Generator(device=dev); g.manual_seed(0)
eng = ShallowNetEngine(n, 112, dtype='bf16', device=dev, save_for_backward=True)
eng.set_weights(syn.shallownet_params(1, 112))
fr = torch.rand(n, 112, 112, 3, device=dev, generator=g)
d = torch.rand(n, 49, 49, device=dev, generator=g)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
def step():
    eng.forward(fr); eng.backward(d); eng.adam_step(0, 1e-4)
print('cfg1 fwd %.3f ms  train step %.3f ms' % (timed(lambda: eng.forward(fr)), timed(step)))
