"""Dev: cascade (config 5's head) forward + backward loop for rocprofv3."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import CascadeEngine

dev = torch.device('cuda:0')
B, T = 16, 35
eng = CascadeEngine(B, T, 98, dtype='bf16', device=dev, save_for_backward=True)
eng.set_weights(syn.cascade_params(0))
frames = torch.rand(B, T, 98, 98, 3, device=dev)
c3d = torch.tensor(syn.c3d_features(1, B, T), device=dev)
gt = torch.rand(B, T, 49, 49, device=dev)
import time
for _ in range(10):
    eng.backward(eng.forward(frames, c3d), gt, want_d_rows=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.forward(frames, c3d)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(20):
    eng.backward(eng.forward(frames, c3d), gt, want_d_rows=True)
torch.cuda.synchronize()
t2 = time.perf_counter()
print('cascade 16 x 35: forward %.3f ms, forward + backward %.3f ms' % ((t1 - t0) * 50, (t2 - t1) * 50))

# the same with the host running ahead (as in config 5's joint step, where the conv stack's forward is still running while the
# cascade is enqueued): a long GEMM first, then device-side time between two events around the cascade
a = torch.randn(16384, 16384, device=dev, dtype=torch.bfloat16)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
fw, fb = [], []
for _ in range(10):
    for _ in range(int(os.environ.get('AHEAD_GEMMS', '3'))):
        a @ a
    e[0].record(); m = eng.forward(frames, c3d); e[1].record(); eng.backward(m, gt, want_d_rows=True); e[2].record()
    torch.cuda.synchronize()
    fw.append(e[0].elapsed_time(e[1])); fb.append(e[0].elapsed_time(e[2]))
fw.sort(); fb.sort()
print('cascade 16 x 35, host ahead: forward %.3f ms, forward + backward %.3f ms' % (fw[len(fw) // 2], fb[len(fb) // 2]))
