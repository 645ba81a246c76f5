import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import GrcnEngine, C3DEngine
what = sys.argv[1] if len(sys.argv) > 1 else 'head'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
if what in ('head', 'all'):
    B, T = 64, 16
    eng = GrcnEngine(B, T, dtype=dtype); eng.set_weights(syn.grcn_params(1, T))
    x = torch.relu(torch.randn(B, T, 1024, 7, 7, device='cuda'))
    lg = torch.empty(B, T, 49, 49, device='cuda'); pr = torch.empty_like(lg)
    dt = timeit(lambda: eng.forward(x, out_logits=lg, out_probs=pr))
    print('head %s B=%d T=%d: %.3f ms  %.0f frames/s  %.1f TFLOP/s (432.79 MFLOP/frame)' % (dtype, B, T, dt*1e3, B*T/dt, B*T*432.79e6/dt/1e12))
if what in ('c3d', 'all'):
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    eng = C3DEngine(n, dtype=dtype); eng.set_weights(syn.c3d_params(1))
    v = torch.rand(n, 16, 112, 112, 3, device='cuda') - 0.5
    rows = torch.empty(n*49, 1024, dtype=eng.torch_dtype, device='cuda')
    dt = timeit(lambda: eng.forward(v, want_features=False, want_rows=True, out_rows=rows), n=3, w=1)
    print('c3d %s n=%d: %.3f ms  %.1f windows/s  %.1f TFLOP/s (76.99 GFLOP/window)' % (dtype, n, dt*1e3, n/dt, n*76.99327e9/dt/1e12))
