"""Dev: wall time of the cascade forward (rgp_cascade_forward) at a few (B, T)."""
import sys
import time

import torch

sys.path.insert(0, '.')
from recurrent_gaze_prediction_amd import synthetic as syn          # noqa: E402
from recurrent_gaze_prediction_amd.engine import CascadeEngine     # noqa: E402

dev = torch.device('cuda:0')
for dtype in ('bf16', 'f32'):
    for B, T in ((5, 35), (16, 35)):
        eng = CascadeEngine(B, T, 98, dtype=dtype, device=dev)
        eng.set_weights(syn.cascade_params(0))
        frames = torch.rand(B, T, 98, 98, 3, device=dev)
        c3d = torch.tensor(syn.c3d_features(1, B, T), device=dev)
        for _ in range(3):
            eng.forward(frames, c3d)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 10
        for _ in range(n):
            eng.forward(frames, c3d)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / n * 1e3
        print('cascade %s B=%d T=%d: %.2f ms/forward, %.0f frames/s, workspace %.1f MiB' %
              (dtype, B, T, ms, B * T / ms * 1e3, eng.workspace.numel() / 2 ** 20), flush=True)
        del eng
for B, T in ((5, 35), (16, 35)):
    eng = CascadeEngine(B, T, 98, dtype='bf16', device=dev, save_for_backward=True)
    eng.set_weights(syn.cascade_params(0))
    frames = torch.rand(B, T, 98, 98, 3, device=dev)
    c3d = torch.tensor(syn.c3d_features(1, B, T), device=dev)
    gt = torch.rand(B, T, 49, 49, device=dev)
    for _ in range(2):
        eng.backward(eng.forward(frames, c3d), gt, want_d_rows=True)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        eng.backward(eng.forward(frames, c3d), gt, want_d_rows=True)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / n * 1e3
    print('cascade bf16 B=%d T=%d forward+backward: %.2f ms, %.0f frames/s, workspace %.1f MiB' %
          (B, T, ms, B * T / ms * 1e3, eng.workspace.numel() / 2 ** 20), flush=True)
    del eng
