"""Dev: folded head vs three-stage head vs oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import GrcnEngine
dev = torch.device('cuda:0')
for dtype in ('f32', 'bf16'):
    for B, T in ((2, 3), (20, 4), (64, 16)):
        p = syn.grcn_params(11, T, gru_std=0.05, random_bn=True)
        x = torch.tensor(syn.c3d_features(12, B, T), device=dev)
        a = GrcnEngine(B, T, dtype=dtype, device=dev)
        b = GrcnEngine(B, T, dtype=dtype, device=dev, unfolded_head=True)
        a.set_weights(p); b.set_weights(p)
        la, _ = a.forward(x); lb, _ = b.forward(x)
        torch.cuda.synchronize()
        la, lb = la.cpu().numpy(), lb.cpu().numpy()
        print(dtype, B, T, 'folded absmax %.4e staged absmax %.4e rel diff %.3e  nonzero frac %.3f' % (
            np.abs(la).max(), np.abs(lb).max(), np.abs(la - lb).max() / np.abs(lb).max(), (la != 0).mean()), flush=True)
        if B == 2:
            ref = torch_ref.grcn_forward(torch.tensor(syn.c3d_features(12, B, T)), {k: torch.tensor(v) for k, v in p.items()}).numpy()
            print('   vs oracle: folded %.3e staged %.3e' % (np.abs(la - ref).max() / np.abs(ref).max(), np.abs(lb - ref).max() / np.abs(ref).max()))
            print('   folded[0,0,:2,:6]', la[0, 0, :2, :6], '\n   staged', lb[0, 0, :2, :6])
