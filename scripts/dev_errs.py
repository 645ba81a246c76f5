import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import GrcnEngine
def errs(a, r):
    a = np.asarray(a, np.float64); r = np.asarray(r, np.float64)
    return 'maxrel %.3e rmsrel %.3e' % (np.abs(a-r).max()/np.abs(r).max(), np.sqrt(((a-r)**2).mean())/np.sqrt((r**2).mean()))
for std in (0.05, 0.02):
  for dtype in ('f32', 'bf16'):
    B,T,P,S = 2,3,512,128
    p = syn.grcn_params(11, T, P, S, gru_std=std, random_bn=True); x = syn.c3d_features(12, B, T)
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    lg, hs, emb = torch_ref.grcn_forward(torch.tensor(x, dtype=torch.float64), pt, want_hidden=True)
    eng = GrcnEngine(B,T,P,S,dtype=dtype); eng.set_weights(p)
    logits, probs = eng.forward(torch.tensor(x, device='cuda'))
    print(std, dtype, 'emb', errs(eng.read_buffer('c3d_embedded').cpu().numpy().reshape(emb.shape), emb.numpy()))
    print(std, dtype, 'h  ', errs(eng.read_buffer('rcn_outputs').cpu().numpy().reshape(hs.shape), hs.numpy()))
    print(std, dtype, 'lg ', errs(logits.cpu().numpy(), lg.numpy()), 'logit range', float(lg.min()), float(lg.max()))
    pr = torch_ref.softmax_maps(lg).numpy()
    print(std, dtype, 'pr ', errs(probs.cpu().numpy(), pr))
