"""Dev: persistent ConvGRU sequence kernel vs the per-step launches (needs a `make DEV=1` build: RGP_SEQ=0/1)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    from recurrent_gaze_prediction_amd import synthetic as syn
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    B, T = int(sys.argv[2]), int(sys.argv[3])
    p = syn.grcn_params(3, T, 512, 128, gru_std=0.05, random_bn=True)
    x = torch.tensor(syn.c3d_features(4, B, T), device='cuda:0')
    eng = GrcnEngine(B, T, dtype='bf16', device='cuda:0', save_for_backward=True)
    eng.set_weights(p)
    import time
    logits, _ = eng.forward(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        eng.forward(x)
    torch.cuda.synchronize()
    print('ms per forward', (time.time() - t0) / 20 * 1e3, file=sys.stderr)
    h = eng.read_buffer('rcn_outputs').cpu().numpy().reshape(B, T, 49, 128)
    np.save(sys.argv[4], h)
    np.save(sys.argv[4] + '.logits.npy', logits.cpu().numpy())
else:
    B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 3)
    outs = []
    for seq in ('0', '1'):
        f = '/tmp/seq_%s.npy' % seq
        subprocess.check_call([sys.executable, __file__, 'child', str(B), str(T), f], env=dict(os.environ, RGP_SEQ=seq))
        outs.append((np.load(f), np.load(f + '.logits.npy')))
    (h0, l0), (h1, l1) = outs
    print('B %d T %d: nan in persistent: %d of %d' % (B, T, np.isnan(h1).sum(), h1.size))
    bad = np.argwhere(np.isnan(h1))
    if len(bad):
        print('first / last nan index (b,t,row,ch):', bad[0], bad[-1], 'distinct b', np.unique(bad[:, 0]), 't', np.unique(bad[:, 1]),
              'rows', np.unique(bad[:, 2])[:60], 'ch', np.unique(bad[:, 3])[:40])
    d = np.abs(np.nan_to_num(h1) - h0)
    print('max abs diff states', d.max(), 'at', np.unravel_index(d.argmax(), d.shape), 'ref max', np.abs(h0).max())
    for t in range(T):
        print(' t', t, 'max diff', d[:, t].max())
    print('logits max diff', np.abs(np.nan_to_num(l1) - l0).max(), 'ref max', np.abs(l0).max())
