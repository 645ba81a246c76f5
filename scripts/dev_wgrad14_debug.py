"""Dev: localise errors of the 14 x 14 patch filter-gradient kernel against wgrad_kernel (same process, DEV library:
RGP_WGPATCH selects per backward call).  usage: python scripts/dev_with_lib.py <dev lib> scripts/dev_wgrad14_debug.py [n]"""
import os, sys
import numpy as np, torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
p = syn.c3d_params(31)
rs = np.random.RandomState(32)
video = torch.tensor((rs.rand(n, 16, 112, 112, 3).astype(np.float32) - 0.5) * 2, device=dev)
g = torch.tensor(rs.randn(n, 1024, 7, 7).astype(np.float32), device=dev)
eng = C3DEngine(n, dtype='bf16', device=dev, save_for_backward=True)
eng.set_weights(p)
res = {}
for mask in ('7', '31'):
    os.environ['RGP_WGPATCH'] = mask
    eng.forward(video)
    eng.backward(d_features=g)
    torch.cuda.synchronize()
    res[mask] = {k: v.cpu().double().clone() for k, v in eng.grad_views().items()}
for name in ('conv4a', 'conv4b', 'conv3a'):
    for kind in ('_w', '_b'):
        a, b = res['31'][name + kind], res['7'][name + kind]
        bad = ~torch.isfinite(a)
        err = (a - b).abs()
        err[bad] = float('inf')
        scale = float(b.abs().max())
        print(name + kind, 'shape', tuple(a.shape), 'non-finite', int(bad.sum()), 'max rel err', float(err[~bad].max() / scale) if (~bad).any() else None)
        if kind == '_w':
            e = (err / scale > 1.5e-2) | bad                     # [3,3,3,cin,cout]
            if e.any():
                print('  wrong per tap (kz,ky,kx):', e.sum(dim=(3, 4)).flatten().tolist())
                cin, cout = a.shape[3], a.shape[4]
                print('  wrong per 32-channel input slice:', e.reshape(27, cin // 32, 32, cout).sum(dim=(0, 2, 3)).tolist())
                print('  wrong per 16-channel input tile parity (ct):', e.reshape(27, cin // 16, 16, cout).sum(dim=(0, 2, 3)).reshape(-1, 2).sum(0).tolist())
                print('  wrong per 64-channel output slice:', e.reshape(27, cin, cout // 64, 64).sum(dim=(0, 1, 3)).tolist())
                print('  wrong per 16-wide output tile j:', e.reshape(27, cin, cout // 16, 16).sum(dim=(0, 1, 3)).reshape(-1, 4).sum(0).tolist())
                print('  sample got/want:', a.flatten()[:6].tolist(), b.flatten()[:6].tolist())
