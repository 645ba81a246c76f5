"""Dev: per-layer gradient errors of rgp_c3d_backward vs torch autograd (CPU), and backward timing."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from oracle import torch_ref                                          # noqa: E402
from recurrent_gaze_prediction_amd import synthetic as syn          # noqa: E402
from recurrent_gaze_prediction_amd.engine import C3DEngine         # noqa: E402

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = syn.c3d_params(31)
rs = np.random.RandomState(32)
video = (rs.rand(n, 16, 112, 112, 3).astype(np.float32) - 0.5) * 2
g = rs.randn(n, 1024, 7, 7).astype(np.float32)
tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
torch.set_num_threads(16)
t0 = time.time()
import torch.nn.functional as F                                      # noqa: E402
x = torch.tensor(video).permute(0, 4, 1, 2, 3)
pre = []
for name, _, _, pool in torch_ref.C3D_LAYERS:
    z = F.conv3d(x, tp[name + '_w'].permute(4, 3, 0, 1, 2), tp[name + '_b'], padding=1)
    z.retain_grad()
    pre.append(z)
    x = torch.relu(z)
    if pool is not None:
        x = F.max_pool3d(x, (pool[0], pool[1], pool[1]), ceil_mode=True)
feat = x.reshape(n, 1024, 7, 7)
(feat * torch.tensor(g)).sum().backward()
print('oracle %.1f s' % (time.time() - t0), flush=True)
ref = {k: v.grad.numpy() for k, v in tp.items()}
for dtype in ('f32', 'bf16'):
    eng = C3DEngine(n, dtype=dtype, device=dev, save_for_backward=True)
    eng.set_weights(p)
    f, _ = eng.forward(torch.tensor(video, device=dev))
    eng.backward(d_features=torch.tensor(g, device=dev))
    got = {k: v.cpu().numpy() for k, v in eng.grad_views().items()}
    print(dtype, 'feat err %.2e' % (np.abs(f.cpu().numpy() - feat.detach().numpy()).max() / np.abs(feat.detach().numpy()).max()))
    for k in sorted(ref):
        r = ref[k].astype(np.float64)
        d = got[k].astype(np.float64) - r
        print('  %-9s max-rel %.2e  rms-rel %.2e  |ref|max %.3e' % (k, np.abs(d).max() / np.abs(r).max(),
                                                                 np.sqrt((d ** 2).mean()) / np.sqrt((r ** 2).mean()), np.abs(r).max()), flush=True)
    for i in range(7, -1, -1):
        r = pre[i].grad.permute(0, 2, 3, 4, 1).numpy().astype(np.float64)
        d = eng.read_grad_image(i, n).cpu().numpy().astype(np.float64) - r
        bad = np.argwhere(np.abs(d) > 1e-3 * np.abs(r).max())
        print('  dYpre[%d] max-rel %.2e rms-rel %.2e  nbad %d of %d  first bad %s' % (
            i, np.abs(d).max() / np.abs(r).max(), np.sqrt((d ** 2).mean()) / np.sqrt((r ** 2).mean()), len(bad), d.size,
            [tuple(b) for b in bad[:6]]), flush=True)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        eng.backward(d_features=torch.tensor(g, device=dev))
    torch.cuda.synchronize()
    print('  backward %.2f ms for %d windows' % ((time.time() - t0) / 3 * 1e3, n), flush=True)
    del eng
