"""Dev (make DEV=1 build): conv4a / conv4b with and without the halo-plane tap-group skipping (RGP_CP_ABLATE=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine
n = 1024
dev = torch.device('cuda:0')
c3d = C3DEngine(n, dtype='bf16', device=dev)
c3d.set_weights(syn.c3d_params(2))
g = torch.Generator(device=dev); g.manual_seed(1)
video = torch.rand(n, 16, 112, 112, 3, device=dev, generator=g) - 0.5
rows = torch.empty(n * 49, 1024, dtype=c3d.torch_dtype, device=dev)
ref = None
for rnd in range(3):
    for abl in (0, 16):
        os.environ['RGP_CP_ABLATE'] = str(abl)
        for _ in range(2):
            c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
        torch.cuda.synchronize()
        c3d.profile(True)
        for _ in range(4):
            c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
        torch.cuda.synchronize()
        pr = c3d.profile_read(); c3d.profile(False)
        if ref is None: ref = rows.clone()
        print('ablate %2d' % abl, ' '.join('%s=%.3f' % (k, pr[k][0] / max(pr[k][1], 1)) for k in ('conv3b', 'conv4a', 'conv4b', 'conv5a')), 'equal', bool(torch.equal(ref, rows)), flush=True)
