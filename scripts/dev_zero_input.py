"""Dev: per-layer time of the C3D forward (1024 windows) on random windows, on all-zero windows and with all-zero
filters -- operand toggling is what the chip's clock reacts to (MI355X_MICROARCH.md, DVFS give-back), so the zero runs
show each kernel's stall-limited time at the full clock.  usage: dev_zero_input.py [windows [case,case]]
(with a DEV=1 build, RGP_CP_ABLATE selects the timing ablations of conv_patch.hip.h)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('RGP_DEV_LIB'):                         # another build of the library (A/B on one box)
    import ctypes
    from recurrent_gaze_prediction_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['RGP_DEV_LIB'])
    _probe = ctypes.CDLL(_lib.LIB_PATH)
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
import torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
only = sys.argv[2].split(',') if len(sys.argv) > 2 else None
dev = torch.device('cuda:0')
c3d = C3DEngine(n, dtype='bf16', device=dev)
p = syn.c3d_params(2)
g = torch.Generator(device=dev); g.manual_seed(1)
rnd = torch.rand(n, 16, 112, 112, 3, device=dev, generator=g) - 0.5
rows = torch.empty(n * 49, 1024, dtype=c3d.torch_dtype, device=dev)
for label, video, params in (('random', rnd, p), ('zero-video', torch.zeros_like(rnd), p),
                             ('zero-filters', rnd, {k: v * 0 for k, v in p.items()}), ('random', rnd, p)):
    if only is not None and label not in only:
        continue
    c3d.set_weights(params)
    for _ in range(2):
        c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
    torch.cuda.synchronize()
    c3d.profile(True)
    for _ in range(5):
        c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
    torch.cuda.synchronize()
    pr = c3d.profile_read()
    c3d.profile(False)
    print('%-13s' % label, ' '.join('%s=%.2f' % (k, pr[k][0] / max(pr[k][1], 1)) for k in
                                   ('conv1a', 'conv2a', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'conv5a', 'conv5b')), flush=True)
