"""Dev (make DEV=1 build): A/B of conv1a variants on one box, interleaved -- per-stage ms of the C3D forward over 1024
windows with RGP_C1VAR = 0, 1, 0, 1, ...   usage: dev_conv1a_ab.py [rounds] [knob] [values...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recurrent_gaze_prediction_amd import synthetic as syn          # noqa: E402
from recurrent_gaze_prediction_amd.engine import C3DEngine         # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
knob = sys.argv[2] if len(sys.argv) > 2 else 'RGP_C1VAR'
values = sys.argv[3:] or ['0', '1']
n = 1024
eng = C3DEngine(n, dtype='bf16')
eng.set_weights(syn.c3d_params(1))
v = torch.rand(n, 16, 112, 112, 3, device='cuda') - 0.5
rows = torch.empty(n * 49, 1024, dtype=eng.torch_dtype, device='cuda')
eng.profile(True)
for r in range(rounds):
    for val in values:
        os.environ[knob] = val
        for _ in range(2):
            eng.forward(v, want_features=False, want_rows=True, out_rows=rows)
        torch.cuda.synchronize()
        eng.profile_read()
        for _ in range(5):
            eng.forward(v, want_features=False, want_rows=True, out_rows=rows)
        torch.cuda.synchronize()
        st = eng.profile_read()
        print('%s=%s  ' % (knob, val) + '  '.join('%s %.3f' % (k, t / 5) for k, (t, _) in st.items() if t > 0), flush=True)
