"""Dev: C3D per-layer times of two builds of librgp_hip.so on one box, alternating processes.
usage: dev_ab_lib.py <other .so> [rounds]   (child mode: dev_ab_lib.py --child <.so>)
(older builds are kept under prev/, which .gpurunignore keeps off the GPU box: take the entry out for an A/B run)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == '--child':
    import ctypes
    from recurrent_gaze_prediction_amd import _lib
    _lib.LIB_PATH = sys.argv[2]
    _probe = ctypes.CDLL(sys.argv[2])                      # an older build lacks the newer entry points: bind what it has
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
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
    for _ in range(3):
        c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
    torch.cuda.synchronize()
    c3d.profile(True)
    for _ in range(8):
        c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
    torch.cuda.synchronize()
    pr = c3d.profile_read()
    names = ('conv1a', 'conv2a', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'conv5a', 'conv5b')
    print(' '.join('%s=%.3f' % (k, pr[k][0] / pr[k][1]) for k in names), 'sum=%.2f' % sum(pr[k][0] / pr[k][1] for k in names),
          'checksum %.6e' % float(rows.float().abs().sum()))
else:
    other = os.path.abspath(sys.argv[1])
    cur = os.path.join(ROOT, 'recurrent_gaze_prediction_amd', 'librgp_hip.so')
    for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
        for tag, lib in (('new ', cur), ('prev', other)):
            out = subprocess.run([sys.executable, __file__, '--child', lib], capture_output=True, text=True)
            print(tag, out.stdout.strip() or out.stderr[-300:], flush=True)
