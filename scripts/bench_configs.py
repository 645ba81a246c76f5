#!/usr/bin/env python
"""One table for the five BASELINE.json configs on one MI355X: forward and training-step throughput of every
model family at the config's shape (synthetic inputs resident in HBM, median of timed repetitions).
Prints one JSON object; the headline metric itself is bench.py's."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from recurrent_gaze_prediction_amd import synthetic as syn                                            # noqa: E402
from recurrent_gaze_prediction_amd.engine import (C3DEngine, CascadeEngine, FcGruEngine, GrcnEngine,   # noqa: E402
                                                  ShallowNetEngine)
from recurrent_gaze_prediction_amd.finetune import EndToEndCascade, EndToEndGaze                      # noqa: E402

dev = torch.device('cuda:0')


def timed(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def entry(frames, sec):
    return {'ms': round(sec * 1e3, 3), 'frames_per_s': round(frames / sec, 1)}


out = {}
g = torch.Generator(device=dev)
g.manual_seed(0)

# config 1: frame-wise ShallowNet, 112x112 frames, 7x7 maps, batch 2 (and a throughput-sized batch)
for n in (2, 512):
    eng = ShallowNetEngine(n, 112, dtype='bf16', device=dev, save_for_backward=True)
    eng.set_weights(syn.shallownet_params(1, 112))
    fr = torch.rand(n, 112, 112, 3, device=dev, generator=g)
    d = torch.rand(n, 49, 49, device=dev, generator=g)
    out['cfg1_shallownet_112_n%d_fwd' % n] = entry(n, timed(lambda: eng.forward(fr, want_7x7=True)))
    out['cfg1_shallownet_112_n%d_fwd_bwd' % n] = entry(n, timed(lambda: (eng.forward(fr), eng.backward(d))))
    del eng

# config 2: fc-GRU over conv5b features, 16-step clips, 7x7 map, fp32
B, T = 64, 16
eng = FcGruEngine(B, T, (7, 7), dtype='f32', device=dev, save_for_backward=True)
eng.set_weights(syn.fcgru_params(2, 7, 7))
x = torch.relu(torch.randn(B, T, 1024, 7, 7, device=dev, generator=g))
gt7 = torch.rand(B, T, 7, 7, device=dev, generator=g)
gt7 = (gt7 / gt7.sum((-1, -2), keepdim=True)).contiguous()
out['cfg2_fcgru_f32_B64_T16_fwd'] = entry(B * T, timed(lambda: eng.forward(x)))


def fc_step():
    lg, pr = eng.forward(x)
    eng.backward(lg, pr, gt7)
    eng.adam_step(0, 1e-4)


out['cfg2_fcgru_f32_B64_T16_train_step'] = entry(B * T, timed(fc_step))
del eng

# config 3: gaze_grcn, 16-step clips, bf16: head on features, and end to end from video windows
eng = GrcnEngine(B, T, dtype='bf16', save_for_backward=True, device=dev)
eng.set_weights(syn.grcn_params(3, T))
gt = torch.rand(B, T, 49, 49, device=dev, generator=g)
gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
out['cfg3_grcn_bf16_B64_T16_head_fwd'] = entry(B * T, timed(lambda: eng.forward(x)))


def grcn_step():
    lg, pr = eng.forward(x)
    eng.backward(lg, pr, gt)
    eng.adam_step(0, 1e-4)


out['cfg3_grcn_bf16_B64_T16_head_train_step'] = entry(B * T, timed(grcn_step))
del eng
c3d = C3DEngine(1024, dtype='bf16', device=dev)
c3d.set_weights(syn.c3d_params(4))
head = GrcnEngine(B, T, dtype='bf16', device=dev)
head.set_weights(syn.grcn_params(3, T))
video = torch.rand(B * T, 16, 112, 112, 3, device=dev, generator=g) - 0.5
rows = torch.empty(B * T * 49, 1024, dtype=torch.bfloat16, device=dev)


def e2e():
    c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
    head.forward_rows(rows)


out['cfg3_grcn_bf16_B64_T16_e2e_fwd'] = entry(B * T, timed(e2e, reps=5))
del c3d, head, video, rows

# config 4: gaze_grcn head, 35-step clips, 8 clips per GPU (64 over 8 GPUs): training step
B4, T4 = 8, 35
eng = GrcnEngine(B4, T4, dtype='bf16', save_for_backward=True, device=dev)
eng.set_weights(syn.grcn_params(5, T4))
x4 = torch.relu(torch.randn(B4, T4, 1024, 7, 7, device=dev, generator=g))
gt4 = torch.rand(B4, T4, 49, 49, device=dev, generator=g)
gt4 = (gt4 / gt4.sum((-1, -2), keepdim=True)).contiguous()


def grcn4_step():
    lg, pr = eng.forward(x4)
    eng.backward(lg, pr, gt4)
    eng.adam_step(0, 1e-4)


out['cfg4_grcn_bf16_B8_T35_train_step'] = entry(B4 * T4, timed(grcn4_step))
del eng

# the reference's own shapes: training B = 28 x T = 42 (/root/reference/models/train_gaze.py:75, gaze_rnn.py:50) and the LSMDC
# map extraction B = 14 x T = 105 (/root/reference/models/extract_map.py:63-65), gaze_grcn head on features
for tag, Br, Tr, train in (('ref_train_B28_T42', 28, 42, True), ('ref_extract_map_B14_T105', 14, 105, False)):
    eng = GrcnEngine(Br, Tr, dtype='bf16', save_for_backward=train, device=dev)
    eng.set_weights(syn.grcn_params(9, Tr))
    xr = torch.relu(torch.randn(Br, Tr, 1024, 7, 7, device=dev, generator=g))
    gtr = torch.rand(Br, Tr, 49, 49, device=dev, generator=g)
    gtr = (gtr / gtr.sum((-1, -2), keepdim=True)).contiguous()
    out['%s_grcn_bf16_head_fwd' % tag] = entry(Br * Tr, timed(lambda: eng.forward(xr)))
    if train:
        def ref_step():
            lg, pr = eng.forward(xr)
            eng.backward(lg, pr, gtr)
            eng.adam_step(0, 1e-4)
        out['%s_grcn_bf16_head_train_step' % tag] = entry(Br * Tr, timed(ref_step))
    eng.status()
    del eng, xr, gtr
ft = EndToEndGaze(16, 16, dtype='bf16', device=dev, seed=6)
v16 = torch.rand(256, 16, 112, 112, 3, device=dev, generator=g) - 0.5
gt16 = torch.rand(16, 16, 49, 49, device=dev, generator=g)
gt16 = (gt16 / gt16.sum((-1, -2), keepdim=True)).contiguous()
out['cfg3_grcn_bf16_B16_T16_end_to_end_finetune_step'] = entry(256, timed(lambda: ft.train_step(v16, gt16, 1e-4), reps=3, warm=1))
del ft, v16

# config 5: two-level cascade, 35-step clips, 16 clips per GPU (128 over 8 GPUs)
B5, T5 = 16, 35
eng = CascadeEngine(B5, T5, 98, dtype='bf16', device=dev, save_for_backward=True)
eng.set_weights(syn.cascade_params(7))
fr5 = torch.rand(B5, T5, 98, 98, 3, device=dev, generator=g)
x5 = torch.relu(torch.randn(B5, T5, 1024, 7, 7, device=dev, generator=g))
gt5 = torch.rand(B5, T5, 49, 49, device=dev, generator=g)
out['cfg5_cascade_bf16_B16_T35_fwd'] = entry(B5 * T5, timed(lambda: eng.forward(fr5, x5)))
out['cfg5_cascade_bf16_B16_T35_fwd_bwd'] = entry(B5 * T5, timed(lambda: eng.backward(eng.forward(fr5, x5), gt5, want_d_rows=True)))
del eng
m = EndToEndCascade(B5, T5, dtype='bf16', device=dev, seed=8)
v5 = torch.rand(B5 * T5, 16, 112, 112, 3, device=dev, generator=g) - 0.5
out['cfg5_c3d_finetune_plus_cascade_B16_T35_train_step'] = entry(B5 * T5, timed(lambda: m.train_step(v5, fr5, gt5, 1e-4), reps=3, warm=1))
print(json.dumps(out, indent=1))
