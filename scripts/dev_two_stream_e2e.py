"""Dev experiment: the headline step (1024 windows -> C3D -> head) as ONE chain of launches against TWO half-batch chains on two
streams (two C3D plans with the same weights, 512 windows each; the head on all rows afterwards): does the other chain's next kernel
fill the tail of this one's?   python scripts/dev_two_stream_e2e.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine, GrcnEngine
dev = torch.device('cuda:0')
B, T = 64, 16
F = B * T
g = torch.Generator(device=dev); g.manual_seed(1)
video = torch.rand(F, 16, 112, 112, 3, device=dev, generator=g) - 0.5
head = GrcnEngine(B, T, dtype='bf16', device=dev); head.set_weights(syn.grcn_params(1, T))
w = syn.c3d_params(2)
one = C3DEngine(F, dtype='bf16', device=dev); one.set_weights(w)
halves = [C3DEngine(F // 2, dtype='bf16', device=dev) for _ in range(2)]
for h in halves: h.set_weights(w)
rows = torch.empty(F * 49, 1024, dtype=one.torch_dtype, device=dev)
logits = torch.empty(B, T, 49, 49, device=dev); probs = torch.empty_like(logits)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def step_one():
    one.forward(video, want_features=False, want_rows=True, out_rows=rows)
    head.forward_rows(rows, out_logits=logits, out_probs=probs)
def step_two():
    cur = torch.cuda.current_stream(dev)
    for i, (h, s) in enumerate(zip(halves, streams)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            h.forward(video[i * F // 2:(i + 1) * F // 2], want_features=False, want_rows=True, out_rows=rows[i * F // 2 * 49:(i + 1) * F // 2 * 49])
    for s in streams: cur.wait_stream(s)
    head.forward_rows(rows, out_logits=logits, out_probs=probs)
def timed(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
step_one(); ref = probs.clone(); step_two(); assert torch.equal(probs, ref), 'two-stream result differs'
for r in range(3):
    print('one chain %.3f ms   two half-batch chains on two streams %.3f ms' % (timed(step_one), timed(step_two)))
