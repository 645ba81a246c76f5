"""The driver's workload with its input in HOST memory: 1024 fp32 windows per step (2.47 GB) copied from pinned buffers on a
side stream, double-buffered under the previous step's kernels, and the same from uint8 frames (16 x 128 x 171 x 3 per window
through rgp_c3d_forward_frames: 1.05 MB per window).  Reported in DESIGN.md; never bench.py's `value` (inputs resident)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recurrent_gaze_prediction_amd import synthetic as syn
from recurrent_gaze_prediction_amd.engine import C3DEngine, GrcnEngine

dev = torch.device('cuda:0')
B, T = 64, 16
F = B * T
c3d = C3DEngine(F, dtype='bf16', device=dev)
c3d.set_weights(syn.c3d_params(2))
head = GrcnEngine(B, T, dtype='bf16', device=dev)
head.set_weights(syn.grcn_params(3, T, gru_std=0.05))
rows = torch.empty(F * 49, 1024, dtype=c3d.torch_dtype, device=dev)
copy_stream = torch.cuda.Stream(device=dev)
out = {}


def run(name, host, dev_bufs, step, steps=8, warmup=2):
    done = [torch.cuda.Event() for _ in range(2)]
    used = [torch.cuda.Event() for _ in range(2)]
    t0 = None
    for it in range(warmup + steps + 1):
        k = it & 1
        with torch.cuda.stream(copy_stream):                       # H2D of batch `it` (waits until its buffer was consumed)
            if it >= 2:
                copy_stream.wait_event(used[k])
            dev_bufs[k].copy_(host[k], non_blocking=True)
            done[k].record(copy_stream)
        if it >= 1:                                                # compute on batch it - 1
            j = (it - 1) & 1
            torch.cuda.current_stream().wait_event(done[j])
            step(dev_bufs[j])
            used[j].record()
        if it == warmup:
            torch.cuda.synchronize()
            t0 = time.time()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    out[name] = {'ms_per_step': round(dt * 1e3, 2), 'frames_per_s': round(F / dt, 1),
                 'host_GB_per_step': round(host[0].numel() * host[0].element_size() / 1e9, 3)}


def step_windows(video):
    c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
    head.forward_rows(rows)


# (a) fp32 windows in pinned host memory
host = [torch.rand(F, 16, 112, 112, 3).sub_(0.5).pin_memory() for _ in range(2)]
bufs = [torch.empty(F, 16, 112, 112, 3, device=dev) for _ in range(2)]
run('fp32_windows_from_pinned_host', host, bufs, step_windows)
# resident reference on the same box
torch.cuda.synchronize()
t0 = time.time()
for _ in range(8):
    step_windows(bufs[0])
torch.cuda.synchronize()
out['resident'] = {'ms_per_step': round((time.time() - t0) / 8 * 1e3, 2)}
del host, bufs
# (b) uint8 frames 128 x 171 (the VIDEO_DATA layer on the device): 16 frames per window, windows back to back
mean = torch.zeros(3, 16, 128, 171, device=dev)
starts = list(range(0, 16 * F, 16))
hostf = [torch.randint(0, 256, (16 * F, 128, 171, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
buff = [torch.empty(16 * F, 128, 171, 3, dtype=torch.uint8, device=dev) for _ in range(2)]


def step_frames(frames):
    c3d.forward_frames(frames, starts, mean, want_features=False, want_rows=True, out_rows=rows)
    head.forward_rows(rows)


run('uint8_frames_128x171_from_pinned_host', hostf, buff, step_frames)
print(json.dumps(out))
