"""HIP-graph replay of the launch-bound training step of the gaze_grcn head (BASELINE config 4: 35-step clips,
8 clips per GPU).

At that shape a step is ~360 small launches (T recurrent steps x (2 forward + 4 backward) kernels, the head's
phases, packing) and the GPU idles between them.  The whole step is static -- fixed (B, T), device-resident
inputs, and the optimizer's step counter and learning-rate schedule live on the device
(rgp_lr_schedule_step / rgp_adam_clip_step_dev) -- so it is captured once and replayed:

    graph 1:  forward -> backward                       (into the flat gradient bucket)
    eager  :  all-reduce of the bucket over RCCL         (only when a process group is attached)
    graph 2:  global-norm clip + TF-Adam + re-pack       (base.py:286-297)

torch.cuda.CUDAGraph is the capture mechanism (hipGraph on ROCm); the captured work is the librgp_hip launches.
"""
import ctypes

import torch

from . import _lib
from . import dist as rdist
from .engine import _ptr, _stream_ptr


class DeviceAdam(object):
    """clip_by_global_norm + TF-Adam over several engines' flat buffers with the schedule on the device."""

    def __init__(self, engines, lr0, decay=0.8, decay_steps=500, max_grad_norm=10.0, beta1=0.9, beta2=0.999, eps=1e-8):
        self.engines = list(engines)
        self.lib, self.device = engines[0].lib, engines[0].device
        self.lr0, self.decay, self.decay_steps = float(lr0), float(decay), int(decay_steps)
        self.max_grad_norm, self.b1, self.b2, self.eps = float(max_grad_norm), beta1, beta2, eps
        npart = _lib.RGP_SQNORM_PARTIALS
        self.partials = torch.zeros(npart * len(self.engines), dtype=torch.float32, device=self.device)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
        for e in self.engines:
            if getattr(e, 'adam_m', None) is None:
                e.adam_m, e.adam_v = torch.zeros_like(e.flat_params), torch.zeros_like(e.flat_params)

    def step(self):
        lib, npart, s = self.lib, _lib.RGP_SQNORM_PARTIALS, _stream_ptr(self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.rgp_lr_schedule_step(_ptr(self.step_dev), self.lr0, self.decay, self.decay_steps, self.b1, self.b2,
                                                _ptr(self.lr_t), s))
            for i, e in enumerate(self.engines):
                _lib.check(lib.rgp_global_sqnorm(_ptr(e.flat_grads), e.flat_grads.numel(), _ptr(self.partials[i * npart:]), s))
            for e in self.engines:
                _lib.check(lib.rgp_adam_clip_step_dev(_ptr(e.flat_params), _ptr(e.flat_grads), _ptr(e.adam_m), _ptr(e.adam_v),
                                                      e.flat_params.numel(), _ptr(self.partials), self.partials.numel(),
                                                      _ptr(self.lr_t), self.b1, self.b2, self.eps, self.max_grad_norm,
                                                      _ptr(self.gnorm), s))
                e.repack()


class GraphedHeadTrainStep(object):
    """Captured training step of a GrcnEngine on static device buffers x [B,T,1024,7,7] and labels [B,T,49,49]
    (copy new batches INTO them).  step() replays; loss is not part of the graph (compute it from .logits)."""

    def __init__(self, head, x, labels, lr0, decay=0.8, decay_steps=500, max_grad_norm=10.0, loss_type='xentropy', dist=None,
                 use_graph=True):
        self.head, self.x, self.labels, self.loss_type, self.dist = head, x, labels, loss_type, dist
        self.logits = torch.empty(head.B, head.T, 49, 49, device=head.device)
        self.probs = torch.empty_like(self.logits)
        self.opt = DeviceAdam([head], lr0, decay, decay_steps, max_grad_norm)
        self.g1 = self.g2 = None
        side = torch.cuda.Stream(device=head.device)
        side.wait_stream(torch.cuda.current_stream(head.device))
        with torch.cuda.stream(side):                       # warm-up: first-call attribute setup, gradient buffers
            for _ in range(2):
                self._fwd_bwd()
            self.head.flat_grads.zero_()
        torch.cuda.current_stream(head.device).wait_stream(side)
        torch.cuda.synchronize(head.device)
        if use_graph:
            self.g1, self.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g1):
                self._fwd_bwd()
            with torch.cuda.graph(self.g2):
                self.opt.step()

    def _fwd_bwd(self):
        self.head.forward(self.x, out_logits=self.logits, out_probs=self.probs)
        self.head.backward(self.logits, self.probs, self.labels, self.loss_type)

    def step(self):
        if self.g1 is not None:
            self.g1.replay()
        else:
            self._fwd_bwd()
        if self.dist is not None:
            rdist.allreduce_mean_(self.dist, [self.head.flat_grads])
        if self.g2 is not None:
            self.g2.replay()
        else:
            self.opt.step()

    @property
    def global_step(self):
        return int(self.opt.step_dev.item())
