"""End-to-end fine-tune: video windows -> C3D conv stack -> gaze_grcn head -> loss, with the gradient
flowing back into conv1a..conv5b (BASELINE config 5 "end-to-end C3D fine-tune"; the reference keeps the
Caffe features frozen in configs 2-4 and names the joint run in its cascade experiment).

One training step = C3DEngine.forward (recording pooling arg-max) -> GrcnEngine.forward_rows -> loss
(gaze_rnn.py:363-408) -> GrcnEngine.backward (+ backward_input: d loss / d conv5b rows) ->
C3DEngine.backward -> [all-reduce of both flat gradient buckets over RCCL, SURVEY 8e] ->
clip_by_global_norm over ALL variables + TF Adam (base.py:286-297).

When B*T exceeds the conv plan's max_windows the windows are processed in chunks: the first pass keeps
only the conv5b rows, and each chunk's activations are recomputed right before its backward.
"""
import numpy as np
import torch

from . import dist as rdist
from . import synthetic
from .engine import C3DEngine, GrcnEngine, adam_clip_step_multi


def _c3d_backward_chunks(m, video, d_rows, reducer):
    """Conv-stack backward over the plan's window chunks; with a reducer, layer i's gradient bucket is handed to
    RCCL as soon as the LAST chunk's layer-i kernels are queued (conv5b first), so the all-reduce of 110.6 MB runs
    under the backward of the layers below instead of after it."""
    chunks = m._chunks()
    m.c3d.flat_grads.zero_()
    for ci, (w0, n) in enumerate(chunks):
        if len(chunks) > 1:          # recompute this chunk's activations (only the last chunk's are resident)
            m.c3d.forward(video[w0:w0 + n], want_features=False)
        m.c3d.backward(d_rows=d_rows[w0 * 49:(w0 + n) * 49], zero_grads=False)
    if reducer is not None and reducer.dist is not None:
        for layer in range(7, -1, -1):
            reducer.reduce(m.c3d.layer_grad_slice(layer),
                           ready=(lambda st, layer=layer: m.c3d.wait_layer_grads(layer, st)) if reducer.cuda else None)


class EndToEndGaze(object):
    def __init__(self, batch, n_steps, dtype='bf16', device='cuda:0', max_windows=None, seed=0, c3d_params=None,
                 grcn_params=None, loss_type='xentropy', per_step=False):
        """per_step=True: the head's ConvGRU recurrence and BPTT as per-timestep launches (RGP_GRCN_PER_STEP) -- needed
        when several processes share one device, where two persistent launches would compete for the CUs."""
        self.B, self.T, self.F = int(batch), int(n_steps), int(batch) * int(n_steps)
        self.loss_type = loss_type
        self.device = torch.device(device)
        self.c3d = C3DEngine(min(self.F, max_windows or self.F), dtype=dtype, device=device, save_for_backward=True)
        self.head = GrcnEngine(self.B, self.T, dtype=dtype, save_for_backward=True, device=device, per_step=per_step)
        self.c3d.set_weights(c3d_params if c3d_params is not None else synthetic.c3d_params(seed))
        self.head.set_weights(grcn_params if grcn_params is not None else synthetic.grcn_params(seed + 1, self.T))
        self.rows = torch.empty(self.F * 49, 1024, dtype=self.c3d.torch_dtype, device=self.device)
        self.d_rows = torch.empty(self.F * 49, 1024, device=self.device)
        self.dist = None
        self.global_step = 0

    @property
    def engines(self):
        return [self.c3d, self.head]

    def attach_process_group(self, dist):
        self.dist = dist
        self.reducer = rdist.GradBucketReducer(dist, self.device)

    def _chunks(self):
        m = self.c3d.max_windows
        return [(w0, min(m, self.F - w0)) for w0 in range(0, self.F, m)]

    def forward(self, video, want_probs=True):
        """video [B*T,16,112,112,3] fp32 device tensor (mean-subtracted) -> (logits, probs) [B,T,49,49]."""
        assert tuple(video.shape) == (self.F, 16, 112, 112, 3), tuple(video.shape)
        for w0, n in self._chunks():
            self.c3d.forward(video[w0:w0 + n], want_features=False, want_rows=True, out_rows=self.rows[w0 * 49:(w0 + n) * 49])
        return self.head.forward_rows(self.rows, want_probs=want_probs)

    def backward(self, video, logits, probs, labels, finish=True):
        """Fills both engines' flat_grads; returns the loss (device scalar).  With a process group attached the
        gradient buckets are all-reduced on a side stream while the conv stack differentiates; finish=True (default)
        joins them before returning, so the gradients a caller reads are the reduced ones.  finish=False leaves the
        collectives in flight (the caller must call reducer.finish() before touching flat_grads)."""
        from .engine import l2_loss, softmax_xent
        red = getattr(self, 'reducer', None)
        if red is not None and red.pending:
            red.finish()                              # collectives of an earlier backward(finish=False) still own flat_grads
        labels = labels.reshape(self.B, self.T, 49, 49).contiguous()
        if self.loss_type == 'xentropy':
            loss = softmax_xent(logits, labels, want_probs=False)[2]
        else:
            loss = l2_loss(logits, labels, self.F)
        self.head.backward(logits, probs, labels, self.loss_type)
        if red is not None:
            red.reduce_buckets(self.head.grad_buckets())   # 12 MB in three buckets (the first leaves before the BPTT)
        self.head.backward_input(self.d_rows)
        _c3d_backward_chunks(self, video, self.d_rows, red)
        if red is not None and finish:
            red.finish()                              # every bucket reduced (mean) before anyone reads the gradients
        return loss

    def train_step(self, video, labels, lr, max_grad_norm=10.0):
        logits, probs = self.forward(video, want_probs=self.loss_type == 'xentropy')
        loss = self.backward(video, logits, probs, labels)       # (joins the all-reduces: the clip needs them all)
        gnorm = adam_clip_step_multi(self.engines, self.global_step, lr, max_grad_norm)
        self.global_step += 1
        return loss, gnorm

    def gradients(self):
        g = {'c3d/' + k: v for k, v in self.c3d.grad_views().items()}
        g.update({'head/' + k: v for k, v in self.head.grads.items()})
        return g


class EndToEndCascade(object):
    """BASELINE config 5 as a whole: video windows -> C3D -> two-level cascade (gaze_grcn_cascade.py) -> l2
    loss, trained jointly (conv stack + every non-ShallowNet variable of the cascade)."""

    def __init__(self, batch, n_steps, dtype='bf16', device='cuda:0', max_windows=None, seed=0, c3d_params=None,
                 cascade_params=None, image_hw=98):
        from .engine import CascadeEngine
        self.B, self.T, self.F = int(batch), int(n_steps), int(batch) * int(n_steps)
        self.device = torch.device(device)
        self.c3d = C3DEngine(min(self.F, max_windows or self.F), dtype=dtype, device=device, save_for_backward=True)
        self.head = CascadeEngine(self.B, self.T, image_hw, dtype=dtype, device=device, save_for_backward=True)
        self.c3d.set_weights(c3d_params if c3d_params is not None else synthetic.c3d_params(seed))
        self.head.set_weights(cascade_params if cascade_params is not None else synthetic.cascade_params(seed + 1, image_hw))
        self.feats = torch.empty(self.F, 1024, 7, 7, device=self.device)
        self.dist = None
        self.global_step = 0

    @property
    def engines(self):
        return [self.c3d, self.head]

    def attach_process_group(self, dist):
        self.dist = dist
        self.reducer = rdist.GradBucketReducer(dist, self.device)

    def _chunks(self):
        m = self.c3d.max_windows
        return [(w0, min(m, self.F - w0)) for w0 in range(0, self.F, m)]

    def forward(self, video, frames):
        """video [B*T,16,112,112,3] fp32 (mean-subtracted), frames [B,T,H,W,3] fp32 in [0,1] -> maps [B,T,49,49]."""
        for w0, n in self._chunks():
            self.feats[w0:w0 + n] = self.c3d.forward(video[w0:w0 + n])[0]
        return self.head.forward(frames, self.feats.reshape(self.B, self.T, 1024, 7, 7))

    def backward(self, video, maps, labels, finish=True):
        """As EndToEndGaze.backward: returns after the gradient all-reduces have been joined unless finish=False."""
        from .engine import l2_loss
        red = getattr(self, 'reducer', None)
        if red is not None and red.pending:
            red.finish()
        labels = labels.reshape(maps.shape).contiguous()
        loss = l2_loss(maps, labels, self.F)
        _, d_rows = self.head.backward(maps, labels, want_d_rows=True)
        if red is not None:
            red.reduce(self.head.flat_grads)          # 216 MB: the largest bucket starts first
        _c3d_backward_chunks(self, video, d_rows, red)
        if red is not None and finish:
            red.finish()
        return loss

    def train_step(self, video, frames, labels, lr, max_grad_norm=10.0):
        maps = self.forward(video, frames)
        loss = self.backward(video, maps, labels)
        gnorm = adam_clip_step_multi(self.engines, self.global_step, lr, max_grad_norm)
        self.global_step += 1
        return loss, gnorm


def flops_per_frame_train():
    """forward 77 426.06 MFLOP per frame (conv stack 76 993.27 + head 432.79); a trained layer
    costs 3x forward (fwd + dgrad + wgrad), conv1a has no dgrad."""
    conv = [2.08e9, 22.20e9, 11.10e9, 22.20e9, 5.55e9, 11.10e9, 1.39e9, 1.39e9]
    return 3 * sum(conv) - conv[0] + 3 * 432.79e6
