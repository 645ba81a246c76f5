"""Device-side engines: thin owners of a librgp_hip plan + its workspace.

PyTorch is plumbing only here (device memory, streams); all arithmetic happens in
the HIP kernels behind the C ABI (include/rgp.h).  These classes play the role of
the reference's ``tf.Session`` + graph for the gaze path (models/gaze_rnn.py:523-531).
"""
import ctypes

import numpy as np
import torch

from . import _lib

GRCN_PARAM_TO_FIELD = {   # reference TF variable name -> rgp_grcn_weights field
    'proj_c3d_W': 'proj_c3d_W', 'proj_c3d_b': 'proj_c3d_b',
    'GRU_Conv_Wz': 'gru_Wz', 'GRU_Conv_Uz': 'gru_Uz', 'GRU_Conv_Wr': 'gru_Wr', 'GRU_Conv_Ur': 'gru_Ur',
    'GRU_Conv_W': 'gru_W', 'GRU_Conv_U': 'gru_U', 'bn_gamma': 'bn_gamma', 'bn_beta': 'bn_beta',
    'weight1': 'up_weight1', 'weight2': 'up_weight2', 'weight3': 'up_weight3', 'out_W': 'out_W', 'out_b': 'out_b',
}
C3D_LAYER_NAMES = ('conv1a', 'conv2a', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'conv5a', 'conv5b')
C3D_EXTENTS = ((16, 112), (16, 56), (8, 28), (8, 28), (4, 14), (4, 14), (2, 7), (2, 7))     # conv output D, H (= W)
C3D_CHANNELS = ((3, 64), (64, 128), (128, 256), (256, 256), (256, 512), (512, 512), (512, 512), (512, 512))


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise _lib.RgpError('no HIP device visible: the gaze path runs only on the GPU (no CPU fallback)')
    return torch.device(device)


def _as_dev_f32(x, device):
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    return t.to(device=device, dtype=torch.float32).contiguous()


OPTIMIZERS = ('adam', 'rmsprop', 'sgd')      # create_train_op (base.py:268-273)


def _opt_slots(e, method):
    """Optimizer slot buffers of one engine, created on first use with TF's initial values."""
    if method == 'adam':
        if getattr(e, 'adam_m', None) is None:
            e.adam_m, e.adam_v = torch.zeros_like(e.flat_params), torch.zeros_like(e.flat_params)
    elif method == 'rmsprop':
        if getattr(e, 'rms_ms', None) is None:       # RMSPropOptimizer: rms slot starts at ones, momentum at zeros
            e.rms_ms, e.rms_mom = torch.ones_like(e.flat_params), torch.zeros_like(e.flat_params)
    elif method == 'sgd':
        if getattr(e, 'mom_accum', None) is None:
            e.mom_accum = torch.zeros_like(e.flat_params)
    else:
        raise ValueError('Invalid optimization method!')          # base.py:274


OPT_STATE_KEYS = ('adam_m', 'adam_v', 'rms_ms', 'rms_mom', 'mom_accum')


def optimizer_state(engine):
    """{slot name: cpu tensor} of the slots that exist (what tf.train.Saver(tf.all_variables()) also saves,
    base.py:236-251: a resumed run must not restart Adam's moments at zero)."""
    return {k: getattr(engine, k).detach().cpu().clone() for k in OPT_STATE_KEYS if getattr(engine, k, None) is not None}


def load_optimizer_state(engine, state):
    for k, v in (state or {}).items():
        if k in OPT_STATE_KEYS:
            setattr(engine, k, torch.as_tensor(v).to(engine.device, torch.float32).contiguous().clone())


def clip_step_multi(engines, step, lr, max_grad_norm=10.0, method='adam', beta1=0.9, beta2=0.999, eps=1e-8,
                    momentum=0.9, rms_decay=0.9, rms_eps=1e-10):
    """tf.clip_by_global_norm over the gradients of ALL engines (base.py:286-292) + the chosen optimizer
    (base.py:268-273; TF defaults) on each engine's flat buffers, then repack.
    engines: objects with flat_params / flat_grads.  Returns a 1-element device tensor with the pre-clip norm."""
    lib, dev = engines[0].lib, engines[0].device
    npart = _lib.RGP_SQNORM_PARTIALS
    partials = torch.zeros(npart * len(engines), dtype=torch.float32, device=dev)
    gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        for i, e in enumerate(engines):
            _lib.check(lib.rgp_global_sqnorm(_ptr(e.flat_grads), e.flat_grads.numel(), _ptr(partials[i * npart:]),
                                             _stream_ptr(dev)))
        for e in engines:
            _opt_slots(e, method)
            n = e.flat_params.numel()
            if method == 'adam':
                _lib.check(lib.rgp_adam_clip_step_ext(_ptr(e.flat_params), _ptr(e.flat_grads), _ptr(e.adam_m), _ptr(e.adam_v),
                                                      n, _ptr(partials), partials.numel(), int(step), float(lr), beta1,
                                                      beta2, eps, float(max_grad_norm), _ptr(gnorm), _stream_ptr(dev)))
            elif method == 'rmsprop':
                _lib.check(lib.rgp_rmsprop_clip_step(_ptr(e.flat_params), _ptr(e.flat_grads), _ptr(e.rms_ms), _ptr(e.rms_mom),
                                                     n, _ptr(partials), partials.numel(), float(lr), rms_decay, momentum,
                                                     rms_eps, float(max_grad_norm), _ptr(gnorm), _stream_ptr(dev)))
            else:
                _lib.check(lib.rgp_momentum_clip_step(_ptr(e.flat_params), _ptr(e.flat_grads), _ptr(e.mom_accum), n,
                                                      _ptr(partials), partials.numel(), float(lr), momentum,
                                                      float(max_grad_norm), _ptr(gnorm), _stream_ptr(dev)))
            e.repack()
    return gnorm


def adam_clip_step_multi(engines, step, lr, max_grad_norm=10.0, beta1=0.9, beta2=0.999, eps=1e-8):
    return clip_step_multi(engines, step, lr, max_grad_norm, 'adam', beta1, beta2, eps)


def l2_loss(maps, labels, frames):
    """sum 0.5 (maps - labels)^2 / frames (gaze_rnn.py:387-389) -> 1-element device tensor."""
    lib = _lib.load()
    assert maps.is_cuda and maps.dtype == torch.float32 and maps.is_contiguous()
    assert labels.is_cuda and labels.dtype == torch.float32 and labels.is_contiguous() and labels.numel() == maps.numel()
    ws = torch.empty(_lib.RGP_SQNORM_PARTIALS, dtype=torch.float32, device=maps.device)
    loss = torch.empty(1, dtype=torch.float32, device=maps.device)
    with torch.cuda.device(maps.device):
        _lib.check(lib.rgp_l2_loss_fwd(_ptr(maps), _ptr(labels), maps.numel(), int(frames), _ptr(ws), _ptr(loss),
                                       _stream_ptr(maps.device)))
    return loss


class DropoutSite(object):
    """Keep-mask owner of one tf.nn.dropout site: draws a fresh Philox mask per training step on the device
    (rgp_dropout_mask), or takes the caller's bytes (parity tests feed the oracle the same mask)."""

    def __init__(self, lib, device, n_elems, setter):
        self.lib, self.device, self.n, self._set = lib, device, int(n_elems), setter
        self.mask = torch.ones(self.n, dtype=torch.uint8, device=device)
        self.keep_prob, self.seed, self.draws = 1.0, 0, 0

    def configure(self, keep_prob, seed=0):
        self.keep_prob, self.seed, self.draws = float(keep_prob), int(seed), 0

    def off(self):
        _lib.check(self._set(ctypes.c_float(1.0), None))

    def draw(self):
        """New mask for this step (counter offset advances by ceil(n/4) blocks per draw)."""
        if self.keep_prob >= 1.0:
            return self.off()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_dropout_mask(_ptr(self.mask), self.n, self.keep_prob, self.seed,
                                                 self.draws * ((self.n + 3) // 4), _stream_ptr(self.device)))
        self.draws += 1
        _lib.check(self._set(ctypes.c_float(self.keep_prob), _ptr(self.mask)))

    def use(self, mask, keep_prob):
        """Fixed mask (uint8 / bool array of n elements, reference order)."""
        m = torch.as_tensor(np.asarray(mask).astype(np.uint8) if not torch.is_tensor(mask) else mask)
        assert m.numel() == self.n, (m.numel(), self.n)
        self.mask.copy_(m.reshape(-1).to(self.device, torch.uint8))
        self.keep_prob = float(keep_prob)
        _lib.check(self._set(ctypes.c_float(self.keep_prob), _ptr(self.mask)))


class GrcnEngine(object):
    """gaze_grcn graph (models/gaze_grcn.py:173-376) at fixed (B, T, P, S, dtype)."""

    def __init__(self, batch, n_steps, dim_proj=512, dim_state=128, dtype='bf16', save_for_backward=False,
                 device='cuda:0', per_step=False, unfolded_head=False):
        """per_step=True: the recurrence (and its BPTT) as per-timestep launches even where the persistent
        kernels apply (RGP_GRCN_PER_STEP): the library's second implementation, used for cross-checks.
        unfolded_head=True: an inference plan that runs the three transposed convolutions one by one
        (RGP_GRCN_UNFOLDED_HEAD) instead of their exact fold into one GEMM (csrc/head_fold.hip.h); training plans
        always do."""
        self.lib = _lib.load()
        self.device = _require_gpu(device)
        self.B, self.T, self.P, self.S = int(batch), int(n_steps), int(dim_proj), int(dim_state)
        self.dtype = dtype
        self.per_step, self.save_for_backward = bool(per_step), bool(save_for_backward)
        self.torch_dtype = torch.bfloat16 if _lib.DTYPES[dtype] == _lib.RGP_BF16 else torch.float32
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            flags = (_lib.RGP_GRCN_SAVE_FOR_BACKWARD if save_for_backward else 0) | (_lib.RGP_GRCN_PER_STEP if per_step else 0) | \
                (_lib.RGP_GRCN_UNFOLDED_HEAD if unfolded_head else 0)
            _lib.check(self.lib.rgp_grcn_create(ctypes.byref(self._h), self.B, self.T, self.P, self.S,
                                                _lib.DTYPES[dtype], flags))
            nbytes = self.lib.rgp_grcn_workspace_bytes(self._h)
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.rgp_grcn_bind_workspace(self._h, _ptr(self.workspace), nbytes, _stream_ptr(self.device)))
        self.weights = None

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self.lib.rgp_grcn_destroy(h)

    def _flat_views(self, like):
        """One flat fp32 buffer + per-variable views (one all-reduce bucket, one Adam launch)."""
        sizes = [(k, tuple(like[k].shape)) for k in GRCN_PARAM_TO_FIELD]
        flat = torch.zeros(sum(int(np.prod(s)) for _, s in sizes), dtype=torch.float32, device=self.device)
        views, off = {}, 0
        for k, shp in sizes:
            n = int(np.prod(shp))
            views[k] = flat[off:off + n].view(shp)
            off += n
        return flat, views

    def _struct(self, views):
        st = _lib.GrcnWeights()
        for k, f in GRCN_PARAM_TO_FIELD.items():
            setattr(st, f, views[k].data_ptr())
        return st

    def set_weights(self, params):
        """params: dict keyed by the reference's TF variable names -> array/tensor (fp32).
        The values are copied into the engine's flat fp32 master buffer."""
        src = {k: _as_dev_f32(params[k], self.device) for k in GRCN_PARAM_TO_FIELD}
        assert tuple(src['bn_gamma'].shape) == (self.T, self.S), 'one batch-norm layer per timestep (SURVEY 9-Q1)'
        if self.weights is None:
            self.flat_params, self.weights = self._flat_views(src)
        for k in GRCN_PARAM_TO_FIELD:
            self.weights[k].copy_(src[k])
        self.repack()

    def repack(self):
        """Re-pack the master weights into MFMA operand form (after set_weights / an optimizer step)."""
        st = self._struct(self.weights)     # the plan keeps raw pointers to biases / BN: views stay alive
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_grcn_set_weights(self._h, ctypes.byref(st), _stream_ptr(self.device)))

    def backward(self, logits, probs, labels, loss_type='xentropy'):
        """Gradients of the reference loss (gaze_rnn.py:363-408) w.r.t. every variable, after a
        forward() on the same inputs.  labels: normalised gt maps [B,T,49,49] fp32 device tensor.
        Returns {TF variable name: fp32 gradient view}; the flat buffer is self.flat_grads."""
        assert labels.is_cuda and labels.dtype == torch.float32 and labels.is_contiguous()
        if getattr(self, 'grads', None) is None:
            self.flat_grads, self.grads = self._flat_views(self.weights)
        st = self._struct(self.grads)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_grcn_backward(self._h, _ptr(logits), _ptr(probs), _ptr(labels), ctypes.byref(st),
                                                  {'xentropy': 0, 'l2': 1}[loss_type], _stream_ptr(self.device)))
        return self.grads

    # gradient groups in the order the backward finishes them (rgp_grcn_wait_grads); each is one contiguous slice of
    # flat_grads because GRCN_PARAM_TO_FIELD lists the variables in this order
    GRAD_GROUPS = ((_lib.RGP_GRCN_GRADS_TOP, 'bn_gamma', 'out_b'),
                   (_lib.RGP_GRCN_GRADS_GRU, 'GRU_Conv_Wz', 'GRU_Conv_U'),
                   (_lib.RGP_GRCN_GRADS_PROJ, 'proj_c3d_W', 'proj_c3d_b'))

    def grad_buckets(self):
        """[(slice of flat_grads, ready)] in completion order: `ready(stream)` makes a torch.cuda.Stream wait until the
        last backward() has finished that slice (data-parallel training: the all-reduce of the upsampling / output
        gradients runs under the BPTT, that of the ConvGRU filters under the projection's filter gradient)."""
        names = list(GRCN_PARAM_TO_FIELD)
        offs, off = {}, 0
        for k in names:
            offs[k] = (off, off + self.grads[k].numel())
            off += self.grads[k].numel()
        out = []
        for group, first, last in self.GRAD_GROUPS:
            assert names.index(first) <= names.index(last)
            lo, hi = offs[first][0], offs[last][1]
            out.append((self.flat_grads[lo:hi],
                        lambda stream, group=group: _lib.check(self.lib.rgp_grcn_wait_grads(
                            self._h, group, ctypes.c_void_p(stream.cuda_stream)))))
        assert sum(b.numel() for b, _ in out) == self.flat_grads.numel()
        return out

    def backward_input(self, out=None):
        """After backward(): d loss / d input as conv5b rows [B*T*49, 1024] fp32 (column d*512+c), the
        gradient C3DEngine.backward(d_rows=...) consumes when the conv stack is fine-tuned."""
        d = out if out is not None else torch.empty(self.B * self.T * 49, 1024, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_grcn_backward_input(self._h, _ptr(d), _stream_ptr(self.device)))
        return d

    def adam_step(self, step, lr, max_grad_norm=10.0, beta1=0.9, beta2=0.999, eps=1e-8, method='adam'):
        """clip_by_global_norm + TF AdamOptimizer on the flat buffers (base.py:286-297), then repack.
        Returns a 1-element device tensor holding the pre-clip global gradient norm.
        method 'rmsprop' / 'sgd': the other two optimizers of base.py:268-273."""
        if method != 'adam':
            return clip_step_multi([self], step, lr, max_grad_norm, method)
        if getattr(self, 'adam_m', None) is None or getattr(self, '_opt_ws', None) is None:
            if getattr(self, 'adam_m', None) is None:       # (slots restored from a checkpoint are kept)
                self.adam_m = torch.zeros_like(self.flat_params)
                self.adam_v = torch.zeros_like(self.flat_params)
            self._opt_ws = torch.zeros(256, dtype=torch.float32, device=self.device)
            self._gnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_adam_clip_step(_ptr(self.flat_params), _ptr(self.flat_grads), _ptr(self.adam_m),
                                                   _ptr(self.adam_v), self.flat_params.numel(), _ptr(self._opt_ws),
                                                   int(step), float(lr), beta1, beta2, eps, float(max_grad_norm),
                                                   _ptr(self._gnorm), _stream_ptr(self.device)))
        self.repack()
        return self._gnorm

    def forward(self, c3d_input, want_probs=True, out_logits=None, out_probs=None):
        """c3d_input [B,T,1024,7,7] fp32 device tensor -> (logits, probs) [B,T,49,49]."""
        x = c3d_input
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert tuple(x.shape) == (self.B, self.T, 1024, 7, 7), tuple(x.shape)
        logits = out_logits if out_logits is not None else torch.empty(self.B, self.T, 49, 49, device=self.device)
        probs = None
        if want_probs:
            probs = out_probs if out_probs is not None else torch.empty_like(logits)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_grcn_forward(self._h, _ptr(x), _ptr(logits), _ptr(probs), _stream_ptr(self.device)))
        return logits, probs

    def forward_rows(self, rows, want_probs=True, out_logits=None, out_probs=None):
        """rows: conv5b rows from C3DEngine.forward (operand dtype, [B*T*49, 1024])."""
        assert rows.is_cuda and rows.dtype == self.torch_dtype and rows.numel() == self.B * self.T * 49 * 1024
        logits = out_logits if out_logits is not None else torch.empty(self.B, self.T, 49, 49, device=self.device)
        probs = None
        if want_probs:
            probs = out_probs if out_probs is not None else torch.empty_like(logits)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_grcn_forward_rows(self._h, _ptr(rows), _ptr(logits), _ptr(probs),
                                                      _stream_ptr(self.device)))
        return logits, probs

    def status(self):
        """Wait for the current stream and raise RgpError (RGP_ETIMEOUT) if a persistent ConvGRU launch of this
        plan lost a group member since the last check (its outputs are NaN)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_grcn_status(self._h, _stream_ptr(self.device)))

    @property
    def persistent_workgroups(self):
        """CUs a persistent ConvGRU / BPTT launch of this plan occupies (0: the recurrence runs as per-timestep launches)."""
        with torch.cuda.device(self.device):
            return int(self.lib.rgp_grcn_persistent_workgroups(self._h))

    @property
    def persistent(self):
        return self.persistent_workgroups > 0

    @property
    def grads_top_early(self):
        """Whether the first gradient bucket (grad_buckets()[0]) is released before the BPTT launch (include/rgp.h: only
        when that launch leaves RGP_RCCL_CU_RESERVE CUs to the collective) or behind it."""
        with torch.cuda.device(self.device):
            return bool(self.lib.rgp_grcn_grads_top_early(self._h))

    def inject_fault(self, kind):
        """Test hook (rgp_grcn_inject_fault): kind 'seq' / 'bptt' -- the next persistent launch loses a member."""
        _lib.check(self.lib.rgp_grcn_inject_fault(self._h, {'seq': _lib.RGP_FAULT_SEQ_LOST_MEMBER,
                                                            'bptt': _lib.RGP_FAULT_BPTT_LOST_MEMBER}[kind]))

    def read_buffer_elems(self, name):
        """fp32 elements read_buffer(name) returns; 0 = this plan has no such intermediate."""
        return int(self.lib.rgp_grcn_buffer_elems(self._h, name.encode()))

    def read_buffer(self, name):
        n = self.lib.rgp_grcn_buffer_elems(self._h, name.encode())
        if n == 0:
            raise _lib.RgpError('unknown intermediate %r' % name)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_grcn_read_buffer(self._h, name.encode(), _ptr(out), _stream_ptr(self.device)))
        return out

    def profile(self, enable=True):
        _lib.check(self.lib.rgp_grcn_profile_enable(self._h, int(enable)))

    def profile_read(self):
        """{stage: (total_ms, launch_groups)} since the last read (HIP events on the launch stream)."""
        n = len(_lib.GRCN_STAGES)
        ms, calls = (ctypes.c_double * n)(), (ctypes.c_longlong * n)()
        _lib.check(self.lib.rgp_grcn_profile_read(self._h, ms, calls))
        return {k: (ms[i], calls[i]) for i, k in enumerate(_lib.GRCN_STAGES)}


def softmax_xent(logits, labels=None, want_probs=True):
    """Per-frame softmax / cross entropy (model_util.py:61-72; gaze_rnn.py:390-407).
    logits [..., H, W] fp32 device tensor -> (probs, frame_loss, loss)."""
    lib = _lib.load()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous()
    npix = logits.shape[-1] * logits.shape[-2]
    frames = logits.numel() // npix
    probs = torch.empty_like(logits) if want_probs else None
    frame_loss = loss = None
    if labels is not None:
        assert labels.shape == logits.shape and labels.dtype == torch.float32 and labels.is_contiguous()
        frame_loss = torch.empty(frames, device=logits.device)
        loss = torch.empty(1, device=logits.device)
    with torch.cuda.device(logits.device):
        _lib.check(lib.rgp_softmax_xent_fwd(_ptr(logits), _ptr(labels), _ptr(probs), _ptr(frame_loss), _ptr(loss),
                                            frames, npix, _stream_ptr(logits.device)))
    return probs, frame_loss, loss


class FcGruEngine(object):
    """fc-GRU gaze model (models/gaze_rnn.py:211-360, BASELINE config 2) at fixed (B, T, GH, GW)."""

    def __init__(self, batch, n_steps, gazemap_hw=(49, 49), dtype='f32', device='cuda:0', save_for_backward=False):
        self.lib = _lib.load()
        self.device = _require_gpu(device)
        self.B, self.T, self.GH, self.GW = int(batch), int(n_steps), int(gazemap_hw[0]), int(gazemap_hw[1])
        self.dtype = dtype
        self.save_for_backward = bool(save_for_backward)
        self.flat_params = self.flat_grads = self.grads = self.adam_m = self.adam_v = None
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_fcgru_create_ex(ctypes.byref(self._h), self.B, self.T, self.GH, self.GW,
                                                    _lib.DTYPES[dtype], int(self.save_for_backward)))
            nbytes = self.lib.rgp_fcgru_workspace_bytes(self._h)
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.rgp_fcgru_bind_workspace(self._h, _ptr(self.workspace), nbytes, _stream_ptr(self.device)))
        self.weights = None
        # tf.nn.dropout on c3d_embedded [B*T*49, 32] (gaze_rnn.py:302-303): off until a training step draws a mask
        self.dropout = DropoutSite(self.lib, self.device, self.B * self.T * 49 * 32,
                                   lambda keep, mask: self.lib.rgp_fcgru_set_dropout(self._h, keep, mask))

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self.lib.rgp_fcgru_destroy(h)

    def _flat(self, like):
        sizes = [(k, tuple(like[k].shape)) for k in _lib.FcGruWeights.FIELDS]
        flat = torch.zeros(sum(int(np.prod(s)) for _, s in sizes), dtype=torch.float32, device=self.device)
        views, off = {}, 0
        for k, shp in sizes:
            n = int(np.prod(shp))
            views[k] = flat[off:off + n].view(shp)
            off += n
        return flat, views

    def _struct(self, views):
        st = _lib.FcGruWeights()
        for k in _lib.FcGruWeights.FIELDS:
            setattr(st, k, views[k].data_ptr())
        return st

    def set_weights(self, params):
        src = {k: _as_dev_f32(params[k], self.device) for k in _lib.FcGruWeights.FIELDS}
        if self.flat_params is None:
            self.flat_params, self.weights = self._flat(src)
        for k in _lib.FcGruWeights.FIELDS:
            self.weights[k].copy_(src[k])
        self.repack()

    def repack(self):
        st = self._struct(self.weights)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_fcgru_set_weights(self._h, ctypes.byref(st), _stream_ptr(self.device)))

    def backward(self, logits, probs, labels, loss_type='xentropy'):
        """Gradients of the loss of gaze_rnn.py:363-408 w.r.t. the 8 variables after forward() on the same
        inputs; returns {field: gradient view}, the flat buffer is flat_grads."""
        assert self.save_for_backward, 'create the engine with save_for_backward=True'
        assert labels.is_cuda and labels.dtype == torch.float32 and labels.is_contiguous()
        if self.flat_grads is None:
            self.flat_grads, self.grads = self._flat(self.weights)
        st = self._struct(self.grads)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_fcgru_backward(self._h, _ptr(logits), _ptr(probs), _ptr(labels), ctypes.byref(st),
                                                   {'xentropy': 0, 'l2': 1}[loss_type], _stream_ptr(self.device)))
        return self.grads

    def adam_step(self, step, lr, max_grad_norm=10.0, method='adam'):
        return clip_step_multi([self], step, lr, max_grad_norm, method)

    def forward(self, c3d_input, want_probs=True, train=False):
        """train=True draws a fresh dropout mask when dropout.configure(keep < 1) was called (single_step feeds
        keep 0.5 in training, gaze_rnn.py:529); train=False runs with the site off, as every evaluation does."""
        x = c3d_input
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert tuple(x.shape) == (self.B, self.T, 1024, 7, 7), tuple(x.shape)
        if train == 'keep':
            pass                                   # a mask installed by dropout.use() stays as it is
        elif train:
            self.dropout.draw()
        else:
            self.dropout.off()
        logits = torch.empty(self.B, self.T, self.GH, self.GW, device=self.device)
        probs = torch.empty_like(logits) if want_probs else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_fcgru_forward(self._h, _ptr(x), _ptr(logits), _ptr(probs), _stream_ptr(self.device)))
        return logits, probs


class ShallowNetEngine(object):
    """Frame-wise ShallowNet (models/saliency_shallownet.py:74-216, BASELINE config 1)."""

    def __init__(self, max_frames, image_hw=98, dtype='f32', device='cuda:0', save_for_backward=False):
        self.lib = _lib.load()
        self.device = _require_gpu(device)
        self.max_frames, self.image_hw, self.dtype = int(max_frames), int(image_hw), dtype
        self.save_for_backward = bool(save_for_backward)
        self.flat_params = self.flat_grads = self.grads = self.adam_m = self.adam_v = None
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_shallownet_create_ex(ctypes.byref(self._h), self.max_frames, self.image_hw,
                                                         _lib.DTYPES[dtype], int(self.save_for_backward)))
            nbytes = self.lib.rgp_shallownet_workspace_bytes(self._h)
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.rgp_shallownet_bind_workspace(self._h, _ptr(self.workspace), nbytes,
                                                              _stream_ptr(self.device)))
        self.weights = None

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self.lib.rgp_shallownet_destroy(h)

    def _flat(self, like):
        sizes = [(k, tuple(like[k].shape)) for k in _lib.ShallowNetWeights.FIELDS]
        flat = torch.zeros(sum(int(np.prod(s)) for _, s in sizes), dtype=torch.float32, device=self.device)
        views, off = {}, 0
        for k, shp in sizes:
            n = int(np.prod(shp))
            views[k] = flat[off:off + n].view(shp)
            off += n
        return flat, views

    def _struct(self, views):
        st = _lib.ShallowNetWeights()
        for k in _lib.ShallowNetWeights.FIELDS:
            setattr(st, k, views[k].data_ptr())
        return st

    def set_weights(self, params):
        src = {k: _as_dev_f32(params[k], self.device) for k in _lib.ShallowNetWeights.FIELDS}
        if self.flat_params is None:
            self.flat_params, self.weights = self._flat(src)
        for k in _lib.ShallowNetWeights.FIELDS:
            self.weights[k].copy_(src[k])
        self.repack()

    def repack(self):
        st = self._struct(self.weights)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_shallownet_set_weights(self._h, ctypes.byref(st), _stream_ptr(self.device)))

    def backward(self, d_saliency):
        """d loss / d saliency [n,49,49] fp32 (frames of the last forward) -> {field: gradient view} for all ten
        variables (FramewiseShallowNet trains them all, gaze_framewise_shallownet.py:43-57)."""
        assert self.save_for_backward, 'create the engine with save_for_backward=True'
        d = d_saliency
        assert d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and tuple(d.shape[1:]) == (49, 49)
        if self.flat_grads is None:
            self.flat_grads, self.grads = self._flat(self.weights)
        st = self._struct(self.grads)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_shallownet_backward(self._h, d.shape[0], _ptr(d), ctypes.byref(st), _stream_ptr(self.device)))
        return self.grads

    def adam_step(self, step, lr, max_grad_norm=10.0, method='adam'):
        return clip_step_multi([self], step, lr, max_grad_norm, method)

    def forward(self, frames, want_7x7=False):
        """frames [n,H,W,3] fp32 device tensor -> (saliency [n,49,49], saliency7 [n,7,7] or None)."""
        assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous()
        n = frames.shape[0]
        assert tuple(frames.shape[1:]) == (self.image_hw, self.image_hw, 3) and n <= self.max_frames
        sal = torch.empty(n, 49, 49, device=self.device)
        sal7 = torch.empty(n, 7, 7, device=self.device) if want_7x7 else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_shallownet_forward(self._h, _ptr(frames), n, _ptr(sal), _ptr(sal7),
                                                       _stream_ptr(self.device)))
        return sal, sal7


class CascadeEngine(object):
    """Two-level cascade (models/gaze_grcn_cascade.py:188-423, BASELINE config 5), forward."""

    # rgp_cascade_weights field -> key of the parameter dictionary (TF variable names)
    KEYS = (('proj_c3d_W', 'proj_c3d_W'), ('proj_c3d_b', 'proj_c3d_b'),
            ('bottom_Wz', 'RCNBottom/GRU_Conv_Wz'), ('bottom_Uz', 'RCNBottom/GRU_Conv_Uz'),
            ('bottom_Wr', 'RCNBottom/GRU_Conv_Wr'), ('bottom_Ur', 'RCNBottom/GRU_Conv_Ur'),
            ('bottom_W', 'RCNBottom/GRU_Conv_W'), ('bottom_U', 'RCNBottom/GRU_Conv_U'),
            ('upsampling_weight', 'Upsampling/weight'),
            ('top_Wz', 'RCNGaze/GRU_Conv_Wz'), ('top_Uz', 'RCNGaze/GRU_Conv_Uz'),
            ('top_Wr', 'RCNGaze/GRU_Conv_Wr'), ('top_Ur', 'RCNGaze/GRU_Conv_Ur'),
            ('top_W', 'RCNGaze/GRU_Conv_W'), ('top_U', 'RCNGaze/GRU_Conv_U'),
            ('fc1_w', 'LastProjection/fc1_w'), ('fc1_b', 'LastProjection/fc1_b'),
            ('fc2_w', 'LastProjection/fc2_w'), ('fc2_b', 'LastProjection/fc2_b'))

    def __init__(self, batch, n_steps, image_hw=98, dtype='bf16', device='cuda:0', save_for_backward=False):
        self.lib = _lib.load()
        self.device = _require_gpu(device)
        self.batch, self.n_steps, self.image_hw, self.dtype = int(batch), int(n_steps), int(image_hw), dtype
        self.save_for_backward = bool(save_for_backward)
        self.flat_params = self.flat_grads = self.grads = self.adam_m = self.adam_v = None
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_cascade_create_ex(ctypes.byref(self._h), self.batch, self.n_steps, self.image_hw,
                                                      _lib.DTYPES[dtype], int(self.save_for_backward)))
            nbytes = self.lib.rgp_cascade_workspace_bytes(self._h)
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.rgp_cascade_bind_workspace(self._h, _ptr(self.workspace), nbytes,
                                                           _stream_ptr(self.device)))
        self.weights = None
        # tf.nn.dropout on relu(fc1) [B*T, 4802] before the maxout (gaze_grcn_cascade.py:401-402)
        self.dropout = DropoutSite(self.lib, self.device, self.batch * self.n_steps * 4802,
                                   lambda keep, mask: self.lib.rgp_cascade_set_dropout(self._h, keep, mask))

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self.lib.rgp_cascade_destroy(h)

    def _flat(self, shapes):
        flat = torch.zeros(sum(int(np.prod(s)) for _, s in shapes), dtype=torch.float32, device=self.device)
        views, off = {}, 0
        for f, shp in shapes:
            n = int(np.prod(shp))
            views[f] = flat[off:off + n].view(shp)
            off += n
        return flat, views

    def _struct(self, views):
        st = _lib.CascadeWeights()
        for field, _ in self.KEYS:
            setattr(st, field, views[field].data_ptr())
        for k in _lib.ShallowNetWeights.FIELDS:                 # frozen (learning rate 0, base.py:264-265)
            setattr(st.shallownet, k, self.weights['shallownet.' + k].data_ptr())
        return st

    def set_weights(self, params):
        """The 19 trainable arrays live in one flat fp32 master buffer (one all-reduce bucket, one Adam launch);
        the ShallowNet's arrays are separate and frozen."""
        src = {field: _as_dev_f32(params[key], self.device) for field, key in self.KEYS}
        if self.flat_params is None:
            self.flat_params, self.weights = self._flat([(f, tuple(src[f].shape)) for f, _ in self.KEYS])
        for f, _ in self.KEYS:
            self.weights[f].copy_(src[f])
        for k in _lib.ShallowNetWeights.FIELDS:
            self.weights['shallownet.' + k] = _as_dev_f32(params['ShallowNet'][k], self.device)
        self.repack()

    def repack(self):
        st = self._struct(self.weights)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_cascade_set_weights(self._h, ctypes.byref(st), _stream_ptr(self.device)))

    def backward(self, maps, gt, want_d_rows=False):
        """Gradients of the l2 loss (gaze_grcn_cascade.py:428-441) after forward() on the same inputs.
        Returns ({rgp_cascade_weights field: gradient view}, d_rows or None); the flat buffer is flat_grads."""
        assert self.save_for_backward, 'create the engine with save_for_backward=True'
        assert gt.is_cuda and gt.dtype == torch.float32 and gt.is_contiguous() and gt.numel() == maps.numel()
        if self.flat_grads is None:
            self.flat_grads, self.grads = self._flat([(f, tuple(self.weights[f].shape)) for f, _ in self.KEYS])
        st = self._struct(self.grads)
        d_rows = torch.empty(self.batch * self.n_steps * 49, 1024, device=self.device) if want_d_rows else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_cascade_backward(self._h, _ptr(maps), _ptr(gt), ctypes.byref(st), _ptr(d_rows),
                                                     _stream_ptr(self.device)))
        return self.grads, d_rows

    def forward(self, frame_images, c3d_input, train=False):
        """frame_images [B,T,H,W,3], c3d_input [B,T,1024,7,7] fp32 device tensors -> maps [B,T,49,49].
        train: True = draw a dropout mask (if configured), 'keep' = leave an installed mask, False = site off."""
        B, T = self.batch, self.n_steps
        if train == 'keep':
            pass
        elif train:
            self.dropout.draw()
        else:
            self.dropout.off()
        assert frame_images.is_cuda and frame_images.dtype == torch.float32 and frame_images.is_contiguous()
        assert c3d_input.is_cuda and c3d_input.dtype == torch.float32 and c3d_input.is_contiguous()
        assert tuple(frame_images.shape) == (B, T, self.image_hw, self.image_hw, 3), tuple(frame_images.shape)
        assert tuple(c3d_input.shape) == (B, T, 1024, 7, 7), tuple(c3d_input.shape)
        maps = torch.empty(B, T, 49, 49, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_cascade_forward(self._h, _ptr(frame_images), _ptr(c3d_input), _ptr(maps),
                                                    _stream_ptr(self.device)))
        return maps

    BUFFERS = {'frm_sal': lambda B, T: (B, T, 49, 49), 'rcn_outputs': lambda B, T: (B, T, 7, 7, 256),
               'rcn_upsampled_outputs': lambda B, T: (B, T, 49, 49, 64), 'gaze_rcn_outputs': lambda B, T: (B, T, 49, 49, 3)}

    def read_buffer(self, name):
        out = torch.empty(self.BUFFERS[name](self.batch, self.n_steps), device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_cascade_read_buffer(self._h, name.encode(), _ptr(out), _stream_ptr(self.device)))
        return out


class C3DEngine(object):
    """C3D conv1a..conv5b (prototxt:22-342) for up to max_windows windows per launch chain."""

    KERNELS = {'patch': 0, 'igemm': _lib.RGP_C3D_KERNELS_IGEMM,
               'igemm128': _lib.RGP_C3D_KERNELS_IGEMM | _lib.RGP_C3D_KERNELS_TILE128,
               'patch-rowwise': _lib.RGP_C3D_CONV2A_ROWWISE}

    def __init__(self, max_windows, dtype='bf16', device='cuda:0', save_for_backward=False, kernels='patch'):
        """kernels: 'patch' (default: the layer-specific kernels for conv2a..conv4b), 'igemm' (the general
        implicit-GEMM / filter-gradient kernels for every layer, tile by problem size) or 'igemm128' (the same on
        the 128x128 tile loop only), 'patch-rowwise' (the patch kernels with conv2a's inference forward on the row-wise
        fetch instead of the plane-slab one: RGP_C3D_CONV2A_ROWWISE) -- rgp_c3d_create_ex flags; the alternatives exist
        for cross-checks."""
        self.lib = _lib.load()
        self.device = _require_gpu(device)
        self.max_windows = int(max_windows)
        self.dtype = dtype
        self.kernels = kernels
        self.save_for_backward = bool(save_for_backward)
        self.torch_dtype = torch.bfloat16 if _lib.DTYPES[dtype] == _lib.RGP_BF16 else torch.float32
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_c3d_create_ex(ctypes.byref(self._h), self.max_windows, _lib.DTYPES[dtype],
                                                  int(self.save_for_backward) | self.KERNELS[kernels]))
            nbytes = self.lib.rgp_c3d_workspace_bytes(self._h)
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.rgp_c3d_bind_workspace(self._h, _ptr(self.workspace), nbytes, _stream_ptr(self.device)))
        # fp32 master parameters in one flat vector, layout w[0], b[0], w[1], b[1], ... (= rgp_c3d_backward's grads)
        n = self.lib.rgp_c3d_param_elems(self._h)
        self.flat_params = torch.zeros(n, device=self.device)
        self.flat_grads = torch.zeros(n, device=self.device) if self.save_for_backward else None
        self.adam_m = self.adam_v = None
        self.weights = {}
        for i, name in enumerate(C3D_LAYER_NAMES):
            cin, cout = C3D_CHANNELS[i]
            ow, ob = self.lib.rgp_c3d_param_offset(self._h, i, 0), self.lib.rgp_c3d_param_offset(self._h, i, 1)
            self.weights[name + '_w'] = self.flat_params[ow:ow + 27 * cin * cout].view(3, 3, 3, cin, cout)
            self.weights[name + '_b'] = self.flat_params[ob:ob + cout]
        self.weights_set = False

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self.lib.rgp_c3d_destroy(h)

    def grad_views(self):
        """{name: view into flat_grads} with the shapes of the parameters."""
        out = {}
        for i, name in enumerate(C3D_LAYER_NAMES):
            cin, cout = C3D_CHANNELS[i]
            ow, ob = self.lib.rgp_c3d_param_offset(self._h, i, 0), self.lib.rgp_c3d_param_offset(self._h, i, 1)
            out[name + '_w'] = self.flat_grads[ow:ow + 27 * cin * cout].view(3, 3, 3, cin, cout)
            out[name + '_b'] = self.flat_grads[ob:ob + cout]
        return out

    def set_weights(self, params):
        """params: {'conv1a_w': [3,3,3,Cin,Cout], 'conv1a_b': [Cout], ...} fp32 -> master copy + packed operands."""
        for name in C3D_LAYER_NAMES:
            for suffix in ('_w', '_b'):
                self.weights[name + suffix].copy_(_as_dev_f32(params[name + suffix], self.device).reshape(
                    self.weights[name + suffix].shape))
        self.repack()

    def repack(self):
        """Re-derive the packed (and, for training, rotated) operand filters from the master parameters."""
        st = _lib.C3DWeights()
        for i, name in enumerate(C3D_LAYER_NAMES):
            st.w[i] = self.weights[name + '_w'].data_ptr()
            st.b[i] = self.weights[name + '_b'].data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_c3d_set_weights(self._h, ctypes.byref(st), _stream_ptr(self.device)))
        self.weights_set = True

    def backward(self, d_features=None, d_rows=None, zero_grads=True):
        """Gradient of the last forward (<= max_windows windows) w.r.t. every conv filter and bias, given the
        gradient w.r.t. conv5b as d_features [n,1024,7,7] or d_rows [n*49,1024] (fp32); fills flat_grads."""
        assert self.save_for_backward, 'create the engine with save_for_backward=True'
        g = d_features if d_features is not None else d_rows
        assert (d_features is None) != (d_rows is None) and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()
        n = g.shape[0] if d_features is not None else g.shape[0] // 49
        if zero_grads:
            self.flat_grads.zero_()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_c3d_backward(self._h, _ptr(d_features), _ptr(d_rows), n, _ptr(self.flat_grads),
                                                 _stream_ptr(self.device)))
        return self.flat_grads

    def forward(self, video, want_features=True, want_rows=False, out_rows=None):
        """video [n,16,112,112,3] fp32 device tensor -> (features [n,1024,7,7] fp32, rows)."""
        assert video.is_cuda and video.dtype == torch.float32 and video.is_contiguous()
        assert tuple(video.shape[1:]) == (16, 112, 112, 3), tuple(video.shape)
        n = video.shape[0]
        feats = torch.empty(n, 1024, 7, 7, device=self.device) if want_features else None
        rows = out_rows
        if want_rows and rows is None:
            rows = torch.empty(n * 49, 1024, dtype=self.torch_dtype, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_c3d_forward(self._h, _ptr(video), n, _ptr(feats), _ptr(rows), _stream_ptr(self.device)))
        return feats, rows

    def layer_grad_slice(self, layer):
        """View of flat_grads holding layer's filter then bias gradient (contiguous: one all-reduce bucket)."""
        cin, cout = C3D_CHANNELS[layer]
        ow = self.lib.rgp_c3d_param_offset(self._h, layer, 0)
        return self.flat_grads[ow:ow + 27 * cin * cout + cout]

    def wait_layer_grads(self, layer, stream):
        """Make `stream` (a torch.cuda.Stream) wait until the last backward() has finished layer's gradient slice."""
        _lib.check(self.lib.rgp_c3d_wait_layer_grads(self._h, int(layer), ctypes.c_void_p(stream.cuda_stream)))

    def read_grad_image(self, layer, n_windows):
        """After backward(): d loss / d (conv output of `layer` before ReLU and pooling), [n,D,H,W,Cout] fp32."""
        d, h = C3D_EXTENTS[layer]
        out = torch.empty(n_windows, d, h, h, C3D_CHANNELS[layer][1], device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_c3d_read_grad_image(self._h, layer, n_windows, _ptr(out), _stream_ptr(self.device)))
        return out

    def _frames_args(self, frames, window_starts, mean_cube):
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 4
        assert frames.shape[3] == 3, tuple(frames.shape)
        starts = [int(v) for v in window_starts]
        arr = (ctypes.c_int * len(starts))(*starts)
        if mean_cube is not None:
            assert mean_cube.is_cuda and mean_cube.dtype == torch.float32 and mean_cube.is_contiguous()
            assert tuple(mean_cube.shape) == (3, 16, 128, 171), tuple(mean_cube.shape)
        return arr, len(starts)

    def forward_frames(self, frames, window_starts, mean_cube=None, want_features=True, want_rows=False, out_rows=None):
        """frames [N,H,W,3] uint8 device tensor, window_starts: first frame of each 16-frame window
        -> (features [n,1024,7,7] fp32, rows): the VIDEO_DATA layer + conv1a..conv5b."""
        arr, n = self._frames_args(frames, window_starts, mean_cube)
        feats = torch.empty(n, 1024, 7, 7, device=self.device) if want_features else None
        rows = out_rows
        if want_rows and rows is None:
            rows = torch.empty(n * 49, 1024, dtype=self.torch_dtype, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_c3d_forward_frames(self._h, _ptr(frames), frames.shape[0], frames.shape[1],
                                                       frames.shape[2], arr, n, _ptr(mean_cube), _ptr(feats), _ptr(rows),
                                                       _stream_ptr(self.device)))
        return feats, rows

    def frames_to_video(self, frames, window_starts, mean_cube=None):
        """The VIDEO_DATA step alone -> video [n,16,112,112,3] fp32 (the input of forward())."""
        arr, n = self._frames_args(frames, window_starts, mean_cube)
        video = torch.empty(n, 16, 112, 112, 3, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_c3d_frames_to_video(self._h, _ptr(frames), frames.shape[0], frames.shape[1],
                                                        frames.shape[2], arr, n, _ptr(mean_cube), _ptr(video),
                                                        _stream_ptr(self.device)))
        return video

    def layer_kernel_name(self, layer, n_windows):
        """Kernel instantiation that runs layer's forward at n_windows windows per launch (as a profiler names it)."""
        return self.lib.rgp_c3d_layer_kernel_name(self._h, int(layer), int(n_windows)).decode()

    def read_layer(self, layer, n_windows):
        n = self.lib.rgp_c3d_layer_elems(self._h, layer, n_windows)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.rgp_c3d_read_layer(self._h, layer, n_windows, _ptr(out), _stream_ptr(self.device)))
        return out

    def profile(self, enable=True):
        _lib.check(self.lib.rgp_c3d_profile_enable(self._h, int(enable)))

    def profile_read(self):
        """{stage: (total_ms, launch_groups)} since the last read (HIP events on the launch stream)."""
        n = len(_lib.C3D_STAGES)
        ms, calls = (ctypes.c_double * n)(), (ctypes.c_longlong * n)()
        _lib.check(self.lib.rgp_c3d_profile_read(self._h, ms, calls))
        return {k: (ms[i], calls[i]) for i, k in enumerate(_lib.C3D_STAGES)}
