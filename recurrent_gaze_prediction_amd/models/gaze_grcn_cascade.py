"""Mirror of /root/reference/models/gaze_grcn_cascade.py (BASELINE config 5): GRU_RCN_Cell and the
two-level GazePredictionGRCN -- ShallowNet saliency + bottom ConvGRU (7x7, 256 maps) -> stride-7
transposed conv -> top ConvGRU (49x49, 3 units) -> two maxout FCs, loss l2 -- on the HIP path
(rgp_cascade_*, forward).  The committed reference file cannot build its graph;
this class implements the evident intent recorded in its commented block (:370-377)."""
from types import SimpleNamespace

import numpy as np
import torch

from .. import synthetic
from ..engine import CascadeEngine
from .gaze_rnn import GazePredictionGRU, GRUModelConfig as _BaseGRUModelConfig

CONSTANTS = SimpleNamespace(image_width=98, image_height=98, gazemap_width=49, gazemap_height=49,
                            saliencymap_width=49, saliencymap_height=49)        # gaze_grcn_cascade.py:40-46


class GRU_RCN_Cell(object):
    """gaze_grcn_cascade.py:49-135: six bias-free kH x kW filters (truncated normal, stddev 1e-4) of a
    convolutional GRU on an H x W map.  The HIP path has the two instances the cascade builds:
    (256 units, 512 features, 7x7, 3x3) and (3 units, 65 features, 49x49, 5x5)."""

    def __init__(self, num_units, dim_feature, spatial_shape=(7, 7), kernel_spatial_shape=(3, 3), seed=0, stddev=1e-4):
        self.spatial_H, self.spatial_W = spatial_shape
        assert self.spatial_H > 0 and self.spatial_W > 0
        geom = (num_units, dim_feature, tuple(spatial_shape), tuple(kernel_spatial_shape))
        assert geom in ((256, 512, (7, 7), (3, 3)), (3, 65, (49, 49), (5, 5))), 'no HIP path for cell %r' % (geom,)
        self._num_units, self.dim_feature = num_units, dim_feature
        rs = np.random.RandomState(seed)
        kh, kw = kernel_spatial_shape
        tn = lambda cin: synthetic._trunc_normal(rs, (kh, kw, cin, num_units), stddev)
        self.W_z, self.U_z, self.W_r, self.U_r, self.W, self.U = (tn(dim_feature), tn(num_units), tn(dim_feature),
                                                                  tn(num_units), tn(dim_feature), tn(num_units))

    @property
    def input_size(self):
        return self._num_units

    @property
    def output_size(self):
        return self._num_units

    @property
    def state_size(self):
        return self._num_units

    def zero_state(self, batch_size, dtype=np.float32):
        return np.zeros([batch_size, self.spatial_H, self.spatial_W, self.state_size], dtype)


class GRUModelConfig(_BaseGRUModelConfig):
    """The cascade is trained with the plain l2 loss (gaze_grcn_cascade.py:427-438)."""

    def __init__(self):
        super(GRUModelConfig, self).__init__()
        self.loss_type = 'l2'
        self.compute_dtype = 'bf16'
        self.image_hw = CONSTANTS.image_height


class _CascadeAdapter(object):
    """CascadeEngine behind the harness' engine contract forward(c3d, want_probs) -> (maps, probs)."""

    def __init__(self, model):
        self.model = model
        self.net = CascadeEngine(model.batch_size, model.n_lstm_steps, getattr(model.config, 'image_hw', 98),
                                 dtype=getattr(model.config, 'compute_dtype', 'bf16'), device=model.session.device,
                                 save_for_backward=getattr(model.config, 'trainable', True))
        self.frames = None

    def set_weights(self, params):
        self.net.set_weights(params)

    # ---- training contract of GazePredictionGRU._train_op (backward -> all-reduce of flat_grads -> adam_step)
    @property
    def flat_grads(self):
        return self.net.flat_grads

    def backward(self, logits, probs, labels, loss_type='l2'):
        assert loss_type == 'l2', 'the cascade is trained with the l2 loss (gaze_grcn_cascade.py:428-441)'
        return self.net.backward(logits, labels.reshape(logits.shape).contiguous())[0]

    def adam_step(self, step, lr, max_grad_norm=10.0, method='adam'):
        from ..engine import clip_step_multi
        return clip_step_multi([self.net], step, lr, max_grad_norm, method)

    def forward(self, c3d, want_probs=False, train=False):
        m = self.model
        assert self.frames is not None, 'the cascade needs frame_images (predict(c3d, frames))'
        x = self.frames if torch.is_tensor(self.frames) else torch.as_tensor(np.asarray(self.frames, np.float32))
        x = x.to(m.session.device, torch.float32).reshape(m.batch_size, m.n_lstm_steps, self.net.image_hw,
                                                          self.net.image_hw, 3).contiguous()
        maps = self.net.forward(x, c3d, train=train)     # train: fc1's dropout draws a mask (gaze_grcn_cascade.py:401-402)
        probs = None
        if want_probs:
            from ..engine import softmax_xent
            probs = softmax_xent(maps.contiguous())[0]
        return maps, probs


class GazePredictionGRCN(GazePredictionGRU):
    """gaze_grcn_cascade.py:138-445."""

    def __init__(self, session, data_sets, config=None, gazemap_height=CONSTANTS.gazemap_height,
                 gazemap_width=CONSTANTS.gazemap_width):
        assert (gazemap_height, gazemap_width) == (49, 49), 'the cascade emits 49x49 maps (gaze_grcn_cascade.py:423)'
        super(GazePredictionGRCN, self).__init__(session, data_sets, config if config is not None else GRUModelConfig(),
                                                 gazemap_height=gazemap_height, gazemap_width=gazemap_width)
        self.dim_cnn_proj = 512          # gaze_grcn_cascade.py:156

    @staticmethod
    def create_gazeprediction_network(frame_images, c3d_input, dropout_keep_prob=1.0, net=None, model=None):
        """gaze_grcn_cascade.py:188-423."""
        assert model is not None
        engine = _CascadeAdapter(model)
        model.variables = synthetic.cascade_params(getattr(model.config, 'init_seed', 0), engine.net.image_hw)
        engine.set_weights(model.variables)
        from .gaze_rnn import dropout_seed
        engine.net.dropout.configure(getattr(model.config, 'train_keep_prob', 0.5), seed=dropout_seed(model.config, 0x2545f491))
        if net is not None:
            net['variables'] = model.variables
        return engine

    def state_dict(self):
        net = self.engine.net                      # the trained values live in the engine's master buffer
        for field, key in net.KEYS:
            self.variables[key] = net.weights[field].detach().cpu().numpy().copy()
        flat = {k: np.array(v, copy=True) for k, v in self.variables.items() if k != 'ShallowNet'}
        flat.update({'ShallowNet/' + k: np.array(v, copy=True) for k, v in self.variables['ShallowNet'].items()})
        return flat

    def load_state_dict(self, state):
        v = {k: np.asarray(a, np.float32) for k, a in state.items() if not k.startswith('ShallowNet/')}
        v['ShallowNet'] = {k[len('ShallowNet/'):]: np.asarray(a, np.float32) for k, a in state.items()
                           if k.startswith('ShallowNet/')}
        self.variables = v
        self.engine.set_weights(v)

    def predict(self, c3d, frames=None, train=False):
        self.engine.frames = frames
        return super(GazePredictionGRCN, self).predict(c3d, frames, train=train)
