"""Mirror of /root/reference/models/gaze_framewise_shallownet.py: FramewiseShallowNet, the
per-frame ShallowNet as a GazePredictionGRU subclass (BASELINE config 1), on the HIP path."""
import numpy as np
import torch

from .. import synthetic
from ..engine import ShallowNetEngine
from .gaze_rnn import CONSTANTS, GazePredictionGRU, GRUModelConfig as _BaseGRUModelConfig


class GRUModelConfig(_BaseGRUModelConfig):
    """gaze_framewise_shallownet.py:43-57: T = 35, B = 5, loss l2."""

    def __init__(self):
        super(GRUModelConfig, self).__init__()
        self.n_lstm_steps = 35
        self.batch_size = 5
        self.loss_type = 'l2'
        self.compute_dtype = 'f32'
        self.image_hw = CONSTANTS.image_height        # 98 in the reference; 112 also supported


class _FramewiseEngine(object):
    """Adapts ShallowNetEngine to the harness' engine contract: forward(c3d, want_probs) -> (maps, probs)
    from the frames stashed by predict()."""

    def __init__(self, model):
        self.model = model
        hw = getattr(model.config, 'image_hw', CONSTANTS.image_height)
        self.net = ShallowNetEngine(model.batch_size * model.n_lstm_steps, hw,
                                    dtype=getattr(model.config, 'compute_dtype', 'f32'), device=model.session.device,
                                    save_for_backward=getattr(model.config, 'trainable', True))
        self.frames = None

    def set_weights(self, params):
        self.net.set_weights(params)

    # ---- training contract of GazePredictionGRU._train_op (backward -> all-reduce of flat_grads -> adam_step)
    @property
    def flat_grads(self):
        return self.net.flat_grads

    @property
    def weights(self):
        return self.net.weights

    def backward(self, logits, probs, labels, loss_type='l2'):
        """l2 loss of gaze_rnn.py:387-389 on the emitted maps; a 7x7 map is the 7x7 average pool of the 49x49 one
        (gaze_rnn.py:262-269), so its gradient spreads uniformly over each 7x7 cell."""
        assert loss_type == 'l2', 'FramewiseShallowNet is trained with the l2 loss (gaze_framewise_shallownet.py:43-57)'
        m = self.model
        F = m.batch_size * m.n_lstm_steps
        d = (logits - labels.reshape(logits.shape)) / float(F)
        if (m.gazemap_height, m.gazemap_width) == (7, 7):
            d = d.reshape(F, 7, 1, 7, 1).expand(F, 7, 7, 7, 7).reshape(F, 49, 49) / 49.0
        return self.net.backward(d.reshape(F, 49, 49).contiguous())

    def adam_step(self, step, lr, max_grad_norm=10.0, method='adam'):
        return self.net.adam_step(step, lr, max_grad_norm, method=method)

    def forward(self, c3d, want_probs=False):
        m = self.model
        assert self.frames is not None, 'FramewiseShallowNet needs frame_images (predict(c3d, frames))'
        x = torch.as_tensor(np.asarray(self.frames, np.float32)).to(m.session.device)
        x = x.reshape(m.batch_size * m.n_lstm_steps, self.net.image_hw, self.net.image_hw, 3).contiguous()
        want7 = (m.gazemap_height, m.gazemap_width) == (7, 7)
        sal, sal7 = self.net.forward(x, want_7x7=want7)
        out = (sal7 if want7 else sal).reshape(m.batch_size, m.n_lstm_steps, m.gazemap_height, m.gazemap_width)
        probs = None
        if want_probs:
            from ..engine import softmax_xent
            probs = softmax_xent(out.contiguous())[0]
        return out, probs


class FramewiseShallowNet(GazePredictionGRU):
    """gaze_framewise_shallownet.py:62-111."""

    def __init__(self, session, data_sets, config=None, gazemap_height=CONSTANTS.gazemap_height,
                 gazemap_width=CONSTANTS.gazemap_width):
        super(FramewiseShallowNet, self).__init__(session, data_sets, config if config is not None else GRUModelConfig(),
                                                  gazemap_height=gazemap_height, gazemap_width=gazemap_width)

    @staticmethod
    def create_gazeprediction_network(frame_images, c3d_input, dropout_keep_prob=1.0, net=None, model=None):
        """gaze_framewise_shallownet.py:74-90: reshape [B*T, IH, IW, 3] -> create_shallownet -> [B,T,49,49]."""
        assert model is not None
        engine = _FramewiseEngine(model)
        model.variables = synthetic.shallownet_params(getattr(model.config, 'init_seed', 0), engine.net.image_hw)
        engine.set_weights(model.variables)
        if net is not None:
            net['variables'] = model.variables
        return engine

    def predict(self, c3d, frames=None, train=False):
        self.engine.frames = frames
        return super(FramewiseShallowNet, self).predict(c3d, frames, train=train)
