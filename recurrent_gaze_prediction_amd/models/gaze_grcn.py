"""Mirror of /root/reference/models/gaze_grcn.py: GRU_RCN_Cell and GazePredictionGRCN,
executed by the HIP path (librgp_hip.so) instead of a TF graph."""
import numpy as np

from .. import synthetic
from ..engine import GRCN_PARAM_TO_FIELD, GrcnEngine
from .gaze_rnn import CONSTANTS, GazePredictionGRU, GRUModelConfig  # noqa: F401  (re-exported, gaze_grcn.py:35-45)


class GRU_RCN_Cell(object):
    """gaze_grcn.py:48-146: the six bias-free 3x3 filters of the convolutional GRU
    (truncated-normal stddev 1e-4) and its geometry.  The step itself
    (u, r, c, new_h = u*h + (1-u)*c, gaze_grcn.py:108-129) runs in the
    ``EpiGruZR`` / ``EpiGruC`` epilogues of the recurrent HIP kernels."""

    def __init__(self, num_units, dim_feature, spatial_shape=(7, 7), kernel_spatial_shape=(3, 3), seed=0,
                 stddev=1e-4):
        self.spatial_H, self.spatial_W = spatial_shape
        assert self.spatial_H > 0 and self.spatial_W > 0
        assert tuple(spatial_shape) == (7, 7) and tuple(kernel_spatial_shape) == (3, 3), \
            'HIP path is built for the 3x3 cell on 7x7 maps (gaze_grcn.py:260)'
        self._num_units, self.dim_feature = num_units, dim_feature
        p = synthetic.grcn_params(seed, 1, dim_feature, num_units, gru_std=stddev)
        self.W_z, self.U_z = p['GRU_Conv_Wz'], p['GRU_Conv_Uz']
        self.W_r, self.U_r = p['GRU_Conv_Wr'], p['GRU_Conv_Ur']
        self.W, self.U = p['GRU_Conv_W'], p['GRU_Conv_U']

    @property
    def output_size(self):
        return self._num_units

    @property
    def state_size(self):
        return self._num_units

    def zero_state(self, batch_size, dtype=np.float32):
        return np.zeros([batch_size, self.spatial_H, self.spatial_W, self.state_size], dtype)


class GazePredictionGRCN(GazePredictionGRU):
    """gaze_grcn.py:152-376."""

    DIM_CNN_PROJ = 512      # gaze_grcn.py:206
    RNN_STATE_SIZE = 128    # gaze_grcn.py:211

    def __init__(self, session, data_sets, config=None):
        super(GazePredictionGRCN, self).__init__(session, data_sets, config=config)

    @staticmethod
    def create_gazeprediction_network(frame_images, c3d_input, dropout_keep_prob=1.0, net=None, model=None):
        """gaze_grcn.py:173-376.  Instead of graph tensors this returns the device engine that
        evaluates the same graph; ``net`` receives the variables.  frame_images is unused by the
        reference graph beyond its shape (9-Q4); dropout is inert there (9-Q2) and is off here."""
        assert model is not None, 'pass the owning model (B, T, dtype, device come from its config)'
        if net is None:
            net = {}
        B, T = model.batch_size, model.n_lstm_steps
        P, S = GazePredictionGRCN.DIM_CNN_PROJ, GazePredictionGRCN.RNN_STATE_SIZE
        engine = GrcnEngine(B, T, P, S, dtype=getattr(model.config, 'compute_dtype', 'bf16'),
                            save_for_backward=getattr(model.config, 'trainable', True), device=model.session.device,
                            per_step=bool(getattr(model.config, 'convgru_per_step', False)))
        # reference initialisers (gaze_grcn.py:64-81,234-237,292-314); BN gamma=1, beta=0 per step
        model.variables = synthetic.grcn_params(getattr(model.config, 'init_seed', 0), T, P, S, gru_std=1e-4)
        engine.set_weights(model.variables)
        net['variables'] = model.variables
        return engine

    def _recover_from_timeout(self):
        """A persistent ConvGRU launch lost a group member (another process on the device took its CUs; include/rgp.h):
        continue on a plan that runs the recurrence and its BPTT as per-timestep launches (RGP_GRCN_PER_STEP), which need no
        co-residency.  A NEW engine object in this process (never a re-exec): master weights and optimizer slots move
        over device to device; the poisoned outputs are recomputed by the caller.  Logged once."""
        from ..engine import OPT_STATE_KEYS
        old = self.engine
        if getattr(old, 'per_step', False):
            return False                                          # already the fallback: the error is something else
        log = __import__('logging').getLogger('rgp')
        log.warning('persistent ConvGRU launch timed out (RGP_ETIMEOUT): switching this model to per-timestep launches')
        new = GrcnEngine(old.B, old.T, old.P, old.S, dtype=old.dtype, save_for_backward=old.save_for_backward,
                         device=old.device, per_step=True)
        new.set_weights(old.weights)
        for k in OPT_STATE_KEYS:
            if getattr(old, k, None) is not None:
                setattr(new, k, getattr(old, k).clone())
        self.engine = new
        self.config.convgru_per_step = True
        return True

    # ---- variables (TF names), for checkpoints and for loading exported weights ----------
    def state_dict(self):
        # the engine's fp32 master weights are the live variables (updated by the optimizer)
        return {k: v.detach().cpu().numpy().copy() for k, v in self.engine.weights.items()}

    def load_state_dict(self, state):
        missing = [k for k in GRCN_PARAM_TO_FIELD if k not in state]
        assert not missing, 'missing variables: %s' % missing
        T, S = self.n_lstm_steps, self.RNN_STATE_SIZE
        assert tuple(np.shape(state['bn_gamma'])) == (T, S), \
            'checkpoint has %s batch-norm layers, model needs %d (one per timestep, SURVEY 9-Q1)' % (
                np.shape(state['bn_gamma']), T)
        self.variables = {k: np.asarray(state[k], np.float32) for k in GRCN_PARAM_TO_FIELD}
        self.engine.set_weights(self.variables)
