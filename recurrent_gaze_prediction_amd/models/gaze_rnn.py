"""Mirror of /root/reference/models/gaze_rnn.py: GRUModelConfig, CONSTANTS and the
GazePredictionGRU harness (build_model / single_step / generate / evaluate).

The harness is model-agnostic exactly as in the reference: subclasses supply
``create_gazeprediction_network``.  The fc-GRU graph of this base class (gaze_rnn.py:211-360,
BASELINE config 2) runs forward AND backward on the HIP path (rgp_fcgru_*), including the
training-time dropout on its projected features."""
import logging
import time
from types import SimpleNamespace

import numpy as np
import torch

from .. import evaluation_metrics
from ..evaluation_metrics import AVAILABLE_METRICS, saliency_score
from .base import BaseModelConfig, ModelBase, Session
from .model_util import normalize_probability_map

log = logging.getLogger('rgp')

CONSTANTS = SimpleNamespace(image_width=98, image_height=98, gazemap_width=49, gazemap_height=49,
                            saliencymap_width=49, saliencymap_height=49)        # gaze_rnn.py:34-40


class GRUModelConfig(BaseModelConfig):
    """gaze_rnn.py:44-61."""

    def __init__(self):
        super(GRUModelConfig, self).__init__()
        self.n_lstm_steps = 42
        self.batch_size = 7
        self.dim_feature = 1024
        self.dim_sal = 1024 * 49
        self.dim_sal_proj = 1024
        self.optimization_method = 'adam'
        self.loss_type = 'xentropy'
        self.use_flip_batch = True
        # additions (not in the reference): MFMA operand dtype of the HIP path, weight seed, augmentation seed
        self.compute_dtype = 'bf16'
        self.init_seed = 0
        self.flip_seed = None          # None: derived from init_seed and the rank (recorded in the log / checkpoint)
        self.train_keep_prob = 0.5     # keep_prob single_step feeds when training (gaze_rnn.py:529)


DROPOUT_RANK_STRIDE = 0x9E3779B1     # key offset between data-parallel ranks (also used by the checkpoint loader)


def dropout_seed(config, salt):
    """Philox key of a model's dropout site: data-parallel ranks must draw DIFFERENT masks (as they mirror different
    clips, flip_seed below), so the rank is mixed in; the key and the draw counter are saved in checkpoints."""
    from .. import dist as rdist
    rank = rdist.env_world()[0]
    return ((int(getattr(config, 'init_seed', 0)) << 20) + int(salt) + DROPOUT_RANK_STRIDE * rank) & 0x7fffffffffffffff


class Placeholder(object):
    """Inert stand-in for a ``tf.placeholder`` attribute of the reference model (gaze_rnn.py:114-129)."""

    def __init__(self, name, shape, dtype='float32'):
        self.name, self.shape, self.dtype = name, tuple(shape), dtype

    def get_shape(self):
        return self.shape

    def __repr__(self):
        return '<Placeholder %s %s %s: feed model.predict(c3d, frames) instead of session.run>' % (self.name, self.shape, self.dtype)


class GazePredictionGRU(ModelBase):
    """gaze_rnn.py:66-680."""

    def __init__(self, session, data_sets, config=None, gazemap_height=CONSTANTS.gazemap_height,
                 gazemap_width=CONSTANTS.gazemap_width):
        self.session = session if session is not None else Session()
        self.data_sets = data_sets
        self.config = config if config is not None else GRUModelConfig()
        super(GazePredictionGRU, self).__init__(self.config)
        self.batch_size = self.config.batch_size
        self.n_lstm_steps = self.config.n_lstm_steps
        self.dim_feature = self.config.dim_feature
        self.dim_sal = self.config.dim_sal
        self.dim_sal_proj = self.config.dim_sal_proj
        self.dim_cnn_proj = 32
        self.initial_learning_rate = self.config.initial_learning_rate
        self.learning_rate_decay = self.config.learning_rate_decay
        self.max_grad_norm = self.config.max_grad_norm
        self.gazemap_height, self.gazemap_width = gazemap_height, gazemap_width
        self.image_height, self.image_width = CONSTANTS.image_height, CONSTANTS.image_width
        self.dropout_keep_prob = 1.0
        # The reference's flip augmentation draws from numpy's GLOBAL RNG (gaze_rnn.py:504-510).  Data-parallel ranks
        # must not mirror the same clips, and a run must be repeatable: each rank owns a seeded stream, seed recorded.
        from .. import dist as rdist
        rank = rdist.env_world()[0]
        cfg_seed = getattr(self.config, 'flip_seed', None)
        self.flip_seed = int(cfg_seed) if cfg_seed is not None else \
            (int(getattr(self.config, 'init_seed', 0)) * 1000003 + 7919 * rank + 12345) & 0x7fffffff
        self.flip_rng = np.random.RandomState(self.flip_seed)
        log.info('flip-augmentation RNG: rank %d seed %d', rank, self.flip_seed)
        self.build_model()
        self.build_train_op()

    # ------------------------------------------------------------------ graph
    def build_model(self):
        """gaze_rnn.py:108-160: create the network (device engine) and its outputs."""
        self.net = {}
        self.engine = self.create_gazeprediction_network(frame_images=None, c3d_input=None,
                                                         dropout_keep_prob=self.dropout_keep_prob, net=self.net,
                                                         model=self)
        self.predicted_gazemaps = None          # filled by predict(): probs for xentropy, raw maps for l2
        self.predicted_gazemaps_logit = None
        self.loss = None
        # The reference's placeholders (gaze_rnn.py:114-129), read by name by callers that build their own feed_dict
        # (extract_map.py:221-227: model.c3d_input, model.frame_images, model.gt_gazemap).  There is no graph to feed:
        # they are inert descriptors (name / shape / dtype) so such code fails at its session.run with a message that
        # names predict(), not with an AttributeError on the model.
        B, T = self.batch_size, self.n_lstm_steps
        self.frame_images = Placeholder('frame_images', (B, T, self.image_height, self.image_width, 3))
        self.c3d_input = Placeholder('c3d_input', (B, T, 1024, 7, 7))
        self.gt_gazemap = Placeholder('gt_gazemap', (B, T, self.gazemap_height, self.gazemap_width))

    @staticmethod
    def create_gazeprediction_network(frame_images, c3d_input, dropout_keep_prob=1.0, net=None, model=None):
        """gaze_rnn.py:211-360: 1024->32 projection, GRUCell(7*7*32+49), linear read-out, on the HIP
        path (rgp_fcgru_*), forward and backward.  The ShallowNet branch the reference also builds (:256-275)
        does not reach the output and is not evaluated.  tf.nn.dropout on the projected features (:302-303)
        is an op of the engine: off at inference, keep = config.train_keep_prob (0.5, :529) with a fresh
        device-drawn mask per training step."""
        from .. import synthetic
        from ..engine import FcGruEngine
        assert model is not None
        if net is None:
            net = {}
        engine = FcGruEngine(model.batch_size, model.n_lstm_steps, (model.gazemap_height, model.gazemap_width),
                             dtype=getattr(model.config, 'compute_dtype', 'f32'), device=model.session.device,
                             save_for_backward=getattr(model.config, 'trainable', True))
        model.variables = synthetic.fcgru_params(getattr(model.config, 'init_seed', 0), model.gazemap_height,
                                                 model.gazemap_width)
        engine.set_weights(model.variables)
        engine.dropout.configure(getattr(model.config, 'train_keep_prob', 0.5), seed=dropout_seed(model.config, 0x5bd1e995))
        net['variables'] = model.variables
        return engine

    def state_dict(self):
        w = getattr(self.engine, 'weights', None)          # trained values live in the engine's master buffer
        if isinstance(w, dict) and all(k in w for k in self.variables):
            self.variables = {k: w[k].detach().cpu().numpy().copy() for k in self.variables}
        return {k: np.array(v, copy=True) for k, v in self.variables.items()}

    def load_state_dict(self, state):
        self.variables = {k: np.asarray(v, np.float32) for k, v in state.items()}
        self.engine.set_weights(self.variables)

    def initialize_pretrained_shallownet(self, checkpoint_path):
        """gaze_rnn.py:412-433: copy the ``ShallowNet/*`` variables of a separately trained checkpoint into this
        model (optimizer slots and tflearn's is_training flags skipped).  checkpoint_path: an exported ``.npz`` of
        {TF variable name: array} (checkpoint.load_tf_export)."""
        from .. import checkpoint
        shallow = checkpoint.import_shallownet_variables(checkpoint.load_tf_export(checkpoint_path))
        state = self.state_dict()
        if any(k.startswith('ShallowNet/') for k in state):          # cascade: nested sub-network
            state.update({'ShallowNet/' + k: v for k, v in shallow.items()})
        elif all(k in state for k in shallow):                        # frame-wise ShallowNet: the whole model
            state.update(shallow)
        else:
            raise KeyError('%s has no ShallowNet variables' % type(self).__name__)
        self.load_state_dict(state)
        for k, v in shallow.items():
            log.info('Using pretrained value for ShallowNet/%s : %s', k, str(v.shape))

    def build_train_op(self):
        """gaze_rnn.py:448-478 + base.py:262-308: gradients of the loss w.r.t. every non-ShallowNet
        variable, clip_by_global_norm(max_grad_norm), then the configured optimizer (adam / rmsprop / sgd with
        momentum 0.9, base.py:268-273) at the scheduled learning rate.  Here: the engine's backward + the fused
        clip+optimizer kernels; with WORLD_SIZE > 1 the flat gradient bucket is all-reduced (mean) over RCCL
        before the clip, so the clip sees the global-batch gradient."""
        from ..engine import OPTIMIZERS
        if self.config.optimization_method not in OPTIMIZERS:
            raise ValueError('Invalid optimization method!')          # base.py:274
        self.dist = None           # set by attach_process_group()
        self.reducer = None
        self.train_op = self._train_op

    def attach_process_group(self, dist):
        """dist: torch.distributed (already initialised, backend nccl = RCCL) or None."""
        from .. import dist as rdist
        self.dist = dist
        self.reducer = rdist.GradBucketReducer(dist, self.session.device) if dist is not None else None
        # plans with persistent ConvGRU launches can fail asynchronously (RGP_ETIMEOUT): the ranks then agree on the outcome
        # of every backward before they enter the collectives.  Decided ONCE, from the plan every rank was built with (a rank
        # that later falls back to per-step launches keeps answering), so the ranks' collective sequences stay the same.
        self._dp_agree = dist is not None and bool(getattr(self.engine, 'persistent', False))

    def _train_op(self, logits, probs, labels_dev):
        loss_type = 'l2' if self.config.loss_type == 'l2' else 'xentropy'
        self.engine.backward(logits, probs, labels_dev, loss_type)
        err = None
        try:
            if self._status_or_recover():         # the BPTT launch timed out (NaN gradients): redo the step's forward
                z, p = self.engine.forward(self._last_input, **self._last_forward_kw)[:2]      # and backward on the new engine
                self.engine.backward(z, p, labels_dev, loss_type)
                self._status_or_recover(final=True)
        except Exception as exc:                  # noqa: BLE001 -- re-raised below, on every rank
            err = exc
        if getattr(self, '_dp_agree', False):
            # a rank that raises here must not leave its peers blocked in the all-reduce: everybody raises, or nobody
            from .. import dist as rdist
            if not rdist.all_ranks_ok(self.dist, err is None, self.session.device):
                raise err if err is not None else RuntimeError('a peer rank failed in its backward pass')
        elif err is not None:
            raise err
        if self.reducer is not None:
            buckets = getattr(self.engine, 'grad_buckets', None)
            if callable(buckets):
                # in completion order on the reducer's side stream.  (On this path the status check above has already waited
                # for the backward, so nothing overlaps; engines driven directly -- dist.dp_train_probe, finetune.py --
                # skip that wait and do overlap.)
                self.reducer.reduce_buckets(buckets())
            else:
                self.reducer.reduce(self.engine.flat_grads)   # in place, fp32, mean over ranks
            self.reducer.finish()
        step_kw = {}
        if self.config.optimization_method != 'adam':
            step_kw['method'] = self.config.optimization_method
        self.grad_norm = self.engine.adam_step(self._global_step, self.learning_rate_at(self._global_step),
                                               max_grad_norm=self.max_grad_norm, **step_kw)
        self._global_step += 1

    def learning_rate_at(self, step):
        """_build_learning_rate (gaze_rnn.py:436-444): lr0 * decay^floor(step/500)."""
        return self.initial_learning_rate * self._learning_rate_scale * self.learning_rate_decay ** (step // 500)

    @property
    def current_learning_rate(self):
        return self.learning_rate_at(self.current_step)

    @property
    def global_step(self):
        """The reference's ``global_step`` variable (gaze_rnn.py:104; read by extract_map.py / the trainers): an int here."""
        return self._global_step

    # ------------------------------------------------------------------ execution
    def predict(self, c3d, frames=None, train=False):
        """Replacement for ``session.run(predicted_gazemaps, feed_dict)`` (gaze_rnn.py:603-611):
        c3d [B,T,1024,7,7] (numpy or device tensor; frames are accepted and ignored)
        -> numpy [B,T,GH,GW]: softmax maps for loss_type xentropy/KLD, raw maps for l2 (9-Q5).
        train=True is single_step's training feed (dropout_keep_prob 0.5, gaze_rnn.py:529): engines that own a
        connected dropout site (fc-GRU, cascade) draw a mask; gaze_grcn's sites are inert."""
        x = torch.as_tensor(np.asarray(c3d, dtype=np.float32) if not torch.is_tensor(c3d) else c3d)
        x = x.to(self.session.device, torch.float32).reshape(self.batch_size, self.n_lstm_steps, 1024, 7, 7).contiguous()
        want_probs = self.config.loss_type in ('xentropy', 'KLD')
        kw = {'train': True} if (train and self._has_dropout()) else {}
        logits, probs = self.engine.forward(x, want_probs=want_probs, **kw)[:2]
        if self._status_or_recover():             # the engine was replaced: compute this batch again with it
            logits, probs = self.engine.forward(x, want_probs=want_probs, **kw)[:2]
            self._status_or_recover(final=True)
        self._last_input, self._last_forward_kw = x, dict(kw, want_probs=want_probs)
        self.predicted_gazemaps_logit = logits
        self.predicted_gazemaps = probs if want_probs else logits
        return self.predicted_gazemaps

    def _status_or_recover(self, final=False):
        """"A TF session either returns or raises" (gaze_rnn.py:603-611).  Engines with persistent launches report a lost
        group member asynchronously (RGP_ETIMEOUT, NaN-poisoned outputs, include/rgp.h): wait for the stream, and on
        that error let the model swap in an engine that does not depend on co-residency (_recover_from_timeout) --
        returns True then, and the caller recomputes the batch.  Any other error, a model without a fallback, or
        final=True: raises."""
        status = getattr(self.engine, 'status', None)
        if not callable(status):
            return False
        if not getattr(self.engine, 'persistent', True):
            return False          # per-timestep launches cannot time out: no host wait for the stream either
        from .. import _lib
        try:
            status()
            return False
        except _lib.RgpError as err:
            if final or err.code != _lib.RGP_ETIMEOUT or not self._recover_from_timeout():
                raise
            return True

    def _recover_from_timeout(self):
        """Models whose engine has a time-out-free variant override this (GazePredictionGRCN)."""
        return False

    def _has_dropout(self):
        e = getattr(self.engine, 'net', self.engine)
        return getattr(e, 'dropout', None) is not None

    def compute_loss(self, gt_gazemap):
        """create_loss_and_summary (gaze_rnn.py:363-408) on the last predict()'s logits."""
        from ..engine import l2_loss, softmax_xent
        g = torch.as_tensor(np.asarray(gt_gazemap, np.float32)).to(self.session.device).contiguous()
        z = self.predicted_gazemaps_logit
        if self.config.loss_type == 'xentropy':
            return float(softmax_xent(z, g.reshape(z.shape), want_probs=False)[2].item())
        if self.config.loss_type == 'l2':
            return float(l2_loss(z.contiguous(), g.reshape(z.shape).contiguous(), z.shape[0] * z.shape[1]).item())
        raise NotImplementedError(str(self.config.loss_type))   # 'KLD' is broken in the reference too (:395-399)

    def single_step(self, train_mode=True, dataset=None):
        """gaze_rnn.py:483-565."""
        _start_time = time.time()
        if dataset is None:
            dataset = self.data_sets.train if train_mode else self.data_sets.valid
        batch_images, batch_maps, batch_fixmaps, batch_c3d, batch_pupil, batch_clipnames = dataset.next_batch(self.batch_size)
        batch_c3d = np.reshape(batch_c3d, [self.batch_size, -1, 1024, 7, 7])
        if self.config.loss_type in ('xentropy', 'KLD'):
            batch_maps = normalize_probability_map(batch_maps)
        if train_mode and self.config.use_flip_batch:
            # gaze_rnn.py:504-510: mirror a random half of the clips left-right (Python-2 integer division,
            # SURVEY 9-Q12); drawn from this rank's seeded stream (self.flip_seed) instead of numpy's global RNG
            batch_images, batch_maps, batch_c3d = np.array(batch_images), np.array(batch_maps), np.array(batch_c3d)
            indices = self.flip_rng.choice(self.batch_size, self.batch_size // 2, replace=False)
            batch_images[indices] = batch_images[indices][:, :, :, ::-1, :]
            batch_maps[indices] = batch_maps[indices][:, :, :, ::-1]
            batch_c3d[indices] = batch_c3d[indices][:, :, :, :, ::-1]
            if isinstance(batch_fixmaps, np.ndarray) and batch_fixmaps.dtype != object:
                batch_fixmaps = np.array(batch_fixmaps)
                batch_fixmaps[indices] = batch_fixmaps[indices][:, :, :, ::-1]
        self.predict(batch_c3d, batch_images, train=train_mode)
        self.loss = loss = self.compute_loss(batch_maps)
        if train_mode:
            # a diverged step must be loud: ReLU / max-pool (maxNum semantics) turn a NaN into a finite value further down
            # the graph, so the loss and the pre-clip gradient norm are the places to look
            if not np.isfinite(loss):
                raise FloatingPointError('non-finite training loss %r at step %d' % (loss, self.current_step))
            labels = torch.as_tensor(np.ascontiguousarray(batch_maps, np.float32)).to(self.session.device)
            self.train_op(self.predicted_gazemaps_logit, self.predicted_gazemaps, labels.reshape(self.predicted_gazemaps_logit.shape).contiguous())
        step = self.current_step
        dt = time.time() - _start_time
        if (not train_mode) or step % max(1, self.config.steps_per_logprint) == 0:
            gn = ''
            if train_mode and getattr(self, 'grad_norm', None) is not None:
                g = float(self.grad_norm.item())
                if not np.isfinite(g):
                    raise FloatingPointError('non-finite gradient norm %r at step %d' % (g, step))
                gn = ' (|g|=%.3g)' % g
            log.info(" [%5s step %4d] batch total-loss: %.5f (%.3f sec/batch, %.3f instances/sec) (lr=%.3g)%s",
                     'train' if train_mode else 'val', step, loss, dt, self.batch_size / dt, self.current_learning_rate, gn)
        return step

    def generate(self, dataset, max_instances=50):
        """gaze_rnn.py:568-650: runs ceil(min(len(dataset), max_instances) / batch_size) batches through the network and
        returns the reference's dictionary -- maps, ground truth and C3D features flattened over (clip, timestep);
        ``fixationmap_list`` a dense array when the loader hands out dense maps, else a flat list of the per-frame
        (sparse) maps; ``images_list`` per frame; ``clipname_list`` per clip."""
        GH, GW, B = self.gazemap_height, self.gazemap_width, self.batch_size
        n_instances = len(dataset) if max_instances is None else min(len(dataset), max_instances)
        n_batches = -(-n_instances // B)
        assert n_batches > 0
        normalise_gt = self.config.loss_type == 'xentropy'
        out = {'pred_gazemap_list': [], 'gt_gazemap_list': [], 'c3d_list': [], 'images_list': [], 'clipname_list': []}
        fixations, dense_fixations = [], True
        for _ in range(n_batches):
            images, maps, fixmaps, c3d, _pupil, clipnames = dataset.next_batch(B)
            images = np.asarray(list(images))
            assert images.dtype == np.float32
            c3d = np.reshape(c3d, [B, -1, 1024, 7, 7])
            pred = self.predict(c3d, images).cpu().numpy()        # (predict() has waited for the engine's status)
            out['pred_gazemap_list'].append(pred.reshape(-1, GH, GW))
            out['gt_gazemap_list'].append(np.reshape(normalize_probability_map(maps) if normalise_gt else maps, (-1, GH, GW)))
            out['c3d_list'].append(c3d.reshape(-1, 1024, 7, 7))
            out['images_list'].extend(images.reshape((-1,) + images.shape[2:]))
            out['clipname_list'].extend(clipnames)
            for clip_fix in fixmaps:                              # one entry per clip: [T, H', W'] dense, or T sparse maps
                dense_fixations = dense_fixations and isinstance(clip_fix, np.ndarray) and clip_fix.dtype != object
                fixations.extend(clip_fix[t] for t in range(len(clip_fix)))
        for key in ('pred_gazemap_list', 'gt_gazemap_list', 'c3d_list'):
            out[key] = np.concatenate(out[key])
        if dense_fixations:
            try:
                fixations = np.stack(fixations)
            except ValueError:                                    # frames of different sizes: stays a list
                pass
        assert len(fixations) == len(out['pred_gazemap_list'])
        out['fixationmap_list'] = fixations
        return out

    def evaluate(self, pred_gazemap_list, gt_gazemap_list, fixationmap_list, images_list, **_ignored):
        """gaze_rnn.py:653-674.  Extra keys of generate()'s dictionary are accepted and ignored
        (the reference raises TypeError there)."""
        assert len(pred_gazemap_list) == len(gt_gazemap_list) == len(fixationmap_list) == len(images_list), \
            "Length mismatch: %d %d %d %d" % (len(pred_gazemap_list), len(gt_gazemap_list),
                                              len(fixationmap_list), len(images_list))
        batch_scores = {}
        for metric in AVAILABLE_METRICS:
            batch_scores[metric] = saliency_score(metric, pred_gazemap_list, gt_gazemap_list, fixationmap_list)
            log.info('Saliency %s : %f', metric, batch_scores[metric])
        self.report_evaluate_summary(batch_scores)
        return batch_scores

    def generate_and_evaluate(self, dataset, max_instances=50):
        ret = self.generate(dataset, max_instances)
        return ret, self.evaluate(**ret)


__all__ = ['CONSTANTS', 'GRUModelConfig', 'GazePredictionGRU', 'evaluation_metrics']
