"""Mirror of /root/reference/models/base.py: BaseModelConfig and ModelBase.

Differences from the reference, all forced by the runtime swap: checkpoints are
``torch.save`` state dicts keyed by the reference's TF variable names (so an exported TF
checkpoint can be mapped in) instead of ``tf.train.Saver`` files; TF
summaries become log lines; ``session`` is a small device/stream handle."""
import json
import logging
import os
import pickle
import re
import tempfile
import time

import numpy as np
import torch

log = logging.getLogger('rgp')


class Session(object):
    """Stands in for ``tf.Session``: the device (and its current HIP stream) the model runs on."""

    def __init__(self, device='cuda:0'):
        self.device = torch.device(device)

    def synchronize(self):
        torch.cuda.synchronize(self.device)


class BaseModelConfig(object):
    """base.py:19-85 (same fields and defaults)."""

    def __init__(self, train_dir=None, max_steps=100000, steps_per_checkpoint=1000, steps_per_validation=100,
                 steps_per_evaluation=2000, steps_per_logprint=5, verbose_level=2, learning_rate_decay=0.80,
                 optimization_method='adam', initial_learning_rate=0.003):
        self.train_dir = train_dir
        self.train_tag = ''
        self.max_steps = max_steps
        self.steps_per_checkpoint = steps_per_checkpoint
        self.steps_per_validation = steps_per_validation
        self.steps_per_evaluation = steps_per_evaluation
        self.steps_per_logprint = steps_per_logprint
        self.verbose_level = int(verbose_level)
        self.learning_rate_decay = learning_rate_decay
        self.optimization_method = optimization_method
        self.initial_learning_rate = initial_learning_rate
        self.max_grad_norm = 10.0

    def __repr__(self):
        return 'ModelConfig{' + ', '.join("%s : %s" % kv for kv in sorted(vars(self).items())) + '}'

    def dump(self, fp):
        """base.py:60-72: a path gets a pickle, a file object gets JSON."""
        if isinstance(fp, str):
            with open(fp, 'wb') as f:
                pickle.dump(self, f)
        else:
            json.dump(self.__dict__, fp, sort_keys=True, indent=4, separators=(',', ': '))
            fp.write('\n')
            fp.flush()

    @staticmethod
    def load(fp):
        if isinstance(fp, str):
            with open(fp, 'r') as f:
                return BaseModelConfig.load(f)
        config = BaseModelConfig()
        for key, value in json.load(fp).items():
            setattr(config, key, value)
        return config


def urlify(s):
    s = re.sub(r"[^\w\s\-_,]", '', s)
    return re.sub(r"\s+", '-', s)


class ModelBase(object):
    """base.py:93-404."""

    def __init__(self, config):
        tempd = None
        if config.train_dir is None:
            tempd = tempfile.mkdtemp(prefix='tmp-' + time.strftime('%Y%m%d-%H%M%S') + '-',
                                     suffix='-' + urlify(config.train_tag) if config.train_tag else '')
        self.train_dir = config.train_dir or tempd
        os.makedirs(self.train_dir, exist_ok=True)
        config_file = os.path.join(self.train_dir, 'config.json')
        if not os.path.exists(config_file):
            self.config.dump(config_file)
        config_pkl = os.path.join(self.train_dir, 'config.pkl')
        if not os.path.exists(config_pkl):
            with open(config_pkl, 'wb') as f:
                pickle.dump(config, f)
        self._global_step = 0
        self._learning_rate_scale = 1.0

    # ---- checkpoints (base.py:188-253) ------------------------------------------------
    def state_dict(self):
        raise NotImplementedError

    def load_state_dict(self, state):
        raise NotImplementedError

    def save_model_checkpoint(self, checkpoint_dir):
        model_name = type(self).__name__ or "Model"
        checkpoint_dir = os.path.join(checkpoint_dir, "model")
        os.makedirs(checkpoint_dir, exist_ok=True)
        path = os.path.join(checkpoint_dir, '%s-%d.pt' % (model_name, self.current_step))
        # tf.train.Saver(tf.all_variables()) (base.py:240-251) also holds the optimizer slots (Adam m / v, beta
        # powers = global_step here) -- without them a resumed run restarts the moments at zero with t ~ 1
        # (first updates ~ lr * sign(g)); the learning-rate scale and the augmentation stream ride along.
        from ..engine import optimizer_state
        ck = {'variables': self.state_dict(), 'global_step': self.current_step,
              'optimizer': [optimizer_state(e) for e in self._optimizer_engines()],
              'learning_rate_scale': self._learning_rate_scale}
        if getattr(self, 'flip_rng', None) is not None:
            ck['flip_seed'] = getattr(self, 'flip_seed', None)
            ck['flip_rng_state'] = self.flip_rng.get_state()
        # dropout sites: Philox key + draw counter, so a resumed run continues the mask stream instead of replaying it
        # (the key carries the SAVING rank's offset, models/gaze_rnn.py dropout_seed: the rank is stored so that a loader
        # on another rank can re-derive its own key instead of copying this one)
        from .. import dist as rdist
        ck['dropout'] = [{'seed': d.seed, 'draws': d.draws, 'keep_prob': d.keep_prob, 'rank': rdist.env_world()[0]}
                         for d in self._dropout_sites()]
        torch.save(ck, path)
        with open(os.path.join(checkpoint_dir, 'checkpoint'), 'w') as f:
            f.write(os.path.basename(path) + '\n')
        log.info(" [Checkpoint] Saved checkpoints into %s !", path)
        return path

    def load_model_from_checkpoint_file(self, checkpoint_path):
        ck = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        self.load_state_dict(ck['variables'])
        self._global_step = int(ck.get('global_step', 0))
        self._learning_rate_scale = float(ck.get('learning_rate_scale', self._learning_rate_scale))
        from ..engine import load_optimizer_state
        for e, st in zip(self._optimizer_engines(), ck.get('optimizer', [])):
            load_optimizer_state(e, st)
        if ck.get('flip_rng_state') is not None and getattr(self, 'flip_rng', None) is not None:
            self.flip_rng.set_state(ck['flip_rng_state'])
            self.flip_seed = ck.get('flip_seed', getattr(self, 'flip_seed', None))
        sites, saved = self._dropout_sites(), ck.get('dropout')
        if saved is None:
            # a checkpoint written before the dropout state was stored: keep this rank's freshly derived key, draws = 0
            if sites:
                log.warning(" [Checkpoint] %s holds no dropout state: %d dropout site(s) start a new mask stream",
                            checkpoint_path, len(sites))
            saved = []
        elif len(saved) != len(sites):
            raise ValueError('checkpoint holds %d dropout site(s), this model has %d' % (len(saved), len(sites)))
        from .. import dist as rdist
        from .gaze_rnn import DROPOUT_RANK_STRIDE
        rank = rdist.env_world()[0]
        for d, st in zip(sites, saved):
            # every rank loads the one saved file: move the key from the saving rank's stream to this rank's, so
            # data-parallel replicas keep drawing DIFFERENT masks after a resume; the draw counter continues
            d.seed = (int(st['seed']) + DROPOUT_RANK_STRIDE * (rank - int(st.get('rank', 0)))) & 0x7fffffffffffffff
            d.draws = int(st['draws'])
        log.info(" [Checkpoint] Successfully loaded from %s", checkpoint_path)

    def _dropout_sites(self):
        """DropoutSite objects of the model's engine(s) (fc-GRU: the projected features; cascade: fc1), if any."""
        sites = []
        e = getattr(self, 'engine', None)
        for obj in (e, getattr(e, 'net', None)):
            d = getattr(obj, 'dropout', None)
            if d is not None and hasattr(d, 'draws') and d not in sites:
                sites.append(d)
        return sites

    def load_model_checkpoint(self, checkpoint_dir):
        checkpoint_dir = os.path.join(checkpoint_dir, "model")
        index = os.path.join(checkpoint_dir, 'checkpoint')
        if not os.path.exists(index):
            return False
        with open(index) as f:
            name = f.read().strip()
        self.load_model_from_checkpoint_file(os.path.join(checkpoint_dir, name))
        return True

    def reload_checkpoint(self):
        if self.load_model_checkpoint(self.train_dir):
            log.info(" [Checkpoint] Successfully loaded model (step %d, lr %.6f)", self.current_step,
                     self.current_learning_rate)
            return True
        log.error(" [Checkpoint] Failed to load model !! (starting from scratch)")
        return False

    def _optimizer_engines(self):
        """Device engines that own optimizer slots (flat_params + adam_m / ...)."""
        e = getattr(self, 'engine', None)
        e = getattr(e, 'net', e)
        return [e] if e is not None and hasattr(e, 'flat_params') else []

    # ---- training loop (base.py:330-358) ---------------------------------------------
    def fit(self):
        assert self.current_learning_rate > 0
        self.reload_checkpoint()
        step = 0
        while step <= self.config.max_steps:
            step = self.single_step(train_mode=True)
            assert step > 0
            if np.mod(step, self.config.steps_per_checkpoint) == 0:
                self.save_model_checkpoint(self.train_dir)
            if np.mod(step, self.config.steps_per_validation) == 0:
                self.single_step(train_mode=False)
            if np.mod(step, self.config.steps_per_evaluation) == 0:
                self.generate_and_evaluate(self.data_sets.valid)

    def report_evaluate_summary(self, batch_scores):
        for metric, score in batch_scores.items():
            log.info('evaluation/%s @ step %d : %f', metric, self.current_step, score)

    @property
    def current_step(self):
        return self._global_step

    @property
    def current_learning_rate(self):
        return self.config.initial_learning_rate * self._learning_rate_scale

    def decay_learning_rate(self, decay_factor):
        self._learning_rate_scale *= decay_factor
        return self.current_learning_rate
