"""TF-checkpoint variable mapping.

A trained reference checkpoint exported to ``{tf_variable_name: ndarray}`` (e.g.
``np.savez(path, **{v.name: session.run(v) for v in tf.global_variables()})`` on a machine that
has TensorFlow 1.x) maps onto this package's state dict as follows
(/root/reference/models/gaze_grcn.py:215,234-237,64-81,292-314,325; gaze_rnn.py:412-433):

  RGP/proj_c3d_W, RGP/proj_c3d_b                  -> proj_c3d_W, proj_c3d_b
  RGP/RCNBottom/GRU_Conv_{Wz,Uz,Wr,Ur,W,U}        -> GRU_Conv_{Wz,Uz,Wr,Ur,W,U}
  RGP/Upsampling/weight{1,2,3}                    -> weight{1,2,3}
  RGP/out_W, RGP/out_b                            -> out_W, out_b
  RGP/batch_normalization{,_1,..,_T-1}/{gamma,beta} -> bn_gamma[t], bn_beta[t]   (one layer per timestep)

Optimizer slots (``.../Adam``, ``.../Adam_1``), ``global_step`` and the moving statistics (never
updated by the reference: moving_mean 0, moving_variance 1) are not parameters.

The other three graphs of the gaze family (``model=`` of import_tf_variables / export_tf_variables):

  'gaze_rnn'  (fc-GRU, gaze_rnn.py:294-320; no enclosing scope, the cell lives in scope "RNN"):
      proj_c3d_W, proj_c3d_b                                  -> proj_c3d_W, proj_c3d_b
      RNN/gru_cell/gates/{kernel,bias}                        -> gates_kernel [1568+1617, 2*1617] ([r | u]), gates_bias
      RNN/gru_cell/candidate/{kernel,bias}                    -> candidate_kernel, candidate_bias
        (TF <= 1.1 spelling RNN/GRUCell/{Gates,Candidate}/Linear/{Matrix,Bias} is accepted too)
      RNN/proj_out_W, RNN/proj_out_b                          -> proj_out_W, proj_out_b
  'shallownet'  (saliency_shallownet.py:92-185, tf.contrib.layers scopes; what
      initialize_pretrained_shallownet copies from a separate checkpoint, gaze_rnn.py:412-433):
      ShallowNet/conv{1,2,3}/{weights,biases}, ShallowNet/fc{1,2}/{weights,biases} -> conv1_w, conv1_b, ... fc2_b
  'gaze_grcn_cascade'  (gaze_grcn_cascade.py:267-423):
      proj_c3d_W, proj_c3d_b; RCNBottom/GRU_Conv_*; Upsampling/weight; RCNGaze/GRU_Conv_*;
      RCNGaze/LastProjection/fc{1,2}/{weights,biases|bias}    -> LastProjection/fc{1,2}_{w,b}
      ShallowNet/*                                            -> ShallowNet/<shallownet names>
"""
import re

import numpy as np

_SIMPLE = {
    'RGP/proj_c3d_W': 'proj_c3d_W', 'RGP/proj_c3d_b': 'proj_c3d_b',
    'RGP/Upsampling/weight1': 'weight1', 'RGP/Upsampling/weight2': 'weight2', 'RGP/Upsampling/weight3': 'weight3',
    'RGP/out_W': 'out_W', 'RGP/out_b': 'out_b',
}
_GRU = ('Wz', 'Uz', 'Wr', 'Ur', 'W', 'U')
_BN = re.compile(r'^RGP/batch_normalization(?:_(\d+))?/(gamma|beta|moving_mean|moving_variance)$')


def _strip(name):
    return name[:-2] if name.endswith(':0') else name


def import_tf_variables(tf_vars, n_steps=None):
    """{tf name: array} -> {state-dict name: float32 array}; raises KeyError listing what is missing."""
    src = {_strip(k): np.asarray(v) for k, v in dict(tf_vars).items()}
    out, bn = {}, {}
    for name, arr in src.items():
        if 'Adam' in name or name.endswith('global_step'):
            continue
        if name in _SIMPLE:
            out[_SIMPLE[name]] = arr.astype(np.float32)
            continue
        m = re.match(r'^RGP/RCNBottom/GRU_Conv_(Wz|Uz|Wr|Ur|W|U)(?:_\d+)?$', name)
        if m:
            out['GRU_Conv_' + m.group(1)] = arr.astype(np.float32)
            continue
        m = _BN.match(name)
        if m:
            t = int(m.group(1) or 0)
            if m.group(2) in ('moving_mean', 'moving_variance'):
                expect = 0.0 if m.group(2) == 'moving_mean' else 1.0
                if not np.allclose(arr, expect):
                    raise ValueError('%s is not at its initial value: the inference-mode BN of the reference '
                                     'assumes it never moves (SURVEY 9-Q1)' % name)
                continue
            bn.setdefault(m.group(2), {})[t] = arr.astype(np.float32)
    if bn:
        T = n_steps if n_steps is not None else 1 + max(max(d) for d in bn.values())
        for key in ('gamma', 'beta'):
            missing = [t for t in range(T) if t not in bn.get(key, {})]
            if missing:
                raise KeyError('batch-norm %s missing for timesteps %s' % (key, missing))
            out['bn_' + key] = np.stack([bn[key][t] for t in range(T)], 0)
    need = set(_SIMPLE.values()) | {'GRU_Conv_' + g for g in _GRU} | {'bn_gamma', 'bn_beta'}
    missing = sorted(need - set(out))
    if missing:
        raise KeyError('TF checkpoint lacks variables for: %s' % ', '.join(missing))
    return out


def export_tf_variables(state):
    """Inverse mapping (state dict -> TF names), e.g. to hand weights trained here back to the reference."""
    inv = {v: k for k, v in _SIMPLE.items()}
    out = {}
    for k, v in state.items():
        if k in inv:
            out[inv[k]] = np.asarray(v)
        elif k.startswith('GRU_Conv_'):
            out['RGP/RCNBottom/' + k] = np.asarray(v)
        elif k in ('bn_gamma', 'bn_beta'):
            for t, row in enumerate(np.asarray(v)):
                layer = 'batch_normalization' + ('_%d' % t if t else '')
                out['RGP/%s/%s' % (layer, k[3:])] = row
    return out


# ---------------------------------------------------------------------------------------------------------------
# fc-GRU, ShallowNet and cascade graphs
_FCGRU = {
    'proj_c3d_W': 'proj_c3d_W', 'proj_c3d_b': 'proj_c3d_b', 'RNN/proj_out_W': 'proj_out_W', 'RNN/proj_out_b': 'proj_out_b',
    'RNN/gru_cell/gates/kernel': 'gates_kernel', 'RNN/gru_cell/gates/bias': 'gates_bias',
    'RNN/gru_cell/candidate/kernel': 'candidate_kernel', 'RNN/gru_cell/candidate/bias': 'candidate_bias',
}
_FCGRU_OLD = {     # TF <= 1.1 spelling of the same GRUCell variables
    'RNN/GRUCell/Gates/Linear/Matrix': 'gates_kernel', 'RNN/GRUCell/Gates/Linear/Bias': 'gates_bias',
    'RNN/GRUCell/Candidate/Linear/Matrix': 'candidate_kernel', 'RNN/GRUCell/Candidate/Linear/Bias': 'candidate_bias',
}
_SHALLOW_LAYERS = ('conv1', 'conv2', 'conv3', 'fc1', 'fc2')
_W_NAMES, _B_NAMES = ('weights', 'W', 'kernel'), ('biases', 'bias', 'b')


def _skip(name):
    return 'Adam' in name or 'RMSProp' in name or 'Momentum' in name or name.endswith('global_step') or 'is_training' in name


def import_shallownet_variables(tf_vars, scope='ShallowNet'):
    """``{'ShallowNet/conv1/weights': ...}`` -> ``{'conv1_w': ..., 'conv1_b': ...}`` (10 arrays); the copy
    initialize_pretrained_shallownet makes from a pre-trained checkpoint (gaze_rnn.py:412-433)."""
    src = {_strip(k): np.asarray(v) for k, v in dict(tf_vars).items()}
    out = {}
    for layer in _SHALLOW_LAYERS:
        for suffix, names in (('_w', _W_NAMES), ('_b', _B_NAMES)):
            hit = [n for n in names if '%s/%s/%s' % (scope, layer, n) in src]
            if hit:
                out[layer + suffix] = src['%s/%s/%s' % (scope, layer, hit[0])].astype(np.float32)
    missing = sorted({l + s for l in _SHALLOW_LAYERS for s in ('_w', '_b')} - set(out))
    if missing:
        raise KeyError('TF checkpoint lacks %s variables for: %s' % (scope, ', '.join(missing)))
    return out


def export_shallownet_variables(params, scope='ShallowNet'):
    out = {}
    for layer in _SHALLOW_LAYERS:
        out['%s/%s/weights' % (scope, layer)] = np.asarray(params[layer + '_w'])
        out['%s/%s/biases' % (scope, layer)] = np.asarray(params[layer + '_b'])
    return out


def import_fcgru_variables(tf_vars):
    src = {_strip(k): np.asarray(v) for k, v in dict(tf_vars).items() if not _skip(_strip(k))}
    out = {}
    for table in (_FCGRU, _FCGRU_OLD):
        for tf_name, key in table.items():
            if tf_name in src and key not in out:
                out[key] = src[tf_name].astype(np.float32)
    missing = sorted(set(_FCGRU.values()) - set(out))
    if missing:
        raise KeyError('TF checkpoint lacks fc-GRU variables for: %s' % ', '.join(missing))
    return out


def export_fcgru_variables(state):
    inv = {v: k for k, v in _FCGRU.items()}
    return {inv[k]: np.asarray(v) for k, v in state.items() if k in inv}


def import_cascade_variables(tf_vars):
    """-> the cascade model's state dict: flat keys, ShallowNet variables as 'ShallowNet/<name>'."""
    src = {_strip(k): np.asarray(v) for k, v in dict(tf_vars).items() if not _skip(_strip(k))}
    out = {}
    for name in ('proj_c3d_W', 'proj_c3d_b', 'Upsampling/weight'):
        if name in src:
            out[name] = src[name].astype(np.float32)
    for name, arr in src.items():
        m = re.match(r'^(RCNBottom|RCNGaze)/GRU_Conv_(Wz|Uz|Wr|Ur|W|U)(?:_\d+)?$', name)
        if m:
            out['%s/GRU_Conv_%s' % (m.group(1), m.group(2))] = arr.astype(np.float32)
            continue
        m = re.match(r'^(?:RCNGaze/)?LastProjection/(fc[12])/(\w+)$', name)
        if m and m.group(2) in _W_NAMES + _B_NAMES:
            out['LastProjection/%s_%s' % (m.group(1), 'w' if m.group(2) in _W_NAMES else 'b')] = arr.astype(np.float32)
    need = {'proj_c3d_W', 'proj_c3d_b', 'Upsampling/weight'} | {'%s/GRU_Conv_%s' % (c, g) for c in ('RCNBottom', 'RCNGaze') for g in _GRU} \
        | {'LastProjection/fc%d_%s' % (i, s) for i in (1, 2) for s in 'wb'}
    missing = sorted(need - set(out))
    if missing:
        raise KeyError('TF checkpoint lacks cascade variables for: %s' % ', '.join(missing))
    out.update({'ShallowNet/' + k: v for k, v in import_shallownet_variables(src).items()})
    return out


def export_cascade_variables(state):
    out = {}
    shallow = {}
    for k, v in state.items():
        if k.startswith('ShallowNet/'):
            shallow[k[len('ShallowNet/'):]] = v
        elif k.startswith('LastProjection/'):
            layer, kind = k[len('LastProjection/'):].split('_')
            out['RCNGaze/LastProjection/%s/%s' % (layer, 'weights' if kind == 'w' else 'biases')] = np.asarray(v)
        else:
            out[k] = np.asarray(v)
    out.update(export_shallownet_variables(shallow))
    return out


_IMPORTERS = {'gaze_rnn': import_fcgru_variables, 'shallownet': import_shallownet_variables,
              'gaze_framewise_shallownet': import_shallownet_variables, 'gaze_grcn_cascade': import_cascade_variables}
_EXPORTERS = {'gaze_rnn': export_fcgru_variables, 'shallownet': export_shallownet_variables,
              'gaze_framewise_shallownet': export_shallownet_variables, 'gaze_grcn_cascade': export_cascade_variables}


def import_model_variables(model, tf_vars, n_steps=None):
    """Dispatch on the reference's --model name (train_gaze.py:41-69)."""
    if model in ('gaze_grcn', 'gaze_grcn77'):
        return import_tf_variables(tf_vars, n_steps)
    return _IMPORTERS[model](tf_vars)


def export_model_variables(model, state):
    if model in ('gaze_grcn', 'gaze_grcn77'):
        return export_tf_variables(state)
    return _EXPORTERS[model](state)


def load_tf_export(path):
    """An exported checkpoint file: ``.npz`` of {tf name: array} (names may carry ':0')."""
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}
