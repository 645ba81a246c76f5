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
