"""Mirror of /root/reference/models/model_util.py (numpy helpers + device softmax/xent)."""
import numpy as np


def normalize_map(t):
    """model_util.py:20-38: each map min-max scaled to [0, 1] (numpy)."""
    t = np.array(t, copy=True)
    if t.ndim not in (3, 4):
        raise ValueError("Unsupported shape : {}".format(t.shape))
    assert t.dtype == np.float32 or t.dtype == float
    for i in range(len(t)):
        t[i] -= t[i].min()
        if t[i].max() > 0:
            t[i] /= t[i].max()
    return t


def normalize_probability_map(t):
    """model_util.py:40-58: each frame divided by its sum (no epsilon: an all-zero
    frame gives NaN labels exactly as in the reference)."""
    assert t.dtype == np.float32 or t.dtype == float
    t = np.array(t, copy=True)
    if t.ndim == 3:
        t /= t.reshape(t.shape[0], -1).sum(-1)[:, None, None]
    elif t.ndim == 4:
        t /= t.reshape(t.shape[0], t.shape[1], -1).sum(-1)[:, :, None, None]
    else:
        raise ValueError("Unsupported shape : {}".format(t.shape))
    return t


def softmax_2d(logits):
    """tf_softmax_2d (model_util.py:61-64) on a device tensor [..., H, W]."""
    from ..engine import softmax_xent
    return softmax_xent(logits.contiguous())[0]


def softmax_cross_entropy_with_logits_2d(logits, labels):
    """tf_softmax_cross_entropy_with_logits_2d (model_util.py:66-72): per-frame loss."""
    from ..engine import softmax_xent
    return softmax_xent(logits.contiguous(), labels.contiguous(), want_probs=False)[1]
