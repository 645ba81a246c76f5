"""float64 numpy restatement of the gaze_grcn graph (oracle; test infrastructure).

Follows /root/reference/models/gaze_grcn.py:173-376 (network),
gaze_grcn.py:95-129 (GRU_RCN_Cell.__call__), gaze_rnn.py:149-159 (per-frame
softmax), gaze_rnn.py:363-408 + model_util.py:66-72 (loss).

Parameter dictionary (names = the reference's TF variable names):
  proj_c3d_W [1024,P]  proj_c3d_b [P]                     (gaze_grcn.py:234-237)
  GRU_Conv_Wz/Wr/W [3,3,P,S]  GRU_Conv_Uz/Ur/U [3,3,S,S]  (gaze_grcn.py:64-81)
  bn_gamma [T,S]  bn_beta [T,S]     one BN layer per timestep (gaze_grcn.py:325, 9-Q1)
  weight1 [5,5,64,S]  weight2 [5,5,32,64]  weight3 [7,7,12,32]   (gaze_grcn.py:292-310)
  out_W [12,1]  out_b [1]                                 (gaze_grcn.py:311-314)
P = dim_cnn_proj (512), S = rnn_state_size (128) in the reference.
"""
import numpy as np

from . import np_ops as ops

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon


def gru_rcn_cell(x, h, p):
    """One GRU_RCN_Cell step (gaze_grcn.py:108-129). x [B,7,7,P], h [B,7,7,S]."""
    u = ops.sigmoid(ops.conv2d_same(x, p['GRU_Conv_Wz']) + ops.conv2d_same(h, p['GRU_Conv_Uz']))
    r = ops.sigmoid(ops.conv2d_same(x, p['GRU_Conv_Wr']) + ops.conv2d_same(h, p['GRU_Conv_Ur']))
    c = np.tanh(ops.conv2d_same(x, p['GRU_Conv_W']) + ops.conv2d_same(r * h, p['GRU_Conv_U']))
    new_h = u * h + (1.0 - u) * c            # gaze_grcn.py:127 (u gates the OLD state, 9-Q3)
    return new_h, dict(u=u, r=r, c=c)


def head_frame(h_t, t, p):
    """BN + three transposed convs + 12->1 for one timestep (gaze_grcn.py:318-366)."""
    y = ops.batchnorm_inference(h_t, p['bn_gamma'][t], p['bn_beta'][t], eps=BN_EPS)
    d1 = ops.conv2d_transpose(y, p['weight1'], 3, 'VALID', (23, 23))
    d2 = ops.conv2d_transpose(d1, p['weight2'], 2, 'VALID', (49, 49))
    d3 = ops.conv2d_transpose(d2, p['weight3'], 1, 'SAME', (49, 49))
    b = d3.shape[0]
    z = d3.reshape(-1, d3.shape[-1]) @ np.asarray(p['out_W'], np.float64) + np.asarray(p['out_b'], np.float64)
    return z.reshape(b, 49, 49), dict(bn=y, d1=d1, d2=d2, d3=d3)


def forward(c3d_input, p, want_intermediates=False):
    """create_gazeprediction_network (gaze_grcn.py:173-376), dropout keep=1
    (inert in the reference).  c3d_input [B,T,1024,7,7] ->
    logits [B,T,49,49]."""
    x = np.asarray(c3d_input, np.float64)
    bsz, tlen = x.shape[:2]
    xr = np.transpose(x, (0, 1, 3, 4, 2))                       # gaze_grcn.py:225-227
    emb = xr.reshape(-1, 1024) @ np.asarray(p['proj_c3d_W'], np.float64) + np.asarray(p['proj_c3d_b'], np.float64)
    emb = emb.reshape(bsz, tlen, 7, 7, -1)                      # gaze_grcn.py:239-250
    s = p['GRU_Conv_Uz'].shape[-1]
    h = np.zeros((bsz, 7, 7, s), np.float64)                    # zero_state, gaze_grcn.py:132-146
    hs, gates, logits, inter = [], [], [], []
    for t in range(tlen):
        h, g = gru_rcn_cell(emb[:, t], h, p)
        hs.append(h)
        gates.append(g)
    for t in range(tlen):
        z, it = head_frame(hs[t], t, p)
        logits.append(z)
        inter.append(it)
    logits = np.stack(logits, axis=1)                           # gaze_grcn.py:371-372
    if not want_intermediates:
        return logits
    return logits, dict(c3d_embedded=emb, rcn_outputs=np.stack(hs, 1), gates=gates, head=inter)


def softmax_maps(logits):
    """tf_softmax_2d per frame (model_util.py:61-64; gaze_rnn.py:149-159)."""
    b, t, h, w = logits.shape
    return ops.softmax_rows(np.asarray(logits, np.float64).reshape(b, t, h * w)).reshape(b, t, h, w)


def loss(logits, gt, loss_type='xentropy'):
    """create_loss_and_summary (gaze_rnn.py:363-408): sum over t and b of the
    per-frame loss, divided by B*T."""
    b, t, h, w = logits.shape
    z = np.asarray(logits, np.float64).reshape(b, t, h * w)
    g = np.asarray(gt, np.float64).reshape(b, t, h * w)
    if loss_type == 'xentropy':
        tot = ops.softmax_xent_rows(z, g).sum()
    elif loss_type == 'l2':                                     # tf.nn.l2_loss = sum(d^2)/2
        tot = 0.5 * ((z - g) ** 2).sum()
    else:
        raise NotImplementedError(loss_type)
    return tot / float(b * t)


def normalize_probability_map(t):
    """model_util.py:40-58: each frame divided by its own sum (no epsilon, 9-Q8)."""
    t = np.array(t, dtype=np.float64, copy=True)
    return t / t.reshape(t.shape[:-2] + (-1,)).sum(-1)[..., None, None]
