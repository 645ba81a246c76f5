"""torch-CPU restatement of the reference graphs (oracle; test infrastructure).

Second, independent restatement (library convolutions, fp32 or fp64) of what
``np_ops``/``grcn`` state with direct loops; also the ``cpu_baseline`` leg of
``bench.py`` ("reference-equivalent CPU restatement, TF1.x unavailable
offline": unfused, T-unrolled, op for op as the TF graph) and the source of
gradients (autograd) for the backward parity tests.

Reference lines followed:
  gaze_grcn graph      /root/reference/models/gaze_grcn.py:95-129,173-376
  loss / softmax       /root/reference/models/gaze_rnn.py:149-159,363-408, model_util.py:61-72
  optimizer            /root/reference/models/base.py:262-308, gaze_rnn.py:436-444 (TF Adam, 9-Q9)
  fc-GRU (config 2)    /root/reference/models/gaze_rnn.py:211-360 + TF-1.x GRUCell semantics
  shallownet (cfg 1)   /root/reference/models/saliency_shallownet.py:74-216
  C3D conv stack       /root/reference/C3D/.../c3d_prototxt/feature_extration.prototxt:22-342
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _t(x, dtype):
    return torch.as_tensor(x, dtype=dtype)


# --------------------------------------------------------------------------- gaze_grcn
def conv2d_same(x_nhwc, w_hwio):
    kh = w_hwio.shape[0]
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w_hwio.permute(3, 2, 0, 1), padding=kh // 2)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose(x_nhwc, f_hwoi, stride, padding):
    """tf.nn.conv2d_transpose with filter [kh,kw,out,in] (gaze_grcn.py:326-358)."""
    w = f_hwoi.permute(3, 2, 0, 1)            # torch: [in, out, kh, kw]
    pad = 0
    if padding == 'SAME':                     # TF SAME: output = input*stride, crop pad_before
        k, n = f_hwoi.shape[0], x_nhwc.shape[1]
        total = max((n - 1) * stride + k - n * stride, 0)
        assert total % 2 == 0, 'asymmetric SAME crop not needed by any reference layer'
        pad = total // 2
    y = F.conv_transpose2d(x_nhwc.permute(0, 3, 1, 2), w, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1)


def grcn_cell(x, h, p):
    """gaze_grcn.py:108-129."""
    u = torch.sigmoid(conv2d_same(x, p['GRU_Conv_Wz']) + conv2d_same(h, p['GRU_Conv_Uz']))
    r = torch.sigmoid(conv2d_same(x, p['GRU_Conv_Wr']) + conv2d_same(h, p['GRU_Conv_Ur']))
    c = torch.tanh(conv2d_same(x, p['GRU_Conv_W']) + conv2d_same(r * h, p['GRU_Conv_U']))
    return u * h + (1 - u) * c


def grcn_forward(c3d_input, p, want_hidden=False):
    """gaze_grcn.py:173-376 -> logits [B,T,49,49] (dropout inert, 9-Q2)."""
    b, t = c3d_input.shape[:2]
    xr = c3d_input.permute(0, 1, 3, 4, 2)
    emb = (xr.reshape(-1, 1024) @ p['proj_c3d_W'] + p['proj_c3d_b']).reshape(b, t, 7, 7, -1)
    s = p['GRU_Conv_Uz'].shape[-1]
    h = torch.zeros(b, 7, 7, s, dtype=c3d_input.dtype)
    hs = []
    for i in range(t):
        h = grcn_cell(emb[:, i], h, p)
        hs.append(h)
    outs = []
    inv = 1.0 / math.sqrt(1.0 + BN_EPS)
    for i in range(t):
        y = p['bn_gamma'][i] * hs[i] * inv + p['bn_beta'][i]           # inference BN, 9-Q1
        y = conv2d_transpose(y, p['weight1'], 3, 'VALID')
        y = conv2d_transpose(y, p['weight2'], 2, 'VALID')
        y = conv2d_transpose(y, p['weight3'], 1, 'SAME')
        z = y.reshape(-1, y.shape[-1]) @ p['out_W'] + p['out_b']
        outs.append(z.reshape(b, 49, 49))
    logits = torch.stack(outs, 1)
    if want_hidden:
        return logits, torch.stack(hs, 1), emb
    return logits


def softmax_maps(logits):
    b, t, h, w = logits.shape
    return torch.softmax(logits.reshape(b, t, h * w), -1).reshape(b, t, h, w)


def gaze_loss(logits, gt, loss_type='xentropy'):
    """gaze_rnn.py:363-408."""
    b, t, h, w = logits.shape
    z = logits.reshape(b, t, h * w)
    g = gt.reshape(b, t, h * w)
    if loss_type == 'xentropy':
        tot = -(g * torch.log_softmax(z, -1)).sum()
    elif loss_type == 'l2':
        tot = 0.5 * ((z - g) ** 2).sum()
    else:
        raise NotImplementedError(loss_type)
    return tot / float(b * t)


def grcn_loss_and_grads(c3d_input, gt, params, dtype=torch.float64, loss_type='xentropy'):
    """loss + d loss / d params by autograd (what tf.gradients does, base.py:278-281)."""
    p = {k: _t(v, dtype).clone().requires_grad_(True) for k, v in params.items()}
    logits = grcn_forward(_t(c3d_input, dtype), p)
    ls = gaze_loss(logits, _t(gt, dtype), loss_type)
    ls.backward()
    return ls.item(), logits.detach(), {k: v.grad.detach() for k, v in p.items()}


# --------------------------------------------------------------------------- optimizer
def learning_rate(lr0, decay, step, decay_steps=500):
    """tf.train.exponential_decay(staircase=True) (gaze_rnn.py:436-444)."""
    return lr0 * decay ** (step // decay_steps)


def clip_by_global_norm(grads, clip):
    """tf.clip_by_global_norm (base.py:286-288): g * clip / max(norm, clip)."""
    norm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    scale = clip / max(norm, clip)
    return {k: g * scale for k, g in grads.items()}, norm


def adam_step_tf(params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer.apply_gradients (base.py:269,294-297; 9-Q9):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t * m / (sqrt(v)+eps); t = step+1."""
    t = step + 1
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    for k in params:
        g = grads[k]
        m[k] = b1 * m[k] + (1 - b1) * g
        v[k] = b2 * v[k] + (1 - b2) * g * g
        params[k] = params[k] - lr_t * m[k] / (torch.sqrt(v[k]) + eps)
    return params, m, v


def momentum_step_tf(params, grads, accum, lr, momentum=0.9):
    """tf.train.MomentumOptimizer(lr, momentum=0.9) (base.py:272-273): accum = momentum*accum + g; var -= lr*accum."""
    for k in params:
        accum[k] = momentum * accum[k] + grads[k]
        params[k] = params[k] - lr * accum[k]
    return params, accum


def rmsprop_step_tf(params, grads, ms, mom, lr, decay=0.9, momentum=0.9, eps=1e-10):
    """tf.train.RMSPropOptimizer(lr, momentum=0.9) (base.py:270-271), TF defaults decay 0.9, epsilon 1e-10;
    slots start at ms = 1, mom = 0:  ms = decay*ms + (1-decay) g^2;  mom = momentum*mom + lr*g/sqrt(ms+eps);
    var -= mom."""
    for k in params:
        g = grads[k]
        ms[k] = decay * ms[k] + (1 - decay) * g * g
        mom[k] = momentum * mom[k] + lr * g / torch.sqrt(ms[k] + eps)
        params[k] = params[k] - mom[k]
    return params, ms, mom


# --------------------------------------------------------------------------- dropout
def dropout(x, keep_prob, mask):
    """tf.nn.dropout with the draw made explicit: x / keep_prob * mask (mask = floor(keep_prob + u) in {0,1})."""
    return x / keep_prob * mask.to(x.dtype)


def philox4x32_10(counter, key):
    """Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
    generator behind TF's and cuRAND's stateless streams).  counter [..., 4], key [..., 2] uint32 -> [..., 4]."""
    import numpy as np
    c = np.array(counter, dtype=np.uint64) & 0xFFFFFFFF
    k = np.array(key, dtype=np.uint64) & 0xFFFFFFFF
    c = np.broadcast_to(c, np.broadcast_shapes(c.shape[:-1], k.shape[:-1]) + (4,)).copy()
    k = np.broadcast_to(k, c.shape[:-1] + (2,)).copy()
    for _ in range(10):
        p0 = 0xD2511F53 * c[..., 0]
        p1 = 0xCD9E8D57 * c[..., 2]
        n0 = ((p1 >> 32) ^ c[..., 1] ^ k[..., 0]) & 0xFFFFFFFF
        n1 = p1 & 0xFFFFFFFF
        n2 = ((p0 >> 32) ^ c[..., 3] ^ k[..., 1]) & 0xFFFFFFFF
        n3 = p0 & 0xFFFFFFFF
        c = np.stack([n0, n1, n2, n3], -1)
        k = np.stack([(k[..., 0] + 0x9E3779B9) & 0xFFFFFFFF, (k[..., 1] + 0xBB67AE85) & 0xFFFFFFFF], -1)
    return c.astype(np.uint32)


def dropout_mask(n, keep_prob, seed, offset=0):
    """The keep mask rgp_dropout_mask draws: element i = word i&3 of Philox block (offset + i//4) under key `seed`;
    u = (word >> 8) / 2^24 (float32); keep = floor(keep_prob + u) >= 1 (tf.nn.dropout's rule)."""
    import numpy as np
    nb = (n + 3) // 4
    ctr = np.uint64(offset) + np.arange(nb, dtype=np.uint64)
    counter = np.stack([ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), np.zeros_like(ctr), np.zeros_like(ctr)], -1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    w = philox4x32_10(counter, key).reshape(-1)[:n]
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.floor(np.float32(keep_prob) + u) >= 1.0).astype(np.uint8)


# --------------------------------------------------------------------------- fc-GRU (config 2)
def tf_gru_cell(x, h, gate_kernel, gate_bias, cand_kernel, cand_bias):
    """TF-1.x rnn_cell.GRUCell.call: [r,u] = sigmoid([x,h] Wg + bg);
    c = tanh([x, r*h] Wc + bc); h' = u*h + (1-u)*c."""
    n = h.shape[1]
    ru = torch.sigmoid(torch.cat([x, h], 1) @ gate_kernel + gate_bias)
    r, u = ru[:, :n], ru[:, n:]
    c = torch.tanh(torch.cat([x, r * h], 1) @ cand_kernel + cand_bias)
    return u * h + (1 - u) * c


def fcgru_forward(c3d_input, p, gh=49, gw=49, keep_prob=1.0, drop_mask=None):
    """gaze_rnn.py:284-357 (the ShallowNet branch :256-275 does not reach the output).
    keep_prob < 1 with drop_mask [B*T*49, 32]: the training-time tf.nn.dropout on c3d_embedded (:302-303)."""
    b, t = c3d_input.shape[:2]
    xr = c3d_input.permute(0, 1, 3, 4, 2)
    emb = xr.reshape(-1, 1024) @ p['proj_c3d_W'] + p['proj_c3d_b']                        # [(B*T*49), 32]
    if drop_mask is not None and keep_prob < 1.0:
        emb = dropout(emb, keep_prob, torch.as_tensor(drop_mask).reshape(emb.shape))
    emb = emb.reshape(b, t, -1)                                                           # [B,T,7*7*32]
    n = p['proj_out_W'].shape[0]
    h = torch.zeros(b, n, dtype=c3d_input.dtype)
    outs = []
    for i in range(t):
        h = tf_gru_cell(emb[:, i], h, p['gates_kernel'], p['gates_bias'], p['candidate_kernel'], p['candidate_bias'])
        outs.append((h @ p['proj_out_W'] + p['proj_out_b']).reshape(b, gh, gw))
    return torch.stack(outs, 1)


# --------------------------------------------------------------------------- shallownet (config 1)
def _max_pool_same(x_nchw, k, s):
    h, w = x_nchw.shape[2:]
    oh, ow = -(-h // s), -(-w // s)
    ph, pw = max((oh - 1) * s + k - h, 0), max((ow - 1) * s + k - w, 0)
    x = F.pad(x_nchw, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float('-inf'))
    return F.max_pool2d(x, k, s)


def shallownet_forward(images_nhwc, p):
    """saliency_shallownet.py:74-216 with dropout off -> [N,49,49]."""
    x = images_nhwc.permute(0, 3, 1, 2)
    x = _max_pool_same(torch.relu(F.conv2d(x, p['conv1_w'].permute(3, 2, 0, 1), p['conv1_b'])), 2, 2)
    x = _max_pool_same(torch.relu(F.conv2d(x, p['conv2_w'].permute(3, 2, 0, 1), p['conv2_b'])), 3, 2)
    x = _max_pool_same(torch.relu(F.conv2d(x, p['conv3_w'].permute(3, 2, 0, 1), p['conv3_b'])), 3, 2)
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)          # NHWC flatten order
    x = torch.relu(x @ p['fc1_w'] + p['fc1_b'])
    x = torch.maximum(x[:, :2401], x[:, 2401:])
    x = torch.relu(x @ p['fc2_w'] + p['fc2_b'])
    x = torch.maximum(x[:, :2401], x[:, 2401:])
    return x.reshape(-1, 49, 49)


# --------------------------------------------------------------------------- cascade (config 5)
def cascade_forward(frame_images, c3d_input, p, want_all=False, keep_prob=1.0, drop_mask=None):
    """gaze_grcn_cascade.py:188-423 as intended (the top cell sees
    concat(upsampled bottom state, ShallowNet saliency), the commented block :370-377).
    frame_images [B,T,H,W,3], c3d_input [B,T,1024,7,7] -> gazemaps [B,T,49,49].
    keep_prob < 1 with drop_mask [B,T,4802]: tf.nn.dropout on relu(fc1) before the maxout (:401-402)."""
    b, t = c3d_input.shape[:2]
    sal = shallownet_forward(frame_images.reshape((-1,) + tuple(frame_images.shape[2:])), p['ShallowNet'])
    sal = sal.reshape(b, t, 49, 49, 1)                                           # :235-244
    xr = c3d_input.permute(0, 1, 3, 4, 2)
    emb = (xr.reshape(-1, 1024) @ p['proj_c3d_W'] + p['proj_c3d_b']).reshape(b, t, 7, 7, -1)   # :262-275
    bot = {k[len('RCNBottom/'):]: v for k, v in p.items() if k.startswith('RCNBottom/')}
    top = {k[len('RCNGaze/'):]: v for k, v in p.items() if k.startswith('RCNGaze/')}
    h = torch.zeros(b, 7, 7, bot['GRU_Conv_Uz'].shape[-1], dtype=c3d_input.dtype)
    g = torch.zeros(b, 49, 49, top['GRU_Conv_Uz'].shape[-1], dtype=c3d_input.dtype)
    outs, ups, hs, gs = [], [], [], []
    for i in range(t):
        h = grcn_cell(emb[:, i], h, bot)                                         # :307
        up = conv2d_transpose(h, p['Upsampling/weight'], 7, 'SAME')              # :327-333
        g = grcn_cell(torch.cat([up, sal[:, i]], -1), g, top)                    # :370-379
        x = g.reshape(b, -1)                                                     # :383
        x = torch.relu(x @ p['LastProjection/fc1_w'] + p['LastProjection/fc1_b'])
        if drop_mask is not None and keep_prob < 1.0:
            x = dropout(x, keep_prob, torch.as_tensor(drop_mask).reshape(b, t, 4802)[:, i])
        x = torch.maximum(x[:, :2401], x[:, 2401:])
        x = torch.relu(x @ p['LastProjection/fc2_w'] + p['LastProjection/fc2_b'])
        x = torch.maximum(x[:, :2401], x[:, 2401:])
        outs.append(x.reshape(b, 49, 49))
        ups.append(up); hs.append(h); gs.append(g)
    maps = torch.stack(outs, 1)
    if want_all:
        return maps, dict(sal=sal[..., 0], bottom=torch.stack(hs, 1), up=torch.stack(ups, 1), top=torch.stack(gs, 1))
    return maps


# --------------------------------------------------------------------------- C3D conv stack
C3D_LAYERS = [  # name, Cin, Cout, pool (kd,k) applied AFTER relu (None = no pool)
    ('conv1a', 3, 64, (1, 2)), ('conv2a', 64, 128, (2, 2)),
    ('conv3a', 128, 256, None), ('conv3b', 256, 256, (2, 2)),
    ('conv4a', 256, 512, None), ('conv4b', 512, 512, (2, 2)),
    ('conv5a', 512, 512, None), ('conv5b', 512, 512, None),
]


def c3d_forward(video_ndhwc, p, upto='conv5b', want_all=False):
    """prototxt:22-342.  video [N,16,112,112,3] (already mean-subtracted) ->
    conv5b after ReLU folded to [N,1024,7,7] with channel index c*2+d
    (gaze_rnn.py:494-497; crc_input_data_seq.py:326-330)."""
    x = video_ndhwc.permute(0, 4, 1, 2, 3)     # NCDHW (Caffe)
    acts = {}
    for name, _, _, pool in C3D_LAYERS:
        w = p[name + '_w'].permute(4, 3, 0, 1, 2)    # DHWIO -> [Co,Ci,kd,kh,kw]
        x = torch.relu(F.conv3d(x, w, p[name + '_b'], padding=1))
        if pool is not None:
            x = F.max_pool3d(x, (pool[0], pool[1], pool[1]), ceil_mode=True)
        acts[name] = x
        if name == upto:
            break
    n, c, d, h, w_ = x.shape
    feat = x.reshape(n, c * d, h, w_) if upto == 'conv5b' else x
    return (feat, acts) if want_all else feat
