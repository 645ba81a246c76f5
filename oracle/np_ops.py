"""float64 numpy direct-loop primitives (oracle; test infrastructure only).

Every op loops explicitly over filter taps / output positions and contracts the
channel axis with a plain matrix product, so it shares no code with torch's
convolution library (the second restatement in ``torch_ref``).

Layouts follow the reference's TensorFlow graph: activations NHWC, conv filters
HWIO, transposed-conv filters ``[kh, kw, out, in]``
(/root/reference/models/gaze_grcn.py:64-81, 292-310).
"""
import numpy as np

F64 = np.float64


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def conv2d_same(x, w):
    """tf.nn.conv2d(x, w, [1,1,1,1], 'SAME'), odd kernel, cross-correlation.

    x [N,H,W,Ci], w [kh,kw,Ci,Co] -> [N,H,W,Co]   (gaze_grcn.py:108-125)
    """
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    n, h, wd, ci = x.shape
    kh, kw, ci2, co = w.shape
    assert ci == ci2 and kh % 2 == 1 and kw % 2 == 1
    ph, pw = kh // 2, kw // 2
    out = np.zeros((n, h, wd, co), F64)
    for a in range(kh):
        for b in range(kw):
            for y in range(h):
                yy = y + a - ph
                if yy < 0 or yy >= h:
                    continue
                for xx_o in range(wd):
                    xx = xx_o + b - pw
                    if xx < 0 or xx >= wd:
                        continue
                    out[:, y, xx_o, :] += x[:, yy, xx, :] @ w[a, b]
    return out


def conv2d_valid(x, w, bias=None):
    """tf conv2d VALID stride 1 (saliency_shallownet.py:90-130)."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    oh, ow = h - kh + 1, wd - kw + 1
    out = np.zeros((n, oh, ow, co), F64)
    for a in range(kh):
        for b in range(kw):
            out += np.einsum('nhwc,co->nhwo', x[:, a:a + oh, b:b + ow, :], w[a, b])
    if bias is not None:
        out += np.asarray(bias, F64)
    return out


def max_pool_same(x, k, s):
    """tf.nn.max_pool(ksize=k, strides=s, padding='SAME') on NHWC
    (saliency_shallownet.py:101,117,134).  SAME: out=ceil(in/s); total pad
    = max((out-1)*s+k-in,0), floor half before; padded cells are ignored."""
    x = np.asarray(x, F64)
    n, h, w, c = x.shape
    oh, ow = -(-h // s), -(-w // s)
    pt = max((oh - 1) * s + k - h, 0) // 2
    pl = max((ow - 1) * s + k - w, 0) // 2
    out = np.full((n, oh, ow, c), -np.inf, F64)
    for i in range(oh):
        for j in range(ow):
            y0, x0 = i * s - pt, j * s - pl
            ys, xs = max(y0, 0), max(x0, 0)
            ye, xe = min(y0 + k, h), min(x0 + k, w)
            out[:, i, j, :] = x[:, ys:ye, xs:xe, :].max(axis=(1, 2))
    return out


def conv2d_transpose(y, f, stride, padding, out_hw):
    """tf.nn.conv2d_transpose, filter f[kh,kw,out,in] (gaze_grcn.py:326-358).

    VALID, stride s:  out[n, s*i+a, s*j+b, o] += y[n,i,j,c] * f[a,b,o,c]
    SAME,  stride s:  same scatter shifted by the SAME padding of the forward
                      conv whose gradient this is (pad_before = total//2 with
                      total = max((in-1)*s + k - out, 0)).
    """
    y = np.asarray(y, F64)
    f = np.asarray(f, F64)
    n, h, w, c = y.shape
    kh, kw, co, ci = f.shape
    assert ci == c
    oh, ow = out_hw
    if padding == 'VALID':
        assert oh == (h - 1) * stride + kh and ow == (w - 1) * stride + kw
        pt = pl = 0
    else:
        assert h == -(-oh // stride) and w == -(-ow // stride)
        pt = max((h - 1) * stride + kh - oh, 0) // 2
        pl = max((w - 1) * stride + kw - ow, 0) // 2
    out = np.zeros((n, oh, ow, co), F64)
    for i in range(h):
        for j in range(w):
            v = y[:, i, j, :]                        # [n, ci]
            for a in range(kh):
                oy = stride * i + a - pt
                if oy < 0 or oy >= oh:
                    continue
                for b in range(kw):
                    ox = stride * j + b - pl
                    if ox < 0 or ox >= ow:
                        continue
                    out[:, oy, ox, :] += v @ f[a, b].T   # f[a,b] is [co,ci]
    return out


def batchnorm_inference(x, gamma, beta, mean=0.0, var=1.0, eps=1e-3):
    """tf.layers.batch_normalization(training=False) (gaze_grcn.py:325):
    y = gamma*(x-mean)/sqrt(var+eps)+beta on the channel axis; TF default
    epsilon 1e-3, moving mean 0 / variance 1 never updated."""
    x = np.asarray(x, F64)
    return np.asarray(gamma, F64) * (x - mean) / np.sqrt(var + eps) + np.asarray(beta, F64)


def softmax_rows(z):
    """tf.nn.softmax over the last axis (model_util.py:61-64)."""
    z = np.asarray(z, F64)
    m = z.max(axis=-1, keepdims=True)
    e = np.exp(z - m)
    return e / e.sum(axis=-1, keepdims=True)


def softmax_xent_rows(logits, labels):
    """tf.nn.softmax_cross_entropy_with_logits (model_util.py:66-72):
    -sum_p labels * log_softmax(logits) per row."""
    z = np.asarray(logits, F64)
    g = np.asarray(labels, F64)
    m = z.max(axis=-1, keepdims=True)
    lse = m + np.log(np.exp(z - m).sum(axis=-1, keepdims=True))
    return -(g * (z - lse)).sum(axis=-1)


def conv3d_pad1(x, w, bias):
    """Caffe CONVOLUTION3D kernel 3x3x3, pad 1, stride 1 (prototxt:22-47).
    x [N,D,H,W,Ci] (NDHWC here), w [kd,kh,kw,Ci,Co], bias [Co]."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    n, d, h, wd, ci = x.shape
    xp = np.zeros((n, d + 2, h + 2, wd + 2, ci), F64)
    xp[:, 1:-1, 1:-1, 1:-1, :] = x
    out = np.zeros((n, d, h, wd, w.shape[-1]), F64)
    for a in range(3):
        for b in range(3):
            for c in range(3):
                out += np.einsum('ndhwc,co->ndhwo', xp[:, a:a + d, b:b + h, c:c + wd, :], w[a, b, c])
    return out + np.asarray(bias, F64)


def max_pool3d(x, kd, k):
    """Caffe POOLING3D MAX kernel (kd,k,k) stride = kernel (prototxt:54-66);
    all C3D extents before conv5b are even so ceil-mode == floor-mode."""
    x = np.asarray(x, F64)
    n, d, h, w, c = x.shape
    assert d % kd == 0 and h % k == 0 and w % k == 0
    return x.reshape(n, d // kd, kd, h // k, k, w // k, k, c).max(axis=(2, 4, 6))
