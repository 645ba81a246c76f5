"""Seeded synthetic inputs and reference-style initial weights (numpy, host side).

There are no datasets or checkpoints offline, so tests and ``bench.py`` use the
synthetic workload BASELINE.md section 3 / SURVEY.md 8(d) define.  Everything is
drawn from ``numpy.random.RandomState(seed)`` (legacy generator: bit-stable across
numpy versions and machines), so a fixture only has to store the seed.

Initialisers follow the reference:
  proj_c3d_W/b, out_W/b  U(-0.1, 0.1)                        gaze_grcn.py:234-237,311-314
  GRU_Conv_*            truncated_normal(stddev)             gaze_grcn.py:64-81 (1e-4 there;
                        parity runs use 0.05)
  Upsampling/weight1-3  Xavier-uniform on [kh,kw,out,in]     gaze_grcn.py:292-310
  batch-norm gamma/beta 1 / 0, one pair per timestep         gaze_grcn.py:325 (9-Q1)
  C3D conv              gaussian std 0.01, bias 0 (conv1a) / 1     prototxt:38-45,83-90
"""
import numpy as np

C3D_SPECS = [  # name, Cin, Cout  (prototxt:22-342)
    ('conv1a', 3, 64), ('conv2a', 64, 128), ('conv3a', 128, 256), ('conv3b', 256, 256),
    ('conv4a', 256, 512), ('conv4b', 512, 512), ('conv5a', 512, 512), ('conv5b', 512, 512),
]


def _trunc_normal(rs, shape, std):
    x = rs.randn(*shape)
    bad = np.abs(x) > 2.0
    while bad.any():                      # tf.truncated_normal: redraw beyond 2 sigma
        x[bad] = rs.randn(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def _xavier_conv(rs, shape):
    kh, kw, a, b = shape                  # TF: fan_in = shape[-2]*kh*kw, fan_out = shape[-1]*kh*kw
    lim = np.sqrt(6.0 / ((a + b) * kh * kw))
    return rs.uniform(-lim, lim, size=shape).astype(np.float32)


def _xavier_fc(rs, n_in, n_out):
    lim = np.sqrt(6.0 / (n_in + n_out))
    return rs.uniform(-lim, lim, size=(n_in, n_out)).astype(np.float32)


def grcn_params(seed, n_steps, dim_proj=512, dim_state=128, gru_std=0.05, random_bn=False):
    """Weights of GazePredictionGRCN keyed by the reference's TF variable names."""
    rs = np.random.RandomState(seed)
    u = lambda *s: rs.uniform(-0.1, 0.1, size=s).astype(np.float32)
    p = {
        'proj_c3d_W': u(1024, dim_proj), 'proj_c3d_b': u(dim_proj),
        'GRU_Conv_Wz': _trunc_normal(rs, (3, 3, dim_proj, dim_state), gru_std),
        'GRU_Conv_Uz': _trunc_normal(rs, (3, 3, dim_state, dim_state), gru_std),
        'GRU_Conv_Wr': _trunc_normal(rs, (3, 3, dim_proj, dim_state), gru_std),
        'GRU_Conv_Ur': _trunc_normal(rs, (3, 3, dim_state, dim_state), gru_std),
        'GRU_Conv_W': _trunc_normal(rs, (3, 3, dim_proj, dim_state), gru_std),
        'GRU_Conv_U': _trunc_normal(rs, (3, 3, dim_state, dim_state), gru_std),
        'weight1': _xavier_conv(rs, (5, 5, 64, dim_state)),
        'weight2': _xavier_conv(rs, (5, 5, 32, 64)),
        'weight3': _xavier_conv(rs, (7, 7, 12, 32)),
        'out_W': u(12, 1), 'out_b': u(1),
    }
    if random_bn:
        p['bn_gamma'] = rs.uniform(0.5, 1.5, size=(n_steps, dim_state)).astype(np.float32)
        p['bn_beta'] = rs.uniform(-0.2, 0.2, size=(n_steps, dim_state)).astype(np.float32)
    else:
        p['bn_gamma'] = np.ones((n_steps, dim_state), np.float32)
        p['bn_beta'] = np.zeros((n_steps, dim_state), np.float32)
    return p


def fcgru_params(seed, gh=49, gw=49, dim_proj=32):
    """Weights of GazePredictionGRU (gaze_rnn.py:294-320; TF GRUCell: gate bias 1)."""
    rs = np.random.RandomState(seed)
    n = 7 * 7 * dim_proj + 49
    n_in = 7 * 7 * dim_proj
    u = lambda *s: rs.uniform(-0.1, 0.1, size=s).astype(np.float32)
    k = n_in + n                       # "orthogonal"-scale kernels: N(0, 1/k), same spectrum scale
    g = lambda *sh: (rs.randn(*sh) / np.sqrt(k)).astype(np.float32)
    return {
        'proj_c3d_W': u(1024, dim_proj), 'proj_c3d_b': u(dim_proj),
        'gates_kernel': g(k, 2 * n),
        'gates_bias': np.ones(2 * n, np.float32),
        'candidate_kernel': g(k, n),
        'candidate_bias': np.zeros(n, np.float32),
        'proj_out_W': u(n, gh * gw), 'proj_out_b': np.zeros(gh * gw, np.float32),
    }


def shallownet_params(seed, image_hw=98):
    """Weights of SaliencyModel.create_shallownet (saliency_shallownet.py:90-170)."""
    rs = np.random.RandomState(seed)
    s = image_hw - 4                       # conv1 5x5 VALID
    s = -(-s // 2) - 2                     # pool1 2x2/2 SAME, conv2 3x3 VALID
    s = -(-s // 2) - 2                     # pool2 3x3/2 SAME, conv3 3x3 VALID
    s = -(-s // 2)                         # pool3
    n_flat = s * s * 32
    xc = lambda kh, kw, ci, co: (rs.uniform(-1, 1, size=(kh, kw, ci, co)) *
                                 np.sqrt(6.0 / ((ci + co) * kh * kw))).astype(np.float32)
    return {
        'conv1_w': xc(5, 5, 3, 32), 'conv1_b': np.zeros(32, np.float32),
        'conv2_w': xc(3, 3, 32, 64), 'conv2_b': np.zeros(64, np.float32),
        'conv3_w': xc(3, 3, 64, 32), 'conv3_b': np.zeros(32, np.float32),
        'fc1_w': _xavier_fc(rs, n_flat, 4802), 'fc1_b': np.zeros(4802, np.float32),
        'fc2_w': _xavier_fc(rs, 2401, 4802), 'fc2_b': np.zeros(4802, np.float32),
    }


def cascade_params(seed, image_hw=98, dim_proj=512, dim_state=256, gru_std=0.03, top_std=0.05):
    """Weights of the cascade model (gaze_grcn_cascade.py:228-423) keyed by TF variable names
    (scopes RCNBottom / RCNGaze / LastProjection / ShallowNet).  The reference initialises the GRU
    filters with stddev 1e-4; tests use larger values so that the recurrence carries signal."""
    rs = np.random.RandomState(seed)
    u = lambda *s: rs.uniform(-0.1, 0.1, size=s).astype(np.float32)
    p = {'proj_c3d_W': u(1024, dim_proj), 'proj_c3d_b': u(dim_proj)}
    for n, cin in (('Wz', dim_proj), ('Uz', dim_state), ('Wr', dim_proj), ('Ur', dim_state), ('W', dim_proj),
                   ('U', dim_state)):
        p['RCNBottom/GRU_Conv_' + n] = _trunc_normal(rs, (3, 3, cin, dim_state), gru_std)
    p['Upsampling/weight'] = _xavier_conv(rs, (11, 11, 64, dim_state))
    for n, cin in (('Wz', 65), ('Uz', 3), ('Wr', 65), ('Ur', 3), ('W', 65), ('U', 3)):
        p['RCNGaze/GRU_Conv_' + n] = _trunc_normal(rs, (5, 5, cin, 3), top_std)
    p['LastProjection/fc1_w'] = _xavier_fc(rs, 49 * 49 * 3, 4802) * 4.0
    p['LastProjection/fc1_b'] = u(4802) * 0.1
    p['LastProjection/fc2_w'] = _xavier_fc(rs, 2401, 4802)
    p['LastProjection/fc2_b'] = u(4802) * 0.1
    p['ShallowNet'] = shallownet_params(seed + 1, image_hw)
    return p


def c3d_params(seed, scale='he'):
    """C3D conv1a..conv5b weights, DHWIO [3,3,3,Cin,Cout] + bias [Cout].

    scale='caffe': the prototxt fillers (gaussian 0.01, bias 0/1) -- with random
    weights every activation sits at ~1 and ReLU never fires, so parity runs use
    scale='he' (std sqrt(2/(27 Cin)), bias N(0,0.1)) which keeps activations O(1)
    with ~50% zeros through all eight layers."""
    rs = np.random.RandomState(seed)
    p = {}
    for name, ci, co in C3D_SPECS:
        if scale == 'caffe':
            std, b = 0.01, np.full(co, 0.0 if name == 'conv1a' else 1.0, np.float32)
        else:
            std, b = np.sqrt(2.0 / (27 * ci)), (rs.randn(co) * 0.1).astype(np.float32)
        p[name + '_w'] = (rs.randn(3, 3, 3, ci, co) * std).astype(np.float32)
        p[name + '_b'] = b
    return p


def c3d_features(seed, batch, n_steps):
    """conv5b is post-ReLU: ReLU(N(0,1)), ~50 % zeros -> [B,T,1024,7,7] f32."""
    rs = np.random.RandomState(seed)
    return np.maximum(rs.randn(batch, n_steps, 1024, 7, 7), 0).astype(np.float32)


def video_windows(seed, n, frames=16, hw=112):
    """Mean-subtracted RGB windows U(0,1)-0.5 -> [n, frames, hw, hw, 3] f32."""
    rs = np.random.RandomState(seed)
    return (rs.uniform(0, 1, size=(n, frames, hw, hw, 3)) - 0.5).astype(np.float32)


def gaze_maps(seed, batch, n_steps, hw=49, sigma=2.0):
    """Gaussian blob at a random centre per frame (mirrors crc_input_data_seq.py:231-233),
    un-normalised; strictly positive sum."""
    rs = np.random.RandomState(seed)
    cy = rs.uniform(5, hw - 5, size=(batch, n_steps, 1, 1))
    cx = rs.uniform(5, hw - 5, size=(batch, n_steps, 1, 1))
    yy, xx = np.mgrid[0:hw, 0:hw]
    g = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sigma ** 2))
    return g.astype(np.float32), np.concatenate([cy, cx], -1).reshape(batch, n_steps, 2)


def fixation_maps(seed, centres, hw=49, n_fix=6, spread=3.0):
    """Binary fixation maps: n_fix points scattered around each gaze centre."""
    rs = np.random.RandomState(seed)
    b, t, _ = centres.shape
    out = np.zeros((b, t, hw, hw), np.float32)
    for i in range(b):
        for j in range(t):
            pts = np.clip(np.round(centres[i, j] + rs.randn(n_fix, 2) * spread), 0, hw - 1).astype(int)
            out[i, j, pts[:, 0], pts[:, 1]] = 1.0
    return out


class SyntheticDataSet(object):
    """Stands in for crc_input_data_seq.CRCDataSet: ``len(ds)`` and ``next_batch(B)`` returning the
    reference's 6-tuple (crc_input_data_seq.py:132-156): images [B,T,98,98,3] f32 in [0,1],
    gazemaps [B,T,49,49] f32, fixationmaps [B,T,49,49], c3d [B,T,512,2,7,7] f32, pupils [B,T],
    clipnames."""

    def __init__(self, n_clips, n_steps, seed=0, image_hw=98):
        self.n_clips, self.n_steps, self.seed, self.image_hw = n_clips, n_steps, seed, image_hw
        self._cursor = 0

    def __len__(self):
        return self.n_clips

    def next_batch(self, batch_size):
        idx = [(self._cursor + i) % self.n_clips for i in range(batch_size)]
        self._cursor = (self._cursor + batch_size) % self.n_clips
        T = self.n_steps
        rs = np.random.RandomState(self.seed)
        images = rs.rand(batch_size, T, self.image_hw, self.image_hw, 3).astype(np.float32)
        maps = np.concatenate([gaze_maps(self.seed + 17 * i + 1, 1, T)[0] for i in idx])
        cents = np.concatenate([gaze_maps(self.seed + 17 * i + 1, 1, T)[1] for i in idx])
        fix = fixation_maps(self.seed + 3, cents)
        feats = np.concatenate([c3d_features(self.seed + 17 * i + 2, 1, T) for i in idx])
        c3d = feats.reshape(batch_size, T, 512, 2, 7, 7)          # channel = c*2+d (gaze_rnn.py:494-497)
        pupils = np.zeros((batch_size, T), np.float32)
        names = ['synthetic_%04d' % i for i in idx]
        return images, maps, fix, c3d, pupils, names
