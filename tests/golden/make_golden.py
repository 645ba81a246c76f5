#!/usr/bin/env python
"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container
(``python tests/golden/make_golden.py``); the GPU box only reads the .npz files.

Fixtures hold seeds + expected outputs only: inputs and weights are regenerated from
``recurrent_gaze_prediction_amd.synthetic`` (numpy RandomState, bit-stable).

* grcn_small.npz / grcn_refdims.npz / grcn_grads_small.npz / c3d_one_window.npz:
  outputs of the float64 oracle (oracle/grcn.py direct loops, oracle/torch_ref.py
  in float64).  The reference's model code needs TensorFlow 1.x / Python 2, absent
  here, so these pin the ORACLE (against drift), not a running reference.
* metrics_ref.npz: outputs of the REFERENCE's own evaluation_metrics.py
  (/root/reference/evaluation_metrics.py), imported here with three in-memory shims
  (SURVEY.md 8c): a ``skimage.transform.resize`` stand-in that is the identity at equal
  shapes (the only case exercised: 49x49 maps), and ``np.bool`` / ``np.int`` aliases
  removed in numpy >= 1.24; and, for AUC_shuffled (evaluation_metrics.py:167-204), a module-level ``map`` that
  returns a list as Python 2's did (``:200-201`` hands the result of ``map`` to ``np.transpose``).
* c3d_arch_ref.json: the layer table parsed from the prototxt the REFERENCE's generate_feature_prototxt writes
  (extract_C3D_features.py:183-650; same import as c3d_wire_ref.npz).
* model_util_ref.npz: outputs of the REFERENCE's numpy map normalisers (models/model_util.py:20-58; ``tensorflow``, which the
  file imports and these functions never touch, replaced by an empty module).
* c3d_wire_ref.npz: the C3D feature files as the REFERENCE reads and writes them: its own ``read_binary_blob`` and
  ``process_c3d_features`` (C3D/.../hollywood_feature_extraction/extract_C3D_features.py:13-76, 763-798), imported with
  two in-memory shims for modules absent here that those two functions do not touch (``cv2``, ``h5py``), run on blob
  files this script writes byte by byte (header of five int32, then float32 data).  The fixture holds the blob bytes,
  the array the reference decodes from them, the ``.c3d`` pickle it writes, and the ``input.txt`` / ``output_prefix.txt``
  its ``create_src_and_output_file`` (:653-686) writes for a given clip and window schedule.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import grcn, torch_ref  # noqa: E402
from recurrent_gaze_prediction_amd import synthetic as syn  # noqa: E402

REFERENCE_METRICS = '/root/reference/evaluation_metrics.py'
REFERENCE_EXTRACT = ('/root/reference/C3D/C3D-v1.0/examples/c3d_feature_extraction/hollywood_feature_extraction/'
                     'extract_C3D_features.py')


def grcn_case(name, B, T, P, S, seed, use_numpy):
    p = syn.grcn_params(seed, T, P, S, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(seed + 1, B, T)
    gt, _ = syn.gaze_maps(seed + 2, B, T)
    g = grcn.normalize_probability_map(gt)
    if use_numpy:
        logits, inter = grcn.forward(x, p, want_intermediates=True)
        emb, hs = inter['c3d_embedded'], inter['rcn_outputs']
        d1 = np.stack([it['d1'] for it in inter['head']], 1)
        d2 = np.stack([it['d2'] for it in inter['head']], 1)
    else:
        pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
        lg, hst, embt = torch_ref.grcn_forward(torch.tensor(x, dtype=torch.float64), pt, want_hidden=True)
        logits, hs, emb, d1, d2 = lg.numpy(), hst.numpy(), embt.numpy(), None, None
    out = dict(config=np.array([B, T, P, S, seed]), logits=logits.astype(np.float32),
               probs=grcn.softmax_maps(logits).astype(np.float32), loss=np.float64(grcn.loss(logits, g)),
               loss_l2=np.float64(grcn.loss(logits, g, 'l2')),
               h_last=hs[:, -1].astype(np.float32), emb_checksum=np.float64(np.abs(emb).sum()),
               h_checksum=np.float64(np.abs(hs).sum()))
    if d1 is not None:
        out.update(d1_checksum=np.float64(np.abs(d1).sum()), d2_checksum=np.float64(np.abs(d2).sum()))
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'loss', out['loss'], 'logit range', logits.min(), logits.max())


def grads_case(name, B, T, P, S, seed):
    p = syn.grcn_params(seed, T, P, S, gru_std=0.05, random_bn=True)
    x = syn.c3d_features(seed + 1, B, T)
    gt, _ = syn.gaze_maps(seed + 2, B, T)
    g = grcn.normalize_probability_map(gt)
    loss, _, grads = torch_ref.grcn_loss_and_grads(x, g, p)
    clipped, norm = torch_ref.clip_by_global_norm(grads, 10.0)
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in pt.items()}
    v = {k: torch.zeros_like(v) for k, v in pt.items()}
    lr = torch_ref.learning_rate(1e-4, 0.8, 0)
    newp, m, v = torch_ref.adam_step_tf(pt, clipped, m, v, 0, lr)
    out = dict(config=np.array([B, T, P, S, seed]), loss=np.float64(loss), global_norm=np.float64(norm))
    for k, gr in grads.items():
        out['gnorm_' + k] = np.float64(gr.norm().item())
        if gr.numel() <= 4096:
            out['grad_' + k] = gr.numpy().astype(np.float32)
            out['adam1_' + k] = newp[k].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'loss', loss, 'global grad norm', norm)


def c3d_case(name, seed):
    p = syn.c3d_params(seed, scale='he')
    v = syn.video_windows(seed + 1, 1)
    with torch.no_grad():
        feat, acts = torch_ref.c3d_forward(torch.tensor(v, dtype=torch.float64),
                                           {k: torch.tensor(a, dtype=torch.float64) for k, a in p.items()}, want_all=True)
    out = dict(config=np.array([seed]), features=feat.numpy().astype(np.float32))
    for k, a in acts.items():
        out['abs_sum_' + k] = np.float64(a.abs().sum().item())
        out['zero_frac_' + k] = np.float64((a == 0).double().mean().item())
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: round(float(out['zero_frac_' + k]), 3) for k in acts})


def load_reference_metrics():
    """Import the reference's evaluation_metrics.py with in-memory shims (see module doc)."""
    import scipy.sparse  # noqa: F401  (before the np.int alias, as SURVEY 8c notes)
    sk = types.ModuleType('skimage')
    skt = types.ModuleType('skimage.transform')

    def resize(img, shape, order=3, mode='constant', **kw):
        img = np.asarray(img, dtype=np.float64)
        assert tuple(img.shape) == tuple(shape), 'shim only supports equal shapes'
        return img.copy()
    skt.resize = resize
    sk.transform = skt
    sys.modules['skimage'] = sk
    sys.modules['skimage.transform'] = skt
    if not hasattr(np, 'bool'):
        np.bool = bool
    if not hasattr(np, 'int'):
        np.int = int
    spec = importlib.util.spec_from_file_location('reference_evaluation_metrics', REFERENCE_METRICS)
    mod = importlib.util.module_from_spec(spec)
    import builtins
    mod.map = lambda f, *a: list(builtins.map(f, *a))      # Python-2 ``map``: a list (AUC_shuffled, :200-201)
    spec.loader.exec_module(mod)
    return mod


def metrics_case(name, seed=40, n=12):
    ref = load_reference_metrics()
    if not hasattr(np, 'trapz'):
        np.trapz = np.trapezoid
    gt, centres = syn.gaze_maps(seed, n, 1)
    fix = syn.fixation_maps(seed + 1, centres)[:, 0]
    gt = gt[:, 0]
    rs = np.random.RandomState(seed + 2)
    pred = (gt + 0.3 * rs.rand(*gt.shape) + 0.2 * np.roll(gt, 3, axis=2)).astype(np.float32)
    out = dict(config=np.array([seed, n]))
    sims, ccs, judd, borji = [], [], [], []
    for i in range(n):
        sims.append(ref.saliency_score_single('sim', pred[i], gt[i], fix[i]))
        ccs.append(ref.saliency_score_single('cc', pred[i], gt[i], fix[i]))
        np.random.seed(1000 + i)
        judd.append(ref.saliency_score_single('AUC_Judd', pred[i], gt[i], fix[i]))
        np.random.seed(2000 + i)
        borji.append(ref.saliency_score_single('AUC_Borji', pred[i], gt[i], fix[i]))
    out.update(sim=np.array(sims), cc=np.array(ccs), AUC_Judd=np.array(judd), AUC_Borji=np.array(borji))
    # AUC_shuffled: negatives from the union of the OTHER frames' fixations (what saliency_score builds, :283-287)
    shuf = []
    for i in range(n):
        other = np.zeros(fix[0].shape)
        for j in range(n):
            if j != i:
                other += (fix[j] > 0).astype(int)
        np.random.seed(4000 + i)
        shuf.append(ref.saliency_score_single('AUC_shuffled', pred[i], gt[i], fix[i], other))
    out['AUC_shuffled'] = np.array(shuf)
    for metric in ('sim', 'cc', 'AUC_Borji', 'AUC_shuffled'):
        np.random.seed(3000)
        out['score_' + metric] = np.float64(ref.saliency_score(metric, list(pred), list(gt), list(fix)))
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: float(np.mean(out[k])) for k in ('sim', 'cc', 'AUC_Judd', 'AUC_Borji')})


def _t64(p):
    return {k: _t64(v) if isinstance(v, dict) else torch.tensor(v, dtype=torch.float64) for k, v in p.items()}


def cascade_case(name, B, T, seed):
    """gaze_grcn_cascade.py (config 5): maps, intermediates' checksums, l2 loss and the gradient norm of every
    trainable variable (ShallowNet frozen), float64 torch oracle."""
    p = syn.cascade_params(seed)
    rs = np.random.RandomState(seed + 7)
    frames = rs.rand(B, T, 98, 98, 3).astype(np.float32)
    c3d = syn.c3d_features(seed + 8, B, T)
    gt, _ = syn.gaze_maps(seed + 9, B, T)
    gt = (gt / gt.max()).astype(np.float32)
    tp = _t64(p)
    for k, v in tp.items():
        if k != 'ShallowNet':
            v.requires_grad_(True)
    maps, mid = torch_ref.cascade_forward(torch.tensor(frames, dtype=torch.float64), torch.tensor(c3d, dtype=torch.float64), tp,
                                          want_all=True)
    loss = torch_ref.gaze_loss(maps, torch.tensor(gt, dtype=torch.float64), 'l2')
    loss.backward()
    out = dict(config=np.array([B, T, seed]), maps=maps.detach().numpy().astype(np.float32), loss=np.float64(loss.item()))
    for k, v in mid.items():
        out['abs_sum_' + k] = np.float64(v.detach().abs().sum().item())
    for k, v in tp.items():
        if k != 'ShallowNet':
            out['gnorm_' + k.replace('/', '.')] = np.float64(v.grad.norm().item())
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'loss', out['loss'], 'maps max', float(maps.max()))


def fcgru_case(name, B, T, GH, seed):
    p = syn.fcgru_params(seed, GH, GH)
    x = syn.c3d_features(seed + 1, B, T)
    rs = np.random.RandomState(seed + 2)
    gt = rs.rand(B, T, GH, GH).astype(np.float32)
    gt /= gt.sum(axis=(2, 3), keepdims=True)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    logits = torch_ref.fcgru_forward(torch.tensor(x, dtype=torch.float64), tp, GH, GH)
    loss = torch_ref.gaze_loss(logits, torch.tensor(gt, dtype=torch.float64))
    loss.backward()
    out = dict(config=np.array([B, T, GH, seed]), logits=logits.detach().numpy().astype(np.float32), loss=np.float64(loss.item()))
    for k, v in tp.items():
        out['gnorm_' + k] = np.float64(v.grad.norm().item())
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'loss', out['loss'])


def frontend_case(name, seed):
    """VIDEO_DATA layer oracle (oracle/c3d_frontend.py): 240x320 frames -> one 16x112x112x3 window; a strided
    sample of the output (the full window is regenerated from the seed in the tests)."""
    from oracle import c3d_frontend as ofe
    rs = np.random.RandomState(seed)
    frames = rs.randint(0, 256, size=(18, 240, 320, 3)).astype(np.uint8)
    mean = (rs.rand(3, 16, 128, 171) * 120).astype(np.float32)
    v = ofe.video_data_layer(frames, [1], mean)
    np.savez_compressed(os.path.join(HERE, name), config=np.array([seed]), sample=v[0, ::3, ::7, ::5].astype(np.float32),
                        checksum=np.float64(np.abs(v.astype(np.float64)).sum()))
    print(name, 'checksum', float(np.abs(v.astype(np.float64)).sum()))


def c3d_wire_case(name, seed=161, n_clips=2):
    import shutil
    import tempfile
    for missing in ('cv2', 'h5py'):                          # imported at the top of the script, unused by the two functions
        if missing not in sys.modules:
            sys.modules[missing] = types.ModuleType(missing)
    spec = importlib.util.spec_from_file_location('reference_extract_c3d', REFERENCE_EXTRACT)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rs = np.random.RandomState(seed)
    tmp = tempfile.mkdtemp()
    try:
        feat_dir = os.path.join(tmp, 'videoA')
        os.makedirs(feat_dir)
        blobs = []
        for k in range(n_clips):
            # sparse (the fixture compresses to a few KB), every value exactly representable; conv5b is post-ReLU anyway
            data = np.zeros((1, 512, 2, 7, 7), '<f4')
            idx = rs.choice(data.size, 400, replace=False)
            data.reshape(-1)[idx] = rs.randint(1, 200, size=400) / 8.0
            raw = np.array([1, 512, 2, 7, 7], dtype='<i4').tobytes() + data.tobytes()
            with open(os.path.join(feat_dir, '%06d.conv5b' % (16 * k + 1)), 'wb') as f:
                f.write(raw)
            blobs.append(np.frombuffer(raw, dtype=np.uint8))
        size, blob, status = ref.read_binary_blob(os.path.join(feat_dir, '000001.conv5b'))
        assert status == 1 and list(size) == [1, 512, 2, 7, 7]
        ref.process_c3d_features(feat_dir, 'conv5b')
        with open(os.path.join(tmp, 'videoA.c3d'), 'rb') as f:
            c3d_bytes = f.read()
        import pickle
        stacked = pickle.loads(c3d_bytes, encoding='latin1')
        # the input / output-prefix lists the Caffe extractor is driven by (extract_C3D_features.py:653-686)
        lists = os.path.join(tmp, 'lists')
        os.makedirs(lists)
        starts = [0, 16, 32, 48]
        ref.create_src_and_output_file('/videos/actioncliptest00012.avi', starts, lists, '/data/frames', '/data/feat')
        input_txt = open(os.path.join(lists, 'input.txt')).read()
        prefix_txt = open(os.path.join(lists, 'output_prefix.txt')).read()
        np.savez_compressed(os.path.join(HERE, name), blob_bytes=np.stack(blobs), ref_size=np.array(list(size)),
                            ref_blob0=np.asarray(blob.data, np.float32), c3d_pickle=np.frombuffer(c3d_bytes, dtype=np.uint8),
                            c3d_array=np.asarray(stacked, np.float32), list_starts=np.array(starts),
                            input_txt=np.array(input_txt), output_prefix_txt=np.array(prefix_txt))
        print(name, 'blob0', np.asarray(blob.data).shape, '.c3d', np.asarray(stacked).shape, len(c3d_bytes), 'bytes')
    finally:
        shutil.rmtree(tmp)


REFERENCE_MODEL_UTIL = '/root/reference/models/model_util.py'


def c3d_arch_case(name):
    """The network the reference's extractor declares: its generate_feature_prototxt (extract_C3D_features.py:183-650) is run
    here (same import as c3d_wire_case) and the text it writes is PARSED into a table -- layer order, types, filter counts,
    kernel / pad / stride extents and the VIDEO_DATA geometry; the fixture holds the table, not the text."""
    import json
    import re
    import tempfile
    for missing in ('cv2', 'h5py'):
        if missing not in sys.modules:
            sys.modules[missing] = types.ModuleType(missing)
    spec = importlib.util.spec_from_file_location('reference_extract_c3d', REFERENCE_EXTRACT)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    with tempfile.TemporaryDirectory() as tmp:
        mean = os.path.join(tmp, 'mean.binaryproto')
        open(mean, 'wb').close()
        out = os.path.join(tmp, 'net.prototxt')
        ref.generate_feature_prototxt(out, os.path.join(tmp, 'input.txt'), mean)
        text = open(out).read()
    text = re.sub(r'#.*', '', text)
    layers = []
    for body in re.findall(r'layers\s*\{(.*?)\n\}', text, flags=re.S):
        get = lambda key, cast=str: [cast(v.strip('"')) for v in re.findall(r'\b%s:\s*("[^"]*"|\S+)' % key, body)]
        entry = {'name': get('name')[0], 'type': get('type')[0], 'bottom': get('bottom'), 'top': get('top')}
        for key in ('num_output', 'kernel_size', 'kernel_depth', 'pad', 'temporal_pad', 'stride', 'temporal_stride', 'crop_size',
                    'new_height', 'new_width', 'new_length', 'batch_size'):
            v = get(key, int)
            if v:
                entry[key] = v[0]
        for key in ('pool', 'mirror', 'use_image', 'shuffle'):
            v = get(key)
            if v:
                entry[key] = v[0]
        layers.append(entry)
    with open(os.path.join(HERE, name), 'w') as f:
        json.dump({'net_name': re.findall(r'^name:\s*"([^"]*)"', text, flags=re.M)[0], 'layers': layers}, f, indent=1)
    print(name, len(layers), 'layers:', ' '.join(l['name'] for l in layers))


def model_util_case(name, seed=171):
    """The reference's numpy map normalisers (models/model_util.py:20-58), imported with an EMPTY stand-in for
    `tensorflow` (absent here; the file imports it at the top, these two functions never touch it)."""
    if 'tensorflow' not in sys.modules:
        sys.modules['tensorflow'] = types.ModuleType('tensorflow')
    spec = importlib.util.spec_from_file_location('reference_model_util', REFERENCE_MODEL_UTIL)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rs = np.random.RandomState(seed)
    maps3 = (rs.rand(4, 9, 9) * 3 + 0.1).astype(np.float32)
    maps3[2] = 0.25                                            # a constant map: max after the shift is 0
    maps4 = (rs.rand(2, 3, 7, 7) + 0.05).astype(np.float32)
    maps4b = (rs.rand(3, 6, 6, 1) * 5).astype(np.float32)      # [B, H, W, 1] for normalize_map
    np.savez_compressed(os.path.join(HERE, name), maps3=maps3, maps4=maps4, maps4b=maps4b,
                        norm3=ref.normalize_map(maps3), norm4b=ref.normalize_map(maps4b),
                        prob3=ref.normalize_probability_map(maps3), prob4=ref.normalize_probability_map(maps4),
                        prob3_f64=ref.normalize_probability_map(maps3.astype(np.float64)))
    print(name, 'ok')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'arch':        # the C3D network as the reference's prototxt generator declares it
        c3d_arch_case('c3d_arch_ref.json')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'util':        # reference-pinned numpy normalisers (round 3)
        model_util_case('model_util_ref.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'wire':        # reference-pinned C3D feature-file fixture (round 3)
        c3d_wire_case('c3d_wire_ref.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'metrics':     # reference-pinned metric scores (round 4: + AUC_shuffled)
        metrics_case('metrics_ref.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'new':         # fixtures added after the first set
        cascade_case('cascade_small.npz', 1, 2, 131)
        fcgru_case('fcgru_small.npz', 2, 3, 7, 141)
        frontend_case('frontend_window.npz', 151)
        sys.exit(0)
    grcn_case('grcn_small.npz', 2, 3, 64, 64, 101, use_numpy=True)
    grcn_case('grcn_refdims.npz', 1, 2, 512, 128, 111, use_numpy=False)
    grads_case('grcn_grads_small.npz', 2, 3, 64, 64, 101)
    c3d_case('c3d_one_window.npz', 121)
    metrics_case('metrics_ref.npz')
