"""GPU: the VIDEO_DATA layer on device (frames_prep_kernel) and the frames -> conv5b extractor."""
import pickle

import numpy as np
import pytest
import torch

from oracle import c3d_frontend as ofe
from recurrent_gaze_prediction_amd import c3d_frontend as fe
from recurrent_gaze_prediction_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine(gpu):
    from recurrent_gaze_prediction_amd.engine import C3DEngine
    eng = C3DEngine(2, dtype='bf16', device=gpu)
    eng.set_weights(syn.c3d_params(7))
    return eng


def test_video_data_layer_exact_at_native_size(gpu, engine):
    """128x171 frames: resize is the identity -> bit-exact crop + mean subtraction; 3 windows on a
    2-window plan exercises the chunk loop; overlapping and unordered window starts."""
    rs = np.random.RandomState(11)
    frames = rs.randint(0, 256, size=(40, 128, 171, 3)).astype(np.uint8)
    mean = (rs.rand(3, 16, 128, 171) * 120).astype(np.float32)
    starts = [16, 0, 24]
    ref = ofe.video_data_layer(frames, starts, mean)
    got = engine.frames_to_video(torch.tensor(frames, device=gpu), starts, torch.tensor(mean, device=gpu)).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)
    nomean = engine.frames_to_video(torch.tensor(frames, device=gpu), [3]).cpu().numpy()
    assert np.array_equal(nomean[0], frames[3:19, 8:120, 29:141].astype(np.float32))


@pytest.mark.parametrize('hw', [(240, 320), (100, 400), (300, 171)])
def test_video_data_layer_resized(gpu, engine, hw):
    """Down- and up-scaling: same float recipe as the oracle; a fused multiply-add may move a value across a
    rounding boundary, so allow one 8-bit level at a handful of pixels."""
    rs = np.random.RandomState(12)
    frames = rs.randint(0, 256, size=(17, hw[0], hw[1], 3)).astype(np.uint8)
    ref = ofe.video_data_layer(frames, [0, 1])
    got = engine.frames_to_video(torch.tensor(frames, device=gpu), [0, 1]).cpu().numpy()
    d = np.abs(got - ref)
    assert d.max() <= 1.0 and float((d > 0).mean()) < 1e-3, (d.max(), float((d > 0).mean()))


def test_forward_frames_equals_two_step_path(gpu, engine):
    rs = np.random.RandomState(13)
    frames = torch.tensor(rs.randint(0, 256, size=(33, 120, 160, 3)).astype(np.uint8), device=gpu)
    mean = torch.tensor((rs.rand(3, 16, 128, 171) * 255).astype(np.float32), device=gpu)
    starts = fe.window_starts(33)
    assert starts == [0, 16]
    f1, r1 = engine.forward_frames(frames, starts, mean, want_rows=True)
    video = engine.frames_to_video(frames, starts, mean)
    f2, r2 = engine.forward(video, want_rows=True)
    assert torch.equal(f1, f2) and torch.equal(r1, r2) and float(f1.abs().max()) > 0


def test_bad_windows_are_rejected(gpu, engine):
    from recurrent_gaze_prediction_amd._lib import RgpError
    frames = torch.zeros(20, 128, 171, 3, dtype=torch.uint8, device=gpu)
    with pytest.raises(RgpError):
        engine.frames_to_video(frames, [5])            # 5 + 16 > 20
    with pytest.raises(RgpError):
        engine.forward_frames(frames, [-1])


def test_extractor_writes_reference_file_set(gpu, engine, tmp_path):
    """frames -> `<feat>/<video>/<start+1:06d>.conv5b` blobs + `<video>.c3d` = [n,1,512,2,7,7] with
    feature[c*2+d] = blob[c][d] (gaze_rnn.py:494-497)."""
    rs = np.random.RandomState(14)
    frames = rs.randint(0, 256, size=(50, 128, 171, 3)).astype(np.uint8)
    mean = (rs.rand(3, 16, 128, 171) * 255).astype(np.float32)
    ex = fe.C3DFeatureExtractor(engine, mean)
    out = ex.extract_to_files(frames, str(tmp_path / 'feat'), 'vid1')
    with open(out, 'rb') as f:
        arr = pickle.load(f)
    assert arr.shape == (3, 1, 512, 2, 7, 7) and arr.dtype == np.float32
    s, blob, ok = fe.read_binary_blob(str(tmp_path / 'feat' / 'vid1' / '000017.conv5b'))
    assert ok == 1 and list(s) == [1, 512, 2, 7, 7] and np.array_equal(blob.data, arr[1])
    feats, _ = engine.forward_frames(torch.tensor(frames, device=gpu), [0, 16, 32], torch.tensor(mean, device=gpu))
    assert np.array_equal(arr.reshape(3, 1024, 7, 7), feats.cpu().numpy())
