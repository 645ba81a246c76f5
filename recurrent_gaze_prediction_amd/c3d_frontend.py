"""Host side of the C3D front-end: everything the reference does around the Caffe
`extract_image_features.bin` run, so real videos and the public Sports-1M model can feed the HIP
conv stack (rgp_c3d_*).  Mirrors /root/reference/C3D/C3D-v1.0/examples/c3d_feature_extraction/
hollywood_feature_extraction/extract_C3D_features.py:

  read_binary_blob / write_binary_blob    :13-76   C3D binary feature blobs (5 int32 dims + fp32 data)
  window_starts                            :866     16-frame windows at stride 16
  input_list_lines                         :667-684 the "<frame_dir>/ <start+1> <label>" list + output prefixes
  process_c3d_features                     :763-798 per-window blobs -> one pickled `.c3d` array
  read_mean_binaryproto / read_caffemodel  -- the two model files the script downloads (:88-110),
                                              parsed from the protobuf wire format directly
  caffemodel_to_c3d_params                 -- Caffe [Cout,Cin,kd,kh,kw] -> DHWIO for rgp_c3d_set_weights
  C3DFeatureExtractor                      -- VIDEO_DATA layer + conv1a..conv5b on the GPU

Third-party formats restated here (not vendored in the reference, so parity is unpinned for them):
the facebook/C3D v1.0 fork's caffe.proto -- BlobProto {num=1, channels=2, length=3, height=4,
width=5, data=6 packed float, diff=7}, NetParameter {name=1, layers=2}, LayerConnection {layer=1},
V0 LayerParameter {name=1, type=2, blobs=50}.  The BlobProto reader does not depend on those field
numbers (dims = the varint fields in order, data = the float field of matching size), so stock
BVLC Caffe blobs (no `length`) parse as well.
"""
import glob
import os
import pickle
import struct
from collections import namedtuple

import numpy as np

NUM_FRAMES_PER_CLIP = 16            # extract_C3D_features.py:858
C3D_LAYER_NAMES = ('conv1a', 'conv2a', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'conv5a', 'conv5b')
Blob = namedtuple('Blob', ['size', 'data'])


# --------------------------------------------------------------------------- C3D binary blobs
def read_binary_blob(filename):
    """extract_C3D_features.py:13-76 -> (size [num,channel,length,height,width], Blob(size, data), read_status).
    data is float32 [num, channel, length, height, width] (row-major, as C3D writes it)."""
    with open(filename, 'rb') as f:
        head = f.read(20)
        if len(head) != 20:
            return [], Blob([], []), 0
        s = list(struct.unpack('<5i', head))
        m = int(np.prod(s))
        data = np.fromfile(f, dtype='<f4', count=m)
    if m <= 0 or data.size != m:
        return [], Blob([], []), 0
    return s, Blob(s, data.reshape(s).astype(np.float32)), 1


def write_binary_blob(filename, data):
    """Inverse of read_binary_blob (what extract_image_features.bin writes per clip)."""
    a = np.ascontiguousarray(data, dtype='<f4')
    assert a.ndim == 5, a.shape
    with open(filename, 'wb') as f:
        f.write(struct.pack('<5i', *a.shape))
        a.tofile(f)


# --------------------------------------------------------------------------- window scheduling
def window_starts(num_frames, length=NUM_FRAMES_PER_CLIP, stride=NUM_FRAMES_PER_CLIP, drop_incomplete=True):
    """extract_C3D_features.py:866: range(0, num_frames, 16).  The Caffe layer reads `length` consecutive
    frame files from each start and fails on a missing one, so a trailing window with fewer than
    `length` frames yields no feature; drop_incomplete removes it up front."""
    starts = list(range(0, int(num_frames), int(stride)))
    if drop_incomplete:
        starts = [s for s in starts if s + length <= num_frames]
    return starts


def input_list_lines(frame_dir, feat_dir, video_id, starts, dummy_label=0):
    """extract_C3D_features.py:667-684: (input.txt lines, output_prefix.txt lines); frame numbers are 1-based."""
    fdir = os.path.join(frame_dir, video_id)
    inputs = ['{}/ {:d} {:d} '.format(fdir, int(s) + 1, int(dummy_label)) for s in starts]
    clip = lambda s: os.path.join(feat_dir, video_id) + '/{0:06d}'.format(int(s) + 1)
    outputs = [os.path.join(feat_dir, clip(s)) for s in starts]
    return inputs, outputs


def process_c3d_features(feature_dir, c3d_layer='conv5b'):
    """extract_C3D_features.py:763-798: gather `<video_dir>/*/*.<layer>` blobs of the video's folder,
    stack them [n,1,512,2,7,7] float32 and pickle (protocol 2) to `<video_dir>/<video_name>.c3d`."""
    video_dir, video_name = os.path.split(feature_dir)
    feats = [read_binary_blob(p)[1].data for p in sorted(glob.glob('{}/*/*.{}'.format(video_dir, c3d_layer)))]
    out = os.path.join(video_dir, video_name + '.c3d')
    with open(out, 'wb') as f:
        pickle.dump(np.array(feats, dtype=np.float32), f, protocol=2)
    return out


# --------------------------------------------------------------------------- protobuf wire format
def _varint(buf, pos):
    val, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise ValueError('malformed varint')


def iter_fields(buf):
    """Yields (field_number, wire_type, value) of one message; value is an int (varint, wire 0),
    8 / 4 raw bytes (wire 1 / 5) or a memoryview (length-delimited, wire 2)."""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wire == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wire == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError('unsupported wire type %d (field %d)' % (wire, field))
        if pos > n:
            raise ValueError('truncated message')
        yield field, wire, val


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _enc_field(field, wire, payload):
    if wire == 0:
        return _enc_varint(field << 3) + _enc_varint(payload)
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + bytes(payload)


def parse_blobproto(buf):
    """BlobProto -> float32 array shaped by its dimension fields (5-D for the C3D fork:
    num, channels, length, height, width; 4-D for stock Caffe)."""
    dims, packed, loose = [], {}, {}
    for field, wire, val in iter_fields(buf):
        if wire == 0:
            dims.append((field, val))
        elif wire == 2:
            packed.setdefault(field, []).append(val)
        elif wire == 5:
            loose.setdefault(field, []).append(bytes(val))
    shape = [v for _, v in sorted(dims)]
    count = int(np.prod(shape)) if shape else 0
    for field in sorted(packed):                               # packed repeated float
        raw = b''.join(bytes(v) for v in packed[field])
        if count and len(raw) == 4 * count:
            return np.frombuffer(raw, dtype='<f4').reshape(shape).astype(np.float32)
    for field in sorted(loose):                                # non-packed repeated float
        if count and len(loose[field]) == count:
            return np.frombuffer(b''.join(loose[field]), dtype='<f4').reshape(shape).astype(np.float32)
    raise ValueError('BlobProto: no float field matches dims %r' % (shape,))


def encode_blobproto(arr):
    """5-D array -> BlobProto bytes in the C3D fork's field numbering (see module docstring)."""
    a = np.ascontiguousarray(arr, dtype='<f4')
    assert a.ndim == 5, a.shape
    out = b''.join(_enc_field(i + 1, 0, int(d)) for i, d in enumerate(a.shape))
    return out + _enc_field(6, 2, a.tobytes())


def read_mean_binaryproto(path):
    """sport1m_train16_128_mean.binaryproto -> mean cube [3,16,128,171] float32 (channels BGR)."""
    with open(path, 'rb') as f:
        a = parse_blobproto(f.read())
    if a.ndim == 5:
        assert a.shape[0] == 1, a.shape
        a = a[0]
    return np.ascontiguousarray(a, np.float32)


def write_mean_binaryproto(path, mean_cube):
    m = np.asarray(mean_cube, np.float32)
    with open(path, 'wb') as f:
        f.write(encode_blobproto(m[None] if m.ndim == 4 else m))


def _layer_name_blobs(entry):
    """One `layers` entry of NetParameter -> (name, [arrays]); V0 (LayerConnection{layer=1{name=1,
    blobs=50}}, what the 2014-era C3D model uses) or V1 (name=4, blobs=6)."""
    fields = list(iter_fields(entry))
    for field, wire, val in fields:
        if field == 1 and wire == 2:                       # V0: nested LayerParameter
            name, blobs = None, []
            try:
                for f2, w2, v2 in iter_fields(val):
                    if f2 == 1 and w2 == 2 and name is None:
                        name = bytes(v2).decode('utf-8')
                    elif f2 == 50 and w2 == 2:
                        blobs.append(parse_blobproto(v2))
            except (ValueError, UnicodeDecodeError, IndexError):
                name = None
            if name is not None:
                return name, blobs
    name, blobs = None, []
    for field, wire, val in fields:                        # V1
        if field == 4 and wire == 2:
            name = bytes(val).decode('utf-8')
        elif field == 6 and wire == 2:
            blobs.append(parse_blobproto(val))
    return name, blobs


def read_caffemodel(path):
    """conv3d_deepnetA_sport1m_iter_1900000 (binary NetParameter) -> {layer name: [blob arrays]}."""
    with open(path, 'rb') as f:
        buf = f.read()
    layers = {}
    for field, wire, val in iter_fields(buf):
        if field == 2 and wire == 2:
            name, blobs = _layer_name_blobs(val)
            if name is not None and blobs:
                layers[name] = blobs
    return layers


def caffemodel_to_c3d_params(layers):
    """{name: [W [Cout,Cin,kd,kh,kw], b [1,1,1,1,Cout]]} -> {'conv1a_w': DHWIO [3,3,3,Cin,Cout], 'conv1a_b': [Cout], ...}
    for C3DEngine.set_weights.  Input channel order stays as trained (BGR)."""
    p = {}
    for name in C3D_LAYER_NAMES:
        if name not in layers:
            raise KeyError('caffemodel has no layer %r (found %s)' % (name, sorted(layers)))
        w, b = layers[name][0], layers[name][1]
        assert w.ndim == 5 and w.shape[2:] == (3, 3, 3), (name, w.shape)
        p[name + '_w'] = np.ascontiguousarray(np.transpose(w, (2, 3, 4, 1, 0)), np.float32)
        p[name + '_b'] = np.ascontiguousarray(b.reshape(-1), np.float32)
        assert p[name + '_b'].shape[0] == w.shape[0], (name, w.shape, b.shape)
    return p


def c3d_params_to_caffemodel_bytes(params):
    """Inverse of read_caffemodel + caffemodel_to_c3d_params (V0 layout): lets a fine-tuned conv stack
    go back to the Caffe tool chain, and gives the parser a file to read in tests."""
    out = b''
    for name in C3D_LAYER_NAMES:
        w = np.transpose(np.asarray(params[name + '_w'], np.float32), (4, 3, 0, 1, 2))
        b = np.asarray(params[name + '_b'], np.float32).reshape(1, 1, 1, 1, -1)
        layer = _enc_field(1, 2, name.encode()) + _enc_field(2, 2, b'conv3d')
        layer += _enc_field(50, 2, encode_blobproto(w)) + _enc_field(50, 2, encode_blobproto(b))
        out += _enc_field(2, 2, _enc_field(1, 2, layer) + _enc_field(2, 2, b'bottom') + _enc_field(3, 2, name.encode()))
    return _enc_field(1, 2, b'DeepConv3DNet_Sport1M_Val') + out


# --------------------------------------------------------------------------- device extractor
class C3DFeatureExtractor(object):
    """run_C3D_extraction + process_c3d_features (extract_C3D_features.py:689-724,763-798) on the GPU:
    frames -> 16-frame windows -> resize/crop/mean -> conv1a..conv5b -> conv5b blobs."""

    def __init__(self, engine, mean_cube=None):
        import torch
        self.engine = engine
        self.mean = None
        if mean_cube is not None:
            m = np.asarray(mean_cube, np.float32)
            assert m.shape == (3, 16, 128, 171), m.shape
            self.mean = torch.as_tensor(m).to(engine.device).contiguous()

    def extract(self, frames, starts=None):
        """frames [N,H,W,3] uint8 (numpy or device tensor) -> (starts, conv5b [n,1,512,2,7,7] float32 numpy),
        the array the reference pickles into `<video>.c3d`."""
        import torch
        x = frames if torch.is_tensor(frames) else torch.as_tensor(np.ascontiguousarray(frames, np.uint8))
        x = x.to(self.engine.device).contiguous()
        if starts is None:
            starts = window_starts(x.shape[0])
        if not starts:
            return starts, np.zeros((0, 1, 512, 2, 7, 7), np.float32)
        feats, _ = self.engine.forward_frames(x, starts, self.mean)
        return starts, feats.reshape(len(starts), 1, 512, 2, 7, 7).cpu().numpy()   # channel c*2+d (gaze_rnn.py:494-497)

    def extract_to_files(self, frames, feat_dir, video_id, layer='conv5b'):
        """Writes `<feat_dir>/<video_id>/<start+1:06d>.<layer>` blobs and the collected `.c3d` pickle."""
        starts, feats = self.extract(frames)
        vdir = os.path.join(feat_dir, video_id)
        os.makedirs(vdir, exist_ok=True)
        for s, f in zip(starts, feats):
            write_binary_blob(os.path.join(vdir, '{0:06d}.{1}'.format(s + 1, layer)), f)
        return process_c3d_features(vdir, layer)
