"""Feature / batch wire format and loader shim.

Mirrors the parts of /root/reference/crc_input_data_seq.py that define what flows INTO the
gaze path, without its dataset paths:
  * ``.c3d`` files: a pickle of float32 ``[N,1,512,2,7,7]`` written by
    extract_C3D_features.py:763-798 and read + squeezed by crc_input_data_seq.py:326-330;
  * ``seq2batch``: T-chunking of a clip, last chunk re-taken from the tail, short clips tiled
    (crc_input_data_seq.py:383-420, Python-2 integer division);
  * ``CRCDataSet``: the 6-tuple ``next_batch`` contract with the reference's epoch wrap and its
    fixed shuffle seed (crc_input_data_seq.py:60-156);
  * ``DeviceFeeder``: pinned-memory, double-buffered host->device staging on a side HIP stream
    so the ~200 KB/frame feature upload overlaps the kernels of the previous batch.
"""
import pickle

import numpy as np


def write_c3d_file(path, conv5b):
    """conv5b [N,512,2,7,7] (or [N,1024,7,7], channel = c*2+d) -> the reference's .c3d pickle."""
    a = np.asarray(conv5b, np.float32)
    if a.ndim == 4:
        a = a.reshape(a.shape[0], 512, 2, 7, 7)
    assert a.shape[1:] == (512, 2, 7, 7), a.shape
    with open(path, 'wb') as f:
        pickle.dump(a[:, None], f, protocol=2)          # [N,1,512,2,7,7], protocol the py2 writer used


def read_c3d_file(path):
    """-> float32 [N,512,2,7,7] (np.squeeze of the stored blob, crc_input_data_seq.py:326-330)."""
    with open(path, 'rb') as f:
        c3d = pickle.load(f, encoding='latin1')
    c3d = np.squeeze(np.asarray(c3d, np.float32))
    if c3d.ndim == 4:                                    # a single window: squeeze ate N
        c3d = c3d[None]
    assert c3d.shape[-2:] == (7, 7)
    return c3d


def fold_c3d(batch_c3d, batch_size):
    """[B,T,512,2,7,7] -> [B,T,1024,7,7] (gaze_rnn.py:494-497)."""
    return np.reshape(batch_c3d, [batch_size, -1, 1024, 7, 7])


def seq2batch(data, seq_len):
    """crc_input_data_seq.py:383-420."""
    is_list = isinstance(data, list)
    data_len = len(data) if is_list else data.shape[0]
    seqs = []
    if data_len > seq_len:
        num_parts = int(data_len / seq_len)
        eq_parts = data[:num_parts * seq_len]
        seqs.extend([eq_parts[i:i + seq_len] for i in range(0, len(eq_parts), seq_len)])
        seqs.append(data[-seq_len:])                     # remainder: the LAST seq_len frames
    else:
        tile_count = seq_len // data_len + 1             # Python-2 "/" on ints
        if is_list:
            seqs.append(np.tile(data, [tile_count])[:seq_len])
        else:
            seqs.append(np.tile(data, [tile_count] + [1] * (data.ndim - 1))[:seq_len])
    return np.asarray(seqs)


class CRCDataSet(object):
    """crc_input_data_seq.py:60-156: arrays indexed by clip-chunk; ``next_batch`` wraps to the start
    of the epoch when the batch would run past the end."""

    SHUFFLE_SEED = 3027300

    def __init__(self, images, gazemaps, fixationmaps, c3ds, pupils, clipnames, shuffle=False):
        self.images, self.gazemaps = np.asarray(images), np.asarray(gazemaps)
        self.fixationmaps, self.c3ds = np.asarray(fixationmaps), np.asarray(c3ds)
        self.pupils, self.clipnames = np.asarray(pupils), list(clipnames)
        assert len(self.gazemaps) == len(self.fixationmaps) == len(self.images) == len(self.c3ds)
        self.epochs_completed = 0
        self.index_in_epoch = 0
        if shuffle:
            perm = list(range(self.image_count()))
            np.random.RandomState(self.SHUFFLE_SEED).shuffle(perm)
            self.images, self.gazemaps = self.images[perm], self.gazemaps[perm]
            self.fixationmaps, self.c3ds = self.fixationmaps[perm], self.c3ds[perm]
            self.pupils = self.pupils[perm]
            self.clipnames = [self.clipnames[i] for i in perm]

    def __len__(self):
        return self.image_count()

    def __repr__(self):
        return 'CRC/Hollywood Dataset Split, %d instances' % len(self)

    def image_count(self):
        return len(self.c3ds)

    def next_batch(self, batch_size):
        start = self.index_in_epoch
        self.index_in_epoch += batch_size
        if self.index_in_epoch > self.image_count():
            self.epochs_completed += 1
            start = 0
            self.index_in_epoch = batch_size
            assert batch_size <= self.image_count()
        end = self.index_in_epoch
        idx = slice(start, end)
        return (self.images[idx], self.gazemaps[idx], self.fixationmaps[idx], self.c3ds[idx], self.pupils[idx],
                self.clipnames[start:end])


def clip_to_dataset(images, gazemaps, fixationmaps, c3d, pupils, clipname, seq_len):
    """One clip's per-frame arrays -> a CRCDataSet of its T-chunks (the per-folder part of
    read_crc_data_sets, crc_input_data_seq.py:423-501)."""
    n = min(len(images), len(gazemaps), len(fixationmaps), len(c3d), len(pupils))
    parts = [seq2batch(np.asarray(a)[:n], seq_len) for a in (images, gazemaps, fixationmaps, c3d, pupils)]
    names = ['%s#%d' % (clipname, i) for i in range(len(parts[0]))]
    return CRCDataSet(parts[0], parts[1], parts[2], parts[3], parts[4], names)


class DeviceFeeder(object):
    """Double-buffered host->device upload of feature batches.

    ``for h in DeviceFeeder(batches, device): use(h.tensor); h.release()`` -- batch i+1 is copied
    from pinned host memory on a side stream while the caller's kernels for batch i run.  The copy
    is ordered before its first use by a stream wait; ``release()`` records, on the consumer's
    stream, the point after which the slot may be re-filled."""

    def __init__(self, batches, device='cuda:0', depth=2):
        import torch
        self.torch, self.device, self.depth = torch, torch.device(device), depth
        self.batches = iter(batches)
        self.copy_stream = torch.cuda.Stream(self.device)
        self.slots = []
        self.n_staged = 0

    def _stage(self, arr):
        torch = self.torch
        host = torch.as_tensor(np.ascontiguousarray(arr, np.float32))
        if len(self.slots) < self.depth:
            slot = {'pin': torch.empty(host.shape, dtype=torch.float32).pin_memory(),
                    'dev': torch.empty(host.shape, dtype=torch.float32, device=self.device), 'free': None}
            self.slots.append(slot)
        else:
            slot = self.slots[self.n_staged % self.depth]
            assert slot['pin'].shape == host.shape, 'batches must share one shape'
            h = slot.get('handle')
            if h is not None and h._slot is not None:
                # the consumer still holds this batch un-released: its kernels may not even be queued yet, so the
                # event recorded at hand-over does not cover them -- drain the consumer's stream before re-filling
                h._stream.synchronize()
                h.release()
            if slot['free'] is not None:
                slot['free'].synchronize()               # the pinned source is also being re-used
        self.n_staged += 1
        slot['pin'].copy_(host)
        with torch.cuda.stream(self.copy_stream):
            if slot['free'] is not None:
                self.copy_stream.wait_event(slot['free'])     # consumer finished with this buffer
            slot['dev'].copy_(slot['pin'], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return slot, ready

    def __iter__(self):
        pending = None
        for arr in self.batches:
            staged = self._stage(arr)
            if pending is not None:
                yield self._hand_over(*pending)
            pending = staged
        if pending is not None:
            yield self._hand_over(*pending)

    def _hand_over(self, slot, ready):
        cur = self.torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        slot['free'] = self.torch.cuda.Event()
        slot['free'].record(cur)             # never an un-recorded event: release() re-records it later on
        slot['handle'] = _Handed(slot['dev'], slot, cur)
        return slot['handle']


class _Handed(object):
    """Device batch + the release() the consumer calls after enqueuing the kernels that read it."""

    def __init__(self, tensor, slot, stream):
        self.tensor, self._slot, self._stream = tensor, slot, stream

    def release(self):
        if self._slot is not None:
            self._slot['free'].record(self._stream)
            self._slot = None

    def __del__(self):
        self.release()
