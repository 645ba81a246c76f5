"""CPU: the C-ABI library loads, exports every symbol include/rgp.h declares, and its
host-only entry points (plan creation, argument validation, call-order errors) behave
as documented.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

from recurrent_gaze_prediction_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'rgp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rgp_[a-z0-9_]+)\s*\(', text)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build()'
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, 'recurrent_gaze_prediction_amd')


def test_every_declared_symbol_is_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), 'missing export ' + s
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)


def test_plan_creation_and_validation_are_host_only():
    lib = _lib.load()
    assert lib.rgp_version() >= 100
    h = ctypes.c_void_p()
    assert lib.rgp_grcn_create(ctypes.byref(h), 64, 16, 512, 128, _lib.RGP_BF16, 0) == 0
    ws_bf16 = lib.rgp_grcn_workspace_bytes(h)
    assert ws_bf16 > 64 * 16 * 49 * 1024 * 2                    # at least the transposed input
    assert lib.rgp_grcn_buffer_elems(h, b'rcn_outputs') == 64 * 16 * 49 * 128
    assert lib.rgp_grcn_buffer_elems(h, b'd2') == 64 * 16 * 2401 * 32
    assert lib.rgp_grcn_buffer_elems(h, b'nope') == 0
    # forward before a workspace is bound is refused with RGP_EWORKSPACE (-3), not a crash
    assert lib.rgp_convgru_seq_fwd(h, None) == -3
    assert b'workspace' in lib.rgp_last_error()
    lib.rgp_grcn_destroy(h)
    h2 = ctypes.c_void_p()
    assert lib.rgp_grcn_create(ctypes.byref(h2), 64, 16, 512, 128, _lib.RGP_F32, 1) == 0
    assert lib.rgp_grcn_workspace_bytes(h2) > ws_bf16           # fp32 operands + saved gates
    lib.rgp_grcn_destroy(h2)
    bad = ctypes.c_void_p()
    assert lib.rgp_grcn_create(ctypes.byref(bad), 0, 16, 512, 128, _lib.RGP_BF16, 0) == -1
    assert lib.rgp_grcn_create(ctypes.byref(bad), 2, 2, 500, 128, _lib.RGP_BF16, 0) == -1
    assert b'multiples of 64' in lib.rgp_last_error()
    assert lib.rgp_grcn_create(ctypes.byref(bad), 2, 2, 512, 128, 7, 0) == -1


def test_c3d_plan_host_only():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.rgp_c3d_create(ctypes.byref(h), 4, _lib.RGP_BF16) == 0
    assert lib.rgp_c3d_workspace_bytes(h) > 4 * 18 * 114 * 116 * 4 * 2
    assert lib.rgp_c3d_layer_elems(h, 0, 4) == 4 * 16 * 56 * 56 * 64      # conv1a after pool1
    assert lib.rgp_c3d_layer_elems(h, 7, 4) == 4 * 2 * 7 * 7 * 512
    assert lib.rgp_c3d_forward(h, None, 1, None, None, None) == -1
    lib.rgp_c3d_destroy(h)
    assert lib.rgp_c3d_create(ctypes.byref(h), 0, _lib.RGP_BF16) == -1


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    with pytest.raises(_lib.RgpError):
        GrcnEngine(1, 1)
