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
    assert lib.rgp_grcn_buffer_elems(h, b'd2') == 0             # inference plans fold the head into one GEMM: no d1 / d2
    assert lib.rgp_grcn_buffer_elems(h, b'nope') == 0
    hu = ctypes.c_void_p()
    assert lib.rgp_grcn_create(ctypes.byref(hu), 64, 16, 512, 128, _lib.RGP_BF16, _lib.RGP_GRCN_UNFOLDED_HEAD) == 0
    assert lib.rgp_grcn_buffer_elems(hu, b'd2') == 64 * 16 * 2401 * 32
    # the folded plan trades the two intermediate maps (293 MB at 1024 frames) for a 30 MB filter + its 60 MB fp32 source
    assert lib.rgp_grcn_workspace_bytes(hu) - ws_bf16 > 150e6
    lib.rgp_grcn_destroy(hu)
    assert lib.rgp_grcn_create(ctypes.byref(hu), 1, 1, 512, 128, _lib.RGP_BF16, 8) != 0 and b'flags' in lib.rgp_last_error()
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


def test_c3d_kernel_names_and_flags_host_only():
    """rgp_c3d_layer_kernel_name restates nothing: it asks the dispatcher (igemm_tile_choice) which kernel a layer runs."""
    lib = _lib.load()

    def names(flags, n, dtype=_lib.RGP_BF16, max_windows=1024):
        h = ctypes.c_void_p()
        assert lib.rgp_c3d_create_ex(ctypes.byref(h), max_windows, dtype, flags) == 0, lib.rgp_last_error()
        out = [lib.rgp_c3d_layer_kernel_name(h, i, n).decode() for i in range(8)]
        lib.rgp_c3d_destroy(h)
        return out
    d = names(0, 1024)
    assert d[0].startswith('conv1a_pool_bf16_kernel')
    # conv2a's inference forward runs the plane-slab variant of the patch kernel (csrc/conv_patch_slab.hip.h); a training
    # plan (arg-max codes recorded) the row-wise one
    assert d[1] == 'conv_patch_slab_bf16_kernel<64,128,56,16,pool8>' and d[2] == 'conv_patch_bf16_kernel<128,256,28,8,pool1>'
    assert names(_lib.RGP_C3D_SAVE_FOR_BACKWARD, 256, max_windows=256)[1] == 'conv_patch_bf16_kernel<64,128,56,16,pool8>'
    assert d[3] == 'conv_patch_bf16_kernel<256,256,28,8,pool8>'
    assert d[4] == 'conv_patch14_bf16_kernel<256,pool1>' and d[5] == 'conv_patch14_bf16_kernel<512,pool8>'
    assert d[6] == 'conv_patch7_bf16_kernel<image>' and d[7] == 'conv_patch7_bf16_kernel<rows>'
    g = names(_lib.RGP_C3D_KERNELS_IGEMM, 1024)
    assert g[1] == 'igemm_wide_kernel<512x128,bf16,pool8>' and g[3] == 'igemm_wide_kernel<256x256,bf16,pool8>'
    assert g[4] == 'igemm_wide_kernel<256x256,bf16,pool1>' and g[6] == g[7] == 'igemm_stagger_kernel<256x128,bf16,pool1>'
    t = names(_lib.RGP_C3D_KERNELS_IGEMM | _lib.RGP_C3D_KERNELS_TILE128, 1024)
    assert all(x.startswith('igemm_kernel<128x128,bf16') for x in t[1:])
    small = names(0, 2, max_windows=2)
    assert small[1].startswith('conv_patch_slab_bf16_kernel') and small[6] == 'conv_patch7_bf16_kernel<image>'
    f32 = names(0, 1024, dtype=_lib.RGP_F32)
    assert f32[0] == 'igemm_kernel<128x64,f32,pool4>' and f32[3] == 'igemm_stagger_kernel<256x128,f32,pool8>'
    h = ctypes.c_void_p()
    assert lib.rgp_c3d_create_ex(ctypes.byref(h), 2, _lib.RGP_BF16, 16) == -1                     # unknown flag
    # RGP_C3D_CONV2A_ROWWISE: conv2a's inference forward on the row-wise fetch, nothing else changes
    r = names(_lib.RGP_C3D_CONV2A_ROWWISE, 1024)
    assert r[1] == 'conv_patch_bf16_kernel<64,128,56,16,pool8>' and r[0] == d[0] and r[2:] == d[2:]
    assert lib.rgp_c3d_create_ex(ctypes.byref(h), 2, _lib.RGP_BF16, _lib.RGP_C3D_KERNELS_TILE128) == -1
    assert lib.rgp_grcn_create(ctypes.byref(h), 2, 2, 512, 128, _lib.RGP_BF16, 8) == -1           # unknown flag
    assert lib.rgp_grcn_create(ctypes.byref(h), 2, 2, 512, 128, _lib.RGP_BF16, _lib.RGP_GRCN_PER_STEP) == 0
    lib.rgp_grcn_destroy(h)


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from recurrent_gaze_prediction_amd.engine import GrcnEngine
    with pytest.raises(_lib.RgpError):
        GrcnEngine(1, 1)


def test_header_is_valid_c_and_a_c_program_links_the_library(tmp_path):
    """The drop-in boundary is a C ABI: include/rgp.h must compile as C99 (no C++-isms, no torch types) and a plain C
    program must link librgp_hip.so and drive the host-only entry points (plan creation, validation, error strings,
    workspace queries) -- what a maintainer's cgo / JNI / ctypes stub sees."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no C compiler')
    src = tmp_path / 'abi.c'
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "rgp.h"
int main(void) {
  rgp_grcn_t* g = 0;
  rgp_c3d_t* c = 0;
  if (rgp_version() < 100) return 1;
  if (rgp_grcn_create(&g, 8, 35, 512, 128, RGP_BF16, RGP_GRCN_SAVE_FOR_BACKWARD) != RGP_OK) return 2;
  if (rgp_grcn_workspace_bytes(g) == 0) return 3;
  if (rgp_grcn_wait_grads(g, RGP_GRCN_GRADS_TOP, 0) != RGP_ESTATE) return 4;         /* no backward has run */
  if (!strstr(rgp_last_error(), "backward")) return 5;
  if (rgp_grcn_create(&g, 0, 35, 512, 128, RGP_BF16, 0) != RGP_EINVAL) return 6;     /* empty batch */
  if (rgp_c3d_create_ex(&c, 16, RGP_BF16, 0) != RGP_OK) return 7;
  if (rgp_c3d_param_elems(c) != 27655936u + 0u) return 8;                            /* 27.66 M conv parameters (SURVEY 8a row C) */
  printf("%s\n", rgp_c3d_layer_kernel_name(c, 1, 16));
  rgp_c3d_destroy(c);
  return 0;
}
''')
    inc = os.path.join(ROOT, 'include')
    libdir = os.path.join(ROOT, 'recurrent_gaze_prediction_amd')
    assert subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-fsyntax-only', '-I', inc, str(src)],
                          capture_output=True, text=True).returncode == 0
    exe = tmp_path / 'abi'
    r = subprocess.run(['gcc', '-std=c99', '-I', inc, str(src), '-o', str(exe), '-L', libdir, '-l:librgp_hip.so',
                        '-Wl,-rpath,' + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-1000:])
    assert r.stdout.strip().startswith('conv_patch')
