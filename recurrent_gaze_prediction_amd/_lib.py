"""ctypes binding of librgp_hip.so (the C ABI declared in include/rgp.h).

The product path has NO fallback: if the HIP library is missing or a call fails,
this module raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C recurrent_gaze_prediction_amd/csrc``).
"""
import ctypes
import os

# torch must be imported BEFORE librgp_hip.so is dlopen'ed: the library's libamdhip64
# dependency then resolves to the HIP runtime torch has already loaded, so device
# pointers, streams and events are shared.  Loading it first would bring in a second
# HIP runtime that sees no context ("no ROCm-capable device is detected").
import torch  # noqa: F401

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'librgp_hip.so')

RGP_F32, RGP_BF16 = 0, 1
RGP_ETIMEOUT = -5
RGP_GRCN_SAVE_FOR_BACKWARD, RGP_GRCN_PER_STEP, RGP_GRCN_UNFOLDED_HEAD = 1, 2, 4
RGP_C3D_SAVE_FOR_BACKWARD, RGP_C3D_KERNELS_IGEMM, RGP_C3D_KERNELS_TILE128, RGP_C3D_CONV2A_ROWWISE = 1, 2, 4, 8
RGP_FAULT_SEQ_LOST_MEMBER, RGP_FAULT_BPTT_LOST_MEMBER = 1, 2
RGP_GRCN_GRADS_TOP, RGP_GRCN_GRADS_GRU, RGP_GRCN_GRADS_PROJ = 0, 1, 2
RGP_SQNORM_PARTIALS = 256          # include/rgp.h
DTYPES = {'f32': RGP_F32, 'fp32': RGP_F32, 'float32': RGP_F32, 'bf16': RGP_BF16, 'bfloat16': RGP_BF16}

c_void_p, c_int, c_size_t, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p


class RgpError(RuntimeError):
    code = None        # the library's RGP_E* return value, when the error came from a library call


class GrcnWeights(ctypes.Structure):
    FIELDS = ('proj_c3d_W', 'proj_c3d_b', 'gru_Wz', 'gru_Uz', 'gru_Wr', 'gru_Ur', 'gru_W', 'gru_U',
              'bn_gamma', 'bn_beta', 'up_weight1', 'up_weight2', 'up_weight3', 'out_W', 'out_b')
    _fields_ = [(n, c_void_p) for n in FIELDS]


class FcGruWeights(ctypes.Structure):
    FIELDS = ('proj_c3d_W', 'proj_c3d_b', 'gates_kernel', 'gates_bias', 'candidate_kernel', 'candidate_bias',
              'proj_out_W', 'proj_out_b')
    _fields_ = [(n, c_void_p) for n in FIELDS]


class ShallowNetWeights(ctypes.Structure):
    FIELDS = ('conv1_w', 'conv1_b', 'conv2_w', 'conv2_b', 'conv3_w', 'conv3_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b')
    _fields_ = [(n, c_void_p) for n in FIELDS]


class CascadeWeights(ctypes.Structure):
    FIELDS = ('proj_c3d_W', 'proj_c3d_b', 'bottom_Wz', 'bottom_Uz', 'bottom_Wr', 'bottom_Ur', 'bottom_W', 'bottom_U',
              'upsampling_weight', 'top_Wz', 'top_Uz', 'top_Wr', 'top_Ur', 'top_W', 'top_U',
              'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b')
    _fields_ = [(n, c_void_p) for n in FIELDS] + [('shallownet', ShallowNetWeights)]


class C3DWeights(ctypes.Structure):
    _fields_ = [('w', c_void_p * 8), ('b', c_void_p * 8)]


# name -> (restype, argtypes); every symbol include/rgp.h declares
SIGNATURES = {
    'rgp_last_error': (c_char_p, []),
    'rgp_version': (c_int, []),
    'rgp_device_arch': (c_int, [c_char_p, c_int]),
    'rgp_grcn_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int]),
    'rgp_grcn_destroy': (c_int, [c_void_p]),
    'rgp_grcn_status': (c_int, [c_void_p, c_void_p]),
    'rgp_grcn_inject_fault': (c_int, [c_void_p, c_int]),
    'rgp_c3d_layer_kernel_name': (c_char_p, [c_void_p, c_int, c_int]),
    'rgp_grcn_workspace_bytes': (c_size_t, [c_void_p]),
    'rgp_grcn_bind_workspace': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'rgp_grcn_set_weights': (c_int, [c_void_p, ctypes.POINTER(GrcnWeights), c_void_p]),
    'rgp_grcn_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rgp_grcn_forward_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rgp_proj_fwd': (c_int, [c_void_p, c_void_p, c_void_p]),
    'rgp_convgru_xconv_fwd': (c_int, [c_void_p, c_void_p]),
    'rgp_convgru_seq_fwd': (c_int, [c_void_p, c_void_p]),
    'rgp_head_fwd': (c_int, [c_void_p, c_void_p, c_void_p]),
    'rgp_grcn_read_buffer': (c_int, [c_void_p, c_char_p, c_void_p, c_void_p]),
    'rgp_grcn_buffer_elems': (c_size_t, [c_void_p, c_char_p]),
    'rgp_grcn_backward_input': (c_int, [c_void_p, c_void_p, c_void_p]),
    'rgp_grcn_backward_from_states': (c_int, [c_void_p, c_void_p, ctypes.POINTER(GrcnWeights), c_void_p]),
    'rgp_softmax_xent_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'rgp_fcgru_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int]),
    'rgp_fcgru_destroy': (c_int, [c_void_p]),
    'rgp_fcgru_workspace_bytes': (c_size_t, [c_void_p]),
    'rgp_fcgru_bind_workspace': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'rgp_fcgru_set_weights': (c_int, [c_void_p, ctypes.POINTER(FcGruWeights), c_void_p]),
    'rgp_fcgru_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rgp_fcgru_create_ex': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int]),
    'rgp_fcgru_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(FcGruWeights), c_int, c_void_p]),
    'rgp_shallownet_create_ex': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int]),
    'rgp_shallownet_backward': (c_int, [c_void_p, c_int, c_void_p, ctypes.POINTER(ShallowNetWeights), c_void_p]),
    'rgp_shallownet_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int]),
    'rgp_shallownet_destroy': (c_int, [c_void_p]),
    'rgp_shallownet_workspace_bytes': (c_size_t, [c_void_p]),
    'rgp_shallownet_bind_workspace': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'rgp_shallownet_set_weights': (c_int, [c_void_p, ctypes.POINTER(ShallowNetWeights), c_void_p]),
    'rgp_shallownet_forward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'rgp_cascade_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int]),
    'rgp_cascade_destroy': (c_int, [c_void_p]),
    'rgp_cascade_workspace_bytes': (c_size_t, [c_void_p]),
    'rgp_cascade_bind_workspace': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'rgp_cascade_set_weights': (c_int, [c_void_p, ctypes.POINTER(CascadeWeights), c_void_p]),
    'rgp_cascade_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rgp_cascade_read_buffer': (c_int, [c_void_p, c_char_p, c_void_p, c_void_p]),
    'rgp_cascade_create_ex': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int]),
    'rgp_cascade_backward': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(CascadeWeights), c_void_p, c_void_p]),
    'rgp_c3d_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int]),
    'rgp_c3d_destroy': (c_int, [c_void_p]),
    'rgp_c3d_workspace_bytes': (c_size_t, [c_void_p]),
    'rgp_c3d_bind_workspace': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'rgp_c3d_set_weights': (c_int, [c_void_p, ctypes.POINTER(C3DWeights), c_void_p]),
    'rgp_c3d_forward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'rgp_c3d_forward_frames': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_int), c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    'rgp_c3d_frames_to_video': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_int), c_int, c_void_p,
                                        c_void_p, c_void_p]),
    'rgp_c3d_create_ex': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int]),
    'rgp_c3d_param_elems': (c_size_t, [c_void_p]),
    'rgp_c3d_param_offset': (c_size_t, [c_void_p, c_int, c_int]),
    'rgp_c3d_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'rgp_c3d_read_grad_image': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rgp_c3d_read_layer': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rgp_c3d_layer_elems': (c_size_t, [c_void_p, c_int, c_int]),
    'rgp_grcn_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(GrcnWeights), c_int, c_void_p]),
    'rgp_adam_clip_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_int,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                   c_void_p, c_void_p]),
    'rgp_global_sqnorm': (c_int, [c_void_p, ctypes.c_longlong, c_void_p, c_void_p]),
    'rgp_adam_clip_step_ext': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_int, c_int,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       c_void_p, c_void_p]),
    'rgp_lr_schedule_step': (c_int, [c_void_p, ctypes.c_float, ctypes.c_float, c_int, ctypes.c_float, ctypes.c_float,
                                     c_void_p, c_void_p]),
    'rgp_adam_clip_step_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_int, c_void_p,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p]),
    'rgp_momentum_clip_step': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_int, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_float, c_void_p, c_void_p]),
    'rgp_rmsprop_clip_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_int,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                      c_void_p, c_void_p]),
    'rgp_l2_loss_fwd': (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p, c_void_p, c_void_p]),
    'rgp_dropout_mask': (c_int, [c_void_p, ctypes.c_longlong, ctypes.c_float, ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p]),
    'rgp_dropout_apply': (c_int, [c_void_p, c_void_p, ctypes.c_longlong, ctypes.c_float, c_void_p]),
    'rgp_fcgru_set_dropout': (c_int, [c_void_p, ctypes.c_float, c_void_p]),
    'rgp_cascade_set_dropout': (c_int, [c_void_p, ctypes.c_float, c_void_p]),
    'rgp_c3d_wait_layer_grads': (c_int, [c_void_p, c_int, c_void_p]),
    'rgp_grcn_wait_grads': (c_int, [c_void_p, c_int, c_void_p]),
    'rgp_grcn_grads_top_early': (c_int, [c_void_p]),
    'rgp_grcn_persistent_workgroups': (c_int, [c_void_p]),
    'rgp_grcn_profile_enable': (c_int, [c_void_p, c_int]),
    'rgp_grcn_profile_read': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]),
    'rgp_c3d_profile_enable': (c_int, [c_void_p, c_int]),
    'rgp_c3d_profile_read': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]),
}
GRCN_STAGES = ('proj', 'xconv', 'convgru_seq', 'head', 'softmax')
C3D_STAGES = ('conv1a', 'conv2a', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'conv5a', 'conv5b', 'video_prep')

_lib = None


def load():
    """Load librgp_hip.so and bind every declared symbol.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RgpError('%s not found: the HIP extension is not built (run __graft_entry__.build()); '
                       'there is no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def kernel_source_hashes():
    """{file name: sha256[:16]} of the library's sources (csrc/ + include/rgp.h).  A PMC summary records them
    (scripts/pmc_summary.py) and bench.py compares: counters taken from another build of a kernel are reported as stale."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, 'csrc', '*.hip')) + glob.glob(os.path.join(here, 'csrc', '*.h')))
    files.append(os.path.join(os.path.dirname(here), 'include', 'rgp.h'))
    out = {}
    for f in files:
        if os.path.exists(f):
            with open(f, 'rb') as fh:
                out[os.path.basename(f)] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


# sources that define a kernel's code and launch geometry (prefix of the name rgp_c3d_layer_kernel_name returns)
KERNEL_SOURCES = {
    'conv_patch_bf16_kernel': ('conv_patch.hip.h', 'rgp_conv_patch.hip', 'rgp_c3d.hip', 'rgp_c3d_plan.h', 'rgp_host.h'),
    'conv_patch_slab_bf16_kernel': ('conv_patch_slab.hip.h', 'conv_patch.hip.h', 'rgp_conv_patch.hip', 'rgp_c3d.hip', 'rgp_c3d_plan.h',
                                    'rgp_host.h'),
    'conv_patch14_bf16_kernel': ('conv_patch14.hip.h', 'rgp_conv_patch.hip', 'rgp_c3d.hip', 'rgp_c3d_plan.h', 'rgp_host.h'),
}


def check(rc):
    if rc != 0:
        err = RgpError('librgp_hip error %d: %s' % (rc, load().rgp_last_error().decode()))
        err.code = int(rc)
        raise err
