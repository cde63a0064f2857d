"""ctypes binding of libspecmi.so (the C ABI in include/specmi.h).

There is NO CPU fallback: if the library is missing or cannot be loaded, importing the ops
raises, and every forward requires device tensors.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SPECMI_LIB') or os.path.join(_HERE, 'lib', 'libspecmi.so')   # SPECMI_LIB: an alternative build (A/B runs)



def source_hash() -> str:
    """sha256 (16 hex digits) over the kernel sources the library is built from (spec_amd/csrc/*, include/specmi.h) - the
    stamp measured artefacts carry (profiles/pmc_traffic_latest.json), so that a number measured on other kernels is
    recognisable as stale.  Sources, not the binary: a rebuild on another box must not change it."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.h', '.inc')):
            with open(os.path.join(csrc, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
    with open(os.path.join(os.path.dirname(_HERE), 'include', 'specmi.h'), 'rb') as f:
        h.update(b'specmi.h\0' + f.read())
    return h.hexdigest()[:16]


OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_MISSING = 0, 1, 2, 3, 4
MODEL_CAMCALIB, MODEL_HMR, MODEL_SMPL = 0, 1, 2

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class HmrOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'pred_cam',
        'pred_shape', 'pred_pose_6d')]


class ProfEntry(C.Structure):
    _fields_ = [('kernel', C.c_char * 48), ('label', C.c_char * 48), ('ms', C.c_double),
                ('flops', C.c_double), ('bytes', C.c_double), ('launches', C.c_int)]


class SpecmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libspecmi error {code}: {msg}')
        self.code = code


# every exported symbol of include/specmi.h with its prototype
PROTOTYPES = {
    'specmi_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    'specmi_destroy': (C.c_int, [C.c_void_p]),
    'specmi_last_error': (C.c_char_p, [C.c_void_p]),
    'specmi_version': (C.c_char_p, []),
    'specmi_set_option_i32': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    'specmi_set_option_f32': (C.c_int, [C.c_void_p, C.c_char_p, C.c_float]),
    'specmi_get_option_i32': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    'specmi_option_info': (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'specmi_set_tensor_f32': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_int64_p, C.c_int]),
    'specmi_set_tensor_i32': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_int64_p, C.c_int]),
    'specmi_commit': (C.c_int, [C.c_void_p]),
    'specmi_camcalib_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_camcalib_head_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    'specmi_camcalib_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_camcalib_bins': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_cam_params': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_hmr_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(HmrOutputs), C.c_void_p]),
    'specmi_hmr_regress': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(HmrOutputs), C.c_void_p]),
    'specmi_hmr_uncertainty': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_trunk_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    'specmi_hmr_head_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_smpl_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    'specmi_smpl_native': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    'specmi_conv2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'specmi_maxpool3x3s2': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    'specmi_avgpool': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    'specmi_trunk_forward_pair': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    'specmi_camcalib_head_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p]),
    'specmi_crop_normalize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_crop_normalize_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_crop_resize_normalize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p]),
    'specmi_resize_normalize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    'specmi_eval_mesh': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'specmi_eval_joints': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    'specmi_regress_joints': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    'specmi_rotate_points': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'specmi_trunk_plan': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    'specmi_sync_status': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    'specmi_sync_reset': (C.c_int, [C.c_void_p, C.c_void_p]),
    'specmi_debug_poison_sync': (C.c_int, [C.c_void_p, C.c_uint32]),
    'specmi_profile_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'specmi_profile_read': (C.c_int, [C.c_void_p, C.POINTER(ProfEntry), C.c_int, C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """Load libspecmi.so and attach prototypes; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64.so; libspecmi.so links the HIP runtime by SONAME.  Whichever copy the
    # process loads first serves both, and tensors / streams must come from the SAME runtime as the kernels that use
    # them - so torch's goes first (loading /opt/rocm's before torch left the process without a visible device).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} not found: build it with `python -m spec_amd.build` '
            '(or __graft_entry__.build()).  spec_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(handle, rc):
    if rc != OK:
        msg = load().specmi_last_error(handle)
        raise SpecmiError(rc, msg.decode() if msg else '?')
