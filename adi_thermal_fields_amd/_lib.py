"""ctypes binding of libadi_hip.so (include/adi_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises at import
time with the command that builds it.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ADI_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libadi_hip.so')   # env: A/B builds of the kernels

ADI_OK, ADI_ERR_ARG, ADI_ERR_HIP, ADI_ERR_UNSUPPORTED, ADI_ERR_STATE = 0, 1, 2, 3, 4
FACE_NONE, FACE_SCALAR, FACE_FIELD = 0, 1, 2
ZBC_KINDS = {'neumann0': 0, 'dirichlet': 1, 'robin': 2}
SWEEP_GENERAL, SWEEP_NO_DIR, SWEEP_NO_Q, SWEEP_LEAN = 0, 1, 2, 3
# bytes per cell each sweep variant moves (its input arrays + the output), SURVEY.md 8(d) variant rule
SWEEP_BYTES_PER_CELL = {SWEEP_GENERAL: 42, SWEEP_NO_DIR: 33, SWEEP_NO_Q: 34, SWEEP_LEAN: 25}
EXPLICIT_BYTES_PER_CELL = 17
FACES = ('x-', 'x+', 'y-', 'y+', 'z-', 'z+')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "adi_thermal_fields_amd: %s not found. Build the HIP library first:\n"
        "    python -m adi_thermal_fields_amd.build\n"
        "(there is no CPU fallback; the MI355X kernels are the product)" % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

c_int, c_double, c_void_p, c_size_t = ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
c_long = ctypes.c_long
c_int_p = ctypes.POINTER(c_int)
c_double_p = ctypes.POINTER(c_double)
c_void_pp = ctypes.POINTER(c_void_p)

# every exported symbol of include/adi_hip.h: name -> (restype, argtypes)
SIGNATURES = {
    'adi_abi_version': (c_int, []),
    'adi_build_stamp': (ctypes.c_char_p, []),
    'adi_last_error': (ctypes.c_char_p, []),
    'adi_device_count': (c_int, [c_int_p]),
    'adi_device_info': (c_int, [c_int, ctypes.c_char_p, c_int_p, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    'adi_recommended_plane_stride': (ctypes.c_long, [c_int, c_int]),
    'adi_recommended_dims': (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'adi_exposed_mask': (c_int, [c_void_p, c_int, c_int, c_int, c_long, c_int, c_void_p, c_void_p]),
    'adi_build_coeffs': (c_int, [c_void_p, c_int, c_int, c_int, c_long, c_double, c_double, c_double,
                                 c_int_p, c_double_p, c_void_pp, c_int_p, c_double_p, c_void_pp,
                                 c_void_pp, c_void_pp, c_void_p]),
    'adi_build_nbr_flags': (c_int, [c_void_p, c_int, c_int, c_int, c_long, c_void_p, c_void_p]),
    'adi_build_nbr_flags_planes': (c_int, [c_void_p, c_int, c_int, c_int, c_long, c_void_p, c_int, c_int, c_void_p]),
    'adi_build_coeffs_planes': (c_int, [c_void_p, c_int, c_int, c_int, c_long, c_double, c_double, c_double,
                                        c_int_p, c_double_p, c_void_pp, c_int_p, c_double_p, c_void_pp,
                                        c_void_pp, c_void_pp, c_int, c_int, c_void_p]),
    'adi_explicit_rhs': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_double, c_double, c_double,
                                 c_double, c_void_p, c_void_p]),
    'adi_explicit_rhs_planes': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_double, c_double, c_double,
                                        c_double, c_void_p, c_int, c_int, c_void_p]),
    'adi_sweep': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_int, c_int, c_int, c_long, c_int, c_double, c_double, c_double, c_double,
                          c_void_p, c_void_p, c_void_p, c_double_p, c_void_p, c_size_t, c_void_p]),
    'adi_face_constants': (c_int, [c_double, c_double, c_double, c_int_p, c_double_p, c_int_p, c_double_p, c_double_p, c_int_p]),
    'adi_sweep_workspace_bytes': (c_int, [c_int, c_int, c_int, c_int, c_long, ctypes.POINTER(c_size_t)]),
    'adi_sweep_condense': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_long, c_int, c_double, c_double, c_double, c_double,
                                   c_void_p, c_double_p, c_void_p, c_size_t, c_void_p]),
    'adi_explicit_fused_supported': (c_int, [c_int, c_int, c_int, c_long, c_int]),
    'adi_explicit_sweep0': (c_int, [c_int, c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_long, c_int, c_double, c_double, c_double, c_double, c_double,
                                    c_void_p, c_void_p, c_void_p, c_double_p, c_void_p, c_size_t, c_void_p]),
    'adi_explicit_condense0': (c_int, [c_int, c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_int, c_long, c_int, c_double, c_double, c_double, c_double,
                                       c_double, c_void_p, c_void_p, c_double_p, c_void_p, c_size_t, c_void_p]),
    'adi_axis0_dots_supported': (c_int, [c_int, c_int, c_int, c_long]),
    'adi_axis0_dots_workspace': (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    'adi_axis0_dots_setup': (c_int, [c_int, c_double, c_double, c_void_p, c_void_p]),
    'adi_axis0_classify': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p]),
    'adi_axis0_dots_ichunk': (c_int, [c_int]),
    'adi_explicit_rhs_dots': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_double, c_double, c_double,
                                      c_double, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'adi_axis0_dots_finish': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_double, c_double,
                                      c_double, c_double, c_long, c_long, c_void_p, c_void_p]),
    'adi_interface_solve': (c_int, [c_void_p, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p]),
    'adi_interface_pair': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    'adi_axis0_deferred_setup': (c_int, [c_int, c_double, c_double, c_double, c_void_p, c_double_p, c_int_p, c_void_p]),
    'adi_interface_deferred': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_long, c_void_p, c_void_p, c_void_p]),
    'adi_deferred_exact_setup': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_double,
                                         c_long, c_void_p, c_void_p, c_void_p]),
    'adi_interface_solve_uniform': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p]),
    'adi_deferred_exact_coef': (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    'adi_sweep_corrected': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_long, c_int, c_double, c_double, c_double, c_double,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_double_p, c_void_p, c_size_t,
                                    c_void_p]),
    'adi_interface_deferred_lines': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'adi_deferred_lines_apply': (c_int, [c_void_p, c_int, c_long, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p, c_int,
                                         c_void_p, c_void_p]),
    'adi_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_pp, c_void_p, c_void_p,
                         c_void_pp, c_int, c_int, c_int, c_int, c_int, c_long, c_double, c_double, c_double, c_double,
                         c_double, c_double, c_double, c_double_p, c_void_p, c_size_t, c_void_p]),
    'adi_step_queued': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_pp, c_void_p, c_void_p,
                         c_void_pp, c_int, c_int, c_int, c_int, c_int, c_long, c_double, c_double, c_double, c_double,
                         c_double, c_double, c_double, c_double_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    'adi_morph6': (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'adi_flood_outside': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int_p, c_void_p]),
    'adi_pack_frame_f32be': (c_int, [c_void_p, c_int, c_int, c_int, c_long, c_void_p, c_void_p]),
    'adi_count_exposed_faces': (c_int, [c_void_p, c_int, c_int, c_int, c_long, c_int, c_void_p, c_void_p]),
    'adi_birth_planes': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_int, c_double, c_void_p,
                                 c_void_p]),
    'adi_copy_planes': (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_int, c_void_p]),
    'adi_masked_fill': (c_int, [c_void_p, c_void_p, c_size_t, c_double, c_void_p]),
    'adi_mask_or': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'adi_cyl_plan_create': (c_int, [c_int, c_int, c_int, c_long, c_double, c_double, c_double, c_double, c_double, c_double,
                                    c_double, c_double, c_double, c_int, c_int, c_double, c_double, c_double,
                                    c_double, c_double, c_double, c_void_pp]),
    'adi_cyl_plan_create_annular': (c_int, [c_int, c_int, c_int, c_long, c_double, c_double, c_double, c_double, c_double, c_double,
                                            c_double, c_double, c_double, c_double, c_int, c_int, c_double, c_double, c_double,
                                            c_double, c_double, c_double, c_void_pp]),
    'adi_cyl_plan_destroy': (c_int, [c_void_p]),
    'adi_cyl_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_double, c_double, c_void_p]),
    'adi_cyl_sweep': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_void_p]),
    'adi_ctx_create': (c_int, [c_int, c_int, c_int, c_double, c_int, c_void_pp]),
    'adi_ctx_destroy': (c_int, [c_void_p]),
    'adi_ctx_set_mask': (c_int, [c_void_p, c_void_p]),
    'adi_ctx_build_coeffs': (c_int, [c_void_p, c_double, c_double, c_int_p, c_double_p, c_void_pp,
                                     c_int_p, c_double_p, c_void_pp, c_void_p, c_void_p]),
    'adi_ctx_upload_T': (c_int, [c_void_p, c_void_p]),
    'adi_ctx_download_T': (c_int, [c_void_p, c_void_p]),
    'adi_ctx_download_pack': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'adi_ctx_step': (c_int, [c_void_p, c_double, c_double, c_double, c_double, c_double, c_double, c_int]),
    'adi_ctx_last_step_ms': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_float)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = the .so does not export what the header declares
    _fn.restype = _res
    _fn.argtypes = _args


class AdiError(RuntimeError):
    pass


def last_error():
    msg = lib.adi_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''


def check(rc):
    """Map a C-ABI status to the exception the reference would have raised."""
    if rc == ADI_OK:
        return
    msg = last_error()
    if rc == ADI_ERR_ARG:
        raise ValueError(msg)          # e.g. ValueError("bad face"), adi3d_numba_coeff.py:54
    raise AdiError('libadi_hip error %d: %s' % (rc, msg))


def ptr_array(ptrs):
    arr = (c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
