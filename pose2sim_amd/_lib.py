"""ctypes binding of the C-ABI in include/p2s.h (csrc/libp2s_hip.so).

The library is the only compute path: there is no CPU fallback.  ``load()`` raises if the shared
object is missing; ``Context()`` raises if no MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('P2S_LIB') or os.path.join(_HERE, 'csrc', 'libp2s_hip.so')   # P2S_LIB: kernel experiments only

P2S_F32, P2S_F64 = 0, 1
P2S_MAX_CAMS = 32
P2S_MAX_PERSONS_TOTAL = 48
P2S_MAX_PERSONS_PER_CAM = 16
P2S_MAX_COMBINATIONS = 1 << 20
P2S_JSON_UNREADABLE, P2S_JSON_NO_PEOPLE_LIST = -1, -2
P2S_JSON_PERSON_NO_LIST, P2S_JSON_PERSON_NOT_NUMERIC = -1, -2


class TriParams(C.Structure):
    _fields_ = [('reproj_error_threshold', C.c_double),
                ('likelihood_threshold', C.c_double),
                ('min_cameras', C.c_int32),
                ('undistort_points', C.c_int32),
                ('handle_lr_swap', C.c_int32),
                ('reserved', C.c_int32)]


class AssocParams(C.Structure):
    _fields_ = [('reconstruction_error_threshold', C.c_double),
                ('min_affinity', C.c_double),
                ('min_cameras', C.c_int32),
                ('max_iter', C.c_int32),
                ('w_rank', C.c_double),
                ('tol', C.c_double),
                ('w_sparse', C.c_double)]


class SingleParams(C.Structure):
    _fields_ = [('reproj_error_threshold', C.c_double),
                ('likelihood_threshold', C.c_double),
                ('min_cameras', C.c_int32),
                ('reserved', C.c_int32)]


# name -> (restype, argtypes); every symbol include/p2s.h declares
SIGNATURES = {
    'p2s_version': (C.c_int, []),
    'p2s_last_error': (C.c_char_p, []),
    'p2s_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'p2s_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'p2s_destroy': (C.c_int, [C.c_void_p]),
    'p2s_set_stream': (C.c_int, [C.c_void_p, C.c_void_p]),
    'p2s_synchronize': (C.c_int, [C.c_void_p]),
    'p2s_set_calibration': (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 6),
    'p2s_triangulate_device': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.POINTER(TriParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'p2s_triangulate_host': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.POINTER(TriParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'p2s_associate_device': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.POINTER(AssocParams), C.c_void_p]),
    'p2s_associate_host': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.POINTER(AssocParams), C.c_void_p]),
    'p2s_associate_single_device': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.POINTER(SingleParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    'p2s_associate_single_host': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(SingleParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    'p2s_set_tuning': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'p2s_get_tri_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'p2s_get_assoc_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'p2s_butterworth_host': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'p2s_filter_columns_host': (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'p2s_trc_metrics_host': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'p2s_timing_begin': (C.c_int, [C.c_void_p]),
    'p2s_timing_end': (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    'p2s_json_parse': (C.c_int, [C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    'p2s_json_free': (C.c_int, [C.c_void_p]),
    'p2s_json_people_counts': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'p2s_json_person_lengths': (C.c_int, [C.c_void_p, C.c_void_p]),
    'p2s_json_gather_keypoints': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                            C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]),
    'p2s_json_gather_people': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                         C.c_void_p, C.POINTER(C.c_int64)]),
    'p2s_assoc_argmax_rows': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'p2s_assoc_unique_rows': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'p2s_assoc_filter_rows': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'p2s_json_rewrite_people': (C.c_int, [C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_int32, C.c_void_p]),
    'p2s_trc_append_rows': (C.c_int, [C.c_char_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    'p2s_format_float_repr': (C.c_int, [C.c_double, C.c_char_p, C.c_int32]),
    'p2s_tri_geometry': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32)]),
}

_lib = None


class P2sError(RuntimeError):
    pass


def load():
    """Load libp2s_hip.so and bind every entry point; raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise P2sError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                       '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().p2s_last_error()
        raise P2sError(f'p2s error {rc}: {msg.decode() if msg else "?"}')


def device_count():
    n = C.c_int(0)
    check(load().p2s_device_count(C.byref(n)))
    return n.value
