"""ctypes wrapper of oracle/tri_oracle.c (TEST INFRASTRUCTURE: checker and CPU baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libtri_oracle.so')
CAL_STRIDE = 30
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, 'tri_oracle.c')
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(['make', '-C', _HERE, '-B'], check=True, capture_output=True)
        _lib = C.CDLL(_SO)
        _lib.tri_oracle_batch.restype = C.c_int
        _lib.tri_oracle_batch.argtypes = [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 4 + \
            [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    return _lib


def pack_cal(cams):
    """fx fy cx cy | k1 k2 p1 p2 k3 | R[9] | T[3] | newK[9] per camera."""
    n = len(cams['K'])
    out = np.zeros((n, CAL_STRIDE))
    for c in range(n):
        K = np.asarray(cams['K'][c], dtype=np.float64)
        d = np.zeros(5)
        dc = np.asarray(cams['dist'][c], dtype=np.float64).ravel()
        d[:min(5, len(dc))] = dc[:5]
        out[c, 0:4] = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
        out[c, 4:9] = d
        out[c, 9:18] = np.asarray(cams['R_mat'][c], dtype=np.float64).ravel()
        out[c, 18:21] = np.asarray(cams['T'][c], dtype=np.float64).ravel()
        out[c, 21:30] = np.asarray(cams['optim_K'][c], dtype=np.float64).ravel()
    return out


def triangulate_batch(xyl, P, cams, swap_idx, lik_thr, thr, min_cams, lr_swap=False, undistort=False, threads=1):
    """Same contract as oracle.triangulation_ref.triangulate_batch, on [..., C, K, 3]."""
    lib = load()
    xyl = np.ascontiguousarray(xyl, dtype=np.float64)
    Cn, K = xyl.shape[-3], xyl.shape[-2]
    lead = xyl.shape[:-3]
    nb = int(np.prod(lead)) if lead else 1
    Pm = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(Cn, 12))
    cal = pack_cal(cams) if (undistort and cams is not None) else None
    sw = np.ascontiguousarray(np.asarray(swap_idx, dtype=np.int32)) if swap_idx is not None else None
    Q = np.empty((nb, K, 3)); err = np.empty((nb, K)); ne = np.empty((nb, K), dtype=np.int32)
    mask = np.empty((nb, K), dtype=np.uint32)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib.tri_oracle_batch(nb, Cn, K, p(xyl), p(sw), p(Pm), p(cal), float(thr), float(lik_thr), int(min_cams),
                              int(bool(lr_swap)), int(bool(undistort)), p(Q), p(err), p(ne), p(mask), int(threads))
    if rc != 0:
        raise ValueError('tri_oracle_batch: bad arguments')
    return (Q.reshape(lead + (K, 3)), err.reshape(lead + (K,)), ne.reshape(lead + (K,)), mask.reshape(lead + (K,)))
