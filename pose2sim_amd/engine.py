"""Python face of the HIP engine: one ``Engine`` per GPU, thin wrappers over the C-ABI.

Host arrays are NumPy (the library copies them in and out); device-resident operands are passed
as raw pointers (``tensor.data_ptr()``), with torch used only as the allocator / stream owner.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import P2S_F32, P2S_F64, AssocParams, P2sError, SingleParams, TriParams  # noqa: F401


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, 'data_ptr'):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


def as_packed(xyl):
    """Pick the narrowest exact dtype for an observation tensor: float32 when every value is
    float32-representable (RTMLib output is, poseEstimation.py:259), else float64, so that the
    kernel sees exactly the numbers the reference would read from the JSON files."""
    xyl = np.asarray(xyl)
    if xyl.dtype == np.float32:
        return np.ascontiguousarray(xyl), P2S_F32
    x64 = np.ascontiguousarray(xyl, dtype=np.float64)
    x32 = x64.astype(np.float32)
    same = (x32.astype(np.float64) == x64) | np.isnan(x64)
    if same.all():
        return x32, P2S_F32
    return x64, P2S_F64


class Engine:
    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.p2s_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.n_cams = 0

    def close(self):
        if getattr(self, '_h', None):
            self._lib.p2s_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- calibration -----------------------------------------------------------------------
    def set_calibration(self, P, cal=None):
        """P: C projection matrices (3x4).  cal: dict with 'K', 'dist', 'R_mat', 'T', 'optim_K'
        lists (retrieve_calib_params, common.py:254-288) -- needed for undistortion / association."""
        P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 12))
        n = P.shape[0]
        args = [None] * 5
        keep = [P]
        if cal is not None:
            K = np.ascontiguousarray(np.asarray(cal['K'], dtype=np.float64).reshape(n, 9))
            d = np.zeros((n, 5))
            for c in range(n):
                dc = np.asarray(cal['dist'][c], dtype=np.float64).ravel()
                if len(dc) > 5 and np.any(dc[5:] != 0):
                    raise P2sError('only k1,k2,p1,p2[,k3] distortion terms are supported')
                d[c, :min(5, len(dc))] = dc[:5]
            R = np.ascontiguousarray(np.asarray(cal['R_mat'], dtype=np.float64).reshape(n, 9))
            T = np.ascontiguousarray(np.asarray(cal['T'], dtype=np.float64).reshape(n, 3))
            nk = np.ascontiguousarray(np.asarray(cal['optim_K'], dtype=np.float64).reshape(n, 9))
            keep += [K, d, R, T, nk]
            args = [_ptr(K), _ptr(d), _ptr(R), _ptr(T), _ptr(nk)]
        _lib.check(self._lib.p2s_set_calibration(self._h, n, _ptr(P), *args))
        self.n_cams = n

    def set_stream(self, stream_handle):
        """Enqueue on this HIP stream (0 / None = HIP's default stream, which is torch's default)."""
        _lib.check(self._lib.p2s_set_stream(self._h, C.c_void_p(int(stream_handle or 0))))

    def synchronize(self):
        _lib.check(self._lib.p2s_synchronize(self._h))

    # p2s_set_tuning keys (include/p2s.h): experiments and tests only, results never depend on them
    TUNE_TRI_PATH, TUNE_FORCE_TILED, TUNE_NO_OVERLAP, TUNE_SEARCH_JOB, TUNE_DIAG_MODE, TUNE_MAX_SUBSETS, TUNE_DEEP_MIN_SUBSETS = 1, 2, 3, 4, 5, 6, 7
    TUNE_ASSOC_FORM, TUNE_POOL_SINGLES_PCT, TUNE_DEEP_PRUNE, TUNE_SCREEN, TUNE_POOL_TILES = 8, 9, 10, 11, 12
    TRI_PATH_AUTO, TRI_PATH_WORKLIST, TRI_PATH_ONE_TILE, TRI_PATH_POOLED, TRI_PATH_TWO_TILES = 0, 1, 2, 3, 4
    ASSOC_FORM_AUTO, ASSOC_FORM_GENERAL = 0, 1

    def tri_stats(self, reset=False):
        """Counters of this engine's triangulation calls: units that entered the camera-subset search, subsets
        evaluated (fp64), 64-lane evaluation passes, units stopped by the 2^26-subsets-per-level safety valve, the pruned
        passes' per-camera errors and candidates, subsets looked at by the pooled kernel's fp32 screen and its passes."""
        out = np.zeros(8, dtype=np.uint64)
        _lib.check(self._lib.p2s_get_tri_stats(self._h, _ptr(out), 1 if reset else 0))
        return {'search_units': int(out[0]), 'subsets_evaluated': int(out[1]), 'passes': int(out[2]), 'capped_units': int(out[3]),
                'pruned_camera_errors': int(out[4]), 'pruned_subsets': int(out[5]),
                'screened_subsets': int(out[6]), 'screen_passes': int(out[7])}

    def assoc_stats(self, reset=False):
        """Counters of this engine's multi-person association calls: frames with detections, ADMM passes, Jacobi sweeps,
        fp64 operations (the kernels' own count)."""
        out = np.zeros(4, dtype=np.uint64)
        _lib.check(self._lib.p2s_get_assoc_stats(self._h, _ptr(out), 1 if reset else 0))
        return {'frames': int(out[0]), 'admm_passes': int(out[1]), 'jacobi_sweeps': int(out[2]), 'fp64_flops': int(out[3])}

    def set_tuning(self, key, value):
        _lib.check(self._lib.p2s_set_tuning(self._h, int(key), int(value)))

    # -- triangulation ---------------------------------------------------------------------
    @staticmethod
    def tri_params(thr, lik_thr, min_cams, undistort=False, lr_swap=False):
        return TriParams(float(thr), float(lik_thr), int(min_cams), int(bool(undistort)), int(bool(lr_swap)), 0)

    def triangulate(self, xyl, params, swap_idx=None):
        """xyl: [..., C, K, 3] float32/float64 host array (leading dims = frames x persons).
        Returns Q [..., K, 3] f64, err [..., K] f32, n_excl [..., K] u8, mask [..., K] u32."""
        xyl, dtype = as_packed(xyl)
        Cn, K = xyl.shape[-3], xyl.shape[-2]
        if Cn != self.n_cams or xyl.shape[-1] != 3:
            raise P2sError(f'xyl has shape {xyl.shape}; expected [..., {self.n_cams}, K, 3]')
        lead = xyl.shape[:-3]
        nb = int(np.prod(lead)) if lead else 1
        Q = np.empty((nb, K, 3), dtype=np.float64)
        err = np.empty((nb, K), dtype=np.float32)
        nex = np.empty((nb, K), dtype=np.uint8)
        mask = np.empty((nb, K), dtype=np.uint32)
        sw = None
        if swap_idx is not None:
            sw = np.ascontiguousarray(np.asarray(swap_idx, dtype=np.int32))
            if sw.shape != (K,) or sw.min() < 0 or sw.max() >= K:
                raise P2sError('swap_idx must be K indices in [0, K)')
        _lib.check(self._lib.p2s_triangulate_host(self._h, nb, K, dtype, _ptr(xyl), _ptr(sw), C.byref(params),
                                                  _ptr(Q), _ptr(err), _ptr(nex), _ptr(mask)))
        return (Q.reshape(lead + (K, 3)), err.reshape(lead + (K,)), nex.reshape(lead + (K,)),
                mask.reshape(lead + (K,)))

    def triangulate_packed(self, xyl, params, swap_idx=None, pad_blocks=None):
        """Multi-GPU form of triangulate(): the observations go up, the kernels run, and the results STAY on this GPU
        in one packed uint8 torch tensor (parallel.section_offsets layout, sections sized for pad_blocks >= n_blocks
        blocks, zero beyond this call's blocks) -- the operand of the path's single all-gather.  Returns a
        parallel.PackedDeviceResults."""
        import torch
        from . import parallel
        xyl, dtype = as_packed(xyl)
        Cn, K = xyl.shape[-3], xyl.shape[-2]
        if Cn != self.n_cams or xyl.shape[-1] != 3:
            raise P2sError(f'xyl has shape {xyl.shape}; expected [..., {self.n_cams}, K, 3]')
        nb = int(np.prod(xyl.shape[:-3])) if xyl.ndim > 3 else 1
        pad = nb if pad_blocks is None else int(pad_blocks)
        if pad < nb:
            raise P2sError(f'pad_blocks={pad} < {nb} blocks')
        dev = torch.device('cuda', self.device)
        off_e, off_m, off_n, total = parallel.section_offsets(pad * K)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev)
            buf = torch.zeros(max(total, 16), dtype=torch.uint8, device=dev)
            if nb:
                d_x = torch.from_numpy(xyl.reshape(-1)).to(dev)
                d_sw = None
                if swap_idx is not None:
                    d_sw = torch.from_numpy(np.ascontiguousarray(np.asarray(swap_idx, dtype=np.int32))).to(dev)
                self.set_stream(stream.cuda_stream)
                b = buf.data_ptr()
                self.triangulate_device(nb, K, dtype, d_x, d_sw, params, b, b + off_e, b + off_n, b + off_m)
                stream.synchronize()                   # d_x / d_sw may be freed when this returns
        return parallel.PackedDeviceResults(buf, nb, pad, K)

    def triangulate_device(self, n_blocks, K, dtype, d_xyl, d_swap, params, d_Q, d_err, d_nexcl, d_mask):
        """Device-resident operands (tensors or raw pointers); enqueues on the engine's stream."""
        _lib.check(self._lib.p2s_triangulate_device(self._h, int(n_blocks), int(K), int(dtype), _ptr(d_xyl),
                                                    _ptr(d_swap), C.byref(params), _ptr(d_Q), _ptr(d_err),
                                                    _ptr(d_nexcl), _ptr(d_mask)))

    def timing_begin(self):
        _lib.check(self._lib.p2s_timing_begin(self._h))

    def timing_end(self):
        ms = C.c_float(0)
        _lib.check(self._lib.p2s_timing_end(self._h, C.byref(ms)))
        return ms.value

    def tri_geometry(self, K, dtype=P2S_F32):
        fb, th, lds = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self._lib.p2s_tri_geometry(self.n_cams, int(K), int(dtype), C.byref(fb), C.byref(th), C.byref(lds)))
        return {'blocks_per_tile': fb.value, 'threads': th.value, 'lds_bytes': lds.value}

    # -- downstream of the .trc -----------------------------------------
    def butterworth(self, data, b, a, zi):
        """Zero-phase Butterworth filter of every column of data [n_frames][n_cols] (filtering.py:437-471)."""
        data = np.ascontiguousarray(data, dtype=np.float64)
        if data.ndim != 2:
            raise P2sError(f'data has shape {data.shape}; expected [n_frames][n_cols]')
        b = np.ascontiguousarray(b, dtype=np.float64); a = np.ascontiguousarray(a, dtype=np.float64)
        zi = np.ascontiguousarray(zi, dtype=np.float64)
        if len(a) != len(b) or len(zi) != len(b) - 1:
            raise P2sError('b, a and zi must have n, n and n - 1 coefficients')
        out = np.empty_like(data)
        _lib.check(self._lib.p2s_butterworth_host(self._h, data.shape[0], data.shape[1], _ptr(data) if data.size else None,
                                                  len(b), _ptr(b), _ptr(a), _ptr(zi), _ptr(out) if out.size else None))
        return out

    def filter_columns(self, kind, data, params):
        """One of the window / recurrence filters of filtering.py on every column of data [n_frames][n_cols]
        (include/p2s.h: P2S_FILTER_HAMPEL = 1, _GAUSSIAN = 2, _MEDIAN = 3, _ONE_EURO = 4, _KALMAN = 5, with their
        parameters)."""
        data = np.ascontiguousarray(data, dtype=np.float64)
        if data.ndim != 2:
            raise P2sError(f'data has shape {data.shape}; expected [n_frames][n_cols]')
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1)
        out = np.empty_like(data)
        _lib.check(self._lib.p2s_filter_columns_host(self._h, int(kind), data.shape[0], data.shape[1], _ptr(data) if data.size else None,
                                                     _ptr(params) if params.size else None, params.size, _ptr(out) if out.size else None))
        return out

    def trc_metrics(self, xyz, bones):
        """trc_evaluate's per-frame quantities for xyz [F][K][3] and bones [n][2] (parent, child marker indices):
        bone_len [n][F], bone_stats [n][3] (mean, population sd, n_valid), accel [K][F-2], missing [K]."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        if xyz.ndim != 3 or xyz.shape[2] != 3:
            raise P2sError(f'xyz has shape {xyz.shape}; expected [F][K][3]')
        bones = np.ascontiguousarray(np.asarray(bones, dtype=np.int32).reshape(-1, 2))
        F, K = xyz.shape[:2]
        nb = bones.shape[0]
        bone_len = np.full((nb, F), np.nan)
        bone_stats = np.full((nb, 3), np.nan)
        accel = np.full((K, max(F - 2, 0)), np.nan)
        missing = np.zeros(K, dtype=np.int64)
        p = lambda x: _ptr(x) if x.size else None                           # noqa: E731
        _lib.check(self._lib.p2s_trc_metrics_host(self._h, F, K, p(xyz), nb, p(bones), p(bone_len), p(bone_stats), p(accel), p(missing)))
        return bone_len, bone_stats, accel, missing

    # -- association -----------------------------------------------------------------------
    @staticmethod
    def assoc_params(recon_thr, min_affinity, min_cams, max_iter=20, w_rank=50.0, tol=1e-4, w_sparse=0.1):
        """matchSVT constants are the reference's call-site values (personAssociation.py:799)."""
        return AssocParams(float(recon_thr), float(min_affinity), int(min_cams), int(max_iter), float(w_rank),
                           float(tol), float(w_sparse))

    def associate(self, n_persons, kpts, params):
        """n_persons [F][C] int, kpts [rows][Kj][3] (camera-major then person, JSON keypoint order).
        Returns the thresholded matchSVT matrices [F][n_max][n_max] (top-left N_f x N_f valid)."""
        n_persons = np.ascontiguousarray(np.asarray(n_persons, dtype=np.int32))
        F, Cn = n_persons.shape
        if Cn != self.n_cams:
            raise P2sError(f'n_persons has {Cn} cameras; calibration has {self.n_cams}')
        kpts, dtype = as_packed(kpts)
        per_frame = n_persons.sum(axis=1, dtype=np.int64)
        offsets = np.zeros(F + 1, dtype=np.int64)
        np.cumsum(per_frame, out=offsets[1:])
        if kpts.ndim != 3 or kpts.shape[0] != offsets[-1] or kpts.shape[2] != 3:
            raise P2sError(f'kpts has shape {kpts.shape}; expected [{offsets[-1]}, Kj, 3]')
        n_max = int(per_frame.max()) if F else 0
        n_max = max(2, (n_max + 1) & ~1)
        aff = np.zeros((F, n_max, n_max), dtype=np.float64)
        _lib.check(self._lib.p2s_associate_host(self._h, F, kpts.shape[1], n_max, dtype, _ptr(n_persons),
                                                _ptr(offsets), _ptr(kpts) if kpts.size else None, C.byref(params),
                                                _ptr(aff)))
        return aff

    def associate_single(self, n_persons, tracked, reproj_thr, lik_thr, min_cams):
        """Single-person association (personAssociation.py:154-257).  n_persons [F][C] int, tracked [rows][3]
        = (x, y, likelihood) of the tracked keypoint of every detected person, camera-major per frame.
        Returns comb int32 [F][C] (chosen person per camera, -1 = camera off), err [F] (inf: none), Q [F][3]."""
        n_persons = np.ascontiguousarray(np.asarray(n_persons, dtype=np.int32))
        F, Cn = n_persons.shape
        if Cn != self.n_cams:
            raise P2sError(f'n_persons has {Cn} cameras; calibration has {self.n_cams}')
        tracked, dtype = as_packed(np.asarray(tracked).reshape(-1, 3))
        offsets = np.zeros(F + 1, dtype=np.int64)
        np.cumsum(n_persons.sum(axis=1, dtype=np.int64), out=offsets[1:])
        if tracked.shape[0] != offsets[-1]:
            raise P2sError(f'tracked has {tracked.shape[0]} rows; n_persons sums to {offsets[-1]}')
        comb = np.full((F, Cn), -1, dtype=np.int32)
        err = np.full(F, np.inf)
        Q = np.full((F, 3), np.nan)
        prm = SingleParams(float(reproj_thr), float(lik_thr), int(min_cams), 0)
        _lib.check(self._lib.p2s_associate_single_host(self._h, F, dtype, _ptr(n_persons), _ptr(offsets),
                                                       _ptr(tracked) if tracked.size else None, C.byref(prm),
                                                       _ptr(comb), _ptr(err), _ptr(Q)))
        return comb, err, Q

    def associate_single_device(self, F, dtype, d_n_persons, d_offsets, d_tracked, reproj_thr, lik_thr, min_cams,
                                d_comb, d_err, d_Q):
        """Device-pointer form of associate_single; the caller has checked persons per camera <= 16 and the
        number of combinations per frame (the host entry point does both)."""
        prm = SingleParams(float(reproj_thr), float(lik_thr), int(min_cams), 0)
        _lib.check(self._lib.p2s_associate_single_device(self._h, int(F), int(dtype), _ptr(d_n_persons), _ptr(d_offsets),
                                                         _ptr(d_tracked), C.byref(prm), _ptr(d_comb), _ptr(d_err),
                                                         _ptr(d_Q)))

    def associate_device(self, F, Kj, n_max, dtype, d_n_persons, d_offsets, d_kpts, params, d_aff):
        _lib.check(self._lib.p2s_associate_device(self._h, int(F), int(Kj), int(n_max), int(dtype),
                                                  _ptr(d_n_persons), _ptr(d_offsets), _ptr(d_kpts),
                                                  C.byref(params), _ptr(d_aff)))
