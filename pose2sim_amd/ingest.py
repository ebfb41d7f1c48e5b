"""Batch OpenPose-JSON ingest through the native parser (csrc/p2s_ingest.cpp, include/p2s.h).

One parse of every file of a trial on host threads instead of the reference's json.load per (frame,
camera, person) (triangulation.py:607-653, :77-90; personAssociation.py:260-274).  The numbers are the
ones Python's float() would produce (correctly rounded), a file is unreadable exactly when json.load
would raise.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (P2S_F32, P2S_F64, P2S_JSON_NO_PEOPLE_LIST, P2S_JSON_PERSON_NO_LIST,  # noqa: F401
                   P2S_JSON_PERSON_NOT_NUMERIC, P2S_JSON_UNREADABLE)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class JsonBatch:
    """Parsed files; paths[i] = '' or None means "no file for this slot" (reads as unreadable)."""

    def __init__(self, paths, n_threads=0):
        self._lib = _lib.load()
        enc = [(p or '').encode() if not isinstance(p, bytes) else p for p in paths]
        self.n_files = len(enc)
        offsets = np.zeros(self.n_files + 1, dtype=np.int64)
        if enc:
            np.cumsum(np.fromiter((len(e) for e in enc), dtype=np.int64, count=len(enc)), out=offsets[1:])
        blob = b''.join(enc)
        h = C.c_void_p()
        _lib.check(self._lib.p2s_json_parse(blob, _ptr(offsets), self.n_files, int(n_threads), C.byref(h)))
        self._h = h
        self.counts = np.zeros(self.n_files, dtype=np.int32)
        self.person_base = np.zeros(self.n_files + 1, dtype=np.int64)
        _lib.check(self._lib.p2s_json_people_counts(self._h, _ptr(self.counts), self.person_base.ctypes.data_as(C.c_void_p)))
        self._lengths = None

    def close(self):
        if getattr(self, '_h', None):
            self._lib.p2s_json_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def person_lengths(self):
        """Per person (file-major): len(pose_keypoints_2d) or a P2S_JSON_PERSON_* code."""
        if self._lengths is None:
            self._lengths = np.zeros(int(self.person_base[-1]), dtype=np.int32)
            if self._lengths.size:
                _lib.check(self._lib.p2s_json_person_lengths(self._h, _ptr(self._lengths)))
        return self._lengths

    def gather_keypoints(self, keypoint_ids, max_persons, file_offsets, person_stride, out):
        """out: preallocated float32 / float64 array; see p2s_json_gather_keypoints.  -> n_inexact."""
        ids = np.ascontiguousarray(keypoint_ids, dtype=np.int32)
        file_offsets = np.ascontiguousarray(file_offsets, dtype=np.int64)
        if file_offsets.shape != (self.n_files,):
            raise ValueError('one offset per parsed file is required')
        dtype = P2S_F32 if out.dtype == np.float32 else P2S_F64
        if out.dtype not in (np.float32, np.float64) or not out.flags.c_contiguous:
            raise ValueError('out must be a C-contiguous float32 / float64 array')
        top = int(file_offsets.max(initial=-1))
        if top >= 0 and top + (max_persons - 1) * person_stride + 3 * len(ids) > out.size:
            raise ValueError('offsets reach beyond the output array')
        bad = C.c_int64(0)
        _lib.check(self._lib.p2s_json_gather_keypoints(self._h, _ptr(ids), len(ids), int(max_persons),
                                                       file_offsets.ctypes.data_as(C.c_void_p), int(person_stride), dtype,
                                                       out.ctypes.data_as(C.c_void_p), C.byref(bad)))
        return bad.value

    def gather_people(self, file_index, person_index, n_values, dtype=np.float64):
        file_index = np.ascontiguousarray(file_index, dtype=np.int64)
        person_index = np.ascontiguousarray(person_index, dtype=np.int32)
        out = np.empty((len(file_index), int(n_values)), dtype=dtype)
        bad = C.c_int64(0)
        _lib.check(self._lib.p2s_json_gather_people(self._h, _ptr(file_index), _ptr(person_index), len(file_index),
                                                    int(n_values), P2S_F32 if out.dtype == np.float32 else P2S_F64,
                                                    _ptr(out), C.byref(bad)))
        return out, bad.value
