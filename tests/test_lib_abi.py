"""CPU-side checks of the C-ABI library: it builds, loads, and exports what include/p2s.h declares."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as entry
from pose2sim_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'p2s.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(p2s_[a-z_0-9]+)\s*\(', txt)))


def test_library_builds_and_exports_header_symbols():
    entry.build_hip()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)


def test_no_gpu_means_loud_failure():
    """Without a GPU the engine must refuse to run rather than fall back to the CPU."""
    if _lib.device_count() > 0:                        # asked of the library itself (torch may not see the device)
        pytest.skip('GPU present')
    from pose2sim_amd.engine import Engine
    with pytest.raises(_lib.P2sError, match='no HIP device'):
        Engine(0)


def test_geometry_query():
    lib = _lib.load()
    fb, th, lds = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    for C, K, dt in [(8, 26, 0), (4, 26, 0), (16, 131, 0), (32, 26, 0), (8, 26, 1), (3, 17, 0), (2, 1, 1)]:
        assert lib.p2s_tri_geometry(C, K, dt, ctypes.byref(fb), ctypes.byref(th), ctypes.byref(lds)) == 0
        elem = 4 if dt == 0 else 8
        assert (fb.value * C * K * 3 * elem) % 16 == 0          # dwordx4 staging stays aligned
        assert th.value % 64 == 0 and 64 <= th.value <= 1024
        assert lds.value <= 160 * 1024
    assert lib.p2s_tri_geometry(40, 26, 0, None, None, None) != 0


def test_tuning_constants_match_the_header():
    """The engine's copies of the p2s_set_tuning keys and values are the header's."""
    from pose2sim_amd.engine import Engine
    txt = open(os.path.join(ROOT, 'include', 'p2s.h')).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define\s+P2S_((?:TUNE|TRI_PATH|ASSOC_FORM)_[A-Z_]+)\s+(\d+)', txt)}
    assert len(defs) >= 13
    for name, value in defs.items():
        assert getattr(Engine, name) == value, name
