"""CPU-side checks of the drop-in boundary: libnmf_hip.so loads here (no GPU needed) and exports every
symbol include/nmf_hip.h declares; argument validation returns negative codes without touching a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names += re.findall(r"\b(nmf_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from nmf_amd import hip
    decl = _declared()
    assert len(decl) >= 14
    lib = C.CDLL(hip.LIB_PATH)
    for n in decl:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert sorted(decl) == sorted(set(hip.EXPORTS)), "python binding and header disagree"
    assert hip.version() >= 100


def test_bad_arguments_are_rejected_without_a_gpu():
    from nmf_amd import hip
    lib = C.CDLL(hip.LIB_PATH)
    lib.nmf_last_error_string.restype = C.c_char_p
    assert lib.nmf_alpha_pack(None, C.c_int64(10), None, None) == -1
    assert b"nmf_alpha_pack" in lib.nmf_last_error_string()
    p = hip.MarchParams()
    p.n_steps = 100000
    assert lib.nmf_march_count(C.byref(p), None, C.c_int64(1), None, None, None, None, None) == -2
    assert lib.nmf_march_scan(None, C.c_int64(5), C.c_int64(-1), None, None, None, None, C.c_int64(0), None) == -1
    assert lib.nmf_segment_sum(None, None, None, C.c_int64(0), C.c_int32(3), None, None) == 0   # empty is ok
    assert lib.nmf_composite_fwd(None, None, None, C.c_int64(4), C.c_float(25.0), None, None, None) == -1


def test_workspace_sizes_are_host_arithmetic():
    """the size functions a caller allocates by: pure host code, so they answer here; the binned env adjoint's workspace is
    header | one slot per (workgroup of 256 lookups, tile) | 12 corner records of 24 bytes per lookup, and a workspace without
    room for header + slots + one record is refused before anything is launched"""
    from nmf_amd import hip
    lib = C.CDLL(hip.LIB_PATH)
    lib.nmf_sat_lookup_bwd_workspace_bytes.restype = C.c_int64
    f = lambda R: int(lib.nmf_sat_lookup_bwd_workspace_bytes(C.c_int64(R)))
    assert f(0) == f(-5) == 8720
    for R in (1, 256, 257, 247431):
        assert f(R) == 8720 + -(-R // 256) * 1024 * 4 + R * 12 * 24, R
    one = C.c_void_p(16)            # (never dereferenced: the call fails on its arguments)
    rc = lib.nmf_sat_lookup_bwd_binned(one, C.c_int32(512), C.c_int32(1024), one, C.c_int32(3), one, C.c_int64(1000),
                                       C.c_float(0.0), None, C.c_int32(1), one, one, one, None, None, one,
                                       C.c_int64(8720 + 4 * 256 * 4), None)
    assert rc == -1
    lib.nmf_last_error_string.restype = C.c_char_p
    assert b"workspace too small" in lib.nmf_last_error_string()


def test_product_has_no_cpu_fallback():
    import torch
    from nmf_amd import hip
    with pytest.raises(hip.NmfHipError):
        hip.alpha_pack(torch.zeros(64))          # CPU tensor -> refused, never silently computed on the host
