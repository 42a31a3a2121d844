"""CPU: the C-ABI library loads and exports every symbol include/ssak_hip.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    lib_path = os.path.join(ROOT, "ssak_amd", "lib", "libssak_hip.so")
    assert os.path.exists(lib_path), "build first: make (or __graft_entry__.build())"
    lib = ctypes.CDLL(lib_path)
    hdr = open(os.path.join(ROOT, "include", "ssak_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(ssak_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 8
    for n in sorted(names):
        assert hasattr(lib, n), n
    lib.ssak_version.restype = ctypes.c_int
    assert lib.ssak_version() >= 100


def test_binding_rejects_without_gpu_compute():
    import ssak_amd.hip as h
    # argument validation happens on the host, before any launch
    d = h.GemmDesc(0, 8, 8, 0, 0, 8, 8, 8, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0, 1, 0, 1, 0.0, 0, 0, 0, 0, 0)
    rc = h.lib.ssak_gemm_bf16(ctypes.byref(d), None, None, None, None, None, None, None, 0, None)
    assert rc == h.SSAK_ERR_INVALID and b"null" in h.lib.ssak_last_error()
