"""The C-ABI library loads on a CPU-only box and exports every symbol the public headers declare."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(chase_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "chase_amd", "lib", "libchase_hip.so"))
    names = _declared("chase_hip.h") + _declared("chase_hip_solver.h") + _declared("chase_hip_grid.h")
    names += ["dchase_init_", "dchase_", "dchase_finalize_", "zchase_init_", "zchase_", "zchase_finalize_"]
    assert len(names) > 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_fails_loudly_without_gpu_or_runs_on_one():
    from chase_amd import capi
    h = ctypes.c_void_p()
    rc = capi.lib.chase_hip_ctx_create(ctypes.byref(h), 0, None)
    if rc != 0:
        assert rc == -1002                      # CHASE_HIP_ENODEV: no silent CPU fallback
        assert b"HIP device" in capi.lib.chase_hip_last_error()
    else:
        capi.lib.chase_hip_ctx_destroy(h)


def test_host_lapack_provider_is_bound():
    from chase_amd import capi
    assert capi.lib.chase_hip_lapack_provider() != b""
