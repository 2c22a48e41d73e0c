"""chase_amd — MI355X-native backend for the ChASE hot path (Chebyshev filter HEMM, CholQR, Rayleigh-Ritz, residuals).

The product is ``chase_amd/lib/libchase_hip.so`` (hand-written HIP kernels for gfx950 + a C++ host solver behind the
C ABI of ``include/chase_hip.h``).  This package is only the ctypes binding used by tests, ``bench.py`` and
``__graft_entry__.py``; there is no CPU fallback: importing :mod:`chase_amd.capi` raises if the library is missing.
"""
from .capi import lib, LIB_PATH, ChaseHipError, Context  # noqa: F401
