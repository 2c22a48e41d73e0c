"""ctypes binding of include/chase_hip.h (the C ABI of libchase_hip.so).

Fails loudly when the HIP extension is missing — there is deliberately no CPU fallback in the product path.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libchase_hip.so")


class ChaseHipError(RuntimeError):
    def __init__(self, code, where, text=""):
        super().__init__(f"{where} failed with status {code}: {text}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "chase_amd has no CPU fallback.")
    # torch (when present in the process) bundles its own libamdhip64.so.7 / librccl.so.1; importing it first makes
    # the dynamic linker resolve our NEEDED entries to the same runtime instead of loading a second HIP runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

c_int, c_long, c_size_t, c_double, c_char, c_void_p = C.c_int, C.c_long, C.c_size_t, C.c_double, C.c_char, C.c_void_p
P = C.POINTER


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("chase_hip_version", C.c_char_p)
_sig("chase_hip_last_error", C.c_char_p)
_sig("chase_hip_ctx_create", c_int, P(c_void_p), c_int, c_void_p)
_sig("chase_hip_ctx_destroy", c_int, c_void_p)
_sig("chase_hip_ctx_sync", c_int, c_void_p)
_sig("chase_hip_ctx_stream", c_void_p, c_void_p)
_sig("chase_hip_device_info", c_int, c_void_p, P(c_int), P(c_int), P(c_size_t), C.c_char_p, c_int)
_sig("chase_hip_malloc", c_int, c_void_p, P(c_void_p), c_size_t)
_sig("chase_hip_free", c_int, c_void_p, c_void_p)
_sig("chase_hip_memcpy_h2d", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_memcpy_d2h", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_memcpy_d2d", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_memset", c_int, c_void_p, c_void_p, c_int, c_size_t)
_sig("chase_hip_timer_start", c_int, c_void_p)
_sig("chase_hip_timer_stop", c_int, c_void_p, P(C.c_float))
_sig("chase_hip_gemm_d", c_int, c_void_p, c_char, c_int, c_int, c_int, c_double, c_void_p, c_long, c_void_p, c_long,
     c_double, c_void_p, c_long)
_sig("chase_hip_gemm_z", c_int, c_void_p, c_char, c_int, c_int, c_int, P(c_double), c_void_p, c_long, c_void_p,
     c_long, P(c_double), c_void_p, c_long)
_sig("chase_hip_mfma_f64_peak", c_int, c_void_p, P(c_double))
_sig("chase_hip_hbm_copy_peak", c_int, c_void_p, c_size_t, P(c_double))


def check(code, where):
    if code != 0:
        raise ChaseHipError(code, where, lib.chase_hip_last_error().decode())
    return code


def _z2(x):
    x = complex(x)
    return (c_double * 2)(x.real, x.imag)


class DeviceArray:
    """A column-major device matrix/vector owned by a Context (fp64 or complex fp64)."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = c_void_p()
        check(lib.chase_hip_malloc(ctx.h, C.byref(p), self.nbytes), "chase_hip_malloc")
        self.ptr = p.value
        ctx._live.append(self)

    @property
    def ld(self):
        return self.shape[0]

    def offset(self, col):
        return self.ptr + col * self.shape[0] * self.dtype.itemsize

    def upload(self, host):
        a = np.asfortranarray(host, dtype=self.dtype)
        assert a.shape == self.shape, (a.shape, self.shape)
        check(lib.chase_hip_memcpy_h2d(self.ctx.h, self.ptr, a.ctypes.data, self.nbytes), "memcpy_h2d")
        return self

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype, order="F")
        check(lib.chase_hip_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, self.nbytes), "memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib.chase_hip_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    def __init__(self, device=0, stream=None):
        h = c_void_p()
        check(lib.chase_hip_ctx_create(C.byref(h), device, stream), "chase_hip_ctx_create")
        self.h = h
        self._live = []

    def close(self):
        if self.h:
            for a in self._live:
                a.free()
            self._live = []
            lib.chase_hip_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        check(lib.chase_hip_ctx_sync(self.h), "ctx_sync")

    def info(self):
        ncu, clk, mem = c_int(), c_int(), c_size_t()
        name = C.create_string_buffer(128)
        check(lib.chase_hip_device_info(self.h, C.byref(ncu), C.byref(clk), C.byref(mem), name, 128), "device_info")
        return {"num_cu": ncu.value, "clock_khz": clk.value, "hbm_bytes": mem.value, "name": name.value.decode()}

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def array(self, host):
        host = np.asarray(host)
        dt = np.complex128 if np.iscomplexobj(host) else np.float64
        if host.ndim == 1:
            host = host.reshape(-1, 1)
        return DeviceArray(self, host.shape, dt).upload(host)

    def timer_start(self):
        check(lib.chase_hip_timer_start(self.h), "timer_start")

    def timer_stop(self):
        ms = C.c_float()
        check(lib.chase_hip_timer_stop(self.h, C.byref(ms)), "timer_stop")
        return ms.value

    def gemm(self, opA, m, n, k, alpha, A, lda, B, ldb, beta, Cm, ldc, cplx):
        """Raw-pointer GEMM: A, B, Cm are device addresses (ints)."""
        op = opA.encode()[0:1]
        if cplx:
            check(lib.chase_hip_gemm_z(self.h, op, m, n, k, _z2(alpha), A, lda, B, ldb, _z2(beta), Cm, ldc), "gemm_z")
        else:
            check(lib.chase_hip_gemm_d(self.h, op, m, n, k, float(alpha), A, lda, B, ldb, float(beta), Cm, ldc),
                  "gemm_d")

    def mfma_f64_peak(self):
        t = c_double()
        check(lib.chase_hip_mfma_f64_peak(self.h, C.byref(t)), "mfma_f64_peak")
        return t.value

    def hbm_copy_peak(self, nbytes=1 << 30):
        g = c_double()
        check(lib.chase_hip_hbm_copy_peak(self.h, nbytes, C.byref(g)), "hbm_copy_peak")
        return g.value
