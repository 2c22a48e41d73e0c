"""ctypes binding of include/chase_hip.h (the C ABI of libchase_hip.so).

Fails loudly when the HIP extension is missing — there is deliberately no CPU fallback in the product path.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHASE_HIP_LIB: another build of the same library (kernel-development variants, scripts/dev_build_variant.sh)
LIB_PATH = os.environ.get("CHASE_HIP_LIB") or os.path.join(_HERE, "lib", "libchase_hip.so")


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
_sig("chase_hip_ctx_oplog", c_int, c_void_p, c_int)
_sig("chase_hip_ctx_oplog_text", C.c_char_p, c_void_p)
_sig("chase_hip_device_info", c_int, c_void_p, P(c_int), P(c_int), P(c_size_t), C.c_char_p, c_int)
_sig("chase_hip_device_bus_id", c_int, c_void_p, C.c_char_p, c_int)
_sig("chase_hip_device_count", c_int)
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
_sig("chase_hip_gemm3m_enabled", c_int)
_sig("chase_hip_host_lapack_warmup", c_int)
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
        self.device = device
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

    def oplog(self, on):
        """Start (fresh) / stop the operator log of this context (chase_hip_ctx_oplog)."""
        check(lib.chase_hip_ctx_oplog(self.h, int(on)), "ctx_oplog")

    def oplog_lines(self):
        return lib.chase_hip_ctx_oplog_text(self.h).decode().splitlines()

    def info(self):
        ncu, clk, mem = c_int(), c_int(), c_size_t()
        name = C.create_string_buffer(128)
        check(lib.chase_hip_device_info(self.h, C.byref(ncu), C.byref(clk), C.byref(mem), name, 128), "device_info")
        return {"num_cu": ncu.value, "clock_khz": clk.value, "hbm_bytes": mem.value, "name": name.value.decode()}

    def bus_id(self):
        b = C.create_string_buffer(64)
        check(lib.chase_hip_device_bus_id(self.h, b, 64), "device_bus_id")
        return b.value.decode()

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

    def hash64(self, ptr, m, n, ld, cplx):
        """64-bit content hash of a device matrix (chase_hip_hash64)"""
        h = C.c_ulonglong()
        check(lib.chase_hip_hash64(self.h, int(cplx), m, n, ptr, ld, C.byref(h)), "hash64")
        return h.value

    def gen_clement(self, N, cplx, scale=1.0, perturb=0.0, seed=42):
        """Whole N x N Clement-type test matrix generated in HBM (see chase_hip_gen_clement)."""
        dH = self.empty((N, N), np.complex128 if cplx else np.float64)
        check(lib.chase_hip_gen_clement(self.h, int(cplx), dH.ptr, N, N, N, N, N, 1, 0, 0, N, 1, 0, 0, float(scale),
                                        float(perturb), seed), "gen_clement")
        return dH

    def load_matrix(self, path, N, cplx):
        """Whole N x N matrix from a raw column-major binary file (the reference's input format) into HBM."""
        dH = self.empty((N, N), np.complex128 if cplx else np.float64)
        check(lib.chase_hip_load_matrix_shard(self.h, str(path).encode(), int(cplx), N, N, N, N, 1, 0, N, 1, 0, dH.ptr, N),
              "load_matrix_shard")
        return dH

    def save_matrix(self, path, dA):
        cplx = dA.dtype == np.complex128
        check(lib.chase_hip_save_matrix(self.h, str(path).encode(), int(cplx), dA.shape[0],
                                        dA.shape[1] if len(dA.shape) > 1 else 1, dA.ptr, dA.shape[0]), "save_matrix")

    def gen_bse(self, N, cplx=True, dmin=1.0, dmax=11.0, offdiag=1e-3, seed=7):
        """Whole N x N synthetic Bethe-Salpeter matrix generated in HBM (see chase_hip_gen_bse)."""
        dH = self.empty((N, N), np.complex128 if cplx else np.float64)
        check(lib.chase_hip_gen_bse(self.h, int(cplx), dH.ptr, N, N, N, N, N, 1, 0, N, 1, 0, float(dmin), float(dmax),
                                    float(offdiag), seed), "gen_bse")
        return dH

    def mfma_f64_peak(self):
        t = c_double()
        check(lib.chase_hip_mfma_f64_peak(self.h, C.byref(t)), "mfma_f64_peak")
        return t.value

    def hbm_copy_peak(self, nbytes=1 << 30):
        g = c_double()
        check(lib.chase_hip_hbm_copy_peak(self.h, nbytes, C.byref(g)), "hbm_copy_peak")
        return g.value


# ---- non-GEMM kernels -------------------------------------------------------------------------------------------------
_sig("chase_hip_set_lapack_lib", c_int, C.c_char_p)
_sig("chase_hip_lapack_provider", C.c_char_p)
_sig("chase_hip_set_host_threads", c_int, c_int)
_sig("chase_hip_ctx_set_phase", c_int, c_void_p, c_int)
_sig("chase_hip_ctx_set_gemm_min_rounds", c_int, c_void_p, c_int)
_sig("chase_hip_fill_normal", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_long, c_long, c_long,
     C.c_ulonglong)
_sig("chase_hip_gen_clement", c_int, c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_long, c_int, c_int, c_int,
     c_long, c_int, c_int, c_int, c_long, c_double, c_double, C.c_ulonglong)
_sig("chase_hip_gen_bse", c_int, c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_int,
     c_int, c_double, c_double, c_double, C.c_ulonglong)
_sig("chase_hip_load_matrix_shard", c_int, c_void_p, C.c_char_p, c_int, c_long, c_int, c_int, c_int, c_int, c_int, c_int,
     c_int, c_int, c_void_p, c_long)
_sig("chase_hip_save_matrix", c_int, c_void_p, C.c_char_p, c_int, c_int, c_int, c_void_p, c_long)
_sig("chase_hip_shift_diag", c_int, c_void_p, c_int, c_int, c_void_p, c_long, c_double)
_sig("chase_hip_shift_list", c_int, c_void_p, c_int, c_void_p, c_long, c_void_p, c_void_p, c_int, c_double)
_sig("chase_hip_lacpy", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long)
_sig("chase_hip_swap_cols", c_int, c_void_p, c_int, c_int, c_void_p, c_long, c_long, c_long)
_sig("chase_hip_permute_cols", c_int, c_void_p, c_int, c_int, c_void_p, c_long, c_void_p, c_long, P(c_int), P(c_int),
     c_int)
_sig("chase_hip_upload_matrix", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long)
_sig("chase_hip_download_matrix", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long)
_sig("chase_hip_scale_rows", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_int, c_double)
_sig("chase_hip_scale_rows_bc", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_long, c_long, c_int, c_int,
     c_double)
_sig("chase_hip_conj", c_int, c_void_p, c_int, c_int, c_void_p, c_long)
_sig("chase_hip_resid_norms", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p,
     c_void_p, c_int)
_sig("chase_hip_gemm_workspace_bytes", c_size_t, c_int, C.c_char, c_int, c_int, c_int, c_int, c_int)
_sig("chase_hip_herk", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long)
_sig("chase_hip_herkx", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int)
_sig("chase_hip_abs_trace", c_int, c_void_p, c_int, c_int, c_void_p, c_long, P(c_double))
_sig("chase_hip_potrf_upper", c_int, c_void_p, c_int, c_int, c_void_p, c_long)
_sig("chase_hip_trsm_right_upper", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long)
_sig("chase_hip_cholqr", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_int, c_long)
_sig("chase_hip_houseqr", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long)
_sig("chase_hip_heevd", c_int, c_void_p, c_int, c_int, c_void_p, c_long, c_void_p)
_sig("chase_hip_heevd_gpu", c_int, c_void_p, c_int, c_int, c_void_p, c_long, c_void_p)
_sig("chase_hip_heevd_host", c_int, c_int, c_int, c_void_p, c_long, c_void_p)
_sig("chase_hip_stedc", c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_long)
_sig("chase_hip_stemr_host", c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int)
_sig("chase_hip_col_dot", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p)
_sig("chase_hip_col_nrm2", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p)
_sig("chase_hip_col_axpy", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_double, c_void_p, c_long,
     c_void_p, c_long)
_sig("chase_hip_col_scal", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_long)
_sig("chase_hip_pack_upper", c_int, c_void_p, c_int, c_int, c_void_p, c_long, c_void_p)
_sig("chase_hip_unpack_upper", c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_long, c_int)


# ---- host solver (include/chase_hip_solver.h) ------------------------------------------------------------------------
class Stats(C.Structure):
    _fields_ = [("iterations", c_size_t), ("filtered_vecs", c_size_t), ("lanczos_vecs", c_size_t),
                ("locked", c_size_t), ("t_all", c_double), ("t_init", c_double), ("t_lanczos", c_double),
                ("t_filter", c_double), ("t_qr", c_double), ("t_rr", c_double), ("t_resid", c_double),
                ("filter_ms_device", c_double), ("lowerb", c_double), ("upperb", c_double), ("lambda_", c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_sig("chase_hip_solver_create", c_int, P(c_void_p), c_void_p, c_int, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t,
     c_void_p, c_size_t, c_void_p, c_int)
_sig("chase_hip_solver_destroy", c_int, c_void_p)
_sig("chase_hip_solver_set", c_int, c_void_p, C.c_char_p, c_double)
_sig("chase_hip_solver_get", c_int, c_void_p, C.c_char_p, P(c_double))
_sig("chase_hip_solver_solve", c_int, c_void_p, c_int)
ITER_FN = C.CFUNCTYPE(c_int, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t)
_sig("chase_hip_solver_set_iteration_hook", c_int, c_void_p, ITER_FN, c_void_p)
_sig("chase_hip_set_gemm3m", c_int, c_int)
_sig("chase_hip_ctx_gemm_counters", c_int, c_void_p, c_int, P(c_double), P(c_double), P(C.c_ulonglong), c_int)
_sig("chase_hip_solver_stats", c_int, c_void_p, P(Stats))
_sig("chase_hip_solver_resid", P(c_double), c_void_p)
_sig("chase_hip_solver_trace", C.c_char_p, c_void_p)
_sig("chase_hip_solver_recompute_residuals", c_int, c_void_p, c_size_t, c_void_p, c_void_p)
_sig("chase_hip_solver_tape_mode", c_int, c_void_p, c_int)
_sig("chase_hip_solver_tape_data", c_int, c_void_p, P(P(c_double)), P(c_size_t))
_sig("chase_hip_solver_tape_load", c_int, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_resid_norms_dev", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p,
     c_void_p, c_int)
_sig("chase_hip_solver_peek_v", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_solver_hash_v", c_int, c_void_p, c_void_p, c_size_t, P(C.c_ulonglong))
_sig("chase_hip_op_start", c_int, c_void_p)
_sig("chase_hip_op_end", c_int, c_void_p)
_sig("chase_hip_op_initvecs", c_int, c_void_p, c_int)
_sig("chase_hip_op_reinit_columns", c_int, c_void_p, c_size_t, P(c_size_t), c_size_t)
_sig("chase_hip_op_shift", c_int, c_void_p, c_double, c_int)
_sig("chase_hip_op_hemm", c_int, c_void_p, c_size_t, P(c_double), P(c_double), c_size_t, c_size_t)
_sig("chase_hip_op_qr", c_int, c_void_p, c_size_t, c_double)
_sig("chase_hip_op_rr", c_int, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_op_resd", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_op_swap", c_int, c_void_p, c_size_t, c_size_t)
_sig("chase_hip_op_lock", c_int, c_void_p, c_size_t)
_sig("chase_hip_op_lanczos", c_int, c_void_p, c_size_t, c_size_t, P(c_double), c_void_p, c_void_p, c_void_p)
_sig("chase_hip_op_lanczos_dos", c_int, c_void_p, c_size_t, c_size_t, c_void_p)
_sig("chase_hip_op_check_symmetry", c_int, c_void_p, P(c_int))
_sig("chase_hip_op_sym_or_herm", c_int, c_void_p, c_char)
_sig("chase_hip_complete_hermitian", c_int, c_void_p, c_int, c_char, c_int, c_void_p, c_long)
_sig("chase_hip_hash64", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, P(C.c_ulonglong))
_sig("chase_hip_cols_indexed", c_int, c_void_p, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_int)
_sig("chase_hip_tri_mask_bc", c_int, c_void_p, c_int, c_char, c_int, c_int, c_void_p, c_long, c_long, c_int, c_int, c_long, c_int,
     c_int)
_sig("chase_hip_conj_transpose_add", c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long)


def set_iteration_hook(solver, fn):
    """Installs fn(iteration, filtered_vecs, locked, unconverged) as the outer-iteration observer of a solver object."""
    if fn is None:
        solver._iter_cb = ITER_FN()
    else:
        def _cb(user, it, filtered, locked, unconverged):
            try:
                return 1 if fn(int(it), int(filtered), int(locked), int(unconverged)) else 0
            except Exception as e:  # pragma: no cover - never unwind through the C++ driver
                print("iteration hook failed:", repr(e), flush=True)
                return 1
        solver._iter_cb = ITER_FN(_cb)                 # keep the thunk alive
    check(lib.chase_hip_solver_set_iteration_hook(solver.h, solver._iter_cb, None), "set_iteration_hook")


def recompute_residuals(solver, ncols, lam=None):
    """|| H v_j - lambda_j v_j || for the first ncols vectors of a solver object, from a fresh four-product H V."""
    lam = np.ascontiguousarray(solver.ritzv[:ncols] if lam is None else lam, dtype=np.float64)
    out = np.zeros(ncols)
    check(lib.chase_hip_solver_recompute_residuals(solver.h, ncols, lam.ctypes.data, out.ctypes.data), "recompute_residuals")
    return out


TAPE_OFF, TAPE_RECORD, TAPE_REPLAY = 0, 1, 2


def tape_mode(solver, mode):
    """0 off, 1: the next solves record everything the Impl tells the driver, 2: they replay the loaded tape (tape.hpp)."""
    check(lib.chase_hip_solver_tape_mode(solver.h, int(mode)), "tape_mode")


def tape_get(solver):
    """The scalar tape of the last recorded solve as a float64 array (frames: tag, count, values...)."""
    p, n = P(c_double)(), c_size_t()
    check(lib.chase_hip_solver_tape_data(solver.h, C.byref(p), C.byref(n)), "tape_data")
    return np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0)


def tape_load(solver, tape):
    t = np.ascontiguousarray(tape, dtype=np.float64)
    check(lib.chase_hip_solver_tape_load(solver.h, t.ctypes.data, t.size), "tape_load")


def gemm_counters(ctx, phase, reset=False):
    """(model flops, executed flops, products) of the GEMMs a context issued in `phase` (1 = Chebyshev filter)."""
    a, b, n = c_double(), c_double(), C.c_ulonglong()
    check(lib.chase_hip_ctx_gemm_counters(ctx.h, phase, C.byref(a), C.byref(b), C.byref(n), int(reset)), "gemm_counters")
    return a.value, b.value, n.value


class Solver:
    """ChaseHip<T> behind the C ABI: the reference's ChASEGPU constructor contract (host H, V, ritzv owned by the
    caller; chase_gpu.hpp:107) plus chase::Solve."""

    def __init__(self, ctx, H, nev, nex, V=None, h_on_device_ptr=None, N=None, cplx=None):
        self.ctx = ctx
        if h_on_device_ptr is None:
            assert H.flags.f_contiguous and H.ndim == 2 and H.shape[0] == H.shape[1]
            self.cplx = bool(np.iscomplexobj(H))
            self.N = H.shape[0]
            self.H = H
            hptr, on_dev = H.ctypes.data, 0
        else:
            self.cplx, self.N, self.H = bool(cplx), int(N), None
            hptr, on_dev = h_on_device_ptr, 1
        self.nev, self.nex = nev, nex
        dt = np.complex128 if self.cplx else np.float64
        self.V = np.zeros((self.N, nev + nex), dtype=dt, order="F") if V is None else V
        assert self.V.flags.f_contiguous and self.V.dtype == dt
        self.ritzv = np.zeros(nev + nex, dtype=np.float64)
        h = c_void_p()
        check(lib.chase_hip_solver_create(C.byref(h), ctx.h, int(self.cplx), self.N, nev, nex, hptr, self.N,
                                          self.V.ctypes.data, self.N, self.ritzv.ctypes.data, on_dev),
              "solver_create")
        self.h = h

    def close(self):
        if self.h:
            lib.chase_hip_solver_destroy(self.h)
            self.h = None

    def set(self, **kw):
        for k, v in kw.items():
            check(lib.chase_hip_solver_set(self.h, k.encode(), float(v)), f"solver_set({k})")

    def get(self, key):
        v = c_double()
        check(lib.chase_hip_solver_get(self.h, key.encode(), C.byref(v)), f"solver_get({key})")
        return v.value

    def solve(self, trace=False):
        check(lib.chase_hip_solver_solve(self.h, int(trace)), "solver_solve")
        return self.stats()

    def set_iteration_hook(self, fn):
        """fn(iteration, filtered_vecs, locked, unconverged) -> truthy to leave the iteration loop; None removes it."""
        set_iteration_hook(self, fn)

    def stats(self):
        s = Stats()
        check(lib.chase_hip_solver_stats(self.h, C.byref(s)), "solver_stats")
        return s.as_dict()

    def resid(self):
        p = lib.chase_hip_solver_resid(self.h)
        return np.ctypeslib.as_array(p, shape=(self.nev + self.nex,)).copy()

    def trace(self):
        return lib.chase_hip_solver_trace(self.h).decode().splitlines()

    def recompute_residuals(self, ncols, lam=None):
        return recompute_residuals(self, ncols, lam)

    def peek_v(self):
        out = np.empty_like(self.V, order="F")
        check(lib.chase_hip_solver_peek_v(self.h, self.ctx.h, out.ctypes.data, self.N), "peek_v")
        return out

    # ChaseBase virtuals
    def Start(self): check(lib.chase_hip_op_start(self.h), "Start")
    def End(self): check(lib.chase_hip_op_end(self.h), "End")
    def initVecs(self, random): check(lib.chase_hip_op_initvecs(self.h, int(random)), "initVecs")

    def ReinitColumns(self, fixednev, cols):
        a = (c_size_t * len(cols))(*cols)
        check(lib.chase_hip_op_reinit_columns(self.h, fixednev, a, len(cols)), "ReinitColumns")
    def Shift(self, c, isunshift=False): check(lib.chase_hip_op_shift(self.h, float(c), int(isunshift)), "Shift")

    def HEMM(self, block, alpha, beta, offset_left, offset_right=0):
        check(lib.chase_hip_op_hemm(self.h, block, _z2(alpha), _z2(beta), offset_left, offset_right), "HEMM")

    def QR(self, fixednev, cond): check(lib.chase_hip_op_qr(self.h, fixednev, float(cond)), "QR")

    def RR(self, block, offset):
        """ritz values are written to self.ritzv[offset:offset+block] (the driver passes GetRitzv()+locked)."""
        check(lib.chase_hip_op_rr(self.h, self.ritzv.ctypes.data + 8 * offset, block), "RR")

    def Resd(self, offset):
        n = self.nev + self.nex - offset
        out = np.zeros(n)
        check(lib.chase_hip_op_resd(self.h, self.ritzv.ctypes.data + 8 * offset, out.ctypes.data, offset), "Resd")
        return out

    def Swap(self, i, j): check(lib.chase_hip_op_swap(self.h, i, j), "Swap")
    def Lock(self, n): check(lib.chase_hip_op_lock(self.h, n), "Lock")

    def Lanczos(self, M, numvec):
        ub = c_double()
        if numvec == 0:
            check(lib.chase_hip_op_lanczos(self.h, M, 0, C.byref(ub), None, None, None), "Lanczos")
            return ub.value
        theta, tau, ritzV = np.zeros(M * numvec), np.zeros(M * numvec), np.zeros(M * M)
        check(lib.chase_hip_op_lanczos(self.h, M, numvec, C.byref(ub), theta.ctypes.data, tau.ctypes.data,
                                       ritzV.ctypes.data), "Lanczos")
        return ub.value, theta, tau, ritzV.reshape(M, M, order="F")

    def checkSymmetryEasy(self):
        f = c_int()
        check(lib.chase_hip_op_check_symmetry(self.h, C.byref(f)), "checkSymmetryEasy")
        return bool(f.value)

    def symOrHermMatrix(self, uplo):
        check(lib.chase_hip_op_sym_or_herm(self.h, uplo.encode()[0:1]), "symOrHermMatrix")


_sig("chase_hip_solver_create_pseudo", c_int, P(c_void_p), c_void_p, c_int, c_size_t, c_size_t, c_size_t, c_void_p,
     c_size_t, c_void_p, c_size_t, c_void_p, c_int)
_sig("chase_hip_op_hemm_h2", c_int, c_void_p, c_size_t, P(c_double), P(c_double), P(c_double), c_size_t, c_size_t)
_sig("chase_hip_op_kconj", c_int, c_void_p, c_size_t)
_sig("chase_hip_pseudo_rr_small", c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p)
_sig("chase_hip_set_identity", c_int, c_void_p, c_int, c_int, c_void_p, c_long)


_sig("chase_hip_solver_lanczos_for_h2", c_int, c_void_p, c_int, c_int, P(c_double), P(c_size_t))


class PseudoSolver(Solver):
    """ChaseHipPseudo<T>: pseudo-Hermitian (BSE) Impl — subspace of 2*(nev+nex) columns, chase::Solve_pseudo."""

    def __init__(self, ctx, H, nev, nex):
        assert H.flags.f_contiguous and H.shape[0] == H.shape[1]
        self.ctx, self.H = ctx, H
        self.cplx = bool(np.iscomplexobj(H))
        self.N, self.nev, self.nex = H.shape[0], nev, nex
        dt = np.complex128 if self.cplx else np.float64
        self.ncol = 2 * (nev + nex)
        self.V = np.zeros((self.N, self.ncol), dtype=dt, order="F")
        self.ritzv = np.zeros(self.ncol)
        h = c_void_p()
        check(lib.chase_hip_solver_create_pseudo(C.byref(h), ctx.h, int(self.cplx), self.N, nev, nex, H.ctypes.data,
                                                 self.N, self.V.ctypes.data, self.N, self.ritzv.ctypes.data, 0),
              "solver_create_pseudo")
        self.h = h

    def resid(self):
        p = lib.chase_hip_solver_resid(self.h)
        return np.ctypeslib.as_array(p, shape=(self.ncol,)).copy()

    def HEMM_H2(self, block, alpha, beta, gamma, offset_left, offset_right=0):
        check(lib.chase_hip_op_hemm_h2(self.h, block, _z2(alpha), _z2(beta), _z2(gamma), offset_left, offset_right),
              "HEMM_H2")

    def ApplyKconjugate(self, block):
        check(lib.chase_hip_op_kconj(self.h, block), "ApplyKconjugate")

    def lanczos_for_H2(self, numvec, m):
        ub, idx = c_double(), c_size_t()
        check(lib.chase_hip_solver_lanczos_for_h2(self.h, numvec, m, C.byref(ub), C.byref(idx)), "lanczos_for_H2")
        return ub.value, idx.value

    def RR(self, block, offset):
        check(lib.chase_hip_op_rr(self.h, self.ritzv.ctypes.data + 8 * offset, block), "RR")

    def Resd(self, offset):
        n = self.nev + self.nex - offset
        out = np.zeros(n)
        check(lib.chase_hip_op_resd(self.h, self.ritzv.ctypes.data + 8 * offset, out.ctypes.data, offset), "Resd")
        return out
