"""Multi-process plumbing for the distributed Impl (one process per GPU): torch.distributed is used ONLY to bootstrap
(exchange RCCL unique ids, barriers, max-over-ranks timing) and, in tests, as the gloo back end of the host-callback
transport that lets several ranks share one GPU.  The data path itself is C++/HIP/RCCL (chase_amd/csrc/grid.hip)."""
import ctypes as C
import os

import numpy as np

from .capi import (lib, check, Context, DeviceArray, Stats, _sig, _z2, c_int, c_long, c_size_t, c_double, c_void_p, P)

_sig("chase_hip_rccl_unique_id", c_int, C.c_char_p)
_sig("chase_hip_grid_create_rccl", c_int, P(c_void_p), c_void_p, c_int, c_int, c_int, C.c_char_p, C.c_char_p)
AR_FN = C.CFUNCTYPE(c_int, c_void_p, c_int, P(c_double), c_size_t)
BC_FN = C.CFUNCTYPE(c_int, c_void_p, c_int, P(c_double), c_size_t, c_int)
_sig("chase_hip_grid_create_host", c_int, P(c_void_p), c_void_p, c_int, c_int, c_int, AR_FN, BC_FN, c_void_p)
_sig("chase_hip_grid_create_loopback", c_int, P(c_void_p), c_void_p, c_int, c_int, c_int)
_sig("chase_hip_fabric_create", c_int, P(c_void_p), c_int, c_int)
_sig("chase_hip_fabric_abort", c_int, c_void_p)
_sig("chase_hip_fabric_destroy", c_int, c_void_p)
_sig("chase_hip_grid_create_shared", c_int, P(c_void_p), c_void_p, c_int, c_int, c_int, c_void_p)
_sig("chase_hip_grid_set_loopback_model", c_int, c_void_p, c_double, c_double, c_int, c_int)
_sig("chase_hip_grid_set_comm_streams", c_int, c_void_p, c_int)
_sig("chase_hip_grid_comm_streams", c_int, c_void_p)
_sig("chase_hip_grid_event_record_on", c_int, c_void_p, c_int, c_int)
_sig("chase_hip_grid_event_record", c_int, c_void_p, c_int)
_sig("chase_hip_grid_event_wait", c_int, c_void_p, c_int)
_sig("chase_hip_grid_destroy", c_int, c_void_p)
_sig("chase_hip_grid_group_active", c_int, c_void_p, c_int)
_sig("chase_hip_grid_info", c_int, c_void_p, P(c_int), P(c_int), P(c_int), P(c_int))
_sig("chase_hip_grid_allreduce", c_int, c_void_p, c_int, c_void_p, c_size_t, c_int)
_sig("chase_hip_grid_bcast", c_int, c_void_p, c_int, c_void_p, c_size_t, c_int, c_int)
_sig("chase_hip_grid_wait", c_int, c_void_p)
_sig("chase_hip_grid_agree_max", c_int, c_void_p, P(c_int))
SR_FN = C.CFUNCTYPE(c_int, c_void_p, c_int, P(c_double), c_size_t, c_int, P(c_double), c_size_t, c_int)
_sig("chase_hip_grid_sendrecv", c_int, c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_int)
_sig("chase_hip_grid_set_host_sendrecv", c_int, c_void_p, SR_FN)
_sig("chase_hip_grid_set_profiling", c_int, c_void_p, c_int)
_sig("chase_hip_grid_comm_exposed_ms", c_int, c_void_p, P(c_double), P(C.c_ulonglong), c_int)
_sig("chase_hip_grid_transport", c_int, c_void_p, P(c_int), P(c_int), P(c_int))
for _n in ("chase_hip_block_len",):
    _sig(_n, c_long, c_long, c_int)
_sig("chase_hip_numroc", c_long, c_long, c_long, c_int, c_int)
_sig("chase_hip_owner", c_int, c_long, c_long, c_int)
_sig("chase_hip_local_index", c_long, c_long, c_long, c_int)
_sig("chase_hip_global_index", c_long, c_long, c_long, c_int, c_int)
_sig("chase_hip_psolver_create", c_int, P(c_void_p), c_void_p, c_void_p, c_int, c_size_t, c_size_t, c_size_t, c_size_t,
     c_size_t, c_void_p, c_size_t, c_void_p)
_sig("chase_hip_psolver_create_pseudo", c_int, P(c_void_p), c_void_p, c_void_p, c_int, c_size_t, c_size_t, c_size_t,
     c_size_t, c_size_t, c_void_p, c_size_t, c_void_p)
_sig("chase_hip_psolver_local_shape", c_int, c_void_p, P(c_size_t), P(c_size_t))
_sig("chase_hip_psolver_upload_v", c_int, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_psolver_download_v", c_int, c_void_p, c_void_p, c_size_t)
_sig("chase_hip_psolver_set_pipeline", c_int, c_void_p, c_int)

ROW, COL = 0, 1


def grid_shape(nranks):
    """The reference requires row_dim >= col_dim (grid/mpiGrid2D.hpp:209-211): 1 -> 1x1, 2 -> 2x1, 4 -> 2x2, 8 -> 4x2."""
    c = int(np.floor(np.sqrt(nranks)))
    while nranks % c:
        c -= 1
    return nranks // c, c


def coords_of(rank, nprow):
    return rank % nprow, rank // nprow          # column-major grid ordering


class Layout:
    """1D block-cyclic index map (block layout = block size equal to the block length)."""

    def __init__(self, N, nb, p):
        self.N, self.p = N, p
        self.nb = nb if nb else lib.chase_hip_block_len(N, p)

    def count(self, q): return lib.chase_hip_numroc(self.N, self.nb, q, self.p)
    def owner(self, g): return lib.chase_hip_owner(g, self.nb, self.p)
    def local(self, g): return lib.chase_hip_local_index(g, self.nb, self.p)

    def globals_of(self, q):
        n = self.count(q)
        l = np.arange(n)
        return ((l // self.nb) * self.p + q) * self.nb + l % self.nb


class GlooFabric:
    """Host-transport back end for ranks that are PROCESSES: torch.distributed (gloo) groups from make_process_groups."""

    def __init__(self, pg, nprow, npcol, rank):
        myrow, mycol = coords_of(rank, nprow)
        self.groups = {ROW: pg["row"], COL: pg["col"]}
        self.ranks = {ROW: [myrow + j * nprow for j in range(npcol)], COL: [i + mycol * nprow for i in range(nprow)]}

    def allreduce(self, group, a):
        import torch
        import torch.distributed as dist
        dist.all_reduce(torch.from_numpy(a), group=self.groups[group])

    def bcast(self, group, a, root):
        import torch
        import torch.distributed as dist
        dist.broadcast(torch.from_numpy(a), src=self.ranks[group][root], group=self.groups[group])

    def sendrecv(self, group, send, peer_send, recv, peer_recv):
        import torch
        import torch.distributed as dist
        ops = []
        if send is not None:
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(send), self.ranks[group][peer_send], group=self.groups[group]))
        if recv is not None:
            ops.append(dist.P2POp(dist.irecv, torch.from_numpy(recv), self.ranks[group][peer_recv], group=self.groups[group]))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()


class Grid:
    def __init__(self, ctx, nprow, npcol, rank, transport="rccl", pg=None):
        """pg: dict {"world": group, "row": group, "col": group} of torch.distributed groups (host transport /
        bootstrap)."""
        self.ctx, self.nprow, self.npcol, self.rank = ctx, nprow, npcol, rank
        self.myrow, self.mycol = coords_of(rank, nprow)
        self.transport = transport
        h = c_void_p()
        if transport == "rccl":
            # bootstrap (the reference broadcasts its ids over MPI, grid/mpiGrid2D.hpp:448-484): through the communicator
            # the caller hands over - rank threads of one process (chase_amd.rank_threads.RankComm) or torch.distributed
            if hasattr(pg, "all_gather_object"):
                gather = pg.all_gather_object
            else:
                import torch.distributed as dist

                def gather(obj):
                    out = [None] * (nprow * npcol)
                    dist.all_gather_object(out, obj)
                    return out
            my_id = C.create_string_buffer(128)
            check(lib.chase_hip_rccl_unique_id(my_id), "rccl_unique_id")
            if nprow * npcol > 1:
                ids = gather(bytes(my_id.raw))
            else:
                ids = [bytes(my_id.raw)]
            row_leader = self.myrow                         # (myrow, 0)
            col_leader = self.mycol * nprow                 # (0, mycol): a different rank than the row leader's id use
            # a rank may lead both its row and its column group (rank 0): use distinct ids -> generate a second one
            my_id2 = C.create_string_buffer(128)
            check(lib.chase_hip_rccl_unique_id(my_id2), "rccl_unique_id")
            if nprow * npcol > 1:
                ids2 = gather(bytes(my_id2.raw))
            else:
                ids2 = [bytes(my_id2.raw)]
            check(lib.chase_hip_grid_create_rccl(C.byref(h), ctx.h, nprow, npcol, rank, ids[row_leader],
                                                 ids2[col_leader]), "grid_create_rccl")
        elif transport == "shared":
            # ranks = threads of this process sharing ONE device: device-side collectives through the fabric object the
            # communicator carries (chase_amd.rank_threads.run_ranks(transport="shared") creates it before the threads start)
            fabric = getattr(getattr(pg, "w", None), "fabric", None)
            if not fabric:
                raise ValueError("transport 'shared' needs the rank threads' fabric (rank_threads.run_ranks)")
            check(lib.chase_hip_grid_create_shared(C.byref(h), ctx.h, nprow, npcol, rank, fabric), "grid_create_shared")
        elif transport == "loopback":
            # ONE rank of the grid with nobody on the other side (single-rank replay, chase_hip_grid_create_loopback)
            check(lib.chase_hip_grid_create_loopback(C.byref(h), ctx.h, nprow, npcol, rank), "grid_create_loopback")
        else:
            # host-callback transport (test plumbing): `pg` is either the dict of torch.distributed groups made by
            # make_process_groups (ranks = processes, payloads through gloo) or any object with
            # allreduce(group, array) / bcast(group, array, root) / sendrecv(group, send, peer_send, recv, peer_recv)
            # working in place on float64 numpy views of the pinned staging buffer (ranks = threads of one process,
            # chase_amd/rank_threads.py)
            fabric = pg if hasattr(pg, "allreduce") else (GlooFabric(pg, nprow, npcol, rank) if pg is not None else None)
            self.fabric = fabric

            def _view(buf, count):
                return np.ctypeslib.as_array(buf, shape=(count,))

            def _ar(user, group, buf, count):
                try:
                    fabric.allreduce(group, _view(buf, count))
                    return 0
                except Exception as e:  # pragma: no cover
                    print("allreduce callback failed:", repr(e), flush=True)
                    return 1

            def _bc(user, group, buf, count, root):
                try:
                    fabric.bcast(group, _view(buf, count), root)
                    return 0
                except Exception as e:  # pragma: no cover
                    print("bcast callback failed:", repr(e), flush=True)
                    return 1

            def _sr(user, group, sbuf, scount, peer_send, rbuf, rcount, peer_recv):
                try:
                    fabric.sendrecv(group, _view(sbuf, scount) if peer_send >= 0 and scount else None, peer_send,
                                    _view(rbuf, rcount) if peer_recv >= 0 and rcount else None, peer_recv)
                    return 0
                except Exception as e:  # pragma: no cover
                    print("sendrecv callback failed:", repr(e), flush=True)
                    return 1

            self._cb = (AR_FN(_ar), BC_FN(_bc), SR_FN(_sr))          # keep the thunks alive
            check(lib.chase_hip_grid_create_host(C.byref(h), ctx.h, nprow, npcol, rank, self._cb[0], self._cb[1], None),
                  "grid_create_host")
            check(lib.chase_hip_grid_set_host_sendrecv(h, self._cb[2]), "set_host_sendrecv")
        self.h = h

    def set_profiling(self, on):
        check(lib.chase_hip_grid_set_profiling(self.h, int(on)), "grid_set_profiling")

    def comm_exposed_ms(self, reset=False):
        """(milliseconds the compute stream spent waiting for collectives with nothing else to run, number of waits)"""
        ms, n = c_double(), C.c_ulonglong()
        check(lib.chase_hip_grid_comm_exposed_ms(self.h, C.byref(ms), C.byref(n), int(reset)), "comm_exposed_ms")
        return ms.value, n.value

    def transport_info(self):
        """(is_rccl, ranks RCCL reports for the row communicator, for the column communicator)"""
        a, r, c = c_int(), c_int(), c_int()
        check(lib.chase_hip_grid_transport(self.h, C.byref(a), C.byref(r), C.byref(c)), "grid_transport")
        return a.value == 1, r.value, c.value

    def set_loopback_model(self, busbw_GBps, latency_us=0.0, touch=False, workgroups=0):
        """loopback grids: collectives hold their stream and `workgroups` RCCL-sized workgroups for latency + wire bytes / bus
        bandwidth (a stated model)"""
        check(lib.chase_hip_grid_set_loopback_model(self.h, float(busbw_GBps), float(latency_us), int(touch), int(workgroups)),
              "set_loopback_model")

    def set_comm_streams(self, n):
        """1 (default): both groups' collectives on one communication stream, 2: one stream per group."""
        check(lib.chase_hip_grid_set_comm_streams(self.h, int(n)), "set_comm_streams")

    def comm_streams(self):
        return lib.chase_hip_grid_comm_streams(self.h)

    def sendrecv(self, group, send, peer_send, recv, peer_recv):
        """send / recv: DeviceArray (or None); counts in doubles are taken from the arrays"""
        def cnt(a):
            return 0 if a is None else int(np.prod(a.shape)) * (2 if a.dtype == np.complex128 else 1)
        check(lib.chase_hip_grid_sendrecv(self.h, group, send.ptr if send is not None else None, cnt(send), peer_send,
                                          recv.ptr if recv is not None else None, cnt(recv), peer_recv), "grid_sendrecv")

    def close(self):
        if self.h:
            lib.chase_hip_grid_destroy(self.h)
            self.h = None


def make_process_groups(nprow, npcol):
    """Row / column torch.distributed groups; every rank must create all of them in the same order."""
    import torch.distributed as dist
    rank = dist.get_rank()
    myrow, mycol = coords_of(rank, nprow)
    out = {"world": dist.group.WORLD}
    for i in range(nprow):
        g = dist.new_group([i + j * nprow for j in range(npcol)])
        if i == myrow:
            out["row"] = g
    for j in range(npcol):
        g = dist.new_group([i + j * nprow for i in range(nprow)])
        if j == mycol:
            out["col"] = g
    return out


class DistSolver:
    """pChaseHip<T> behind the C ABI (collective: every rank of the grid constructs it and calls the same methods)."""
    _create = "chase_hip_psolver_create"
    _colmul = 1                                   # vector columns = _colmul * (nev + nex)

    def __init__(self, ctx, grid, dH_loc, N, nev, nex, cplx, mb=0, nb=0, ldh=None):
        self.ctx, self.grid, self.N, self.nev, self.nex, self.cplx = ctx, grid, N, nev, nex, bool(cplx)
        self.ncol = self._colmul * (nev + nex)
        self.ritzv = np.zeros(self.ncol)
        h = c_void_p()
        ptr = dH_loc.ptr if isinstance(dH_loc, DeviceArray) else dH_loc
        ldh = ldh or (dH_loc.ld if isinstance(dH_loc, DeviceArray) else None)
        check(getattr(lib, self._create)(C.byref(h), ctx.h, grid.h, int(cplx), N, nev, nex, mb, nb, ptr, ldh,
                                         self.ritzv.ctypes.data), self._create)
        self.h = h
        m, n = c_size_t(), c_size_t()
        check(lib.chase_hip_psolver_local_shape(h, C.byref(m), C.byref(n)), "local_shape")
        self.m_loc, self.n_loc = m.value, n.value
        self.dt = np.complex128 if cplx else np.float64

    def close(self):
        if self.h:
            lib.chase_hip_solver_destroy(self.h)
            self.h = None

    def set(self, **kw):
        for k, v in kw.items():
            if k == "pipeline":
                check(lib.chase_hip_psolver_set_pipeline(self.h, int(v)), "set_pipeline")
            else:
                check(lib.chase_hip_solver_set(self.h, k.encode(), float(v)), f"solver_set({k})")

    def get(self, key):
        v = c_double()
        check(lib.chase_hip_solver_get(self.h, key.encode(), C.byref(v)), f"solver_get({key})")
        return v.value

    def solve(self, trace=False):
        check(lib.chase_hip_solver_solve(self.h, int(trace)), "solver_solve")
        s = Stats()
        check(lib.chase_hip_solver_stats(self.h, C.byref(s)), "solver_stats")
        return s.as_dict()

    def resid(self):
        p = lib.chase_hip_solver_resid(self.h)
        return np.ctypeslib.as_array(p, shape=(self.ncol,)).copy()

    def trace(self):
        return lib.chase_hip_solver_trace(self.h).decode().splitlines()

    def set_iteration_hook(self, fn):
        from .capi import set_iteration_hook
        set_iteration_hook(self, fn)

    def recompute_residuals(self, ncols, lam=None):
        from .capi import recompute_residuals
        return recompute_residuals(self, ncols, lam)

    def local_V(self):
        out = np.empty((self.m_loc, self.ncol), dtype=self.dt, order="F")
        check(lib.chase_hip_psolver_download_v(self.h, out.ctypes.data, self.m_loc), "download_v")
        return out

    def hash_V(self, ncols):
        """64-bit content hash of the first ncols columns of this rank's vector block, computed on the device"""
        h = C.c_ulonglong()
        check(lib.chase_hip_solver_hash_v(self.h, self.ctx.h, ncols, C.byref(h)), "hash_v")
        return h.value

    def upload_local_V(self, V):
        V = np.asfortranarray(V, dtype=self.dt)
        check(lib.chase_hip_psolver_upload_v(self.h, V.ctypes.data, V.shape[0]), "upload_v")

    # ChaseBase virtuals (collective)
    def Start(self): check(lib.chase_hip_op_start(self.h), "Start")
    def End(self): check(lib.chase_hip_op_end(self.h), "End")
    def initVecs(self, random): check(lib.chase_hip_op_initvecs(self.h, int(random)), "initVecs")
    def Shift(self, c, isunshift=False): check(lib.chase_hip_op_shift(self.h, float(c), int(isunshift)), "Shift")

    def HEMM(self, block, alpha, beta, offset_left, offset_right=0):
        check(lib.chase_hip_op_hemm(self.h, block, _z2(alpha), _z2(beta), offset_left, offset_right), "HEMM")

    def QR(self, fixednev, cond): check(lib.chase_hip_op_qr(self.h, fixednev, float(cond)), "QR")
    def RR(self, block, offset): check(lib.chase_hip_op_rr(self.h, self.ritzv.ctypes.data + 8 * offset, block), "RR")

    def Resd(self, offset):
        out = np.zeros(self.nev + self.nex - offset)
        check(lib.chase_hip_op_resd(self.h, self.ritzv.ctypes.data + 8 * offset, out.ctypes.data, offset), "Resd")
        return out

    def Swap(self, i, j): check(lib.chase_hip_op_swap(self.h, i, j), "Swap")
    def Lock(self, n): check(lib.chase_hip_op_lock(self.h, n), "Lock")

    def checkSymmetryEasy(self):
        f = c_int()
        check(lib.chase_hip_op_check_symmetry(self.h, C.byref(f)), "checkSymmetryEasy")
        return bool(f.value)

    def symOrHermMatrix(self, uplo):
        """collective: complete the Hermitian matrix on the device shards from its stored triangle 'U' / 'L'"""
        check(lib.chase_hip_op_sym_or_herm(self.h, uplo.encode()[0:1]), "symOrHermMatrix")

    def Lanczos(self, M, numvec):
        ub = c_double()
        if numvec == 0:                     # single-vector form: upper bound only (interface.hpp Lanczos(m, upperb))
            check(lib.chase_hip_op_lanczos(self.h, M, 0, C.byref(ub), None, None, None), "Lanczos")
            return ub.value
        theta, tau, ritzV = np.zeros(M * numvec), np.zeros(M * numvec), np.zeros(M * M)
        check(lib.chase_hip_op_lanczos(self.h, M, numvec, C.byref(ub), theta.ctypes.data, tau.ctypes.data,
                                       ritzV.ctypes.data), "Lanczos")
        return ub.value, theta, tau, ritzV.reshape(M, M, order="F")


class DistPseudoSolver(DistSolver):
    """pChaseHipPseudo<T>: distributed pseudo-Hermitian (BSE) Impl, 2*(nev+nex) vector columns, chase::Solve_pseudo."""
    _create = "chase_hip_psolver_create_pseudo"
    _colmul = 2

    def HEMM_H2(self, block, alpha, beta, gamma, offset_left, offset_right=0):
        check(lib.chase_hip_op_hemm_h2(self.h, block, _z2(alpha), _z2(beta), _z2(gamma), offset_left, offset_right),
              "HEMM_H2")

    def ApplyKconjugate(self, block):
        check(lib.chase_hip_op_kconj(self.h, block), "ApplyKconjugate")


def local_block_of(H, rl, cl, myrow, mycol):
    """Extract this rank's block of a full host matrix (tests)."""
    return np.asfortranarray(H[np.ix_(rl.globals_of(myrow), cl.globals_of(mycol))])


def gen_clement_local(ctx, N, cplx, rl, cl, myrow, mycol, scale=1.0, perturb=0.0, seed=42):
    """This rank's shard of the Clement-type test matrix, generated in HBM."""
    m, n = rl.count(myrow), cl.count(mycol)
    dH = ctx.empty((m, n), np.complex128 if cplx else np.float64)
    check(lib.chase_hip_gen_clement(ctx.h, int(cplx), dH.ptr, m, m, n, N, rl.nb, rl.p, myrow, 0, cl.nb, cl.p, mycol, 0,
                                    float(scale), float(perturb), seed), "gen_clement")
    return dH


def gen_bse_local(ctx, N, cplx, rl, cl, myrow, mycol, dmin=1.0, dmax=11.0, offdiag=1e-3, seed=7):
    """This rank's shard of the synthetic Bethe-Salpeter matrix (BASELINE config 5), generated in HBM."""
    m, n = rl.count(myrow), cl.count(mycol)
    dH = ctx.empty((m, n), np.complex128 if cplx else np.float64)
    check(lib.chase_hip_gen_bse(ctx.h, int(cplx), dH.ptr, m, m, n, N, rl.nb, rl.p, myrow, cl.nb, cl.p, mycol,
                                float(dmin), float(dmax), float(offdiag), seed), "gen_bse")
    return dH


def load_matrix_local(ctx, path, N, cplx, rl, cl, myrow, mycol):
    """This rank's shard of a raw column-major binary matrix file, read straight into HBM (chase_hip_load_matrix_shard)."""
    m, n = rl.count(myrow), cl.count(mycol)
    dH = ctx.empty((m, n), np.complex128 if cplx else np.float64)
    check(lib.chase_hip_load_matrix_shard(ctx.h, str(path).encode(), int(cplx), N, m, n, rl.nb, rl.p, myrow, cl.nb, cl.p,
                                          mycol, dH.ptr, m), "load_matrix_shard")
    return dH
