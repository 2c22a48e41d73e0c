"""dist_oracle.py — CPU restatement of the reference's DISTRIBUTED hot-path kernels on torch.distributed (gloo).
*** TEST INFRASTRUCTURE, NOT PRODUCT ***  (only tests/ may import it).

Restates, on numpy blocks + gloo collectives (file:line relative to /root/reference):
  MatrixMultiplyMultiVectors (column-type <-> row-type, beta on grid row/col 0)   linalg/internal/mpi/hemm.hpp:114-229
  cholQR1 with the Gram all-reduce over the column communicator                    linalg/internal/mpi/cholqr.hpp:51-110
  rayleighRitz                                                                      linalg/internal/mpi/rayleighRitz.hpp:103-186
  residuals                                                                         linalg/internal/mpi/residuals.hpp:61-107
  column-type -> row-type redistribution                                            linalg/distMatrix/distMultiVector.hpp:2585-2720
with its OWN index arithmetic (round 5; rounds 2-4 borrowed the product's chase_amd.dist.Layout, so a layout bug would have
been shared): OLayout below restates the block rule and numroc (linalg/distMatrix/distMatrix.hpp:44-67,1992-2052), the grid
coordinates and communicators are the reference's column-major MpiGrid2D (grid/mpiGrid2D.hpp:402-446: row_comm = ranks of
one grid row, col_comm = ranks of one grid column).  tests/test_dist_cpu.py then compares these index maps with the
product's layout helpers instead of sharing them.
Pinned by the reference's distributed HEMM known answer (tests/linalg/internal/mpi/hemm.cpp:36-119) and by agreement with
the serial oracle, which is itself pinned in tests/test_oracle_pins.py."""
import numpy as np
import scipy.linalg as sla
import torch
import torch.distributed as dist



class OLayout:
    """1D block-cyclic map of N indices over p ranks, block size nb (0: the reference's BLOCK layout, whose block length is
    N / p when that divides and N / p + 1 otherwise, the last rank taking the remainder - distMatrix.hpp:2000-2039)."""

    def __init__(self, N, nb, p):
        self.N, self.p = int(N), int(p)
        if nb:
            self.nb = int(nb)
        else:
            self.nb = self.N // self.p if self.N % self.p == 0 else min(self.N // self.p + 1, self.N)

    def count(self, q):
        """ScaLAPACK NUMROC with source process 0 (distMatrix.hpp:44-67)"""
        nblocks = self.N // self.nb
        n = (nblocks // self.p) * self.nb
        extra = nblocks % self.p
        if q < extra:
            n += self.nb
        elif q == extra:
            n += self.N % self.nb
        return n

    def owner(self, g):
        return (int(g) // self.nb) % self.p

    def local(self, g):
        g = int(g)
        return (g // (self.nb * self.p)) * self.nb + g % self.nb

    def globals_of(self, q):
        return np.array([g for g in range(self.N) if self.owner(g) == q], dtype=np.int64)


def grid_coords(rank, nprow):
    """column-major grid ordering (grid/mpiGrid2D.hpp:402-432): rank = row + col * nprow"""
    return rank % nprow, rank // nprow


def make_groups(nprow, npcol):
    """row_comm / col_comm of every rank (grid/mpiGrid2D.hpp:433-446); all ranks create all groups in the same order"""
    rank = dist.get_rank()
    myrow, mycol = grid_coords(rank, nprow)
    out = {}
    for i in range(nprow):
        g = dist.new_group([i + j * nprow for j in range(npcol)])
        if i == myrow:
            out["row"] = g
    for j in range(npcol):
        g = dist.new_group([i + j * nprow for i in range(nprow)])
        if j == mycol:
            out["col"] = g
    return out


class DistBlocks:
    def __init__(self, N, nprow, npcol, mb=0, nb=0):
        self.rank = dist.get_rank()
        self.nprow, self.npcol = nprow, npcol
        self.myrow, self.mycol = grid_coords(self.rank, nprow)
        self.rl, self.cl = OLayout(N, mb, nprow), OLayout(N, nb, npcol)
        self.rows = self.rl.globals_of(self.myrow)            # global rows of column-type blocks / of H_loc
        self.cols = self.cl.globals_of(self.mycol)            # global rows of row-type blocks / columns of H_loc
        self.pg = make_groups(nprow, npcol)
        self.N = N

    def allreduce(self, a, group):
        a = np.ascontiguousarray(a)
        cplx = np.iscomplexobj(a)
        t = torch.from_numpy(a.view(np.float64) if cplx else a)
        dist.all_reduce(t, group=self.pg[group])
        return a

    # mpi/hemm.hpp:114-200: W_loc = alpha * H_loc^H * V_loc + beta' * W_loc ; all-reduce over the column communicator
    def hemm_c2r(self, H_loc, V_loc, W_loc, alpha, beta):
        b = beta if self.myrow == 0 else 0.0
        out = alpha * (H_loc.conj().T @ V_loc) + (b * W_loc if b != 0 else 0)
        return self.allreduce(out, "col")

    # mpi/hemm.hpp:201-229: V_loc = alpha * H_loc * W_loc + beta' * V_loc ; all-reduce over the row communicator
    def hemm_r2c(self, H_loc, W_loc, V_loc, alpha, beta):
        b = beta if self.mycol == 0 else 0.0
        out = alpha * (H_loc @ W_loc) + (b * V_loc if b != 0 else 0)
        return self.allreduce(out, "row")

    # column-type -> row-type: every rank of a grid column needs the rows owned by that grid column; they are spread over
    # the members' column-type blocks -> one broadcast per source member (the product packs exactly these index lists)
    def redistribute_c2r(self, V_loc):
        out = np.zeros((len(self.cols), V_loc.shape[1]), dtype=V_loc.dtype)
        col_ranks = [i + self.mycol * self.nprow for i in range(self.nprow)]
        for ip in range(self.nprow):
            src_rows = self.rl.globals_of(ip)
            sel = [k for k, g in enumerate(src_rows) if self.cl.owner(int(g)) == self.mycol]
            if not sel:
                continue
            buf = np.ascontiguousarray(V_loc[sel, :]) if ip == self.myrow else np.zeros((len(sel), V_loc.shape[1]), dtype=V_loc.dtype)
            t = torch.from_numpy(buf.view(np.float64) if np.iscomplexobj(buf) else buf)
            dist.broadcast(t, src=col_ranks[ip], group=self.pg["col"])
            dst = [self.cl.local(int(src_rows[k])) for k in sel]
            out[dst, :] = buf
        return out

    # mpi/cholqr.hpp:51-110
    def cholqr1(self, V_loc):
        A = self.allreduce(V_loc.conj().T @ V_loc, "col")
        R = sla.cholesky(A, lower=False)
        return sla.solve_triangular(R, V_loc.conj().T, trans="C", lower=False).conj().T

    # mpi/rayleighRitz.hpp:103-186
    def rayleigh_ritz(self, H_loc, V_loc):
        W1 = self.hemm_c2r(H_loc, V_loc, None, 1.0, 0.0)
        W2 = self.redistribute_c2r(V_loc)
        A = self.allreduce(W2.conj().T @ W1, "row")
        w, Z = sla.eigh(A, lower=True, driver="evd")
        return w, V_loc @ Z

    # mpi/residuals.hpp:61-107
    def residuals(self, H_loc, V_loc, lam):
        W1 = self.hemm_c2r(H_loc, V_loc, None, 1.0, 0.0)
        W2 = self.redistribute_c2r(V_loc)
        r = np.sum(np.abs(W1 - W2 * lam[None, :]) ** 2, axis=0)
        return np.sqrt(self.allreduce(r, "row"))
