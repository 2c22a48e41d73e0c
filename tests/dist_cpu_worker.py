"""CPU (gloo) worker: the distributed oracle vs the serial oracle; pins layout math and group structure."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist  # noqa: E402
from chase_amd import dist as cd  # noqa: E402
from oracle import chase_oracle as O  # noqa: E402
from oracle.dist_oracle import DistBlocks  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nprow, npcol = cd.grid_shape(world)
    mb = int(sys.argv[1])
    cplx = sys.argv[2] == "z"
    # 1) reference known answer (tests/linalg/internal/mpi/hemm.cpp:36-119): H == 1 (10x10), V == 2, W == 3, alpha 2, beta 3
    D = DistBlocks(10, nprow, npcol)
    H = np.ones((10, 10))
    Hl = H[np.ix_(D.rows, D.cols)]
    V = np.full((len(D.rows), 2), 2.0)
    W = np.full((len(D.cols), 2), 3.0)
    W = D.hemm_c2r(Hl, V, W, 2.0, 3.0)
    assert np.all(W == 49.0), W
    V = D.hemm_r2c(Hl, W, V, 2.0, 3.0)
    assert np.all(V == 986.0), V
    # 2) against the serial oracle
    N, n = 203, 12
    D = DistBlocks(N, nprow, npcol, mb, mb)
    H = O.clement(N, cplx)
    Hl = H[np.ix_(D.rows, D.cols)]
    X = O.random_start_vectors(N, n, cplx)
    Vl = X[D.rows, :]
    Wl = D.hemm_c2r(Hl, Vl, None, 0.5, 0.0)
    assert np.allclose(Wl, 0.5 * (H.conj().T @ X)[D.cols, :], atol=1e-12)
    assert np.array_equal(D.redistribute_c2r(Vl), X[D.cols, :])
    Ql = D.cholqr1(Vl)
    Q, info = O.cholQR1(X)
    assert info == 0 and np.allclose(Ql, Q[D.rows, :], atol=1e-12)
    w, Vr = D.rayleigh_ritz(Hl, Ql)
    w0, V0 = O.rayleighRitz(H, Q)
    assert np.allclose(w, w0, atol=1e-9 * np.abs(w0).max())
    r = D.residuals(Hl, Vr, w)
    r0 = O.residuals(H, w0, V0)
    assert np.allclose(r, r0, atol=1e-8 * max(1.0, r0.max()))
    # 3) the oracle's OWN index arithmetic (oracle/dist_oracle.py OLayout, restating distMatrix.hpp:44-67,1992-2052) and the
    #    PRODUCT's layout helpers (chase_hip_numroc / owner / local_index through chase_amd.dist.Layout) are two independent
    #    implementations of the reference's rule: they must agree index for index
    for (olay, p, nbsz) in ((D.rl, nprow, mb), (D.cl, npcol, mb)):
        play = cd.Layout(N, nbsz, p)
        assert play.nb == olay.nb
        for q in range(p):
            assert play.count(q) == olay.count(q)
            assert np.array_equal(play.globals_of(q), olay.globals_of(q))
        for g in range(N):
            assert play.owner(g) == olay.owner(g) and play.local(g) == olay.local(g)
    assert cd.coords_of(rank, nprow) == (D.myrow, D.mycol)
    # every global index is owned exactly once and round-trips
    for lay, p in ((D.rl, nprow), (D.cl, npcol)):
        seen = np.zeros(N, dtype=int)
        for q in range(p):
            g = lay.globals_of(q)
            assert len(g) == lay.count(q)
            seen[g] += 1
            assert all(lay.owner(int(x)) == q for x in g[:5])
            assert [lay.local(int(x)) for x in g[:7]] == list(range(min(7, len(g))))
        assert np.all(seen == 1)
    dist.barrier()
    if rank == 0:
        print("DIST_CPU_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
