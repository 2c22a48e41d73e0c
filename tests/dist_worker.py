"""Ranks as PROCESSES (launched by torch.distributed.run; several may share one GPU) - kept for the few tests that need a
process per rank (the global solver objects of the C interface, the RCCL transport).  Everything else runs its ranks as
threads of the pytest process (tests/rank_threads.py); the scenarios are shared (tests/dist_scenarios.py).

usage: dist_worker.py <transport: host|rccl> <scenario> [args] [-- <scenario> [args] ...]
Several scenarios separated by "--" run one after the other on the same context and grid (one process start-up, one set of RCCL
communicators for all of them).  Exit code 0 = all assertions passed on this rank."""
import os
import sys

import faulthandler

faulthandler.enable()
if os.environ.get("CHASE_TEST_FAKE_HOSTS") == "1":
    # Real RCCL collectives between ranks on a ONE-GPU box: RCCL refuses two ranks of a communicator on one device of one
    # host ("Duplicate GPU detected"), but it tells hosts apart by NCCL_HOSTID - give every rank process its own "host" and
    # the ranks talk through RCCL's socket transport over loopback (slow, but every ncclAllReduce / ncclBroadcast /
    # ncclSend / ncclRecv of the grid really runs between distinct ranks).  Must be set before the first RCCL call.
    os.environ["NCCL_HOSTID"] = "chase-test-host-" + os.environ.get("RANK", "0")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    os.environ.setdefault("NCCL_NET", "Socket")
    os.environ.setdefault("NCCL_SHM_DISABLE", "1")
    os.environ.setdefault("NCCL_P2P_DISABLE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from chase_amd.capi import Context  # noqa: E402
from chase_amd import dist as cd  # noqa: E402
import dist_scenarios as S  # noqa: E402
from rank_threads import GlooComm  # noqa: E402


def setup(transport, nprow=None, npcol=None):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if nprow is None:
        nprow, npcol = cd.grid_shape(world)
    ndev = torch.cuda.device_count()
    ctx = Context(int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1))
    pg = cd.make_process_groups(nprow, npcol)
    grid = cd.Grid(ctx, nprow, npcol, rank, transport=transport, pg=pg)
    return ctx, grid, GlooComm()


def main():
    transport = sys.argv[1]
    jobs, cur = [], []
    for a in sys.argv[2:]:
        if a == "--":
            jobs.append(cur); cur = []
        else:
            cur.append(a)
    jobs.append(cur)
    ctx, grid, comm = setup(transport)
    try:
        for job in jobs:
            S.run_named(job[0], ctx, grid, comm, job[1:])
            comm.barrier()
            if comm.rank == 0:
                print("DIST_WORKER_DONE", job[0], flush=True)
        if comm.rank == 0:
            print("DIST_WORKER_OK", " ".join(j[0] for j in jobs), flush=True)
    finally:
        grid.close()
        ctx.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
