"""N ranks of the distributed Impl as THREADS of one process.

Two users.  (1) The tests: the GPU box allows few processes on its card, and the reference runs its distributed tests as
ranks sharing one box (tests/CMakeLists.txt:23-31).  Every rank is a Python thread with its own chase_hip_ctx (own HIP
stream) and its own chase_hip_grid on the host-callback transport; the callbacks meet in this module instead of in gloo.
ctypes releases the GIL around every library call, so the ranks' kernels really run concurrently on the one device, and any
grid shape (2x1, 2x2, 3x2, 4x2 ...) runs with ONE process holding the GPU.  (2) `bench.py --ranks threads`: one process,
one thread per GPU, every thread with the context of ITS device and RCCL communicators created by ncclCommInitRank from
the N threads ("single process, one host thread per GPU") - here this module only carries the bootstrap
(unique ids), the barriers and the max / sum over ranks of the timing figures; the data path is RCCL.

`RankComm` is what a scenario sees as its communicator: rank / world, all_gather_object, barrier, once(key, fn) (compute a
value — an oracle solve, say — on the first rank that asks, hand the same object to the others), and the three fabric
calls chase_amd.dist.Grid needs (allreduce / bcast / sendrecv on float64 views).  `GlooComm` offers the same surface on
torch.distributed for the few tests that still run ranks as processes."""
import queue
import threading
import time
import traceback

import numpy as np

ROW, COL = 0, 1
TIMEOUT = 600.0


class _World:
    def __init__(self, nprow, npcol, shared_device_fabric=False):
        self.nprow, self.npcol, self.n = nprow, npcol, nprow * npcol
        self.fabric = None                     # chase_hip_fabric* of the shared-device transport (ranks on ONE GPU, no host staging)
        if shared_device_fabric:
            import ctypes
            from chase_amd.capi import lib, check
            from chase_amd import dist as _cd  # noqa: F401  (declares the fabric entry points)
            f = ctypes.c_void_p()
            check(lib.chase_hip_fabric_create(ctypes.byref(f), nprow, npcol), "fabric_create")
            self.fabric = f
        self.failed = threading.Event()
        self.world_barrier = threading.Barrier(self.n)
        # group id: (ROW, myrow) has npcol members, (COL, mycol) has nprow members
        self.group_barrier = {}
        self.group_slots = {}
        for i in range(nprow):
            self.group_barrier[(ROW, i)] = threading.Barrier(npcol)
            self.group_slots[(ROW, i)] = [None] * (npcol + 1)
        for j in range(npcol):
            self.group_barrier[(COL, j)] = threading.Barrier(nprow)
            self.group_slots[(COL, j)] = [None] * (nprow + 1)
        self.gather_slots = [None] * self.n
        self.mail = {}
        self.mail_lock = threading.Lock()
        self.once_lock = threading.Lock()
        self.once_vals = {}

    def abort(self):
        self.failed.set()
        if self.fabric:
            from chase_amd.capi import lib
            lib.chase_hip_fabric_abort(self.fabric)          # ranks waiting inside a device-side collective return an error
        self.world_barrier.abort()
        for b in self.group_barrier.values():
            b.abort()

    def box(self, key):
        with self.mail_lock:
            q = self.mail.get(key)
            if q is None:
                q = self.mail[key] = queue.Queue()
            return q


class RankComm:
    def __init__(self, world, rank):
        self.w, self.rank, self.world = world, rank, world.n
        self.nprow, self.npcol = world.nprow, world.npcol
        self.myrow, self.mycol = rank % world.nprow, rank // world.nprow          # column-major grid ordering
        self._once_seen = {}

    # ---- what scenarios use ------------------------------------------------------------------------------------------
    def _wait(self, barrier):
        barrier.wait(timeout=TIMEOUT)

    def barrier(self):
        self._wait(self.w.world_barrier)

    def all_gather_object(self, obj):
        self.w.gather_slots[self.rank] = obj
        self._wait(self.w.world_barrier)
        out = list(self.w.gather_slots)
        self._wait(self.w.world_barrier)
        return out

    def allreduce_max(self, values):
        """elementwise maximum over all ranks of a short list of floats (timing figures)"""
        rows = self.all_gather_object([float(v) for v in values])
        return [max(r[i] for r in rows) for i in range(len(values))]

    def allreduce_sum(self, values):
        rows = self.all_gather_object([float(v) for v in values])
        return [sum(r[i] for r in rows) for i in range(len(values))]

    def allreduce_min(self, values):
        rows = self.all_gather_object([float(v) for v in values])
        return [min(r[i] for r in rows) for i in range(len(values))]

    def once(self, key, fn):
        """Every rank calls once(key, fn) at the same point of its program; fn() runs on ONE of them (repeated keys are told
        apart by the number of times this rank has asked for them)."""
        k = (key, self._once_seen.get(key, 0))
        self._once_seen[key] = k[1] + 1
        with self.w.once_lock:
            if k not in self.w.once_vals:
                self.w.once_vals[k] = fn()
            return self.w.once_vals[k]

    # ---- fabric of chase_amd.dist.Grid (group = ROW: the ranks of my grid row, COL: of my grid column) ----------------
    def _group(self, group):
        gid = (ROW, self.myrow) if group == ROW else (COL, self.mycol)
        me = self.mycol if group == ROW else self.myrow
        return self.w.group_barrier[gid], self.w.group_slots[gid], me

    def allreduce(self, group, a):
        bar, slots, me = self._group(group)
        size = len(slots) - 1
        slots[me] = a
        self._wait(bar)
        if me == 0:                                    # fixed summation order, one result for all: replicas stay identical
            tot = slots[0].copy()
            for i in range(1, size):
                tot += slots[i]
            slots[size] = tot
        self._wait(bar)
        a[:] = slots[size]
        self._wait(bar)

    def bcast(self, group, a, root):
        bar, slots, me = self._group(group)
        if me == root:
            slots[len(slots) - 1] = a
        self._wait(bar)
        if me != root:
            a[:] = slots[len(slots) - 1]
        self._wait(bar)

    def sendrecv(self, group, send, peer_send, recv, peer_recv):
        gid = (ROW, self.myrow) if group == ROW else (COL, self.mycol)
        me = self.mycol if group == ROW else self.myrow
        if send is not None:
            self.w.box((gid, me, peer_send)).put(send.copy())
        if recv is not None:
            q = self.w.box((gid, peer_recv, me))
            waited = 0.0
            while True:
                try:
                    recv[:] = q.get(timeout=0.5)
                    break
                except queue.Empty:
                    waited += 0.5
                    if self.w.failed.is_set() or waited > TIMEOUT:
                        raise RuntimeError("sendrecv: peer never sent (another rank failed or timed out)")


def run_threads(nprow, npcol, body, shared_device_fabric=False):
    """body(comm) on nprow*npcol threads; re-raises the first failure in the caller (a failing rank breaks every barrier, so
    the others fail fast instead of waiting for it)."""
    world = _World(nprow, npcol, shared_device_fabric)
    errors = [None] * world.n

    def wrapped(rank):
        try:
            comm = RankComm(world, rank)
            body(comm)
            comm.barrier()
        except BaseException as e:  # noqa: BLE001 - reported to the caller below
            errors[rank] = (e, traceback.format_exc(), time.monotonic())
            world.abort()

    threads = [threading.Thread(target=wrapped, args=(r,), name=f"rank{r}", daemon=True) for r in range(world.n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(TIMEOUT + 60)
    alive = [t.name for t in threads if t.is_alive()]
    if world.fabric and not alive:
        from chase_amd.capi import lib
        lib.chase_hip_fabric_destroy(world.fabric)           # every grid on it is closed (run_ranks closes them in `finally`)
        world.fabric = None
    # the FIRST failure in time is the cause; the ranks that then found their barriers broken (directly, or as a failed
    # transport callback inside the library) are consequences
    failed = sorted(((e[2], r, e) for r, e in enumerate(errors) if e), key=lambda t: t[0])
    if failed:
        _, r, (e, tb, _) = failed[0]
        raise AssertionError(f"rank {r} of the {nprow}x{npcol} grid failed first:\n{tb}") from e
    assert not alive, f"ranks still running after the timeout: {alive}"


def run_ranks(nprow, npcol, fn, *args, device=0, transport="host", **kw):
    """Runs fn(ctx, grid, comm, *args, **kw) on nprow*npcol rank threads, each with its own Context and Grid.  device: one
    ordinal for all ranks (tests: the ranks share a GPU over the host transport) or a callable rank -> ordinal
    (`bench.py --ranks threads`: one GPU per thread, transport "rccl").  transport "shared": the ranks share ONE GPU and their
    collectives are device-side sums / copies ordered by events (chase_hip_grid_create_shared) - no host staging."""
    from chase_amd.capi import Context
    from chase_amd import dist as cd

    def body(comm):
        ctx = grid = None
        try:
            ctx = Context(device(comm.rank) if callable(device) else device)
            grid = cd.Grid(ctx, nprow, npcol, comm.rank, transport=transport, pg=comm)
            fn(ctx, grid, comm, *args, **kw)
        finally:
            if grid is not None:
                grid.close()
            if ctx is not None:
                ctx.close()

    run_threads(nprow, npcol, body, shared_device_fabric=(transport == "shared"))


class GlooComm:
    """The same surface for ranks that are processes (torch.distributed already initialised)."""

    def __init__(self):
        import torch.distributed as dist
        self.d = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.d.barrier()

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.d.all_gather_object(out, obj)
        return out

    def once(self, key, fn):
        return fn()

    def _reduce(self, values, op):
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        self.d.all_reduce(t, op=op)
        return [float(x) for x in t]

    def allreduce_max(self, values):
        return self._reduce(values, self.d.ReduceOp.MAX)

    def allreduce_sum(self, values):
        return self._reduce(values, self.d.ReduceOp.SUM)

    def allreduce_min(self, values):
        return self._reduce(values, self.d.ReduceOp.MIN)
