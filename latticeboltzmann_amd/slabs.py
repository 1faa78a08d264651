"""y-slab decomposition of one D2Q9 domain over the GPUs of a node, one process per GPU.

No reference code exists for this (the reference is single-GPU; README.md:12 only quotes a
4-GPU number): SURVEY.md 8(e) defines it.  Rank r owns rows [r*N/P, (r+1)*N/P) plus `halo`
ghost rows on each seam.  The halo is deep: after one exchange the ghost rows are good for `halo`
updates (each update recomputes one ghost row less), so neighbours talk once every `halo` steps.

Schedule of one cycle (halo = H), on the compute stream C and the communication stream K:
    C: H-d full updates (row range shrinking by one ghost row per update)
    C: last d updates in one jump, edge rows only: [0, H) and [N-H, N)   -> event E1
    K: wait E1; grouped send/recv of those rows into the neighbours' ghost rows -> event E2
    C: the same jump for the interior rows [H, N-H)  (runs while the halos travel)
    host: wait for E2 (bounded) before the next cycle's first launch is enqueued
Every wait on this path is a HOST-side poll with a deadline (`SlabSolver.timeout_s`): the compute
stream never holds a device-side barrier on another stream's event, so its forward progress does not
depend on how the runtime places streams on hardware queues, and a rank that is stuck raises
SlabTimeout naming the wait, its rank and the stream states instead of blocking for ever (DESIGN.md
section 6 has the round-2 hang this answers).  The host is released as soon as the exchange is done,
i.e. while the interior jump still runs, so the compute stream does not run dry.
d = 1 with the plain kernel; with the LDS kernel (tb_max_depth > 1) d = min(tb_max_depth, H), so a
cycle of H <= 6 updates is three launches: two edge bands, then the interior.
The exchange is a grouped torch.distributed send/recv (RCCL over xGMI on the GPU box, gloo in the
CPU tests): `exchange_halos` below is backend-agnostic and is what tests/test_slabs_gloo.py drives.
There is no reduction or other collective on the data path.
"""
import os
import time

import numpy as np

from . import lb_hip_kernel as hk


class SlabTimeout(RuntimeError):
    """A bounded wait of the slab schedule ran out (a hang made visible)."""


def init_process_group_rccl(device, **kw):
    """torch.distributed over RCCL (backend "nccl") for the halo exchange, bound to `device` at once
    (eager communicator).  kw: init_method / rank / world_size when not taken from the torchrun
    environment.

    Stream priorities are deliberately left alone.  Round 2 tried RCCL's own stream and the
    communication stream at high priority, so that the halo messages would not queue behind the
    interior launch they overlap with; on this runtime (ROCm 7.x, 4 hardware queues per process)
    either of the two made the schedule hang now and then -- 3 of 4 randomised soaks
    (tools/stress_slabs.py) stopped within 70 cases, every time with the host in
    `compute.synchronize()` AFTER `compute.wait_event(E2)` had been enqueued: the compute stream
    never drained, against 0 of 1650 cases with default priorities -- and the measured gain had been
    0.1 %.  Since round 3 the compute stream no longer waits for E2 on the device at all (module
    docstring)."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device}"), **kw)
    return dist


def slab_rows(global_rows, world, rank):
    """Row range [lo, hi) of `rank`: equal slabs, the remainder spread over the first ranks."""
    q, r = divmod(global_rows, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def neighbours(world, rank, boundary, self_seam=False):
    """(up, down) ranks across the top / bottom seam, None at a true domain edge.  self_seam: a
    single periodic rank treats its own wrap-around as a seam and exchanges halos with itself
    (exercises the whole schedule, RCCL included, on one GPU)."""
    if world == 1:
        return (0, 0) if self_seam and boundary == "periodic" else (None, None)
    if boundary == "periodic":
        return (rank - 1) % world, (rank + 1) % world
    return (rank - 1 if rank > 0 else None), (rank + 1 if rank < world - 1 else None)


def exchange_halos(dist, send_top, send_bot, recv_top, recv_bot, up, down, group=None):
    """Post one grouped exchange; returns the list of work handles (call .wait() on each).

    send_top/send_bot: my first/last `halo` owned rows; recv_top/recv_bot: my ghost rows.
    Order matters when up == down (two ranks, periodic): both sides send [top, bottom], so the
    first message I receive from that peer is ITS top rows, which are my BOTTOM ghost rows.
    """
    ops = []
    if up is not None:
        ops.append(dist.P2POp(dist.isend, send_top, up, group))
    if down is not None:
        ops.append(dist.P2POp(dist.isend, send_bot, down, group))
    if down is not None:
        ops.append(dist.P2POp(dist.irecv, recv_bot, down, group))
    if up is not None:
        ops.append(dist.P2POp(dist.irecv, recv_top, up, group))
    if not ops:
        return []
    return dist.batch_isend_irecv(ops)


def advance(b, n):
    """The slab schedule (module docstring) over a backend `b` that provides is_fresh(),
    halo_valid(), last_depth(), full_updates(k), edge_update(d), exchange(), interior_update(d).
    SlabSolver is the GPU backend; tests/test_slabs_gloo.py drives the same function with a numpy
    backend.  last_depth() = how many updates the split jump at the end of a cycle covers (1 for
    the plain kernel; up to 6 with the LDS kernel, so that the whole cycle can run deep)."""
    while n > 0:
        if b.is_fresh():  # first update after an upload only collides: consumes no ghost row
            b.full_updates(1)
            n -= 1
            continue
        hv = b.halo_valid()
        # the hv updates left in the cycle go in ceil(hv / D) launches of balanced depth; the last
        # of them is the split jump (7 = 4 + 3 at D = 6, not 1 + 6: a depth-1 launch costs more
        # than half a deep one)
        dl = hv // -(-hv // b.last_depth())
        if hv > dl:  # everything before the split jump: full launches
            k = min(n, hv - dl)
            b.full_updates(k)
            n -= k
            continue
        if n < hv:  # the cycle does not end within this call
            b.full_updates(n)
            return
        b.edge_update(hv)      # last hv updates of the cycle: rows the neighbours need, first
        b.exchange()           # ... travel on the communication stream ...
        b.interior_update(hv)  # ... while the interior rows jump the same hv updates
        n -= hv


class _DevMem:
    """Raw device memory exposed through __cuda_array_interface__ so torch can wrap it."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {
            "shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}


class SlabSolver:
    """One rank's slab.  torch is used for streams, events and torch.distributed only."""

    def __init__(self, global_rows, M, omega, rank, world, boundary="periodic", halo=4,
                 dtype=np.float32, device=None, dist=None, group=None, host_staged=None,
                 tb_max_depth=1, self_seam=False):
        import torch
        self.torch = torch
        self.dist = dist
        self.group = group
        self.rank, self.world, self.boundary = rank, world, boundary
        self.global_rows, self.M = global_rows, M
        self.row_lo, self.row_hi = slab_rows(global_rows, world, rank)
        self.N = self.row_hi - self.row_lo
        self.up, self.down = neighbours(world, rank, boundary, self_seam)
        self.seams = self.up is not None or self.down is not None
        self.halo = halo if self.seams else 0
        if self.seams and self.N < 2 * halo:
            raise ValueError(f"slab of {self.N} rows is too thin for halo {halo}")
        self.device = torch.cuda.current_device() if device is None else device
        self.solver = hk.Solver(
            self.N, M, omega, boundary=boundary, dtype=dtype, device=self.device, halo=self.halo,
            top_edge=hk.YEDGE_SEAM if self.up is not None else hk.YEDGE_DOMAIN,
            bottom_edge=hk.YEDGE_SEAM if self.down is not None else hk.YEDGE_DOMAIN)
        self.solver.set_tuning(tb_max_depth, 0)
        self.tb_max_depth = tb_max_depth
        if M < 8 or self.N < 16:  # the LDS kernel does not take grids this small
            self.tb_max_depth = 1
        self.compute = torch.cuda.Stream(device=self.device)
        self.comm = torch.cuda.Stream(device=self.device)  # default priority: see init_process_group_rccl
        self.solver.set_stream(self.compute.cuda_stream)
        self._tensors = {}
        self._fresh = True
        self._pending = None  # event (end of the halo exchange) the next update has to wait for
        # deadline of every host-side wait on this path, seconds
        self.timeout_s = float(os.environ.get("LBM_SLAB_TIMEOUT_S", "3600"))
        # gloo cannot send/recv device memory: stage the halos through the host (no overlap).
        # Only for rehearsing several ranks on one GPU; RCCL moves device memory directly.
        if host_staged is None:
            host_staged = self.seams and dist is not None and dist.get_backend(group) == "gloo"
        self.host_staged = host_staged

    # -- scene -------------------------------------------------------------------------------------
    def set_surroundings(self, surroundings):
        self.solver.set_surroundings(surroundings)

    def init_device_scene(self, rho, u, amp=0.01, seed=0, ndiscs=0, radius=0):
        """Synthetic scene generated on the device, consistent across ranks (ghost rows included)."""
        wrap = self.global_rows if self.boundary == "periodic" else 0
        self.solver.init_equilibrium(rho, u, amp, seed, self.row_lo, wrap)
        self.solver.init_blocked_discs(ndiscs, radius, seed + 1, self.row_lo, self.global_rows)
        self._fresh = True

    def upload_global(self, cells, blocked):
        """Cut this rank's slab (with ghost rows) out of host arrays of the whole domain."""
        H = self.halo
        rows = np.arange(self.row_lo - H, self.row_hi + H)
        if self.boundary == "periodic":
            rows = rows % self.global_rows
        else:
            rows = np.clip(rows, 0, self.global_rows - 1)  # beyond a domain edge: never read
        self.solver.upload_blocked(np.ascontiguousarray(blocked[rows]), -H)
        self.solver.upload_cells(np.ascontiguousarray(cells[rows]), -H)
        self.solver.halo_mark_filled()
        self._fresh = True

    # -- stepping ----------------------------------------------------------------------------------
    def _wrap(self, ptr, nbytes):
        t = self._tensors.get(ptr)
        if t is None:
            t = self.torch.as_tensor(_DevMem(ptr, nbytes), device=f"cuda:{self.device}")
            self._tensors[ptr] = t
        return t

    def _bounded_wait(self, what, done):
        """Poll `done()` until it is true or `timeout_s` has passed (then: SlabTimeout).  A poll, not
        hipStreamSynchronize / hipEventSynchronize: those cannot be bounded, and in round 2's hangs
        a second thread's stream query blocked behind the main thread's synchronize."""
        t0 = time.monotonic()
        spins = 0
        while not done():
            spins += 1
            if spins > 200:  # the first ~200 polls spin: an exchange is tens of microseconds
                time.sleep(0.0002)
                if time.monotonic() - t0 > self.timeout_s:
                    raise SlabTimeout(
                        f"rank {self.rank}/{self.world} (device {self.device}): {what} not complete after "
                        f"{self.timeout_s:.0f} s; halo_valid={self.solver.halo_valid}, "
                        f"exchange pending={self._pending is not None}")

    def _wait_pending(self):
        """The ghost rows of the last exchange must have arrived before the next update reads them:
        a host-side, bounded wait for E2 (the interior jump is still running on the compute stream
        meanwhile, so nothing runs dry).  Completion of E2 seen by the host orders the exchange's
        writes before every kernel launched afterwards."""
        if self._pending is not None:
            ev = self._pending
            self._bounded_wait("halo exchange (event E2 on the communication stream)", ev.query)
            self._pending = None

    def step(self, n=1):
        """Advance by n updates (asynchronous)."""
        if not self.seams:
            self.solver.advance(n)
            return
        advance(self, n)

    # backend protocol of `advance` -------------------------------------------------------------
    def is_fresh(self):
        return self._fresh

    def halo_valid(self):
        return self.solver.halo_valid

    def last_depth(self):
        return max(1, min(self.tb_max_depth, self.halo))

    def full_updates(self, k):
        self._wait_pending()
        self.solver.advance(k)  # LDS temporal blocking when tb_max_depth > 1 (set_tuning)
        self._fresh = False

    def edge_update(self, depth=1):
        self._wait_pending()
        self.solver.advance_rows(0, self.halo, depth, False)
        self.solver.advance_rows(self.N - self.halo, self.halo, depth, False)

    def exchange(self):
        torch, s = self.torch, self.solver
        e1 = torch.cuda.Event()
        e1.record(self.compute)
        ptrs = [s.halo_region(0, False, next=True), s.halo_region(1, False, next=True),
                s.halo_region(0, True, next=True), s.halo_region(1, True, next=True)]
        st, sb, rt, rb = (self._wrap(p, b) for p, b in ptrs)
        if self.host_staged:
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(e1)
                hst, hsb = st.cpu(), sb.cpu()  # synchronous D2H on the comm stream
                hrt, hrb = torch.empty_like(hst), torch.empty_like(hsb)
                for w in exchange_halos(self.dist, hst, hsb, hrt, hrb, self.up, self.down, self.group):
                    w.wait()
                if self.up is not None:
                    rt.copy_(hrt)
                if self.down is not None:
                    rb.copy_(hrb)
                e2 = torch.cuda.Event()
                e2.record(self.comm)
            self._pending = e2
            return
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(e1)
            for w in exchange_halos(self.dist, st, sb, rt, rb, self.up, self.down, self.group):
                w.wait()
            e2 = torch.cuda.Event()
            e2.record(self.comm)
        self._pending = e2

    def interior_update(self, depth=1):
        # not ordered after the exchange: it runs while the halos travel
        self.solver.advance_rows(self.halo, self.N - 2 * self.halo, depth, True)
        self.solver.halo_mark_filled()

    def sync(self):
        self._wait_pending()
        self._bounded_wait("compute stream drain", self.compute.query)
        self._bounded_wait("communication stream drain", self.comm.query)

    # -- readback (this rank's rows) ---------------------------------------------------------------
    def read_moments(self, rows=None, cols=None):
        self.sync()
        return self.solver.read_moments(rows, cols)

    def read_cells(self, rows=None, cols=None):
        self.sync()
        return self.solver.read_cells(rows, cols)

    def close(self):
        self.sync()
        self.solver.close()
