"""The N>1 path on CPU: world_size-2/3 gloo runs of the slab schedule (latticeboltzmann_amd.slabs:
slab_rows, neighbours, exchange_halos, advance) with a numpy backend built from the oracle, checked
bit for bit against the single-domain oracle.  The GPU backend (SlabSolver) plugs the HIP solver
into the same `advance`; what is covered here is everything around the kernels: row ownership, which
rows travel where (including the two-rank periodic case, where both neighbours are the same peer),
cycle length, the collide-only first update and the ghost-row bookkeeping."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class NumpySlab:
    """Holds g = collide(f) on the slab plus ghost rows, like the HIP solver does."""

    def __init__(self, dist, rank, world, cells, blocked, boundary, H, omega, surr, last_depth=1):
        from latticeboltzmann_amd import slabs
        from oracle import lbm_numpy as onp
        self.onp, self.dist, self.slabs = onp, dist, slabs
        self.boundary, self.H, self.surr = boundary, H, surr
        G = cells.shape[0]
        self.lo, self.hi = slabs.slab_rows(G, world, rank)
        self.N = self.hi - self.lo
        self.up, self.down = slabs.neighbours(world, rank, boundary)
        self.pt = H if self.up is not None else 0
        self.pb = H if self.down is not None else 0
        rows = np.arange(self.lo - self.pt, self.hi + self.pb)
        if boundary == "periodic":
            rows = rows % G
        self.state = cells[rows].copy()
        self.blocked = blocked[rows].copy()
        self.omega = onp.make_omega(self.blocked, omega)
        self.fresh, self.hv, self.nxt = True, H, None
        self.D = last_depth

    def _stream_reflect(self, g):
        f = g.copy()
        if self.boundary == "periodic":
            self.onp.stream_periodic(f)
        else:
            self.onp.stream(f, self.surr)
        self.onp.reflect(f, self.blocked)
        return f

    def _collide(self, f):
        u, p = self.onp.moments(f)
        self.onp.collide(f, u, p, self.omega)
        return f

    def _update(self):
        return self._collide(self._stream_reflect(self.state))

    # backend protocol of slabs.advance
    def is_fresh(self):
        return self.fresh

    def halo_valid(self):
        return self.hv

    def last_depth(self):
        return self.D

    def full_updates(self, k):
        for _ in range(k):
            if self.fresh:
                self._collide(self.state)
                self.fresh = False
            else:
                assert self.hv >= 1
                self.state = self._update()
                self.hv -= 1

    def edge_update(self, depth=1):
        assert self.hv == depth  # the jump uses up the ghost rows that are still valid
        keep = self.state
        for _ in range(depth):
            self.state = self._update()
        self.nxt, self.state = self.state, keep  # the edge rows are what matters; rest reused below

    def exchange(self):
        import torch
        n, H, pt = self.nxt, self.H, self.pt
        t = torch.from_numpy
        send_top, send_bot = t(n[pt:pt + H]), t(n[pt + self.N - H:pt + self.N])
        recv_top = t(n[0:H]) if self.up is not None else None
        recv_bot = t(n[pt + self.N:pt + self.N + H]) if self.down is not None else None
        for w in self.slabs.exchange_halos(self.dist, send_top, send_bot, recv_top, recv_bot,
                                           self.up, self.down):
            w.wait()

    def interior_update(self, depth=1):
        self.state, self.nxt, self.hv = self.nxt, None, self.H

    def owned_f(self):
        assert self.hv >= 1
        f = self._stream_reflect(self.state) if not self.fresh else self.state
        return f[self.pt:self.pt + self.N]


def _worker(rank, world, port, boundary, H, steps_list, G, M, outdir, last_depth=1):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from latticeboltzmann_amd import slabs
    from oracle import lbm_numpy as onp
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    try:
        rng = np.random.default_rng(5)
        blocked = rng.random((G, M)) < 0.08
        cells = onp.perturbed_cells(G, M, seed=2, amp=0.05)
        surr = onp.ambient()
        slab = NumpySlab(dist, rank, world, cells, blocked, boundary, H, 0.2, surr, last_depth)
        ref = cells.copy()
        for n in steps_list:
            slabs.advance(slab, n)
            onp.run(ref, blocked, n, 0.2, surr, boundary)
            got = slab.owned_f()
            assert np.array_equal(got, ref[slab.lo:slab.hi]), (rank, n, np.abs(got - ref[slab.lo:slab.hi]).max())
        open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,boundary,H", [(2, "periodic", 4), (2, "reference", 4),
                                              (3, "periodic", 2), (3, "reference", 3)])
def test_slab_schedule_gloo(tmp_path, world, boundary, H):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, boundary, H, [1, 3, 4, 6, 1], 37, 21, str(tmp_path)),
             nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("world,boundary,H,D", [(2, "periodic", 4, 4), (3, "reference", 6, 4),
                                                (2, "reference", 3, 6)])
def test_slab_schedule_with_a_deep_last_jump(tmp_path, world, boundary, H, D):
    """The cycle's split jump covers several updates (LDS kernel on the GPU): H - D full updates,
    then D updates at once for the edge rows, exchange, the same jump for the interior."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, boundary, H, [1, 3, 4, 6, 1, 9, 2], 40, 21, str(tmp_path), D),
             nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_slab_rows_cover():
    from latticeboltzmann_amd.slabs import slab_rows, neighbours
    for G in (8, 37, 131072):
        for P in (1, 2, 3, 8):
            spans = [slab_rows(G, P, r) for r in range(P)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(spans[i][1] == spans[i + 1][0] for i in range(P - 1))
    assert neighbours(1, 0, "periodic") == (None, None)
    assert neighbours(2, 0, "periodic") == (1, 1)
    assert neighbours(4, 0, "reference") == (None, 1)
    assert neighbours(4, 3, "reference") == (2, None)


def test_bounded_wait_raises_instead_of_hanging():
    """Every wait of the slab schedule is a host-side poll with a deadline (slabs.py): a wait that
    never completes raises SlabTimeout naming the wait and the rank; one that does returns."""
    import types
    from latticeboltzmann_amd import slabs
    fake = types.SimpleNamespace(rank=3, world=8, device=3, timeout_s=0.05, _pending=object(),
                                 solver=types.SimpleNamespace(halo_valid=2))
    with pytest.raises(slabs.SlabTimeout, match=r"rank 3/8 .*halo exchange.* not complete"):
        slabs.SlabSolver._bounded_wait(fake, "halo exchange (event E2)", lambda: False)
    calls = []
    slabs.SlabSolver._bounded_wait(fake, "x", lambda: calls.append(1) or len(calls) > 300)
    assert len(calls) == 301
    # _wait_pending polls the event and clears it
    ev = types.SimpleNamespace(query=lambda: True)
    fake._pending = ev
    fake._bounded_wait = lambda what, done: slabs.SlabSolver._bounded_wait(fake, what, done)
    slabs.SlabSolver._wait_pending(fake)
    assert fake._pending is None
