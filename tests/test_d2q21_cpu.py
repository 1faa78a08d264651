"""D2Q21 (scoped): what the reference's draft pins - its weight and velocity
tables and their four asserts (latticeboltzmann.jl:3-22) - and the invariants of the oracle built on
them (oracle/lbm_d2q21.py).  PARITY UNPINNED beyond that: the draft is not a runnable program."""
import numpy as np

from oracle import lbm_d2q21 as q


def test_the_drafts_tables_and_asserts():
    q.lattice_asserts()                      # latticeboltzmann.jl:12-13, 21-22
    assert q.w.shape == (21,) and q.e.shape == (21, 2) and q.r2 == 1.5
    assert q.w[10] == np.float32(91 / 324) and tuple(q.e[10]) == (0, 0)   # the rest population
    assert tuple(q.e[0]) == (0, -3) and tuple(q.e[20]) == (0, 3) and tuple(q.e[7]) == (-3, 0)
    # second moment of the weights = 1 / r2 (the lattice's speed of sound squared), isotropic
    e2 = (q.w.astype(np.float64)[:, None] * q.e.astype(np.float64) ** 2).sum(0)
    assert np.allclose(e2, 1 / q.r2, rtol=1e-6)
    assert abs((q.w.astype(np.float64) * q.e[:, 0] * q.e[:, 1]).sum()) < 1e-9


def test_conservation_asserts_of_the_ambient_state():
    """The reference's two import-time asserts (latticeboltzmann.py:93-94) with the draft's tables and
    parameters (latticeboltzmann.jl:27-29: rho = 100, u = [0, 0.3])."""
    s = q.ambient()
    assert np.isclose(s.sum(), q.p_ambient, rtol=1e-6)
    assert np.isclose(s @ q.e_f / q.p_ambient, q.u_ambient, rtol=1e-5, atol=1e-7).all()


def test_pairwise21_is_numpys_sum():
    rng = np.random.default_rng(0)
    a = (rng.random((50, 70, 21)) * rng.choice([1e-3, 1.0, 1e3], size=(50, 70, 21))).astype(np.float32)
    assert np.array_equal(q.pairwise21(a), np.sum(a, axis=2))


def test_mass_and_momentum_are_conserved_on_the_periodic_grid():
    N, M = 24, 40
    cells = q.perturbed_cells(N, M, seed=1, amp=0.02)
    blocked = np.zeros((N, M), bool)
    m0 = cells.astype(np.float64).sum()
    j0 = (cells.astype(np.float64) @ q.e.astype(np.float64)).sum((0, 1))
    q.run(cells, blocked, 30, 0.3)
    assert abs(cells.astype(np.float64).sum() - m0) / m0 < 1e-6
    assert np.allclose((cells.astype(np.float64) @ q.e.astype(np.float64)).sum((0, 1)), j0, rtol=1e-4, atol=1e-2)


def test_uniform_flow_stays_uniform_and_obstacles_reflect():
    N, M = 12, 16
    s = q.ambient()
    cells = np.broadcast_to(s, (N, M, 21)).copy()
    q.run(cells, np.zeros((N, M), bool), 5, 0.5)
    assert np.allclose(cells, s, rtol=1e-5)
    # reflect is k <-> 20 - k at blocked cells only
    rng = np.random.default_rng(2)
    c = rng.random((N, M, 21)).astype(np.float32)
    b = rng.random((N, M)) < 0.2
    want = c.copy()
    want[b] = want[b][:, ::-1]
    q.reflect(c, b)
    assert np.array_equal(c, want)


def test_stream_moves_population_k_by_e_k():
    N, M = 9, 11
    c = np.zeros((N, M, 21), np.float32)
    c[4, 5, :] = np.arange(1, 22)
    q.stream_periodic(c)
    for k, (dy, dx) in enumerate(q.e):
        assert c[(4 + dy) % N, (5 + dx) % M, k] == k + 1
    assert np.count_nonzero(c) == 21


def test_reference_stream_slices_reach_three_cells():
    """latticeboltzmann.py:115-120 with velocities of up to three cells: the first dx columns / dy rows
    take surroundings[k], the last |dx| / |dy| keep their own value (the empty `-3:0` slices)."""
    N, M = 8, 9
    surr = np.arange(100, 121, dtype=np.float32)
    c = (np.arange(N * M * 21, dtype=np.float32).reshape(N, M, 21) + 1000)
    want = c.copy()
    for k, (dy, dx) in enumerate(q.e):
        for y in range(N):
            for x in range(M):
                ys, xs = y - dy, x - dx
                if (dx > 0 and x < dx) or (dy > 0 and y < dy):
                    want[y, x, k] = surr[k]
                elif 0 <= ys < N and 0 <= xs < M:
                    want[y, x, k] = c[ys, xs, k]
    q.stream(c, surr)
    assert np.array_equal(c, want)
