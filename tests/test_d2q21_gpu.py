"""GPU: the D2Q21 lattice (lbm_lattice LBM_LATTICE_D2Q21, csrc/lbm_d2q21.hip) through the C-ABI against
oracle/lbm_d2q21.py, bit for bit.  PARITY UNPINNED by the reference beyond the draft's tables
(tests/test_d2q21_cpu.py)."""
import numpy as np
import pytest

from oracle import lbm_d2q21 as q

pytestmark = pytest.mark.gpu


def _solver(N, M, omega, blocked, cells, boundary="periodic", surr=None):
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    s = Solver(N, M, omega, boundary=boundary, lattice="d2q21")
    s.set_surroundings(q.ambient() if surr is None else surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    return s


@pytest.mark.parametrize("boundary", ["periodic", "reference"])
@pytest.mark.parametrize("shape", [(37, 130), (64, 64), (4, 4), (5, 67), (33, 200), (129, 7), (6, 128), (7, 129)])
@pytest.mark.parametrize("omega", [q.OMEGA, 0.6, 1e-14])
def test_d2q21_equals_its_oracle(shape, omega, boundary):
    N, M = shape
    rng = np.random.default_rng(N * 1000 + M)
    blocked = rng.random((N, M)) < 0.08
    cells = q.perturbed_cells(N, M, seed=3, amp=0.05)
    surr = q.ambient(100, [0.02, 0.25])
    s = _solver(N, M, omega, blocked, cells, boundary, surr)
    assert np.array_equal(s.read_blocked(), blocked)
    assert np.array_equal(s.read_cells(), cells)          # state right after upload
    ref = cells.copy()
    done = 0
    for n in (1, 1, 3, 12):
        s.fused_collide_stream(n)
        q.run(ref, blocked, n, omega, surr, boundary)
        done += n
        assert s.steps_done == done
        got = s.read_cells()
        assert got.shape == (N, M, 21)
        assert np.array_equal(got, ref), f"{shape} omega={omega} after {done}: max abs diff {np.abs(got - ref).max()}"
        u, rho = s.read_moments()
        ur, rr = q.moments(ref)
        assert np.array_equal(rho, rr) and np.array_equal(u, ur)
    # windows
    assert np.array_equal(s.read_cells(rows=(1, 3), cols=(M - 3, M)), ref[1:3, M - 3:])
    s.close()


def test_d2q21_zero_density_and_device_initial_state():
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 40, 130
    blocked = np.zeros((N, M), bool)
    blocked[10:20, 30:50] = True
    cells = q.perturbed_cells(N, M, seed=5)
    cells[blocked] = 0                      # latticeboltzmann.jl:94: nothing inside obstacles
    cells[3, 7, :] = 0                      # rho = 0 with u = 0/0 -> nan_to_num
    s = _solver(N, M, 0.3, blocked, cells)
    s.fused_collide_stream(25)
    ref = q.run(cells.copy(), blocked, 25, 0.3)
    assert np.array_equal(s.read_cells(), ref, equal_nan=True)
    s.close()
    # device-side equilibrium start: amp = 0 is the broadcast ambient state (latticeboltzmann.py:142)
    s = Solver(N, M, 0.3, boundary="periodic", lattice="d2q21")
    s.init_equilibrium(100.0, (0.0, 0.3), amp=0.0)
    got = s.read_cells()
    assert np.array_equal(got, np.broadcast_to(q.ambient(), (N, M, 21)))
    s.close()


def test_d2q21_the_drafts_own_scene():
    """latticeboltzmann.jl:25-32: the 400 x 800 cylinder, OMEGA = 0.01, rho = 100, u = [0, 0.3], inlet /
    outflow edges, from the broadcast ambient state: 150 steps against the oracle, bit for bit."""
    N, M = 400, 800
    blocked = q.blocked_cylinder(N, M)
    surr = q.ambient()
    cells = np.ascontiguousarray(np.broadcast_to(surr, (N, M, 21)))
    s = _solver(N, M, q.OMEGA, blocked, cells, "reference", surr)
    s.fused_collide_stream(150)
    ref = q.run(cells.copy(), blocked, 150, q.OMEGA, surr, "reference")
    assert np.array_equal(s.read_cells(), ref)
    u, rho = s.read_moments()
    ur, rr = q.moments(ref)
    assert np.array_equal(rho, rr) and np.array_equal(u, ur)
    s.close()


def test_d2q21_scope_is_enforced():
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    import numpy as np
    with pytest.raises(RuntimeError, match="single-domain"):
        Solver(16, 16, 0.3, boundary="cuda_kernel", lattice="d2q21")
    with pytest.raises(RuntimeError, match="single-domain"):
        Solver(16, 16, 0.3, boundary="periodic", lattice="d2q21", dtype=np.float16)
    with pytest.raises(RuntimeError, match="N >= 4"):
        Solver(3, 16, 0.3, boundary="periodic", lattice="d2q21")
    s = Solver(16, 16, 0.3, boundary="periodic", lattice="d2q21", inner_timesteps=3)
    s.init_equilibrium(100.0, (0.0, 0.3), amp=0.01, seed=1)
    s.fused_collide_stream(2)               # inner_timesteps = 3 updates per outer iteration, one per launch
    assert s.steps_done == 6
    for call in (lambda: s.render(), lambda: s.set_tuning(6, 0), lambda: s.step_rows(0, 16, True),
                 lambda: s.halo_region(0, False)):
        with pytest.raises(RuntimeError, match="not available for LBM_LATTICE_D2Q21"):
            call()
    s.close()
