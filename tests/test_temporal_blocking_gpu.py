"""GPU parity of the INNER_TIMESTEPS > 1 path (LDS temporal blocking, lbm_tb.hip): the same scenes
as the T = 1 tests, bit for bit against the oracle, with strips narrower than 256 columns, partial
64-column blocks, chunk seams (small tb_rows), obstacles on strip and chunk seams, both edge rules,
and slab seams with ghost rows."""
import numpy as np
import pytest

from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc

pytestmark = pytest.mark.gpu


def _run_case(N, M, T, boundary, tb_rows, n_outer=3, omega=0.3, seed=0, density=0.08, tb_max=6):
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    rng = np.random.default_rng(seed + N * 131 + M)
    blocked = rng.random((N, M)) < density
    cells = onp.perturbed_cells(N, M, seed=seed + 1, amp=0.05)
    surr = onp.ambient()
    s = Solver(N, M, omega, inner_timesteps=T, boundary=boundary)
    s.set_tuning(tb_max, tb_rows)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    ref = cells.copy()
    for it in range(n_outer):
        s.fused_collide_stream(1)
        oc.run(ref, blocked, T, omega, surr, boundary)
        assert s.steps_done == (it + 1) * T
        got = s.read_cells()
        bad = np.argwhere(got != ref)
        assert bad.size == 0, (f"N={N} M={M} T={T} {boundary} rows={tb_rows} outer={it}: "
                               f"{len(bad)} mismatches, first {bad[:5].tolist()}")
    u, rho = s.read_moments()
    ur, rr = onp.moments(ref)
    assert onp.rel_linf(rho, rr) <= 1e-5 and onp.rel_linf(u, ur) <= 1e-5
    s.close()


@pytest.mark.parametrize("boundary", ["reference", "periodic"])
@pytest.mark.parametrize("T", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("shape,tb_rows", [((40, 300), 16), ((37, 130), 256), ((64, 512), 7),
                                           ((9, 64), 3), ((50, 257), 20), ((21, 8), 5)])
def test_tb_small(boundary, T, shape, tb_rows):
    _run_case(shape[0], shape[1], T, boundary, tb_rows, tb_max=6)  # one launch of depth T


@pytest.mark.parametrize("T,tb_max", [(5, 4), (6, 4), (7, 6), (9, 6), (10, 6), (15, 6), (5, 2), (3, 1)])
def test_tb_split_depths(T, tb_max):
    """inner_timesteps beyond the deepest launch in use are split into balanced launches
    (cap 4: 5=3+2, 6=3+3; cap 6: 7=4+3, 9=6+3, 10=6+4, 15=6+6+3 - depth 5 is avoided when more launches follow;
    cap 2: 5=2+2+1; cap 1: the T=1 kernel only)."""
    _run_case(48, 320, T, "reference", 16, n_outer=2, tb_max=tb_max)
    _run_case(48, 320, T, "periodic", 16, n_outer=2, tb_max=tb_max)


def test_tb_cylinder():
    blocked = onp.blocked_cylinder(400, 800)
    surr = onp.ambient(100, [0, 0.3])
    cells = onp.initial_cells(blocked, surr)
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    s = Solver(400, 800, 0.01, inner_timesteps=4, boundary="reference")
    s.set_tuning(4, 0)  # pin the depth (automatic tuning depends on the grid size)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    s.fused_collide_stream(250)  # 1000 updates
    ref = oc.run(cells.copy(), blocked, 1000, 0.01, surr)
    assert np.array_equal(s.read_cells(), ref)
    s.close()


def test_tb_equals_t1_on_large_periodic():
    """Size-independent property at a size the oracle would take too long for: T=3 launches give
    the same bits as single updates (4096 x 4096, random discs, device-generated scene)."""
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    outs = []
    for T in (1, 3):
        s = Solver(4096, 4096, 0.05, inner_timesteps=T, boundary="periodic")
        s.set_surroundings(onp.ambient())
        s.init_equilibrium(100.0, (0.0, 0.1), amp=0.02, seed=3, global_rows=4096)
        s.init_blocked_discs(40, 37, seed=5, global_rows=4096)
        s.fused_collide_stream(12 // T)
        outs.append(s.read_cells(rows=(0, 4096), cols=(1000, 1400)))
        outs.append(s.read_cells(rows=(4000, 4096), cols=(0, 4096)))
        s.close()
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3])
    assert np.isfinite(outs[0]).all()


def test_random_soak():
    """tools/stress_parity.py: 300 random (shape, depth, launch cap, chunk rows, edge rule, omega,
    obstacle density, ambient flow) cases against the oracle, bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_parity.py"), "300", "7"],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "300 cases, 0 mismatching" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
