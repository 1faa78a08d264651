"""GPU: LBM_BOUNDARY_CUDA_KERNEL (the reference's CUDA-kernel edge / gate / density semantics,
SURVEY.md 8 f3; never the default; PARITY-UNPINNED by the reference) against its oracle
oracle/lbm_numpy.cuda_kernel_step, bit for bit."""
import numpy as np
import pytest

from oracle import lbm_numpy as onp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(40, 130), (64, 64), (5, 2), (1, 64), (33, 200), (2, 65)])
@pytest.mark.parametrize("omega", [onp.OMEGA, 0.3])
def test_cuda_kernel_mode_matches_its_oracle(shape, omega):
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = shape
    rng = np.random.default_rng(N * 1000 + M)
    blocked = rng.random((N, M)) < 0.1
    cells = onp.perturbed_cells(N, M, seed=5, amp=0.05)
    surr = onp.ambient()
    s = Solver(N, M, omega, boundary="cuda_kernel", inner_timesteps=2)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    assert np.array_equal(s.read_cells(), cells)
    ref = cells
    done = 0
    for n in (1, 1, 3):
        s.fused_collide_stream(n)       # n outer iterations of 2 updates, all in the plain kernel
        for _ in range(2 * n):
            ref = onp.cuda_kernel_step(ref, blocked, surr, omega)
        done += 2 * n
        assert s.steps_done == done
        got = s.read_cells()
        assert np.array_equal(got, ref), f"max abs diff {np.abs(got - ref).max()}"
        u, rho = s.read_moments()
        ur, rr = onp.moments(ref)
        assert np.array_equal(rho, rr) and np.array_equal(u, ur)
    s.close()


def test_cuda_kernel_mode_is_single_domain_and_depth_1():
    from latticeboltzmann_amd import lb_hip_kernel as hk
    with pytest.raises(RuntimeError, match="single-domain"):
        hk.Solver(32, 64, 0.3, boundary="cuda_kernel", halo=2, top_edge=hk.YEDGE_SEAM)
    s = hk.Solver(32, 64, 0.3, boundary="cuda_kernel")
    s.set_surroundings(onp.ambient())
    s.upload_cells(onp.perturbed_cells(32, 64, seed=1))
    s.fused_collide_stream(1)
    with pytest.raises(RuntimeError, match="depth 1"):
        s.advance_rows(0, 32, 2, True)
    s.close()


def test_cuda_kernel_mode_differs_from_the_default_where_survey_says():
    """A smoke check that the mode is really on: after a few steps of a cylinder scene the two
    edge rules disagree at the right / bottom edge and agree (to the fudge) in the interior."""
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 64, 128
    blocked = onp.blocked_cylinder(N, M)
    surr = onp.ambient()
    cells = onp.initial_cells(blocked, surr)
    out = {}
    for b in ("reference", "cuda_kernel"):
        s = Solver(N, M, onp.OMEGA, boundary=b)
        s.set_surroundings(surr)
        s.upload_blocked(blocked)
        s.upload_cells(cells)
        s.fused_collide_stream(30)
        out[b] = s.read_moments()[1]
        s.close()
    assert not np.array_equal(out["reference"], out["cuda_kernel"])
    assert onp.rel_linf(out["cuda_kernel"][:8, 100:], out["reference"][:8, 100:]) < 1e-3
