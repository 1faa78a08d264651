"""CPU: the host logic of the driver surface that needs no GPU."""
import numpy as np

from oracle import lbm_numpy as onp


def test_constants_and_equilibrium_match_the_oracle():
    from latticeboltzmann_amd import latticeboltzmann as lb
    assert np.array_equal(lb.w, onp.w) and np.array_equal(lb.e, onp.e)
    assert np.array_equal(lb.ambient(), onp.ambient())
    u = np.random.default_rng(0).standard_normal((5, 7, 2)).astype(np.float32) * 0.1
    p = np.full((5, 7), 100, np.float32)
    assert np.array_equal(lb.getEquilibrium(u, p), onp.getEquilibrium(u, p))
    assert np.array_equal(lb.cylinder_blocked(), onp.blocked_cylinder())


def test_stats_line_arithmetic():
    """latticeboltzmann.py:176: us per iteration, GLUPS, GB/s = GLUPS x 72."""
    from latticeboltzmann_amd import latticeboltzmann as lb
    line = lb.stats_line(2160, 3840, 1, 1.587, 999)  # 1000 iterations of a 4K grid in 1.587 s
    us, glups, gbps = (float(line.split(" / ")[i].split()[0]) for i in range(3))
    assert abs(us - 1587.0) < 1e-9
    assert abs(glups - 2160 * 3840 / 1e6 / 1.587) < 1e-9
    assert abs(gbps / glups - 72) < 1e-9
    assert "(iter 1000)" in line


def test_unknown_tuning_is_rejected():
    import pytest
    from latticeboltzmann_amd import latticeboltzmann as lb
    with pytest.raises(NotImplementedError):
        lb.Simulation(np.zeros((8, 8), bool), tune="RTX_2070")
