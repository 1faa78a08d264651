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


def test_numpy_display_variant():
    """latticeboltzmann.py:101-105: hue from the flow direction, saturation 100000 |u|, value =
    density, all clipped to [0, 1], * 255.99 truncated."""
    import numpy as np
    from latticeboltzmann_amd import latticeboltzmann as lb
    u = np.zeros((2, 3, 2), np.float32)
    p = np.ones((2, 3), np.float32)
    u[0, 1] = [0, 1e-3]      # along +col: atan2(0, +) = 0 -> h = 0.5 (cyan), saturated
    u[0, 2] = [0, -1e-3]     # along -col: atan2(0, -) = pi -> h = 1.0 == red
    u[1, 0] = [1e-3, 0]      # along +row: h = 0.75
    p[1, 1] = 0.5            # still fluid at half density: grey 127
    p[1, 2] = 0.0
    class V:
        frames = []
        def append_data(self, f):
            self.frames.append(f)
    v = V()
    f = lb.display(u, p, v)
    assert f.dtype == np.uint8 and f.shape == (2, 3, 3) and v.frames[0] is f
    assert f[0, 0].tolist() == [255, 255, 255]          # at rest, full density: white
    assert f[0, 1].tolist() == [0, 255, 255]
    assert f[0, 2].tolist() == [255, 0, 0]
    assert f[1, 0].tolist() == [127, 0, 255]
    assert f[1, 1].tolist() == [127, 127, 127]
    assert f[1, 2].tolist() == [0, 0, 0]
    # the numpy fallback of hsv_to_rgb agrees with matplotlib's
    import builtins, importlib
    hsv = np.random.default_rng(0).random((5, 7, 3))
    ref = lb._hsv_to_rgb(hsv)
    real_import = builtins.__import__
    def no_mpl(name, *a, **k):
        if name.startswith("matplotlib"):
            raise ImportError(name)
        return real_import(name, *a, **k)
    builtins.__import__ = no_mpl
    try:
        alt = lb._hsv_to_rgb(hsv)
    finally:
        builtins.__import__ = real_import
    assert np.allclose(alt, ref, atol=1e-12)
