"""GPU parity: the HIP path (through the C-ABI, via the ctypes layer) against the CPU oracle on
identical seeded scenes.  fp32 results are compared bit for bit first (the kernels restate numpy's
operation order) and, as the north star states it, to relative L-inf <= 1e-5 on rho and u."""
import numpy as np
import pytest

from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc

pytestmark = pytest.mark.gpu

TOL = 1e-5  # BASELINE.json north_star: relative L-inf after 1000 fp32 steps


def _solver(N, M, blocked, cells, omega=onp.OMEGA, boundary="reference", surr=None, **kw):
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    s = Solver(N, M, omega, boundary=boundary, **kw)
    s.set_surroundings(onp.ambient() if surr is None else surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    return s


def _compare(s, ref_cells, exact=True):
    got = s.read_cells()
    u, rho = s.read_moments()
    ur, rr = onp.moments(ref_cells)
    if exact:
        assert np.array_equal(got, ref_cells), f"max abs diff {np.abs(got - ref_cells).max()}"
        assert np.array_equal(rho, rr) and np.array_equal(u, ur)
    assert onp.rel_linf(rho, rr) <= TOL
    assert onp.rel_linf(u, ur) <= TOL


@pytest.mark.parametrize("boundary", ["reference", "periodic"])
@pytest.mark.parametrize("shape", [(37, 130), (64, 64), (5, 2), (33, 200), (2, 65), (1, 64), (1, 2), (3, 129)])
def test_small_random_scene(boundary, shape):
    N, M = shape
    rng = np.random.default_rng(N * 1000 + M)
    blocked = rng.random((N, M)) < 0.1
    cells = onp.perturbed_cells(N, M, seed=7, amp=0.05)
    s = _solver(N, M, blocked, cells, omega=0.3, boundary=boundary)
    assert np.array_equal(s.read_blocked(), blocked)
    _compare(s, cells)  # state right after upload
    ref = cells.copy()
    done = 0
    for n in (1, 1, 2, 9):
        s.fused_collide_stream(n)
        done += n
        oc.run(ref, blocked, n, 0.3, onp.ambient(), boundary)
        assert s.steps_done == done
        _compare(s, ref)
    # ragged and empty windows
    assert s.read_cells(rows=(0, 0)).shape == (0, M, 9)
    assert s.read_cells(cols=(M - 1, M - 1)).shape == (N, 0, 9)
    assert np.array_equal(s.read_cells(rows=(N - 1, N), cols=(M - 1, M)), ref[N - 1:, M - 1:])
    u, rho = s.read_moments(rows=(0, 1), cols=(0, min(M, 3)))
    ur, rr = onp.moments(ref[:1, :min(M, 3)])
    assert np.array_equal(rho, rr) and np.array_equal(u, ur)
    s.fused_collide_stream(0)  # zero iterations: a no-op
    assert s.steps_done == done
    s.close()


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize("tb_max_depth", [1, 6])
def test_cylinder_1000_steps(tb_max_depth):
    """BASELINE config 1 scene: 400x800 cylinder, OMEGA=1e-14, rho=100, u=[0,0.1], 1000 steps, with
    the plain kernel (launch depth 1) and the LDS kernel (depth 6).  Compared with the oracle bit for
    bit AND with tests/golden/sha_cyl400x800_reference.npz: digests of cells / rho / u after 1, 100
    and 1000 steps of the reference's OWN numpy lines on this scene (tests/golden/make_golden.py)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sha_cyl400x800_reference.npz"))
    assert str(z["source"]).startswith("reference text")
    blocked = onp.blocked_cylinder(400, 800)
    assert np.array_equal(_sha(np.packbits(blocked)), z["blocked_sha"])
    surr = onp.ambient()
    assert np.array_equal(surr, z["surroundings"]) and float(z["omega"]) == onp.OMEGA
    cells = onp.initial_cells(blocked, surr)
    s = _solver(400, 800, blocked, cells)
    s.set_tuning(tb_max_depth, 0)
    ref = cells.copy()
    done = 0
    for chunk in (1, 99, 900):
        s.fused_collide_stream(chunk)
        oc.run(ref, blocked, chunk, onp.OMEGA, surr)
        done += chunk
        _compare(s, ref)
        u, rho = s.read_moments()
        assert np.array_equal(_sha(s.read_cells()), z[f"cells_sha_{done}"])
        assert np.array_equal(_sha(rho), z[f"rho_sha_{done}"]) and np.array_equal(_sha(u), z[f"u_sha_{done}"])
    s.close()


def test_cylinder_vortex_variant():
    """latticeboltzmann.jl:27-29 variant: OMEGA=0.01, u=[0,0.3] (unsteady wake)."""
    blocked = onp.blocked_cylinder(400, 800)
    surr = onp.ambient(100, [0, 0.3])
    cells = onp.initial_cells(blocked, surr)
    s = _solver(400, 800, blocked, cells, omega=0.01, surr=surr)
    ref = cells.copy()
    s.fused_collide_stream(1000)
    oc.run(ref, blocked, 1000, 0.01, surr)
    _compare(s, ref)
    s.close()


def test_numpy_oracle_agrees_too():
    """Same check against the numpy restatement itself (not only its C twin), 100 steps."""
    blocked = onp.blocked_cylinder(100, 200)
    surr = onp.ambient()
    cells = onp.perturbed_cells(100, 200, seed=3)
    s = _solver(100, 200, blocked, cells)
    ref = cells.copy()
    s.fused_collide_stream(100)
    onp.run(ref, blocked, 100)
    _compare(s, ref)
    s.close()


def test_zero_density_and_non_finite_cells():
    """np.nan_to_num semantics (latticeboltzmann.py:183): rho == 0 gives u = 0 (0/0 -> nan -> 0),
    and +-inf becomes +-FLT_MAX.  Scene: the .jl draft's start (latticeboltzmann.jl:94: zero
    populations inside obstacles) plus a few cells with zero density but non-zero momentum."""
    N, M = 40, 130
    blocked = onp.blocked_cylinder(N, M)
    surr = onp.ambient()
    cells = onp.initial_cells(blocked, surr)
    cells[blocked] = 0
    cells[5, 7] = [1, 0, 0, 0, 0, 0, 0, 0, -1]
    cells[30, 100] = [0, 2, 0, 0, 0, 0, 0, -2, 0]
    for T in (1, 3):
        from latticeboltzmann_amd.lb_hip_kernel import Solver
        s = Solver(N, M, 0.5, inner_timesteps=T, boundary="reference")
        s.set_tuning(6, 0)  # pin the LDS kernel (automatic tuning uses the plain kernel on tiny grids)
        s.set_surroundings(surr)
        s.upload_blocked(blocked)
        s.upload_cells(cells)
        ref = cells.copy()
        with np.errstate(all="ignore"):
            for _ in range(2):
                s.fused_collide_stream(1)
                oc.run(ref, blocked, T, 0.5, surr)
                got = s.read_cells()
                assert np.array_equal(got, ref, equal_nan=True)
        refn = cells.copy()
        with np.errstate(all="ignore"):
            onp.run(refn, blocked, 2 * T, 0.5)
        assert np.array_equal(refn, ref, equal_nan=True)  # the C twin follows numpy here too
        s.close()


@pytest.mark.parametrize("T", [1, 2, 4])
def test_extreme_magnitudes_take_the_exact_division_path(T):
    """The packed kernels divide on a shared-reciprocal fast path that is exact only for ordinary
    magnitudes and must hand every other cell to the IEEE division: regions scaled to 1e-30 and
    1e+25, mixed signs, denormal-range moments and exact zeros, against the oracle bit for bit."""
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 48, 640
    rng = np.random.default_rng(99)
    blocked = rng.random((N, M)) < 0.03
    cells = onp.perturbed_cells(N, M, seed=8, amp=0.2)
    cells[:, 100:200] *= np.float32(1e-30)
    cells[:, 200:260] *= np.float32(1e-38)      # denormal populations
    cells[:, 300:400] *= np.float32(1e25)
    cells[10:20, 420:500] *= rng.choice(np.float32([-1, 1]), size=(10, 80, 9))
    cells[30:35, 520:600] = 0
    cells[40, 610] = [1e-20, 0, 0, 0, 1, 0, 0, 0, -1e-20]
    surr = onp.ambient()
    s = Solver(N, M, 0.4, inner_timesteps=T, boundary="periodic")
    s.set_tuning(6, 0)  # pin the LDS kernel (automatic tuning uses the plain kernel on tiny grids)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    ref = cells.copy()
    with np.errstate(all="ignore"):
        for _ in range(3):
            s.fused_collide_stream(1)
            oc.run(ref, blocked, T, 0.4, surr, "periodic")
            got = s.read_cells()
            bad = np.argwhere(~((got == ref) | (np.isnan(got) & np.isnan(ref))))
            assert bad.size == 0, f"T={T}: {len(bad)} mismatches, first {bad[:5].tolist()}"
    s.close()


def test_error_behaviour_of_the_abi():
    """Nonzero status -> RuntimeError with lbm_last_error() (the analogue of pycuda raising):
    bad windows, null outputs, a partial upload after stepping, tuning out of range."""
    import ctypes
    from latticeboltzmann_amd import lb_hip_kernel as hk
    s = hk.Solver(24, 100, 0.3)
    s.set_surroundings(onp.ambient())
    cells = onp.perturbed_cells(24, 100, seed=1)
    s.upload_cells(cells[:10], 0)       # piecewise before the first step is fine
    s.upload_cells(cells[10:], 10)
    s.fused_collide_stream(2)
    with pytest.raises(RuntimeError, match="must cover all"):
        s.upload_cells(cells[:10], 0)
    with pytest.raises(RuntimeError, match="outside"):
        s.read_cells(rows=(0, 25))
    with pytest.raises(RuntimeError, match="columns"):
        s.read_moments(cols=(90, 101))
    with pytest.raises(RuntimeError, match="tb_max_depth"):
        s.set_tuning(7, 0)
    L = hk.load_library()
    assert L.lbm_read_cells(s._h, None, 0, 1, 0, 1) != 0 and b"null" in L.lbm_last_error()
    assert L.lbm_read_moments(s._h, None, None, 0, 1, 0, 1) != 0
    assert L.lbm_fused_collide_stream(s._h, -1) != 0
    assert L.lbm_fused_collide_stream(None, 1) != 0 and b"null solver" in L.lbm_last_error()
    # an update left open by step_rows(finish=0) blocks further stepping until it is finished
    s.step_rows(0, 10, False)
    with pytest.raises(RuntimeError, match="unfinished"):
        s.fused_collide_stream(1)
    with pytest.raises(RuntimeError, match="unfinished"):
        s.advance(2)
    s.step_rows(10, 14, True)
    assert s.steps_done == 3
    cells3 = oc.run(cells.copy(), np.zeros((24, 100), bool), 3, 0.3, onp.ambient())
    assert np.array_equal(s.read_cells(), cells3)
    s.upload_cells(cells)
    s.fused_collide_stream(2)
    # the solver is still usable and correct after the rejected calls
    ref = oc.run(cells.copy(), np.zeros((24, 100), bool), 2, 0.3, onp.ambient())
    assert np.array_equal(s.read_cells(), ref)
    s.upload_cells(cells)               # a full upload restarts
    assert s.steps_done == 0
    s.close()
    with pytest.raises(RuntimeError):   # a second solver with an absurd size: NOMEM, not a crash
        hk.Solver(1 << 20, 1 << 20, 0.3)
    # ... and the failed allocation leaves no stale HIP error behind for the next solver
    s2 = hk.Solver(8, 64, 0.3)
    s2.upload_blocked(np.zeros((8, 64), bool))
    s2.upload_cells(onp.perturbed_cells(8, 64, seed=2))
    s2.fused_collide_stream(2)
    assert np.isfinite(s2.read_cells()).all()
    s2.close()


@pytest.mark.parametrize("T", [1, 6])
@pytest.mark.parametrize("tag,boundary", [("cyl64x128_reference", "reference"),
                                          ("cyl64x128_vortex", "reference"),
                                          ("rand40x72_omega0p6", "reference"),
                                          ("rand48x80_periodic", "periodic")])
def test_committed_golden_fields(tag, boundary, T):
    """The HIP path against the committed fixtures alone (tests/golden/fields_*.npz: rho, u and a
    sha256 of the cells after 1 / 10 / 100 / 1000 (/ 2000) steps), plain kernel and LDS kernel.
    The fixtures were produced by the reference's OWN numpy lines, executed from its file by
    tests/golden/ref_extract.py -- this is the test that pins the device to the reference's text,
    float64 omega (latticeboltzmann.py:98) included."""
    import hashlib
    import os
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"fields_{tag}.npz"))
    assert str(z["source"]).startswith("reference text")
    N, M = (int(v) for v in z["shape"])
    blocked = np.unpackbits(z["blocked_bits"])[:N * M].reshape(N, M).astype(bool)
    surr, omega = z["surroundings"], float(z["omega"])
    cells = z["cells_0"].copy() if "cells_0" in z.files else onp.initial_cells(blocked, surr)
    s = Solver(N, M, omega, boundary=boundary)
    s.set_tuning(T, 0)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    done = 0
    for c in sorted(int(k.split("_")[1]) for k in z.files if k.startswith("rho_")):
        s.advance(c - done)
        done = c
        u, p = s.read_moments()
        assert np.array_equal(p, z[f"rho_{c}"]) and np.array_equal(u, z[f"u_{c}"]), (tag, c)
        sha = np.frombuffer(hashlib.sha256(s.read_cells().tobytes()).digest(), np.uint8)
        assert np.array_equal(sha, z[f"cells_sha_{c}"]), (tag, c)
    s.close()


@pytest.mark.parametrize("T", [1, 2, 5])
@pytest.mark.parametrize("omega", [0.5, 1.0, float(np.float32(0.3)), 0.3, 1e-14, 1.0 - 2.0 ** -40])
def test_omega_is_a_double_as_in_the_reference(omega, T):
    """latticeboltzmann.py:98 makes omega float64, so `cells *= omega` is a double product rounded
    once.  The device takes the plain fp32 multiply when omega is a float32 value (same bits,
    provably) and the double product otherwise: both against the oracle, bit for bit, on a scene
    where the product decides the last bit from the first step on (large f - eq)."""
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 40, 200
    rng = np.random.default_rng(5)
    blocked = rng.random((N, M)) < 0.05
    cells = onp.perturbed_cells(N, M, seed=4, amp=0.05)
    cells *= (1 + 0.3 * (rng.random(cells.shape, dtype=np.float32) - 0.5))  # far from equilibrium
    surr = onp.ambient()
    s = Solver(N, M, omega, inner_timesteps=T, boundary="reference")
    s.set_tuning(6, 0)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    ref = cells.copy()
    for _ in range(3):
        s.fused_collide_stream(1)
        oc.run(ref, blocked, T, omega, surr)
        assert np.array_equal(s.read_cells(), ref)
    refn = onp.run(cells.copy(), blocked, 3 * T, omega, surr)
    assert np.array_equal(refn, ref)
    s.close()


@pytest.mark.parametrize("T", [1, 2, 5])
@pytest.mark.parametrize("omega", [1e-14, 2.0 ** -34, 2.0 ** -34 * (1 + 2.0 ** -20), 0.0, -1e-14])
def test_tiny_omega_shortcut_is_exact(omega, T):
    """For |omega| <= 2^-34 (the reference's OMEGA = 1e-14) collide() returns eq without the
    subtract / multiply / add wherever that is provably the same bits (lbm_device.h:
    relax_is_identity) and falls back to the general path per wave elsewhere.  Scene: ordinary
    cells next to everything the proof excludes -- negative populations, |u| > 0.29, zero density,
    magnitudes beyond 2^+-100, non-finite cells, obstacles -- against the oracle, bit for bit."""
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 48, 704
    rng = np.random.default_rng(17)
    blocked = rng.random((N, M)) < 0.02
    cells = onp.perturbed_cells(N, M, seed=9, amp=0.1)
    cells[:, 64:128] = onp.getEquilibrium(                      # fast flow: c = 1.5 |u|^2 > 1/8
        np.broadcast_to(np.float32([0.25, 0.3]), (N, 64, 2)), np.full((N, 64), 80, np.float32))
    cells[:, 128:192] = onp.getEquilibrium(                     # just around the c = 1/8 threshold
        np.stack([np.zeros((N, 64), np.float32),
                  np.float32(0.2886) + np.float32(1e-4) * rng.standard_normal((N, 64)).astype(np.float32)], -1),
        np.full((N, 64), 100, np.float32))
    cells[:, 192:256] *= rng.choice(np.float32([-1, 1]), size=(N, 64, 9), p=[0.1, 0.9])  # negative f_k
    cells[:, 256:320] *= np.float32(1e-32)                      # p < 2^-100
    cells[:, 320:384] *= np.float32(1e31)                       # p > 2^100
    cells[5:9, 384:448] = 0                                     # zero density
    cells[20, 400] = [np.inf, 0, 0, 0, 1, 0, 0, 0, 0]
    cells[21, 401] = [np.nan, 1, 1, 1, 1, 1, 1, 1, 1]
    cells[:, 448:512, 0] = -0.0                                 # a negative zero is "not ordinary"
    surr = onp.ambient()
    s = Solver(N, M, omega, inner_timesteps=T, boundary="periodic")
    s.set_tuning(6, 0)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    ref = cells.copy()
    with np.errstate(all="ignore"):
        for _ in range(3):
            s.fused_collide_stream(1)
            oc.run(ref, blocked, T, omega, surr, "periodic")
            got = s.read_cells()
            bad = np.argwhere(~((got == ref) | (np.isnan(got) & np.isnan(ref))))
            assert bad.size == 0, f"omega={omega} T={T}: {len(bad)} mismatches, first {bad[:5].tolist()}"
        refn = cells.copy()
        onp.run(refn, blocked, 3 * T, omega, surr, "periodic")
    assert np.array_equal(refn, ref, equal_nan=True)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1, 4, 6])
def test_common_path_window_ends_and_tiny_momentum(T):
    """The LDS kernel's common path (lbm_device.h: moments2_scaled / ordinary2 / collide2_to) takes
    the quotients u = m / p from a numerator-scaled refinement with no test of the operands'
    exponents, accepted by one vote: sign bits clear, 2^-60 <= p <= 2^39, 1.5|u|^2 <= 1/8.  Scene:
    densities either side of both window ends, momentum sums of 2^-100 .. 2^-149 (denormal) over
    p ~ 1 (only the rest population ordinary), u0 = 0 exactly, blocked cells holding values that are
    not ordinary (zero, negative, huge) next to ordinary free cells -- against the oracle, bit for bit."""
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 40, 768
    rng = np.random.default_rng(23)
    blocked = rng.random((N, M)) < 0.01
    cells = onp.perturbed_cells(N, M, seed=3, amp=0.1)
    for j, e in enumerate([-68, -67, -66, -65, 30, 31, 32, 33]):   # p = 100 * 2^e around 2^-60 and 2^39
        cells[:, 64 * j:64 * j + 40] *= np.float32(2.0) ** np.float32(e)
    t = rng.random((N, 64, 9)).astype(np.float32)                   # tiny momentum sums over p ~ 1
    t *= (np.float32(2.0) ** -rng.integers(100, 127, size=(N, 64, 1))).astype(np.float32)
    t[..., 4] = rng.uniform(0.5, 200, size=(N, 64)).astype(np.float32)
    cells[:, 512:576] = t
    sym = cells[:, 576:640].copy()                                  # u0 = 0 exactly: k <-> mirrored in y
    for k in range(9):
        ky = 3 * (k // 3) + (2 - k % 3)
        sym[..., ky] = cells[:, 576:640, k]
    cells[:, 576:640] = np.float32(0.5) * (sym + cells[:, 576:640])
    blocked[10:14, 640:704] = True                                  # blocked cells out of the ordinary
    cells[10, 640:704] = 0
    cells[11, 640:704] *= np.float32(-1)
    cells[12, 640:704] *= np.float32(1e30)
    cells[13, 640:704, 2] = -0.0
    surr = onp.ambient()
    s = Solver(N, M, 1e-14, inner_timesteps=T, boundary="periodic")
    s.set_tuning(6, 0)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    ref = cells.copy()
    with np.errstate(all="ignore"):
        for _ in range(3):
            s.fused_collide_stream(1)
            oc.run(ref, blocked, T, 1e-14, surr, "periodic")
            got = s.read_cells()
            bad = np.argwhere(~((got == ref) | (np.isnan(got) & np.isnan(ref))))
            assert bad.size == 0, f"T={T}: {len(bad)} mismatches, first {bad[:5].tolist()}"
        refn = cells.copy()
        onp.run(refn, blocked, 3 * T, 1e-14, surr, "periodic")
    assert np.array_equal(refn, ref, equal_nan=True)
    s.close()
