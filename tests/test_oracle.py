"""CPU: pins of the oracle.  (1) what the reference itself pins: its import-time asserts
(latticeboltzmann.py:28-29,33-34,93-94), the known-answer `surroundings` vector and the obstacle
masks of its two scene images; (2) the C restatement against the numpy restatement, bit for bit;
(3) the committed golden fields; (4) physical invariants."""
import hashlib
import os

import numpy as np
import pytest

from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_lattice_asserts():
    onp.lattice_asserts()
    assert [tuple(v) for v in onp.e] == [(-1, -1), (0, -1), (1, -1), (-1, 0), (0, 0), (1, 0),
                                         (-1, 1), (0, 1), (1, 1)]
    # opposite direction of k is 8 - k (latticeboltzmann.py:33)
    assert all((onp.e[k] == -onp.e[8 - k]).all() for k in range(9))


def test_reference_conservation_asserts_and_known_answer():
    s = onp.ambient()  # runs the two asserts of latticeboltzmann.py:93-94
    want = [2.027778, 8.111111, 2.027778, 10.944444, 43.777778, 10.944444, 3.694444, 14.777778,
            3.694444]
    assert s.dtype == np.float32
    np.testing.assert_allclose(s, want, rtol=2e-7)
    np.testing.assert_allclose(s.sum(), 100, rtol=1e-6)
    assert np.array_equal(oc.equilibrium(0.0, 0.1, 100.0), s)  # C twin: same bits


def test_reference_masks():
    """Obstacle masks of the reference's scene images (made by tests/golden/make_golden.py with
    latticeboltzmann.py:40-41): shape, popcount, hash."""
    z = np.load(os.path.join(GOLD, "masks.npz"))
    for key, shape, pop, sha in (("yeet", (1355, 3782), 22429, "3c9ad3802cbea8e5"),
                                 ("yeet2", (4264, 12032), 67287, "ac7b163ef157e3f8")):
        assert tuple(z[key + "_shape"]) == shape
        bits = z[key + "_bits"]
        assert hashlib.sha256(bits.tobytes()).hexdigest()[:16] == sha
        b = np.unpackbits(bits)[:shape[0] * shape[1]].reshape(shape).astype(bool)
        assert int(b.sum()) == pop == int(z[key + "_popcount"])
        # no blocked pixel on any border
        assert not (b[0].any() or b[-1].any() or b[:, 0].any() or b[:, -1].any())


def test_cylinder_scene():
    b = onp.blocked_cylinder(400, 800)  # latticeboltzmann.jl:25-32
    assert b.shape == (400, 800) and b.dtype == bool and int(b.sum()) == 1961
    assert b[200, 200] and b[200, 225] and not b[200, 226] and not b[0, 0]


def test_stream_boundary_indexing():
    """latticeboltzmann.py:115-120 cell by cell on a tiny grid: column 0 / row 0 take
    surroundings[k] for dx / dy = +1; the empty `-1:0` slices leave the last column / row with
    their own value."""
    N, M = 4, 5
    g = np.arange(N * M * 9, dtype=np.float32).reshape(N, M, 9) + 1000
    surr = -np.arange(1, 10, dtype=np.float32)
    f = g.copy()
    onp.stream(f, surr)
    for k, (dy, dx) in enumerate(onp.e):
        for y in range(N):
            for x in range(M):
                ys, xs = y - dy, x - dx
                if 0 <= ys < N and 0 <= xs < M:
                    want = g[ys, xs, k]
                elif (dx == 1 and x == 0) or (dy == 1 and y == 0):
                    want = surr[k]
                else:
                    want = g[y, x, k]
                assert f[y, x, k] == want, (k, y, x)
    # and the C twin agrees
    blocked = np.zeros((N, M), bool)
    assert np.array_equal(oc.stream_reflect(g, blocked, surr), f)


def test_reflect_is_k_to_8_minus_k():
    rng = np.random.default_rng(0)
    c = rng.random((6, 7, 9), dtype=np.float32)
    b = rng.random((6, 7)) < 0.3
    r = c.copy()
    onp.reflect(r, b)
    assert np.array_equal(r[b], c[b][:, ::-1]) and np.array_equal(r[~b], c[~b])


@pytest.mark.parametrize("boundary", ["reference", "periodic"])
def test_c_restatement_is_bit_identical_to_numpy(boundary):
    rng = np.random.default_rng(3)
    N, M = 57, 91
    blocked = onp.blocked_cylinder(N, M) | (rng.random((N, M)) < 0.02)
    cells = onp.perturbed_cells(N, M, seed=1, amp=0.05)
    a, b = cells.copy(), cells.copy()
    onp.run(a, blocked, 60, 0.05, boundary=boundary)
    oc.run(b, blocked, 60, 0.05, onp.ambient(), boundary)
    assert np.array_equal(a, b)
    ua, pa = onp.moments(a)
    ub, pb = oc.moments(b)
    assert np.array_equal(ua, ub) and np.array_equal(pa, pb)


GOLDEN_FIELDS = [("cyl64x128_reference", "reference"), ("cyl64x128_vortex", "reference"),
                 ("rand40x72_omega0p6", "reference"), ("rand48x80_periodic", "periodic")]


def load_golden(tag):
    """(z, N, M, blocked, surroundings, omega, initial cells) of a committed fixture."""
    z = np.load(os.path.join(GOLD, f"fields_{tag}.npz"))
    N, M = (int(v) for v in z["shape"])
    blocked = np.unpackbits(z["blocked_bits"])[:N * M].reshape(N, M).astype(bool)
    surr, omega = z["surroundings"], float(z["omega"])
    cells = z["cells_0"].copy() if "cells_0" in z.files else onp.initial_cells(blocked, surr)
    return z, N, M, blocked, surr, omega, cells


@pytest.mark.parametrize("tag,boundary", GOLDEN_FIELDS)
def test_golden_fields(tag, boundary):
    """The restatements (numpy and C) against fields produced by the reference's OWN numpy lines
    (tests/golden/ref_extract.py executed latticeboltzmann.py:16-35,80-95,98,107-123,142,180-183
    from the reference file, in the order of latticeboltzmann.jl:107-120), bit for bit."""
    z, N, M, blocked, surr, omega, cells = load_golden(tag)
    assert str(z["source"]).startswith("reference text")
    ccells = cells.copy()
    done = 0
    for c in sorted(int(k.split("_")[1]) for k in z.files if k.startswith("rho_")):
        onp.run(cells, blocked, c - done, omega, surr, boundary)
        oc.run(ccells, blocked, c - done, omega, surr, boundary)
        done = c
        for got in (cells, ccells):
            u, p = onp.moments(got)
            assert np.array_equal(p, z[f"rho_{c}"]) and np.array_equal(u, z[f"u_{c}"])
            sha = np.frombuffer(hashlib.sha256(got.tobytes()).digest(), np.uint8)
            assert np.array_equal(sha, z[f"cells_sha_{c}"])


def cyl400x800_digests():
    """sha256 digests of BASELINE config 1's own scene (400 x 800 cylinder, latticeboltzmann.jl:25-32)
    after 1 / 100 / 1000 steps of the reference's own numpy lines (make_golden.py `sha_only`)."""
    z = np.load(os.path.join(GOLD, "sha_cyl400x800_reference.npz"))
    assert str(z["source"]).startswith("reference text") and tuple(z["shape"]) == (400, 800)
    assert float(z["omega"]) == onp.OMEGA and np.array_equal(z["surroundings"], onp.ambient())
    return z


def sha_of(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def test_config1_cylinder_digests():
    """BASELINE config 1 pinned directly: the C restatement on the 400 x 800 cylinder for 1000 steps
    against digests of the reference's own lines (the numpy restatement runs the first 100)."""
    z = cyl400x800_digests()
    blocked = onp.blocked_cylinder(400, 800)
    assert np.array_equal(sha_of(np.packbits(blocked)), z["blocked_sha"])
    surr = onp.ambient()
    c = onp.initial_cells(blocked, surr)
    n = c.copy()
    done = 0
    for k in (1, 100, 1000):
        oc.run(c, blocked, k - done, onp.OMEGA, surr, "reference")
        if k <= 100:
            onp.run(n, blocked, k - done, onp.OMEGA, surr, "reference")
            assert np.array_equal(n, c)
        done = k
        u, p = oc.moments(c)
        assert np.array_equal(sha_of(c), z[f"cells_sha_{k}"])
        assert np.array_equal(sha_of(p), z[f"rho_sha_{k}"]) and np.array_equal(sha_of(u), z[f"u_sha_{k}"])


def test_float64_omega_is_what_the_fixtures_pin():
    """latticeboltzmann.py:98 builds omega as float64; a float32 omega (round 1's restatement)
    must NOT reproduce the fixtures, or they would not pin that line."""
    z, N, M, blocked, surr, omega, cells = load_golden("rand40x72_omega0p6")
    assert onp.make_omega(blocked, omega).dtype == np.float64
    om32 = onp.make_omega(blocked, omega).astype(np.float32)
    for _ in range(10):
        onp.step(cells, blocked, om32, surr, "reference")
    assert not np.array_equal(onp.moments(cells)[1], z["rho_10"])


@pytest.mark.skipif(not os.path.exists("/root/reference/latticeboltzmann.py"),
                    reason="build container only: the GPU box has no /root/reference")
@pytest.mark.parametrize("omega,u_amb,steps", [(onp.OMEGA, [0, 0.1], 150), (0.01, [0, 0.3], 150),
                                               (0.9, [0.02, 0.05], 80), (1.7, [0, 0.1], 40)])
def test_live_reference_text_equals_the_restatement(omega, u_amb, steps):
    """Executes the reference's own numpy lines (ref_extract) beside the restatement on scenes the
    committed fixtures do not hold: same bits for cells, rho and u."""
    import sys
    sys.path.insert(0, GOLD)
    import ref_extract
    rng = np.random.default_rng(11)
    N, M = 37, 53
    blocked = onp.blocked_cylinder(N, M) | (rng.random((N, M)) < 0.04)
    start = onp.perturbed_cells(N, M, seed=2, amp=0.03, u=u_amb)
    ref = ref_extract.ReferenceNumpyPath(blocked, {"OMEGA": omega, "u_ambient": u_amb}, cells=start)
    assert ref.omega.dtype == np.float64 and ref.omega.shape == (N, M, 1)
    assert np.array_equal(ref.omega, onp.make_omega(blocked, omega))
    surr = onp.ambient(100, u_amb)
    assert np.array_equal(ref.surroundings, surr)
    a, c = start.copy(), start.copy()
    with np.errstate(all="ignore"):
        ref.step(steps)
        onp.run(a, blocked, steps, omega, surr)
    oc.run(c, blocked, steps, omega, surr)
    assert np.array_equal(a, ref.cells, equal_nan=True) and np.array_equal(c, ref.cells, equal_nan=True)
    ur, pr = ref.moments()
    ua, pa = onp.moments(a)
    assert np.array_equal(ua, ur, equal_nan=True) and np.array_equal(pa, pr, equal_nan=True)
    # the reference's own initial state (latticeboltzmann.py:142) as well
    ref0 = ref_extract.ReferenceNumpyPath(blocked, {"OMEGA": omega, "u_ambient": u_amb})
    assert np.array_equal(ref0.cells, onp.initial_cells(blocked, surr))


def test_mass_is_conserved_on_a_periodic_scene():
    rng = np.random.default_rng(1)
    N, M = 40, 60
    blocked = rng.random((N, M)) < 0.05
    cells = onp.perturbed_cells(N, M, seed=5, amp=0.02)
    m0 = cells.astype(np.float64).sum()
    onp.run(cells, blocked, 50, 0.3, boundary="periodic")
    assert abs(cells.astype(np.float64).sum() - m0) / m0 < 1e-6


def test_uniform_flow_is_a_fixed_point_without_obstacles():
    N, M = 32, 48
    surr = onp.ambient()
    blocked = np.zeros((N, M), bool)
    cells = onp.initial_cells(blocked, surr)
    onp.run(cells, blocked, 100)
    np.testing.assert_allclose(cells, np.broadcast_to(surr, cells.shape), rtol=2e-6)


def test_fp32_oracle_is_close_to_an_fp64_run():
    """Cross-check of the restatement's arithmetic (SURVEY.md section 4 d): the same algorithm in
    float64 stays within 1e-4 of the float32 oracle over 200 steps of the small cylinder."""
    N, M = 64, 128
    blocked = onp.blocked_cylinder(N, M)
    surr = onp.ambient()
    c32 = onp.initial_cells(blocked, surr)
    onp.run(c32, blocked, 200)
    e = onp.e.astype(np.float64)
    w = onp.w.astype(np.float64)
    c = np.broadcast_to(surr.astype(np.float64), (N, M, 9)).copy()
    om = np.where(blocked[..., None], 1.0, onp.OMEGA)
    for _ in range(200):
        p = c.sum(axis=2)
        u = c @ e / p[..., None]
        eu = u @ e.T
        eq = p[..., None] * w * (1 + 3 * eu + 4.5 * eu ** 2 - 1.5 * (u ** 2).sum(-1)[..., None])
        c = eq + om * (c - eq)
        onp.stream(c, surr.astype(np.float64))
        onp.reflect(c, blocked)
    u32, p32 = onp.moments(c32)
    p64 = c.sum(axis=2)
    u64 = c @ e / p64[..., None]
    assert onp.rel_linf(p32, p64) < 1e-4 and onp.rel_linf(u32, u64) < 1e-4


def test_cuda_kernel_frame_colour_map():
    """oracle.cuda_kernel_frame (lb_cuda_kernel.cu:27-67,308-319; parity-unpinned) against an
    independent HSV -> RGB (matplotlib's, the one the reference's numpy display uses,
    latticeboltzmann.py:105) and on hand-checkable pixels."""
    matplotlib = pytest.importorskip("matplotlib")
    import matplotlib.colors
    rng = np.random.default_rng(4)
    N, M = 24, 40
    u = (rng.random((N, M, 2), dtype=np.float32) - 0.5) * np.float32(0.003)
    p = rng.random((N, M), dtype=np.float32) * np.float32(1.2)
    blocked = rng.random((N, M)) < 0.1
    f = onp.cuda_kernel_frame(u, p, blocked)
    assert f.dtype == np.uint8 and f.shape == (N, M, 3) and (f[blocked] == 0).all()
    h = (np.arctan2(u[..., 1], u[..., 0]).astype(np.float64) + np.pi) / (2 * np.pi)
    s = np.clip(1000 * np.hypot(u[..., 0], u[..., 1]).astype(np.float64), 0, 1)
    v = np.clip(p.astype(np.float64), 0, 1)
    want = np.rint(255 * matplotlib.colors.hsv_to_rgb(np.stack([np.clip(h, 0, 1), s, v], -1)))
    want[blocked] = 0
    assert np.abs(f.astype(int) - want.astype(int)).max() <= 1
    # zero velocity: atan2(0, 0) + pi = pi -> hue pi, saturation 0 -> grey of the density
    g = onp.cuda_kernel_frame(np.zeros((1, 2, 2), np.float32), np.float32([[0.5, 2.0]]), np.zeros((1, 2), bool))
    assert g[0, 0].tolist() == [128, 128, 128] and g[0, 1].tolist() == [255, 255, 255]
