"""GPU: the fp16-storage / fp32-compute variant (lb_cuda_kernel.cu:152-179, `half_enable`;
BASELINE config 5) and the RGB frame (lb_cuda_kernel.cu:27-67,308-319)."""
import numpy as np
import pytest

from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc

pytestmark = pytest.mark.gpu


def cavity_scene(N=512, M=256):
    """A tall box with an open-topped box obstacle (the shape at the bottom-left of the
    reference's yeet.png; SURVEY.md section 7, hard part 3): ambient flow passes over its mouth."""
    b = np.zeros((N, M), bool)
    y0, y1, x0, x1, t = N // 2, N - 40, 40, M - 60, 6
    b[y0:y1, x0:x0 + t] = True       # left wall
    b[y0:y1, x1 - t:x1] = True       # right wall
    b[y1 - t:y1, x0:x1] = True       # floor
    return b


def r16(a):
    return a.astype(np.float16).astype(np.float32)


def emulate_fp16(cells, blocked, steps, omega, surr, boundary):
    """What the device does with fp16 storage and one update per launch: the stored state is
    fp16(g), g = collide(f) computed in fp32 from the fp16 values."""
    h = r16(cells)
    h = r16(oc.collide(h.copy(), blocked, omega))
    for _ in range(steps - 1):
        f = oc.stream_reflect(h, blocked, surr, boundary)
        h = r16(oc.collide(f, blocked, omega))
    return oc.stream_reflect(h, blocked, surr, boundary)


def test_fp16_storage_is_bit_exact_against_its_emulation():
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 96, 200
    blocked = cavity_scene(N, M)
    surr = onp.ambient()
    cells = onp.perturbed_cells(N, M, seed=2, amp=0.02)
    s = Solver(N, M, 0.2, boundary="reference", dtype=np.float16)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    assert np.array_equal(s.read_cells(), r16(cells))
    for n in (1, 2, 10):
        s.upload_cells(cells)
        s.fused_collide_stream(n)
        assert np.array_equal(s.read_cells(), emulate_fp16(cells, blocked, n, 0.2, surr, "reference")), n
    s.close()


def emulate_fp16_launches(cells, blocked, depths, omega, surr, boundary):
    """fp16 storage with LDS temporal blocking: every time level is rounded through fp16 on its
    way into the LDS ring, as a store between two single updates would round it (and as the
    reference rounds `newcells` on every inner timestep, lb_cuda_kernel.cu:292,361), so the result
    only depends on the NUMBER of updates, not on how they are cut into launches.
    depths = updates per launch after the initial collide-only launch."""
    return emulate_fp16(cells, blocked, 1 + sum(depths), omega, surr, boundary)


@pytest.mark.parametrize("boundary", ["reference", "periodic"])
def test_fp16_storage_with_temporal_blocking_is_bit_exact_against_its_emulation(boundary):
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 96, 200
    blocked = cavity_scene(N, M)
    surr = onp.ambient()
    cells = onp.perturbed_cells(N, M, seed=2, amp=0.02)
    for T, cap, depths in [(3, 3, [2, 3, 3]), (6, 6, [5, 6]), (4, 2, [2, 1, 2, 2])]:
        s = Solver(N, M, 0.2, inner_timesteps=T, boundary=boundary, dtype=np.float16)
        s.set_tuning(cap, 0)
        s.set_surroundings(surr)
        s.upload_blocked(blocked)
        s.upload_cells(cells)
        s.fused_collide_stream((1 + sum(depths)) // T)
        assert s.steps_done == 1 + sum(depths)
        want = emulate_fp16_launches(cells, blocked, depths, 0.2, surr, boundary)
        assert np.array_equal(s.read_cells(), want), (T, cap, boundary)
        s.close()


@pytest.mark.parametrize("T,shape", [(1, (512, 256)), (3, (512, 256)), (1, (2048, 512)), (6, (2048, 512))])
def test_fp16_cavity_within_1e2_of_fp32(T, shape):
    """BASELINE config 5, tolerance 1e-2 (latticeboltzmann.py:18-19: rtol of the fp16 build).
    Temporal blocking rounds every time level through fp16 like single updates do, so T = 3 gives
    the T = 1 bits.  fp16 rounding error accumulates with the step count (measured on MI355X at
    96 / 240 / 480 steps, u: 8e-3 / 1.8e-2 / 4.2e-2): the reference's README (:94) already calls
    fp16 unphysical; the 1e-2 bound is checked at 96 steps, the horizon where it holds.  2048 x 512
    is the tall box SURVEY.md section 8(d) names for config 5 (there: rho 2.1e-3, u 8.1e-3)."""
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    (N, M), steps = shape, 96
    blocked = cavity_scene(N, M)
    surr = onp.ambient()
    cells = onp.initial_cells(blocked, surr)
    s = Solver(N, M, 0.3, inner_timesteps=T, boundary="reference", dtype=np.float16)
    s.set_tuning(6, 0)  # pin the LDS kernel (automatic tuning uses the plain kernel on tiny grids)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    s.fused_collide_stream(steps // T)
    u, rho = s.read_moments()
    s.close()
    ref = oc.run(cells.copy(), blocked, steps, 0.3, surr, "reference")
    ur, rr = onp.moments(ref)
    assert onp.rel_linf(rho, rr) <= 1e-2
    assert onp.rel_linf(u, ur) <= 1e-2


def test_render_matches_the_reference_colour_map():
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    N, M = 100, 200
    blocked = onp.blocked_cylinder(N, M)
    # density in [0, 1] so that value is not saturated everywhere, and a visible velocity field
    surr = onp.ambient(0.8, [0.0005, 0.0008])
    cells = onp.perturbed_cells(N, M, seed=1, amp=0.3, p=0.6, u=[0.0003, -0.0006])
    s = Solver(N, M, 0.5, boundary="reference")
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    s.fused_collide_stream(5)
    frame = np.zeros((N, M, 3), np.uint8)
    s.fused_collide_stream_display(frame)  # 6th update + render, as latticeboltzmann.py:187-191
    s.sync()
    again = s.render()
    u, p = s.read_moments()
    s.close()
    assert np.array_equal(frame, again)
    want = onp.cuda_kernel_frame(u, p, blocked)  # oracle/: restates lb_cuda_kernel.cu:27-67,308-319
    diff = np.abs(frame.astype(int) - want.astype(int))
    assert (frame[blocked] == 0).all()
    assert diff.max() <= 1, f"max channel difference {diff.max()}"
    assert frame.std() > 10  # a real picture, not a constant


def _launch_depths(T, cap, n_outer, tiny):
    """The library's launch schedule after the initial collide-only launch (lbm_api.hip:
    outer_iteration / advance): balanced launches no deeper than cap; tiny grids use depth 1."""
    depths, first = [], True
    for _ in range(n_outer):
        rem = T - 1 if first else T
        first = False
        if rem <= 0:
            continue
        while rem > 0:
            nl = -(-rem // cap)
            d = -(-rem // nl)
            if d == 5 and rem > 5 and cap >= 6:
                d = 6  # depth 5 is the slow one: 10 = 6 + 4
            depths += [1] * d if tiny else [d]
            rem -= d
    return depths


def test_fp16_random_schedules_are_bit_exact_against_the_emulation():
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    rng = np.random.default_rng(5)
    for case in range(40):
        N, M = int(rng.integers(2, 90)), int(rng.integers(2, 300))
        T, cap, n_outer = int(rng.integers(1, 8)), int(rng.integers(1, 7)), int(rng.integers(1, 4))
        boundary = str(rng.choice(["reference", "periodic"]))
        blocked = rng.random((N, M)) < float(rng.choice([0.0, 0.05, 0.3]))
        cells = onp.perturbed_cells(N, M, seed=int(rng.integers(1 << 30)), amp=0.03)
        surr = onp.ambient()
        s = Solver(N, M, 0.25, inner_timesteps=T, boundary=boundary, dtype=np.float16)
        s.set_tuning(cap, int(rng.choice([0, 5, 40])))
        s.set_surroundings(surr)
        s.upload_blocked(blocked)
        s.upload_cells(cells)
        s.fused_collide_stream(n_outer)
        want = emulate_fp16_launches(cells, blocked, _launch_depths(T, cap, n_outer, M < 8 or N < 16),
                                     0.25, surr, boundary)
        got = s.read_cells()
        s.close()
        assert np.array_equal(got, want, equal_nan=True), (case, N, M, T, cap, n_outer, boundary)
