"""GPU: the allocation-free entry point lbm_fused_collide_stream_display_noalloc (no counterpart in
the reference; for callers that cannot call hipMalloc) against the existing
lbm_fused_collide_stream_display: same frames and same state, bit for bit, on two calls in a row,
the second on a smaller grid (both share the GPU's one static scratch array, which keeps the
first call's bytes).  Collected last on purpose (the round's end runs pytest -x)."""
import numpy as np
import pytest

from oracle import lbm_numpy as onp

pytestmark = pytest.mark.gpu


def _pair(N, M, T, boundary, seed):
    from latticeboltzmann_amd import lb_hip_kernel as hk
    rng = np.random.default_rng(seed)
    blocked = onp.blocked_cylinder(N, M) | (rng.random((N, M)) < 0.02)
    cells = onp.perturbed_cells(N, M, seed=seed, amp=0.05)
    out = []
    for _ in range(2):
        s = hk.Solver(N, M, 0.3, inner_timesteps=T, boundary=boundary)
        s.set_tuning(6, 0)
        s.set_surroundings(onp.ambient())
        s.upload_blocked(blocked)
        s.upload_cells(cells)
        out.append(s)
    return out


def test_noalloc_display_equals_display_two_calls_in_a_row():
    from latticeboltzmann_amd import lb_hip_kernel as hk
    # first a larger frame, then a smaller one that overwrites only the head of the scratch
    for (N, M, T, boundary, seed) in ((300, 500, 1, "reference", 1), (96, 130, 3, "periodic", 2)):
        a, b = _pair(N, M, T, boundary, seed)
        fa = hk.pagelocked_empty((N, M, 3))
        fb = hk.pagelocked_empty((N, M, 3))
        for it in range(3):
            fa[:] = 7
            fb[:] = 9
            a.fused_collide_stream_display(fa)           # existing entry point (allocates its staging)
            b.fused_collide_stream_display_noalloc(fb)   # static scratch, no allocation
            a.sync()
            b.sync()
            assert np.array_equal(fa, fb), (N, M, it)
            assert fb.any() and np.array_equal(b.render(), fb)
            assert a.steps_done == b.steps_done == (it + 1) * T
        # frame == NULL is the plain step in both
        a.fused_collide_stream_display(None)
        b.fused_collide_stream_display_noalloc(None)
        assert np.array_equal(a.read_cells(), b.read_cells())
        hk.release_pagelocked(fa)
        hk.release_pagelocked(fb)
        a.close()
        b.close()


def test_noalloc_display_rejects_what_it_cannot_serve():
    from latticeboltzmann_amd import lb_hip_kernel as hk
    # a frame beyond the static scratch: clear error, state untouched
    N, M = 3400, 3400
    assert N * M * 3 > hk.NOALLOC_FRAME_MAX_BYTES
    s = hk.Solver(N, M, 0.3, boundary="periodic")
    s.set_surroundings(onp.ambient())
    s.init_equilibrium(100.0, (0.0, 0.1), amp=0.01, seed=0)
    f = hk.pagelocked_empty((N, M, 3))
    with pytest.raises(RuntimeError, match="static scratch"):
        s.fused_collide_stream_display_noalloc(f)
    assert s.steps_done == 0
    s.fused_collide_stream_display_noalloc(None)  # the plain step has no size limit
    assert s.steps_done == 1
    hk.release_pagelocked(f)
    s.close()
    # a pageable destination would make the runtime stage the copy: rejected, state untouched
    s = hk.Solver(64, 64, 0.3)
    s.set_surroundings(onp.ambient())
    s.upload_cells(onp.perturbed_cells(64, 64, seed=1))
    with pytest.raises(RuntimeError, match="page-locked"):
        s.fused_collide_stream_display_noalloc(np.empty((64, 64, 3), np.uint8))
    assert s.steps_done == 0
    # and the solver still works
    f = hk.pagelocked_empty((64, 64, 3))
    s.fused_collide_stream_display_noalloc(f)
    s.sync()
    assert np.array_equal(f, s.render())
    hk.release_pagelocked(f)
    s.close()


def test_noalloc_display_into_device_memory():
    """`frame` may be device memory too (the copy is issued with hipMemcpyDefault)."""
    import torch
    from latticeboltzmann_amd import lb_hip_kernel as hk
    N, M = 120, 200
    a, b = _pair(N, M, 2, "reference", 5)
    dev = torch.full((N, M, 3), 5, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    fa = hk.pagelocked_empty((N, M, 3))
    for it in range(2):
        a.fused_collide_stream_display_noalloc(fa)
        b.fused_collide_stream_display_noalloc(device_ptr=dev.data_ptr())
        a.sync()
        b.sync()
        assert np.array_equal(dev.cpu().numpy(), fa) and fa.any()
    hk.release_pagelocked(fa)
    a.close()
    b.close()


@pytest.mark.parametrize("entry", ["fused_collide_stream_display", "fused_collide_stream_display_noalloc"])
def test_display_error_at_a_seam_leaves_the_state_untouched(entry):
    """At a slab seam a frame after the step needs one more valid ghost row: that is checked BEFORE
    the step, so the error return leaves the step count and the state as they were."""
    from latticeboltzmann_amd import lb_hip_kernel as hk
    N, M = 40, 70
    s = hk.Solver(N, M, 0.3, boundary="periodic", halo=1, top_edge=hk.YEDGE_SEAM, bottom_edge=hk.YEDGE_SEAM)
    s.set_surroundings(onp.ambient())
    s.upload_cells(onp.perturbed_cells(N + 2, M, seed=3), -1)
    s.halo_mark_filled()
    f = hk.pagelocked_empty((N, M, 3))
    getattr(s, entry)(f)   # first update after an upload only collides: consumes no ghost row
    s.sync()
    assert s.steps_done == 1 and s.halo_valid == 1
    before = s.read_cells()
    with pytest.raises(RuntimeError, match="valid ghost rows"):
        getattr(s, entry)(f)   # would leave no valid ghost row to render from
    assert s.steps_done == 1 and s.halo_valid == 1
    assert np.array_equal(s.read_cells(), before)
    getattr(s, entry)(None)    # the plain step is still possible
    assert s.steps_done == 2
    hk.release_pagelocked(f)
    s.close()
