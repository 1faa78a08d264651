"""GPU: the slab backend (SlabSolver: HIP solver with ghost rows, row-range updates, split last
update) on 2 ranks that share the one GPU of the box, halos staged through the host over gloo
(RCCL refuses two ranks on one device; the RCCL path differs only in handing the same device
regions to torch.distributed directly).  Checked bit for bit against the single-domain oracle."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, boundary, H, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from latticeboltzmann_amd.slabs import SlabSolver
    from oracle import lbm_numpy as onp
    from oracle import lbm_ref as oc
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    try:
        G, M = 150, 200
        rng = np.random.default_rng(11)
        blocked = rng.random((G, M)) < 0.05
        cells = onp.perturbed_cells(G, M, seed=4, amp=0.05)
        surr = onp.ambient()
        slab = SlabSolver(G, M, 0.2, rank, world, boundary=boundary, halo=H, device=0, dist=dist)
        slab.set_surroundings(surr)
        slab.upload_global(cells, blocked)
        ref = cells.copy()
        for n in (1, 3, 4, 9, 2):
            slab.step(n)
            oc.run(ref, blocked, n, 0.2, surr, boundary)
            got = slab.read_cells()
            want = ref[slab.row_lo:slab.row_hi]
            assert np.array_equal(got, want), (rank, n, np.abs(got - want).max())
        slab.close()
        open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("boundary,H", [("periodic", 4), ("reference", 3)])
def test_two_slabs_one_gpu(tmp_path, boundary, H):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), boundary, H, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


def test_device_memory_wraps_as_torch_tensor():
    """The RCCL path hands library-owned device memory to torch.distributed through
    __cuda_array_interface__; check that wrapping aliases the memory (no copy)."""
    import torch
    from latticeboltzmann_amd.lb_hip_kernel import Solver
    from latticeboltzmann_amd.slabs import _DevMem
    from latticeboltzmann_amd import lb_hip_kernel as hk
    s = Solver(16, 64, 0.5, boundary="periodic", halo=2, top_edge=hk.YEDGE_SEAM,
               bottom_edge=hk.YEDGE_SEAM)
    ptr, nbytes = s.halo_region(0, True)
    t = torch.as_tensor(_DevMem(ptr, nbytes), device="cuda:0")
    assert t.data_ptr() == ptr and t.numel() == nbytes and t.is_cuda
    assert nbytes == 2 * 1 * 9 * 64 * 4
    s.close()


def _worker_tb(rank, world, port, boundary, outdir):
    """Slabs with LDS temporal blocking inside a cycle (halo 6, launches up to 3 deep)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from latticeboltzmann_amd.slabs import SlabSolver
    from oracle import lbm_numpy as onp
    from oracle import lbm_ref as oc
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    try:
        G, M = 160, 300
        rng = np.random.default_rng(12)
        blocked = rng.random((G, M)) < 0.05
        cells = onp.perturbed_cells(G, M, seed=6, amp=0.05)
        surr = onp.ambient()
        slab = SlabSolver(G, M, 0.2, rank, world, boundary=boundary, halo=6, device=0, dist=dist,
                          tb_max_depth=3)
        slab.set_surroundings(surr)
        slab.upload_global(cells, blocked)
        ref = cells.copy()
        for n in (7, 12, 5):
            slab.step(n)
            oc.run(ref, blocked, n, 0.2, surr, boundary)
            got = slab.read_cells()
            want = ref[slab.row_lo:slab.row_hi]
            assert np.array_equal(got, want), (rank, n, np.abs(got - want).max())
        slab.close()
        open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("boundary", ["periodic", "reference"])
def test_two_slabs_with_temporal_blocking(tmp_path, boundary):
    """reference: the top rank has the inlet row and a seam below, the bottom rank a seam above
    and the outflow row (phantom row of the LDS kernel) below."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_tb, args=(2, _free_port(), boundary, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


_RCCL_SELF = r"""
import faulthandler, sys
faulthandler.dump_traceback_later(120, exit=True)  # a hang names its line instead of being killed silently
sys.path.insert(0, {root!r})
import numpy as np
import torch
import torch.distributed as dist
from latticeboltzmann_amd.slabs import SlabSolver
from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc
torch.cuda.set_device(0)
from latticeboltzmann_amd.slabs import init_process_group_rccl
init_process_group_rccl(0, init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1)
rng = np.random.default_rng(21)
surr = onp.ambient()
for G, M, halo, depth, steps in [(192, 320, 4, 1, (1, 3, 4, 9, 2, 16)), (64, 130, 1, 1, (1, 2, 5)),
                                 (96, 200, 2, 6, (3, 4, 7)), (160, 300, 7, 6, (1, 7, 14, 5)),
                                 (128, 64, 5, 3, (11, 2, 9))]:
    blocked = rng.random((G, M)) < 0.05
    cells = onp.perturbed_cells(G, M, seed=8 + halo, amp=0.05)
    slab = SlabSolver(G, M, 0.2, 0, 1, boundary="periodic", halo=halo, device=0, dist=dist,
                      self_seam=True, tb_max_depth=depth)
    assert slab.seams and not slab.host_staged and slab.up == 0 and slab.down == 0
    slab.set_surroundings(surr)
    slab.upload_global(cells, blocked)
    ref = cells.copy()
    for n in steps:
        slab.step(n)
        oc.run(ref, blocked, n, 0.2, surr, "periodic")
        got = slab.read_cells()
        assert np.array_equal(got, ref), (G, M, halo, depth, n, float(np.abs(got - ref).max()))
    slab.close()
dist.destroy_process_group()
print("rccl self-seam ok")
"""


def test_rccl_path_with_a_self_seam():
    """The RCCL data path on the one GPU of the box: a single periodic rank treats its own
    wrap-around as a seam, so every cycle sends its edge rows to itself through
    torch.distributed/nccl (device memory handed over directly, communication stream, events,
    interior update overlapped) — the code the multi-GPU bench runs, bit for bit against the oracle."""
    import subprocess
    code = _RCCL_SELF.format(root=ROOT, port=_free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=150)
    assert r.returncode == 0 and "rccl self-seam ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_advance_rows_checks_its_inputs():
    """lbm_advance_rows: a row band may only jump as deep as the valid ghost rows behind it reach,
    and all bands of one open update share the depth."""
    from latticeboltzmann_amd import lb_hip_kernel as hk
    from oracle import lbm_numpy as onp
    N, M = 40, 128
    s = hk.Solver(N, M, 0.3, boundary="periodic", halo=2, top_edge=hk.YEDGE_SEAM, bottom_edge=hk.YEDGE_SEAM)
    s.set_surroundings(onp.ambient())
    cells = onp.perturbed_cells(N + 4, M, seed=1)
    s.upload_cells(cells, -2)
    s.halo_mark_filled()
    with pytest.raises(RuntimeError, match="stepped state"):
        s.advance_rows(0, 4, 2, False)
    s.fused_collide_stream(1)          # collide-only: consumes no ghost row
    assert s.halo_valid == 2
    with pytest.raises(RuntimeError, match="valid ghost rows"):
        s.advance_rows(0, 4, 3, False)
    with pytest.raises(RuntimeError, match="depth must be"):
        s.advance_rows(0, 4, 7, False)
    s.advance_rows(0, 4, 2, False)
    with pytest.raises(RuntimeError, match="open update has depth 2"):
        s.advance_rows(N - 4, 4, 1, False)
    s.advance_rows(N - 4, 4, 2, False)
    # finishing with owned rows uncovered would leave stale rows in the new state: refused
    with pytest.raises(RuntimeError, match="do not cover"):
        s.advance_rows(4, N - 9, 2, True)
    s.advance_rows(4, N - 8, 2, True)
    assert s.steps_done == 3 and s.halo_valid == 0
    s.close()
    # ghost rows exist at seams only: beyond a DOMAIN edge nothing may be computed, whatever `halo`
    # says (the kernel would read rows in front of the buffer)
    s = hk.Solver(N, M, 0.3, boundary="reference", halo=3, top_edge=hk.YEDGE_DOMAIN,
                  bottom_edge=hk.YEDGE_SEAM)
    s.set_surroundings(onp.ambient())
    s.upload_cells(onp.perturbed_cells(N + 6, M, seed=2), -3)
    s.halo_mark_filled()
    s.fused_collide_stream(1)
    with pytest.raises(RuntimeError, match="can be computed"):
        s.advance_rows(-1, 5, 1, False)
    with pytest.raises(RuntimeError, match="can be computed"):
        s.advance_rows(N - 4, 7, 1, False)     # only 2 ghost rows stay computable at depth 1
    s.advance_rows(N - 4, 6, 1, False)         # owned rows + the 2 ghost rows that stay valid
    s.advance_rows(0, N - 4, 1, True)
    assert s.steps_done == 2 and s.halo_valid == 2
    s.close()


@pytest.mark.parametrize("n,rows,cols", [(2, 2048, 2048), (8, 512, 1024)])
def test_bench_n2_rehearsal(tmp_path, n, rows, cols):
    """bench.py's N > 1 branch (slab timing, the max-over-ranks all_reduce, the
    `slabs_temporal_blocking` side line) run end to end before the driver runs it on an 8-GPU
    node: a fresh child `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with
    two ranks sharing this box's one GPU, halos staged through the host over gloo
    (LBM_BENCH_BACKEND=gloo; RCCL refuses two ranks per device).  Checks the JSON contract, not
    the speed.  n = 8 is the driver's largest case: a ring of eight slabs, eight processes on the one
    GPU (the box allows 16)."""
    import json
    import subprocess
    env = dict(os.environ, LBM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rows", str(rows), "--cols", str(cols),
           "--steps", "8", "--warmup", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    (tmp_path / f"bench_n{n}.log").write_text(r.stdout + "\n--- stderr ---\n" + r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 8 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["unit"] == "GLUPS" and d["value"] > 0 and d["higher_is_better"] is True
    assert "gloo (rehearsal)" in d["config"]["decomposition"] and f"{n} y-slabs" in d["config"]["decomposition"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
    tb = d["slabs_temporal_blocking"]
    assert "error" not in tb, tb
    assert tb["finite"] is True and tb["depth"] == 4 and tb["halo"] == 4 and tb["value"] > 0
    assert "cpu_baseline" not in d and "temporal_blocking" not in d
