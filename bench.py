#!/usr/bin/env python3
"""Headline benchmark: GLUPS of the fused D2Q9 collide+stream path on MI355X.

    python bench.py --gpus N --steps K --warmup W
(for N > 1 launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`).

Workload (BASELINE.json configs[1]): fp32 D2Q9, 16384 x 16384 cells per GPU, periodic channel,
INNER_TIMESTEPS = 1, obstacle-free, state = equilibrium(rho=100, u=[0,0.1]) with a seeded +-1 %
density perturbation generated on the device.  A "step" is one lattice update of every cell.
N > 1 is weak scaling: N y-slabs of 16384 rows of one (N*16384) x 16384 periodic domain, 4-deep
halo exchanged every 4 steps by RCCL send/recv (latticeboltzmann_amd/slabs.py).

One JSON line on rank 0.  `roofline`: the fused kernel's algorithmic bytes (72 B per cell-update,
/root/reference/latticeboltzmann.py:176) per launch over its average launch time measured with HIP
events on the solver's stream, against the 8 TB/s HBM peak.  `cpu_baseline`: the oracle's numpy
path (the reference's vectorised algorithm) timed on this host on a bounded twin of the workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_UPDATE = 72   # 9 populations x 4 B read + written


def cpu_baseline(budget_s=12.0):
    """numpy oracle (the reference's vectorised CPU path) and its C twin on a 2048 x 2048 periodic
    twin of the bench scene; bounded to roughly `budget_s` seconds each."""
    from oracle import lbm_numpy as onp
    from oracle import lbm_ref as oc
    n = 2048
    cells = onp.perturbed_cells(n, n, seed=0, amp=0.01)
    blocked = np.zeros((n, n), bool)
    omega = onp.make_omega(blocked, onp.OMEGA)
    surr = onp.ambient()
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = 1
    onp.step(cells, blocked, omega, surr, "periodic")  # warm-up
    steps, t0 = 0, time.perf_counter()
    while True:
        onp.step(cells, blocked, omega, surr, "periodic")
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 64:
            break
    np_glups = n * n * steps / dt / 1e9
    out = {"value": np_glups, "unit": "GLUPS", "cores": blas_threads, "kind": "port",
           "sample": f"oracle/lbm_numpy.py (numpy {np.__version__}; elementwise single-threaded, "
                     f"BLAS matmul up to {blas_threads} threads), {n}x{n} periodic twin, {steps} steps "
                     f"in {dt:.1f} s; host has {os.cpu_count()} logical CPUs"}
    # the reference's own default scene size (BASELINE.json configs[0]: 400 x 800 cylinder)
    cb = onp.blocked_cylinder(400, 800)
    cc = onp.initial_cells(cb, surr)
    co = onp.make_omega(cb, onp.OMEGA)
    onp.step(cc, cb, co, surr, "reference")
    t0 = time.perf_counter()
    for _ in range(20):
        onp.step(cc, cb, co, surr, "reference")
    dt = time.perf_counter() - t0
    out["numpy_cylinder"] = {"value": 400 * 800 * 20 / dt / 1e9, "unit": "GLUPS", "cores": blas_threads,
                             "sample": f"oracle/lbm_numpy.py, 400x800 cylinder scene, 20 steps in {dt:.1f} s"}
    # the C restatement (OpenMP), a fairer CPU figure
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(16, os.cpu_count() or 1)
    csteps = 8
    oc.run(cells, blocked, 1, onp.OMEGA, surr, "periodic")
    t0 = time.perf_counter()
    oc.run(cells, blocked, csteps, onp.OMEGA, surr, "periodic")
    dt = time.perf_counter() - t0
    out["c_port"] = {"value": n * n * csteps / dt / 1e9, "unit": "GLUPS", "cores": omp,
                     "sample": f"oracle/lbm_ref.c (OpenMP), same scene, {csteps} steps in {dt:.1f} s"}
    return out


def traffic_per_launch():
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic.json; how it was collected and corrected is in DESIGN.md), or None."""
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=16384, help="rows per GPU")
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--halo", type=int, default=4)
    ap.add_argument("--depth", type=int, default=1,
                    help="N > 1 only: deepest LDS launch inside a halo cycle (1 = the headline kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from latticeboltzmann_amd import lb_hip_kernel as hk
    from latticeboltzmann_amd.slabs import SlabSolver
    from latticeboltzmann_amd import latticeboltzmann as drv  # the product driver: OMEGA, ambient()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import faulthandler
        import torch.distributed as dist
        # A hang on the N > 1 path must end the run with a traceback and a non-zero status, not at
        # the driver's limit with nothing written: every slab wait is a bounded host-side poll
        # (SlabSolver.timeout_s -> SlabTimeout), and this is the backstop for everything else
        # (collectives, process-group set-up).  Never a re-exec.
        hang_s = float(os.environ.get("LBM_BENCH_HANG_S", "600"))
        faulthandler.dump_traceback_later(hang_s, exit=True)
        # RCCL on a real node; LBM_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks
        # on one GPU (halos staged through the host), never for a reported number
        backend = os.environ.get("LBM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            from latticeboltzmann_amd.slabs import init_process_group_rccl
            init_process_group_rccl(local_rank)
        else:
            dist.init_process_group(backend)

    rows, cols = args.rows, args.cols
    surr = drv.ambient()
    OMEGA = drv.OMEGA
    if world == 1:
        solver = hk.Solver(rows, cols, OMEGA, inner_timesteps=1, boundary="periodic",
                           device=local_rank)
        solver.set_surroundings(surr)
        solver.init_equilibrium(100.0, (0.0, 0.1), amp=0.01, seed=0, row_offset=0, global_rows=rows)
        solver.fused_collide_stream(args.warmup)
        solver.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel_ms = solver.fused_collide_stream_timed(args.steps)  # HIP events on the solver's stream
        solver.sync()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        u, rho = solver.read_moments(rows=(0, 8), cols=(0, 64))
        if not (np.isfinite(rho).all() and np.isfinite(u).all()):
            raise SystemExit("non-finite fields after the timed region")
        solver.close()
        # BASELINE.json configs[2]: the same grid with INNER_TIMESTEPS = 4 / 5 / 6 (LDS temporal
        # blocking, one launch per outer iteration) and multi-obstacle bounce-back.  Reported beside
        # the headline, never as `value`.
        # obstacles: the reference's own scene (the mask of its yeet.png, tests/golden/masks.npz) tiled
        # over the grid; seeded discs if the fixture is missing
        scene, scene_name = None, "64 discs r=73"
        try:
            z = np.load(os.path.join(ROOT, "tests", "golden", "masks.npz"))
            sh = tuple(int(v) for v in z["yeet_shape"])
            yeet = np.unpackbits(z["yeet_bits"])[:sh[0] * sh[1]].reshape(sh)
            scene = np.tile(yeet, (-(-rows // sh[0]), -(-cols // sh[1])))[:rows, :cols]
            scene_name = f"yeet.png mask tiled ({100.0 * scene.mean():.2f} % blocked)"
        except Exception:
            pass
        tb = []
        tr_all = traffic_per_launch() or {}
        tb_traffic = {int(k.split(",")[1]): v for k, v in tr_all.get("temporal_blocking", {}).items()}
        SIDE_LAUNCHES = 8  # side lines: at least this many timed launches whatever --steps says
        for T in (4, 5, 6):
            s2 = hk.Solver(rows, cols, OMEGA, inner_timesteps=T, boundary="periodic",
                           device=local_rank)
            s2.set_surroundings(surr)
            s2.init_equilibrium(100.0, (0.0, 0.1), amp=0.01, seed=0, row_offset=0, global_rows=rows)
            if scene is not None:
                s2.upload_blocked(scene)
            else:
                s2.init_blocked_discs(64, 73, seed=1, global_rows=rows)
            n_outer = max(SIDE_LAUNCHES, args.steps // (2 * T))
            s2.fused_collide_stream(3)  # collide-only first update + 2 warm-up launches
            s2.sync()
            ms = s2.fused_collide_stream_timed(n_outer)
            u2, rho2 = s2.read_moments(rows=(0, 8), cols=(0, 64))
            ok = bool(np.isfinite(rho2).all() and np.isfinite(u2).all())
            s2.close()
            g = rows * cols * n_outer * T / ms / 1e6
            tt = tb_traffic.get(T) if (rows, cols) == (16384, 16384) else None
            tb.append({"inner_timesteps": T, "value": g, "unit": "GLUPS", "launches": n_outer,
                       "ms_per_launch": ms / n_outer, "finite": ok,
                       "traffic_bytes_per_update": tt["bytes_per_cell_update"] if tt else None,
                       "frac_of_72B_roofline": g * BYTES_PER_UPDATE / HBM_PEAK_GBS,
                       "workload": f"{rows}x{cols} periodic, {scene_name} (bounce-back), T={T}"})
        # BASELINE.json configs[4]'s storage variant on the headline grid: fp16 storage, fp32
        # arithmetic, 36 B per cell-update.  Reported beside the headline, never as `value`.
        s3 = hk.Solver(rows, cols, OMEGA, inner_timesteps=1, boundary="periodic",
                       device=local_rank, dtype=np.float16)
        s3.set_surroundings(surr)
        s3.init_equilibrium(100.0, (0.0, 0.1), amp=0.01, seed=0, row_offset=0, global_rows=rows)
        s3.fused_collide_stream(3)
        s3.sync()
        n3 = max(SIDE_LAUNCHES, args.steps // 4)
        ms3 = s3.fused_collide_stream_timed(n3)
        u3, rho3 = s3.read_moments(rows=(0, 8), cols=(0, 64))
        s3.close()
        g3 = rows * cols * n3 / ms3 / 1e6
        fp16 = {"value": g3, "unit": "GLUPS", "launches": n3, "ms_per_launch": ms3 / n3,
                "finite": bool(np.isfinite(rho3).all() and np.isfinite(u3).all()),
                "hbm_GBps_at_36B": g3 * 36, "frac_of_36B_roofline": g3 * 36 / HBM_PEAK_GBS,
                "workload": f"{rows}x{cols} periodic, fp16 storage / fp32 compute, T=1"}
        # SURVEY.md 8(f4), scoped: the draft's D2Q21 lattice (latticeboltzmann.jl:3-22, its own
        # parameters :27-29) on the same grid, one update per launch, 2 x 21 x 4 = 168 B per
        # cell-update.  Parity-unpinned by the reference; reported beside the headline, never as `value`.
        s5 = hk.Solver(rows, cols, 0.01, inner_timesteps=1, boundary="periodic", device=local_rank,
                       lattice="d2q21")
        s5.init_equilibrium(100.0, (0.0, 0.3), amp=0.01, seed=0)
        s5.fused_collide_stream(3)
        s5.sync()
        n5 = max(SIDE_LAUNCHES, args.steps // 10)
        ms5 = s5.fused_collide_stream_timed(n5)
        u5, rho5 = s5.read_moments(rows=(0, 8), cols=(0, 64))
        s5.close()
        g5 = rows * cols * n5 / ms5 / 1e6
        d2q21 = {"value": g5, "unit": "GLUPS", "launches": n5, "ms_per_launch": ms5 / n5,
                 "finite": bool(np.isfinite(rho5).all() and np.isfinite(u5).all()),
                 "hbm_GBps_at_168B": g5 * 168, "frac_of_168B_roofline": g5 * 168 / HBM_PEAK_GBS,
                 "workload": f"{rows}x{cols} periodic, D2Q21 (latticeboltzmann.jl:3-22), OMEGA=0.01, u=[0,0.3], T=1"}
    else:
        slab = SlabSolver(rows * world, cols, OMEGA, rank, world, boundary="periodic",
                          halo=args.halo, device=local_rank, dist=dist, tb_max_depth=args.depth)
        slab.timeout_s = min(slab.timeout_s, hang_s / 2)
        slab.set_surroundings(surr)
        slab.init_device_scene(100.0, (0.0, 0.1), amp=0.01, seed=0)
        slab.step(args.warmup)
        slab.sync()
        dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(slab.compute)
        slab.step(args.steps)
        slab._wait_pending()
        ev1.record(slab.compute)
        slab.sync()
        torch.cuda.synchronize()
        dist.barrier()
        wall = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1)
        t = torch.tensor([wall, kernel_ms], dtype=torch.float64,
                         device=f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kernel_ms = float(t[0]), float(t[1])
        u, rho = slab.read_moments(rows=(0, 8), cols=(0, 64))
        if not (np.isfinite(rho).all() and np.isfinite(u).all()):
            raise SystemExit("non-finite fields after the timed region")
        slab.close()
        # BASELINE.md config 4 reads "halo 4, T = 4": the same slabs with the LDS kernel inside the
        # halo cycle (one split depth-`halo` launch per cycle).  Reported beside the headline,
        # never as `value` (the headline keeps the N = 1 kernel so that scaling compares like with like).
        slab_tb = None
        if args.depth == 1:
            d = min(args.halo, 6)
            dev = f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu"
            s4, err = None, None
            try:
                s4 = SlabSolver(rows * world, cols, OMEGA, rank, world, boundary="periodic",
                                halo=args.halo, device=local_rank, dist=dist, tb_max_depth=d)
                s4.timeout_s = min(s4.timeout_s, hang_s / 2)
                s4.set_surroundings(surr)
                s4.init_device_scene(100.0, (0.0, 0.1), amp=0.01, seed=0)
            except Exception as e:  # never let the extra line break the headline ...
                err = repr(e)[:300]
            # ... nor leave the other ranks waiting in a collective: agree first
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 1:
                n4 = max(8 * args.halo, (args.steps // args.halo) * args.halo)  # >= 8 halo cycles
                s4.step(1 + 2 * args.halo)
                s4.sync()
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s4.step(n4)
                s4.sync()
                torch.cuda.synchronize()
                dist.barrier()
                w4 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                dist.all_reduce(w4, op=dist.ReduceOp.MAX)
                u4, rho4 = s4.read_moments(rows=(0, 8), cols=(0, 64))
                slab_tb = {"depth": d, "halo": args.halo, "steps": n4, "unit": "GLUPS",
                           "value": rows * cols * world * n4 / float(w4[0]) / 1e9,
                           "ms_per_step": float(w4[0]) * 1e3 / n4,
                           "finite": bool(np.isfinite(rho4).all() and np.isfinite(u4).all())}
            else:
                slab_tb = {"error": err or "another rank failed to set up"}
            if s4 is not None:
                s4.close()

    if rank == 0:
        cells = rows * cols
        glups = cells * world * args.steps / wall / 1e9
        launch_ms = kernel_ms / args.steps
        achieved = BYTES_PER_UPDATE * cells / (launch_ms * 1e-3) / 1e9
        tr = traffic_per_launch() if world == 1 and (rows, cols) == (16384, 16384) else None
        line = {
            "metric": "GLUPS (fp32 D2Q9)", "value": glups, "unit": "GLUPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"D2Q9 BGK fp32 {rows}x{cols} per GPU, periodic channel, "
                                   f"INNER_TIMESTEPS=1 (BASELINE.json configs[1])",
                       "rows_per_gpu": rows, "cols": cols, "inner_timesteps": 1,
                       "decomposition": "single domain" if world == 1 else
                                        f"{world} y-slabs, halo {args.halo}, "
                                        f"{'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend() + ' (rehearsal)'} "
                                        f"send/recv every {args.halo} steps"
                                        + (f", LDS launches up to {args.depth} deep" if args.depth > 1 else "")},
            "frac_of_hbm_roofline": glups * BYTES_PER_UPDATE / (HBM_PEAK_GBS * world),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": tr["bytes_per_launch"] if tr else None,
                         "kernel": "lbm::step_kernel", "launch_ms": launch_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_UPDATE * cells,
                         "traffic_source": tr.get("source") if tr else None},
        }
        if world > 1 and slab_tb is not None:
            line["slabs_temporal_blocking"] = slab_tb
        if world == 1:
            line["temporal_blocking"] = tb
            line["fp16_storage"] = fp16
            line["d2q21"] = d2q21
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
