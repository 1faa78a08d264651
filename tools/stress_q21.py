"""Randomised parity soak of the D2Q21 lattice against oracle/lbm_d2q21.py, bit for bit.
usage: stress_q21.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from latticeboltzmann_amd.lb_hip_kernel import Solver
from oracle import lbm_d2q21 as q

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, t0 = 0, time.time()
for i in range(cases):
    N = int(rng.integers(4, 120))
    M = int(rng.choice([rng.integers(4, 70), rng.integers(60, 300)]))
    omega = float(rng.choice([1e-14, 0.01, 0.3, 0.6, 1.0, 1.7]))
    dens = float(rng.choice([0.0, 0.03, 0.2]))
    boundary = str(rng.choice(["periodic", "reference"]))
    surr = q.ambient(100, [rng.uniform(-0.05, 0.05), rng.uniform(-0.1, 0.3)])
    blocked = rng.random((N, M)) < dens
    cells = q.perturbed_cells(N, M, seed=int(rng.integers(1 << 30)), amp=0.05,
                              u=[rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.3)])
    if rng.random() < 0.3:
        y, x = int(rng.integers(0, N - 2)), int(rng.integers(0, M - 3))
        kind = int(rng.integers(0, 4))
        if kind == 0: cells[y:y + 2, x:x + 3] *= np.float32(-1)
        elif kind == 1: cells[y:y + 2, x:x + 3] = 0
        elif kind == 2: cells[y:y + 2, x:x + 3] *= np.float32(1e32)
        else: cells[y:y + 2, x:x + 3] *= np.float32(1e-33)
    s = Solver(N, M, omega, boundary=boundary, lattice="d2q21")
    s.set_surroundings(surr)
    s.upload_blocked(blocked); s.upload_cells(cells)
    ref = cells.copy()
    for n in rng.integers(1, 8, size=int(rng.integers(1, 4))):
        s.fused_collide_stream(int(n))
        with np.errstate(all="ignore"):
            q.run(ref, blocked, int(n), omega, surr, boundary)
        got = s.read_cells()
        if not np.array_equal(got, ref, equal_nan=True):
            w = np.argwhere((got != ref) & ~(np.isnan(got) & np.isnan(ref)))
            print(f"MISMATCH case {i}: N={N} M={M} {boundary} omega={omega} dens={dens}: {len(w)} values, first {w[:3].tolist()}", flush=True)
            bad += 1
            break
    s.close()
    if i % 50 == 49:
        print(f"{i + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"done: {cases} d2q21 cases, {bad} mismatching")
sys.exit(1 if bad else 0)
