"""Randomised parity soak: many random (shape, depth, chunk rows, edge rule, storage, obstacle density,
omega) cases of the HIP path against the C oracle, bit for bit.  A race in the LDS pipeline would be
intermittent; this runs far more cases than the test-suite.  usage: stress_parity.py [cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from latticeboltzmann_amd.lb_hip_kernel import Solver
from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    for i in range(cases):
        N = int(rng.integers(1, 260))
        M = int(rng.choice([rng.integers(2, 70), rng.integers(60, 400), rng.integers(380, 900)]))
        T = int(rng.integers(1, 10))
        cap = int(rng.integers(1, 7))
        rows = int(rng.choice([0, 0, rng.integers(1, 40), rng.integers(16, 300)]))
        boundary = str(rng.choice(["reference", "periodic"]))
        omega = float(rng.choice([1e-14, 1e-14, 0.01, 0.3, 0.5, 1.0, 1.7, 2.0 ** -34, -1e-14]))
        dens = float(rng.choice([0.0, 0.02, 0.1, 0.4]))
        n_outer = int(rng.integers(1, 4))
        blocked = rng.random((N, M)) < dens
        cells = onp.perturbed_cells(N, M, seed=int(rng.integers(1 << 30)), amp=0.05)
        if rng.random() < 0.3 and N > 4 and M > 8:
            # patches the wave-uniform fast paths must refuse: negative / zero / huge / tiny
            # populations, flow fast enough for 1.5 |u|^2 > 1/8
            for _ in range(int(rng.integers(1, 4))):
                y, x = int(rng.integers(0, N - 2)), int(rng.integers(0, M - 4))
                kind = int(rng.integers(0, 7))
                if kind == 0: cells[y:y + 2, x:x + 4] *= np.float32(-1)
                elif kind == 1: cells[y:y + 2, x:x + 4] = 0
                elif kind == 2: cells[y:y + 2, x:x + 4] *= np.float32(1e32)
                elif kind == 3: cells[y:y + 2, x:x + 4] *= np.float32(1e-33)
                elif kind == 4: cells[y:y + 2, x:x + 4] = onp.getEquilibrium(np.float32([[0.3, 0.35]]), np.float32([90]))[0]
                elif kind == 5:
                    # momentum sums far below the density: only the rest population is ordinary, the
                    # moving ones are tiny or denormal (numerators of 2^-100 .. 2^-149 over p ~ 1)
                    t = rng.random((2, 4, 9)).astype(np.float32) * np.float32(2.0) ** np.float32(-int(rng.integers(100, 127)))
                    t[..., 4] = np.float32(rng.uniform(0.5, 200))
                    cells[y:y + 2, x:x + 4] = t
                else:
                    # densities at the ends of the window the fast quotient accepts (2^-60 .. 2^39)
                    cells[y:y + 2, x:x + 4] *= np.float32(2.0) ** np.float32(int(rng.choice([-67, -66, -65, 31, 32, 33])))
        u_amb = np.array([[rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.3)]], np.float32)
        surr = np.ascontiguousarray(onp.getEquilibrium(u_amb, np.array([100], np.float32))[0], dtype=np.float32)
        s = Solver(N, M, omega, inner_timesteps=T, boundary=boundary)
        s.set_tuning(cap, rows)
        s.set_surroundings(surr)
        s.upload_blocked(blocked)
        s.upload_cells(cells)
        ref = cells.copy()
        ok = True
        for _ in range(n_outer):
            s.fused_collide_stream(1)
            with np.errstate(all="ignore"):
                oc.run(ref, blocked, T, omega, surr, boundary)
            got = s.read_cells()
            if not np.array_equal(got, ref, equal_nan=True):  # a blown-up run (omega 1.7) is NaN on both sides
                ok = False
                w = np.argwhere((got != ref) & ~(np.isnan(got) & np.isnan(ref)))
                print(f"MISMATCH case {i}: N={N} M={M} T={T} cap={cap} rows={rows} {boundary} omega={omega} "
                      f"dens={dens}: {len(w)} values, first {w[:3].tolist()}", flush=True)
                break
        s.close()
        bad += not ok
        if i % 25 == 24:
            print(f"{i + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} mismatching")
    sys.exit(1 if bad else 0)


main()
