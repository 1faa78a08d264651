"""BASELINE config 5 honesty check: the fp16-storage / fp32-compute variant on the tall-box cavity
(2048 x 512, open-topped box obstacle, ambient flow over its mouth) against the fp32 oracle:
relative L-inf of rho and u at 96 / 480 / 1000 steps, T = 1 and T = 6 launches (same bits since
the LDS rings round through half).  Development aid; the numbers go into DESIGN.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from latticeboltzmann_amd.lb_hip_kernel import Solver  # noqa: E402
from oracle import lbm_numpy as onp  # noqa: E402
from oracle import lbm_ref as oc  # noqa: E402


def cavity_scene(N, M):
    b = np.zeros((N, M), bool)
    y0, y1, x0, x1, t = N // 2, N - 40, 40, M - 60, 6
    b[y0:y1, x0:x0 + t] = True
    b[y0:y1, x1 - t:x1] = True
    b[y1 - t:y1, x0:x1] = True
    return b


def main():
    N, M = 2048, 512
    omega = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    blocked = cavity_scene(N, M)
    surr = onp.ambient()
    cells = onp.initial_cells(blocked, surr)
    ref = cells.copy()
    sol = {}
    for T in (1, 6):
        s = Solver(N, M, omega, inner_timesteps=1, boundary="reference", dtype=np.float16)
        s.set_tuning(T, 0)
        s.set_surroundings(surr)
        s.upload_blocked(blocked)
        s.upload_cells(cells)
        sol[T] = s
    done = 0
    for c in (96, 480, 1000):
        oc.run(ref, blocked, c - done, omega, surr, "reference")
        ur, rr = onp.moments(ref)
        line = [f"omega={omega} steps={c:5d}"]
        for T, s in sol.items():
            s.advance(c - done)
            u, rho = s.read_moments()
            line.append(f"T={T}: rho {onp.rel_linf(rho, rr):.2e} u {onp.rel_linf(u, ur):.2e}")
        done = c
        same = np.array_equal(sol[1].read_cells(), sol[6].read_cells())
        print("  ".join(line) + f"  T1==T6 bits: {same}", flush=True)


main()
