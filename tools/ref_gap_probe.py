"""Why are reference edges slower than periodic in the LDS kernel?  Same kernel, the perturbation of the
initial state switched off (a uniform flow stays uniform: every wave on the common path) against the
usual amp = 0.01.  Development aid; output: gpurun_out/r3_ref_gap.log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latticeboltzmann_amd.lb_hip_kernel import Solver
from oracle import lbm_numpy as onp
N = M = 16384
for T in (4, 6):
    for boundary in ("periodic", "reference"):
        for amp in (0.01, 0.0):
            s = Solver(N, M, 1e-14, inner_timesteps=T, boundary=boundary)
            s.set_surroundings(onp.ambient())
            s.set_tuning(6, 0)
            s.init_equilibrium(100.0, (0.0, 0.1), amp=amp, seed=0)
            s.fused_collide_stream(3); s.sync()
            best = min(s.fused_collide_stream_timed(20) for _ in range(3))
            print(f"T={T} {boundary} amp={amp}: {N * M * 20 * T / best / 1e6:.2f} GLUPS", flush=True)
            s.close()
