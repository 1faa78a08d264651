"""D2Q21 throughput probe (development aid): 168 B per cell-update, one update per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latticeboltzmann_amd.lb_hip_kernel import Solver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
omega = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
s = Solver(N, M, omega, boundary="periodic", lattice="d2q21")
s.init_equilibrium(100.0, (0.0, 0.3), amp=0.01, seed=0)
s.fused_collide_stream(3); s.sync()
for rep in range(3):
    ms = s.fused_collide_stream_timed(steps)
    g = N * M * steps / ms / 1e6
    print(f"D2Q21 {N}x{M} omega={omega}: {ms/steps:.3f} ms/update  {g:.2f} GLUPS  {g*168/1000:.3f} TB/s(168B)  frac={g*168/8000:.3f}", flush=True)
u, rho = s.read_moments(rows=(0, 2), cols=(0, 4)); print(rho[0], u[0, 0])
