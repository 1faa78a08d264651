"""Host <-> device rates of the boundary (upload / readback go through a 256 MB device staging buffer
and layout-conversion kernels).  Outside bench.py's timed region; reported in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from latticeboltzmann_amd.lb_hip_kernel import Solver
from oracle import lbm_numpy as onp

N = M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cells = np.broadcast_to(onp.ambient().astype(np.float32), (N, M, 9)).copy()
blocked = np.zeros((N, M), bool)
s = Solver(N, M, 0.3, boundary="periodic")
s.set_surroundings(onp.ambient())
for name, fn, nbytes in [
    ("upload_blocked", lambda: s.upload_blocked(blocked), N * M),
    ("upload_cells", lambda: s.upload_cells(cells), N * M * 36),
    ("read_cells", lambda: s.read_cells(), N * M * 36),
    ("read_moments", lambda: s.read_moments(), N * M * 12),
    ("render", lambda: s.render(), N * M * 3),
]:
    fn()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name:15s} {N}x{M}: {dt*1e3:8.1f} ms  {nbytes/dt/1e9:6.2f} GB/s (host bytes)", flush=True)
s.close()
