"""GPU: the opt-in two-levels-per-wave LDS kernel (csrc/lbm_tb2.hip, LBM_TB2=1 in the environment,
read once per process) against the oracle, bit for bit, in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CODE = r"""
import faulthandler, sys
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, {root!r})
import numpy as np
from latticeboltzmann_amd.lb_hip_kernel import Solver
from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc
rng = np.random.default_rng(3)
surr = onp.ambient()
for (N, M, T, boundary, omega, dtype) in [(70, 700, 6, "periodic", 1e-14, np.float32),
                                          (64, 1100, 4, "reference", 0.3, np.float32),
                                          (33, 130, 6, "reference", 1e-14, np.float32),
                                          (90, 520, 4, "periodic", 0.5, np.float32)]:
    blocked = rng.random((N, M)) < 0.04
    cells = onp.perturbed_cells(N, M, seed=N, amp=0.05)
    s = Solver(N, M, omega, inner_timesteps=T, boundary=boundary, dtype=dtype)
    s.set_tuning(6, 0)
    s.set_surroundings(surr)
    s.upload_blocked(blocked)
    s.upload_cells(cells)
    ref = cells.copy()
    s.fused_collide_stream(1)          # collide-only first update + one launch of depth T-1
    s.advance(2 * T)                   # two launches of depth T
    oc.run(ref, blocked, 3 * T, omega, surr, boundary)
    assert np.array_equal(s.read_cells(), ref), (N, M, T, boundary, omega)
    s.close()
print("tb2 ok")
"""


def test_two_levels_per_wave_kernel_is_bit_exact():
    env = dict(os.environ, LBM_TB2="1")
    r = subprocess.run([sys.executable, "-c", _CODE.format(root=ROOT)], capture_output=True, text=True,
                       timeout=150, env=env)
    assert r.returncode == 0 and "tb2 ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
