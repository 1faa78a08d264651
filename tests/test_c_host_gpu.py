"""GPU: the C-ABI driven from a plain C99 program (tests/c/abi_host.c, built with gcc against
include/lbm_hip.h and liblbm_hip.so) - the boundary as a compiled caller sees it, without Python or
torch in the process - checked bit for bit against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "abi_host")
    lib = os.path.join(ROOT, "latticeboltzmann_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_host.c"), "-o", exe, "-L", lib, "-llbm_hip",
                           f"-Wl,-rpath,{lib}", "-Wl,--allow-shlib-undefined"])
    return exe


@pytest.mark.parametrize("boundary", ["reference", "periodic"])
def test_c_program_drives_the_library(tmp_path, boundary):
    exe = _build(tmp_path)
    N, M, steps, omega = 70, 150, 40, 0.3
    rng = np.random.default_rng(5)
    blocked = onp.blocked_cylinder(N, M) | (rng.random((N, M)) < 0.03)
    surr = onp.ambient()
    cells = onp.perturbed_cells(N, M, seed=9, amp=0.04)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(surr.astype(np.float32).tobytes())
        f.write(blocked.astype(np.uint8).tobytes())
        f.write(cells.tobytes())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, str(N), str(M), repr(omega), str(steps), boundary, fin, fout], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "abi_host ok" in r.stdout, (r.returncode, r.stdout, r.stderr[-2000:])
    out = np.fromfile(fout, np.float32)
    got = out[:N * M * 9].reshape(N, M, 9)
    rho = out[N * M * 9:N * M * 10].reshape(N, M)
    u = out[N * M * 10:].reshape(N, M, 2)
    ref = oc.run(cells.copy(), blocked, steps, omega, surr, boundary)
    ur, rr = onp.moments(ref)
    assert np.array_equal(got, ref)
    assert np.array_equal(rho, rr) and np.array_equal(u, ur)
