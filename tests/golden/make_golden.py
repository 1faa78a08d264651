"""Generates the committed fixtures under tests/golden/ (run in the build container, where
/root/reference is mounted; nothing under tests/ reads /root/reference at test time).

  masks.npz        np.packbits of `np.array(Image.open(p).convert('L')) > 128`
                   (/root/reference/latticeboltzmann.py:40-41) for the reference's two scene images
                   yeet.png / yeet2.png, plus shape and popcount: the reference's own data, used
                   to pin obstacle-mask bit-exactness.
  fields_*.npz     rho / u after 1, 10, 100, 1000 steps of oracle/lbm_numpy.py on small scenes.
                   FIELD PARITY IS UNPINNED BY THE REFERENCE (it ships no golden fields and cannot
                   run here): these vectors pin the oracle against itself over time (regressions,
                   numpy upgrades) and pin the C restatement and the HIP path to it.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import lbm_numpy as onp  # noqa: E402

REF = "/root/reference"


def masks():
    out = {}
    for name in ("yeet.png", "yeet2.png"):
        b = onp.blocked_from_image(os.path.join(REF, name))
        key = name.split(".")[0]
        out[key + "_bits"] = np.packbits(b)
        out[key + "_shape"] = np.array(b.shape)
        out[key + "_popcount"] = np.array(int(b.sum()))
        print(name, b.shape, int(b.sum()), hashlib.sha256(np.packbits(b).tobytes()).hexdigest()[:16])
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **out)


def fields(tag, N, M, blocked, cells, omega, surr, boundary, checkpoints):
    out = {"blocked_bits": np.packbits(blocked), "shape": np.array([N, M]),
           "omega": np.array(omega, np.float64), "surroundings": surr}
    done = 0
    for c in checkpoints:
        onp.run(cells, blocked, c - done, omega, surr, boundary)
        done = c
        u, p = onp.moments(cells)
        out[f"rho_{c}"] = p.copy()
        out[f"u_{c}"] = u.copy()
        out[f"cells_sha_{c}"] = np.frombuffer(hashlib.sha256(cells.tobytes()).digest(), np.uint8)
    np.savez_compressed(os.path.join(HERE, f"fields_{tag}.npz"), **out)
    print(tag, "done")


def main():
    masks()
    surr = onp.ambient()
    # 64 x 128 cylinder-like scene, reference boundaries, reference parameters
    N, M = 64, 128
    b = onp.blocked_cylinder(N, M)
    fields("cyl64x128_reference", N, M, b, onp.initial_cells(b, surr), onp.OMEGA, surr, "reference",
           (1, 10, 100, 1000))
    # unsteady variant (latticeboltzmann.jl:27-29): OMEGA = 0.01, u = [0, 0.3]
    surr3 = onp.ambient(100, [0, 0.3])
    fields("cyl64x128_vortex", N, M, b, onp.initial_cells(b, surr3), 0.01, surr3, "reference",
           (1, 10, 100, 1000))
    # periodic, random obstacles, perturbed start
    rng = np.random.default_rng(0)
    N, M = 48, 80
    b = rng.random((N, M)) < 0.05
    fields("rand48x80_periodic", N, M, b, onp.perturbed_cells(N, M, seed=0, amp=0.01), 0.1, surr,
           "periodic", (1, 10, 100))


if __name__ == "__main__":
    main()
