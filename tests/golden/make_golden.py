"""Generates the committed fixtures under tests/golden/ (run in the build container, where
/root/reference is mounted; nothing under tests/ reads /root/reference at test time).

  masks.npz        np.packbits of `np.array(Image.open(p).convert('L')) > 128`
                   (/root/reference/latticeboltzmann.py:40-41) for the reference's two scene images
                   yeet.png / yeet2.png, plus shape and popcount: the reference's own data, used
                   to pin obstacle-mask bit-exactness.
  fields_*.npz     rho / u (and a sha256 of the cells) after 1, 10, 100, 1000 steps of THE
                   REFERENCE'S OWN numpy lines, executed from /root/reference/latticeboltzmann.py
                   by tests/golden/ref_extract.py (which reads that file at run time and stores
                   none of it).  The oracle (numpy and C), and through it the HIP path, are pinned
                   to these.  The periodic scene replaces the reference's stream() by np.roll (no
                   reference counterpart); everything else in it is the reference's text too.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import lbm_numpy as onp  # noqa: E402
import ref_extract  # noqa: E402

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


def fields(tag, N, M, blocked, cells, omega, u_amb, boundary, checkpoints):
    """cells=None: the reference's own initial state (latticeboltzmann.py:142)."""
    ref = ref_extract.ReferenceNumpyPath(
        blocked, {"OMEGA": omega, "u_ambient": u_amb}, cells=cells,
        stream=onp.stream_periodic if boundary == "periodic" else None)
    surr = ref.surroundings
    assert np.array_equal(surr, onp.ambient(100, u_amb))
    out = {"blocked_bits": np.packbits(blocked), "shape": np.array([N, M]),
           "omega": np.array(omega, np.float64), "surroundings": surr,
           "source": np.array("reference text via tests/golden/ref_extract.py")}
    if cells is not None:
        out["cells_0"] = np.asarray(cells, np.float32)  # else: the reference's own initial state
    done = 0
    for c in checkpoints:
        cells = ref.step(c - done)
        done = c
        u, p = ref.moments()
        out[f"rho_{c}"] = p.copy()
        out[f"u_{c}"] = u.copy()
        out[f"cells_sha_{c}"] = np.frombuffer(hashlib.sha256(cells.tobytes()).digest(), np.uint8)
    np.savez_compressed(os.path.join(HERE, f"fields_{tag}.npz"), **out)
    print(tag, "done")


def main():
    masks()
    # 64 x 128 cylinder-like scene, reference boundaries, reference parameters
    N, M = 64, 128
    b = onp.blocked_cylinder(N, M)
    fields("cyl64x128_reference", N, M, b, None, onp.OMEGA, [0, 0.1], "reference", (1, 10, 100, 1000))
    # unsteady variant (latticeboltzmann.jl:27-29): OMEGA = 0.01, u = [0, 0.3]; long enough for
    # the float64 omega of latticeboltzmann.py:98 to show (a float32 omega first differs after
    # step 900 of this scene)
    fields("cyl64x128_vortex", N, M, b, None, 0.01, [0, 0.3], "reference", (1, 10, 100, 1000, 2000))
    # under-relaxed, strongly perturbed: the omega product matters from the first step
    rng = np.random.default_rng(7)
    N, M = 40, 72
    b = onp.blocked_cylinder(N, M) | (rng.random((N, M)) < 0.03)
    b[0] = b[-1] = False
    fields("rand40x72_omega0p6", N, M, b, onp.perturbed_cells(N, M, seed=3, amp=0.05), 0.6, [0, 0.1],
           "reference", (1, 10, 100, 400))
    # periodic, random obstacles, perturbed start
    rng = np.random.default_rng(0)
    N, M = 48, 80
    b = rng.random((N, M)) < 0.05
    fields("rand48x80_periodic", N, M, b, onp.perturbed_cells(N, M, seed=0, amp=0.01), 0.1, [0, 0.1],
           "periodic", (1, 10, 100))


if __name__ == "__main__":
    main()
