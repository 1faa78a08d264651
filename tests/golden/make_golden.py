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
  sha_cyl400x800_reference.npz
                   sha256 of cells / rho / u after 1, 100, 1000 steps of the same reference lines
                   on BASELINE config 1's own scene (400 x 800 cylinder, latticeboltzmann.jl:25-32):
                   digests only, so the fixture stays a few hundred bytes.
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


def sha_only(tag, N, M, blocked, omega, u_amb, checkpoints):
    """BASELINE config 1 at its own size: only sha256 digests of cells / rho / u (a few hundred
    bytes) after the given step counts of the reference's own lines, from its own initial state."""
    ref = ref_extract.ReferenceNumpyPath(blocked, {"OMEGA": omega, "u_ambient": u_amb})
    out = {"shape": np.array([N, M]), "omega": np.array(omega, np.float64),
           "surroundings": ref.surroundings, "blocked_sha": _sha(np.packbits(blocked)),
           "source": np.array("reference text via tests/golden/ref_extract.py")}
    done = 0
    for c in checkpoints:
        cells = ref.step(c - done)
        done = c
        u, p = ref.moments()
        out[f"cells_sha_{c}"] = _sha(cells)
        out[f"rho_sha_{c}"] = _sha(p)
        out[f"u_sha_{c}"] = _sha(u)
        print(tag, c, flush=True)
    np.savez_compressed(os.path.join(HERE, f"sha_{tag}.npz"), **out)


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "cyl400x800":  # only the (2 min) full-size cylinder digest
        sha_only("cyl400x800_reference", 400, 800, onp.blocked_cylinder(400, 800), onp.OMEGA, [0, 0.1],
                 (1, 100, 1000))
        return
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
    # BASELINE config 1 itself: the 400 x 800 cylinder of latticeboltzmann.jl:25-32, digests only
    sha_only("cyl400x800_reference", 400, 800, onp.blocked_cylinder(400, 800), onp.OMEGA, [0, 0.1],
             (1, 100, 1000))


if __name__ == "__main__":
    main()
