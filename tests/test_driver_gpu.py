"""GPU: the reference's own scene (the obstacle mask of its yeet.png, from tests/golden/masks.npz)
through the driver surface (latticeboltzmann_amd/latticeboltzmann.py): mask bit-exactness on the
device, multi-obstacle bounce-back with and without temporal blocking against the oracle, and the
driver loop with frames."""
import os

import numpy as np
import pytest

from oracle import lbm_numpy as onp
from oracle import lbm_ref as oc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def yeet_mask():
    z = np.load(os.path.join(GOLD, "masks.npz"))
    shape = tuple(z["yeet_shape"])
    return np.unpackbits(z["yeet_bits"])[:shape[0] * shape[1]].reshape(shape).astype(bool)


@pytest.mark.parametrize("tune,steps", [("MI355X_T1", 30), ("MI355X", 30)])
def test_yeet_scene_through_the_driver(tune, steps):
    from latticeboltzmann_amd import latticeboltzmann as lb
    blocked = yeet_mask()  # (1355, 3782): M is not a multiple of 64, 22 429 blocked cells
    sim = lb.Simulation(blocked, tune)
    assert np.array_equal(sim.solver.read_blocked(), blocked)  # mask bit-exact on the device
    assert np.array_equal(sim.surroundings, onp.ambient())
    T = sim.INNER_TIMESTEPS
    for _ in range(steps // T):
        sim.iterate()
    ref = onp.initial_cells(blocked, onp.ambient())
    oc.run(ref, blocked, (steps // T) * T, onp.OMEGA, onp.ambient(), "reference")
    u, p = sim.moments()
    ur, pr = onp.moments(ref)
    assert np.array_equal(p, pr) and np.array_equal(u, ur)
    rows = (600, 700)
    assert np.array_equal(sim.cells(rows=rows), ref[rows[0]:rows[1]])
    sim.close()


def test_driver_main_loop(tmp_path, capsys):
    """main(): scene from an image file, frames every OUTPUT_INTERVAL, the 1000-iteration stats
    line (latticeboltzmann.py:176) and the drain."""
    from PIL import Image
    from latticeboltzmann_amd import latticeboltzmann as lb
    blocked = onp.blocked_cylinder(96, 200)
    path = str(tmp_path / "scene.png")
    Image.fromarray((blocked * 255).astype(np.uint8)).save(path)
    assert np.array_equal(lb.load_blocked(path), blocked)
    lb.main([path, "MI355X", "--steps", "1001"])
    out = capsys.readouterr().out
    assert "(96, 200)" in out and "GLUPS" in out and "Done" in out
    frames = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npy") or f.endswith(".mp4"))
    pngdir = path + ".frames"
    if os.path.isdir(pngdir):  # no imageio / cv2: PNG frames shrunk by 8 (bicubic), as the reference's mp4 frames are
        frames += sorted(os.listdir(pngdir))
        im = Image.open(os.path.join(pngdir, frames[-1]))
        assert im.size == (200 // 8, 96 // 8) and im.mode == "RGB"
        assert len(os.listdir(pngdir)) == 1001 // lb.OUTPUT_INTERVAL + 1
    assert frames, "no frame was written"
