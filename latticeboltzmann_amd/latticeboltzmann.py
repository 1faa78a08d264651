#!/usr/bin/env python3
"""Driver surface of the reference's latticeboltzmann.py, over the HIP library instead of pycuda.

Same knobs, same order of operations, same stdout line, citations into /root/reference:
  constants / getEquilibrium / ambient      latticeboltzmann.py:16-35, 80-95
  scene from an obstacle image              latticeboltzmann.py:38-47
  tuning presets                            latticeboltzmann.py:50-72  (an MI355X preset instead)
  cells, device upload                      latticeboltzmann.py:142-152
  the loop: launch / stats / frame / swap   latticeboltzmann.py:168-206
  Ctrl-C drain                              latticeboltzmann.py:208-219
What differs: everything is a function (the reference runs at import time and cannot be imported
without an NVIDIA GPU), geometry is a run-time parameter, and density / velocity can be read back
(the reference only reads RGB frames).  Video goes through imageio + cv2 when they are installed
(they are not in this image); otherwise frames are kept as raw .npy / handed to a callback.

    python -m latticeboltzmann_amd.latticeboltzmann scene.png [MI355X|MI355X_T1] [--steps N]
"""
import math
import os
import sys
import time
from itertools import count
from threading import Thread

import numpy as np

from . import lb_hip_kernel

dtype = np.float32
rtol = 1e-6

# Constants for D2Q9 (latticeboltzmann.py:24-35)
r2 = 3
w = np.array([1 / 36, 1 / 9, 1 / 36,
              1 / 9, 4 / 9, 1 / 9,
              1 / 36, 1 / 9, 1 / 36], dtype=dtype)
assert np.all(w == np.flip(w, axis=0))
assert math.isclose(sum(w), 1, rel_tol=rtol)
e = np.array([[-1, -1], [0, -1], [1, -1],
              [-1, 0], [0, 0], [1, 0],
              [-1, 1], [0, 1], [1, 1]])
assert np.all(e == -np.flip(e, axis=0))
assert (np.sum(e, axis=0) == [0, 0]).all()
e_f = np.asarray(e, dtype=dtype)

OMEGA = 0.00000000000001  # latticeboltzmann.py:80
p_ambient = 100
u_ambient = [0, 0.1]
OUTPUT_INTERVAL = 201     # latticeboltzmann.py:72

# latticeboltzmann.py:55-71 has RTX_2070 / RTX_3090 / A100_80GB; the MI355X presets only choose
# INNER_TIMESTEPS (launch geometry is the library's business).
TUNINGS = {
    "MI355X": {"INNER_TIMESTEPS": 6},     # LDS temporal blocking, one launch per 6 updates
    "MI355X_T1": {"INNER_TIMESTEPS": 1},  # one update per launch (pure bandwidth)
}


def getEquilibrium(velocity, density):
    """latticeboltzmann.py:87-89."""
    eu = velocity @ e_f.T
    return np.expand_dims(density, -1) * w * (
        1 + r2 * eu + r2 ** 2 / 2 * np.square(eu)
        - r2 / 2 * np.expand_dims(np.sum(np.square(velocity), axis=-1), -1))


def ambient(p=p_ambient, u=u_ambient):
    """surroundings / insides with the two conservation asserts (latticeboltzmann.py:92-95)."""
    s = getEquilibrium(np.array([u], dtype=dtype), np.array([p], dtype=dtype))[0]
    assert np.isclose(sum(s), p, rtol=rtol)
    assert np.isclose(s @ e_f / p, u, rtol=rtol).all()
    return s


def load_blocked(path):
    """latticeboltzmann.py:40-41."""
    from PIL import Image
    return np.array(Image.open(path).convert('L')) > 128


def cylinder_blocked(N=400, M=800):
    """The only cylinder scene the reference defines (latticeboltzmann.jl:25-32,51)."""
    return np.fromfunction(lambda y, x: (x - N / 2) ** 2 + (y - N / 2) ** 2 <= (N / 16) ** 2, (N, M))


class Simulation:
    """What latticeboltzmann.py holds in module globals between :125 and :206."""

    def __init__(self, blocked, tune="MI355X", omega=OMEGA, p=p_ambient, u=u_ambient,
                 boundary="reference", dtype=np.float32, device=0, inner_timesteps=None):
        if tune not in TUNINGS:
            raise NotImplementedError(f"Unknown tuning `{tune}`")  # latticeboltzmann.py:71
        self.blocked = np.ascontiguousarray(blocked, dtype=bool)
        self.N, self.M = self.blocked.shape
        self.INNER_TIMESTEPS = inner_timesteps or TUNINGS[tune]["INNER_TIMESTEPS"]
        self.surroundings = ambient(p, u)
        self.insides = ambient(p, u)
        # latticeboltzmann.py:142: every cell, blocked or not, starts at `insides`
        cells = np.where(np.expand_dims(self.blocked, -1),
                         np.array(self.insides, ndmin=3, dtype=np.float32),
                         np.array(self.insides, ndmin=3, dtype=np.float32))
        self.solver = lb_hip_kernel.Solver(self.N, self.M, omega, self.INNER_TIMESTEPS, boundary,
                                           dtype, device)
        # drv.to_device(...) x4, latticeboltzmann.py:147-150
        self.solver.set_surroundings(self.surroundings)
        self.solver.upload_blocked(self.blocked)
        self.solver.upload_cells(np.ascontiguousarray(cells))
        # two host frames, as the pinned frame1 / frame2 of latticeboltzmann.py:158-159
        self.frames = [lb_hip_kernel.pagelocked_empty((self.N, self.M, 3), np.uint8) for _ in range(2)]
        self.curr_iter = 0

    # latticeboltzmann.py:186-196: one outer iteration, rendering every OUTPUT_INTERVAL
    def iterate(self, render=False):
        frame = self.frames[self.curr_iter & 1] if render else None
        self.solver.fused_collide_stream_display(frame)
        self.curr_iter += 1
        return frame

    def synchronize(self):
        self.solver.sync()

    def moments(self, rows=None, cols=None):
        """(u [row, col], p) of the current state, as latticeboltzmann.py:180-183 computes them."""
        return self.solver.read_moments(rows, cols)

    def cells(self, rows=None, cols=None):
        return self.solver.read_cells(rows, cols)

    def close(self):
        self.solver.sync()
        for f in self.frames:
            lb_hip_kernel.release_pagelocked(f)
        self.frames = []
        self.solver.close()


def _hsv_to_rgb(hsv):
    """matplotlib.colors.hsv_to_rgb when matplotlib is installed (the reference's choice), else the
    same piecewise formula in numpy."""
    try:
        import matplotlib.colors
        return matplotlib.colors.hsv_to_rgb(hsv)
    except ImportError:
        h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
        i = (h * 6.0).astype(int)
        f = h * 6.0 - i
        p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
        i = i % 6
        sel = [i == k for k in range(6)]
        r = np.select(sel, [v, q, p, p, t, v])
        g = np.select(sel, [t, v, v, q, p, p])
        b = np.select(sel, [p, p, t, v, v, q])
        return np.stack([r, g, b], -1)


def display(u, p, video=None):
    """The host-side frame of the numpy path (latticeboltzmann.py:101-105): hue = flow direction,
    saturation = 100000 |u|, value = density, clipped to [0, 1].  (The device renderer,
    lbm_render / lbm_fused_collide_stream_display, follows the CUDA kernel's variant instead:
    lb_cuda_kernel.cu:308-319.)  Returns the uint8 (N, M, 3) frame and appends it to `video`."""
    h = (np.arctan2(u[..., 0], u[..., 1]) + math.pi) / 2 / math.pi
    s = np.sum(u ** 2, axis=-1) ** 0.5 * 100000
    v = p
    frame = (_hsv_to_rgb(np.clip(np.stack([h, s, v], 2), 0, 1)) * 255.99).astype(np.uint8)
    if video is not None:
        video.append_data(frame)
    return frame


def stats_line(N, M, inner_timesteps, dt, curr_iter):
    """The reference's 1000-iteration line (latticeboltzmann.py:176), same arithmetic."""
    return (f"{dt * 1000 / inner_timesteps} us per iteration /  "
            f"{N * M / 1000000 / dt * inner_timesteps} GLUPS /  "
            f"{2 * N * M / 1000000 * 9 * 4 / dt * inner_timesteps} GBps "
            f"(iter {(curr_iter + 1) * inner_timesteps})")


def _open_video(path):
    """The reference's writer (latticeboltzmann.py:45,166): frames shrunk by 8 with a cubic filter into
    `<image>.mp4` at 60 fps.  That needs imageio(-ffmpeg) and cv2; without them (this image) the same
    frames, shrunk by 8 with PIL's bicubic filter, are written as `<image>.frames/NNNNNNNN.png` (encode
    with e.g. `ffmpeg -framerate 60 -i %08d.png out.mp4`); with neither, None (the caller saves .npy)."""
    try:
        import cv2
        import imageio
        writer = imageio.get_writer(path + '.mp4', fps=60)  # latticeboltzmann.py:45

        def append(frame, M, N):
            writer.append_data(cv2.resize(frame, dsize=(M // 8, N // 8), interpolation=cv2.INTER_CUBIC))
        append.close = writer.close
        return append
    except ImportError:
        pass
    try:
        from PIL import Image
    except ImportError:
        return None
    outdir = path + '.frames'
    os.makedirs(outdir, exist_ok=True)
    count_ = [0]

    def append(frame, M, N):
        img = Image.fromarray(frame)
        img = img.resize((max(M // 8, 1), max(N // 8, 1)), Image.BICUBIC)
        img.save(os.path.join(outdir, f"{count_[0]:08d}.png"))
        count_[0] += 1
    append.close = lambda: None
    return append


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        raise ValueError("No base image provided")  # latticeboltzmann.py:47
    steps = None
    if "--steps" in argv:
        i = argv.index("--steps")
        steps = int(argv[i + 1])
        argv = argv[:i] + argv[i + 2:]
    path = argv[0]
    tune = argv[1] if len(argv) > 1 else "MI355X"
    blocked = load_blocked(path)
    print(blocked.shape)
    sim = Simulation(blocked, tune)
    video = _open_video(path)
    N, M, T = sim.N, sim.M, sim.INNER_TIMESTEPS
    threads = [None, None]

    def appendData(frame):  # latticeboltzmann.py:164-166
        sim.synchronize()
        if video is not None:
            video(frame, M, N)
        else:
            np.save(f"{path}.frame{sim.curr_iter:08d}.npy", frame[::8, ::8])

    prev_time = time.time()
    curr_iter = 0
    try:
        for curr_iter in (count() if steps is None else range(steps)):
            if curr_iter % 1000 == 999:
                sim.synchronize()
                curr_time = time.time()
                print(stats_line(N, M, T, curr_time - prev_time, curr_iter))
                prev_time = curr_time
            if curr_iter % OUTPUT_INTERVAL == 0:
                slot = curr_iter & 1
                if threads[slot] is not None:
                    threads[slot].join()
                frame = sim.iterate(render=True)
                threads[slot] = Thread(target=appendData, args=(frame,))
                threads[slot].start()
            else:
                sim.iterate()
    except KeyboardInterrupt:
        print("KeyboardInterrupt")
        print("current iteration: ", (curr_iter + 1) * T)
    print("Waiting for streams and threads to finish...")
    sim.synchronize()
    for t in threads:
        if t is not None:
            t.join()
    if video is not None:
        video.close()
    sim.close()
    print("Done")


if __name__ == "__main__":
    main()
