"""How ordinary does a scene stay?  Runs the reference's yeet.png scene (mask from tests/golden/masks.npz,
reference edges, OMEGA = 1e-14) and reports, every few hundred updates, the launch time and what share
of the cells would fail the common-path conditions of collide2_to (sign, density window, 1.5|u|^2 <= 1/8).
usage: scene_health.py [T] [launches per report] [reports]   (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from latticeboltzmann_amd.lb_hip_kernel import Solver
from oracle import lbm_numpy as onp

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
per = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reports = int(sys.argv[3]) if len(sys.argv) > 3 else 8
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "masks.npz"))
sh = tuple(int(v) for v in z["yeet_shape"])
blocked = np.unpackbits(z["yeet_bits"])[:sh[0] * sh[1]].reshape(sh).astype(bool)
N, M = sh
s = Solver(N, M, 1e-14, inner_timesteps=T, boundary="reference")
s.set_surroundings(onp.ambient())
s.init_equilibrium(100.0, (0.0, 0.1), amp=0.0, seed=0)
s.upload_blocked(blocked)
done = 0
for r in range(reports):
    ms = s.fused_collide_stream_timed(per)
    done += per * T
    cells = s.read_cells()
    free = ~blocked
    p = cells.sum(-1)
    with np.errstate(all="ignore"):
        u0 = (cells * onp.e_f[:, 0]).sum(-1) / p
        u1 = (cells * onp.e_f[:, 1]).sum(-1) / p
        c = 1.5 * (u0 * u0 + u1 * u1)
    neg = (cells < 0).any(-1)
    bad_c = ~(c <= 0.125)
    bad_p = ~((p >= 2.0 ** -60) & (p <= 2.0 ** 39))
    nonfinite = ~np.isfinite(cells).all(-1)
    out = neg | bad_c | bad_p
    # a wave-row of the LDS kernel = 128 consecutive columns of one row
    W = 128
    Mp = -(-M // W) * W
    o2 = np.zeros((N, Mp), bool); o2[:, :M] = out
    wave_rows = o2.reshape(N, Mp // W, W).any(-1).mean()
    print(f"after {done} updates: {ms / per:.3f} ms/launch {N * M * per * T / ms / 1e6:.1f} GLUPS | cells out of the "
          f"ordinary: {out.mean():.4%} (free {out[free].mean():.4%}, blocked {out[blocked].mean():.4%}); neg {neg.mean():.4%} "
          f"c>1/8 {bad_c.mean():.4%} p-window {bad_p.mean():.4%} nonfinite {nonfinite.mean():.4%}; wave-rows hit {wave_rows:.2%}; "
          f"max|u| {np.nanmax(np.sqrt(u0*u0+u1*u1)[free]):.3f} rho [{np.nanmin(p[free]):.3g}, {np.nanmax(p[free]):.3g}]", flush=True)
s.close()
