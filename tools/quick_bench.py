"""Quick single-GPU throughput probe (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from latticeboltzmann_amd.lb_hip_kernel import Solver

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    boundary = sys.argv[4] if len(sys.argv) > 4 else "periodic"
    T = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    tb_rows = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    tb_max = int(sys.argv[7]) if len(sys.argv) > 7 else 6
    dtype = np.float16 if (len(sys.argv) > 8 and sys.argv[8] == "f16") else np.float32
    omega = float(sys.argv[9]) if len(sys.argv) > 9 else 1e-14
    s = Solver(N, M, omega, inner_timesteps=T, boundary=boundary, dtype=dtype)
    from oracle import lbm_numpy as onp
    s.set_surroundings(onp.ambient())
    s.set_tuning(tb_max, tb_rows)
    s.init_equilibrium(100.0, (0.0, 0.1), amp=0.01, seed=0)
    if len(sys.argv) > 10 and sys.argv[10] == "yeet":  # the reference's yeet.png mask, tiled (bench.py's config 3)
        z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "masks.npz"))
        sh = tuple(int(v) for v in z["yeet_shape"])
        yeet = np.unpackbits(z["yeet_bits"])[:sh[0] * sh[1]].reshape(sh)
        s.upload_blocked(np.tile(yeet, (-(-N // sh[0]), -(-M // sh[1])))[:N, :M])
    if len(sys.argv) > 10 and sys.argv[10].startswith("stripes"):  # a blocked column every k columns: every wave sees an obstacle
        k = int(sys.argv[10][7:] or 128)
        b = np.zeros((N, M), np.uint8); b[:, ::k] = 1
        s.upload_blocked(b)
    if len(sys.argv) > 10 and sys.argv[10].startswith("dots"):  # one blocked cell every k columns and k rows
        k = int(sys.argv[10][4:] or 128)
        b = np.zeros((N, M), np.uint8); b[::k, ::k] = 1
        s.upload_blocked(b)
    s.fused_collide_stream(3)
    s.sync()
    for rep in range(3):
        ms = s.fused_collide_stream_timed(steps)
        glups = N * M * steps * T / ms / 1e6
        print(f"N={N} M={M} T={T} omega={omega} {boundary} {np.dtype(dtype).name}: {ms/steps:.3f} ms/outer  {glups:.2f} GLUPS  {glups*72/1000:.3f} TB/s(72B)  frac={glups*72/8000:.3f}", flush=True)
    u, rho = s.read_moments(rows=(0, 4), cols=(0, 8))
    print(rho[0, :4], u[0, :2])
main()
