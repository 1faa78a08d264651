"""CPU oracle (TEST INFRASTRUCTURE, never the product path): numpy restatement of the
reference's vectorised D2Q9 BGK path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it restates (all citations are into /root/reference):
  constants           latticeboltzmann.py:16-35
  mask from an image  latticeboltzmann.py:40-41
  getEquilibrium      latticeboltzmann.py:87-89
  surroundings        latticeboltzmann.py:92-95
  omega               latticeboltzmann.py:98
  collide             latticeboltzmann.py:107-113
  stream              latticeboltzmann.py:115-120   (incl. the empty `-1:0` slices)
  reflect             latticeboltzmann.py:122-123
  initial cells       latticeboltzmann.py:142
  moments             latticeboltzmann.py:180-183
  loop order          latticeboltzmann.jl:107-120   (moments, collide, display, stream, reflect)
  cylinder scene      latticeboltzmann.jl:25-32,51

Pinning status: PINNED to the reference's own text.  The reference cannot be imported as a module
(top-level `import pycuda.autoinit`, latticeboltzmann.py:11, and the numpy path is commented out)
and it ships no golden fields, but its numpy lines can be executed from its file:
tests/golden/ref_extract.py reads /root/reference/latticeboltzmann.py at run time (storing none
of it), takes :16-35, :80-95, :142 verbatim and the commented-out :98, :107-123, :180-183 with the
`# ` removed, and steps them in the order of latticeboltzmann.jl:107-120.  The golden fields under
tests/golden/ (fields_*.npz, sha_cyl400x800_reference.npz) are the OUTPUT OF THOSE LINES
(tests/golden/make_golden.py), and tests/test_oracle.py checks that this restatement and its C
twin reproduce them bit for bit (test_golden_fields, test_config1_cylinder_digests), live on
further scenes in the build container (test_live_reference_text_equals_the_restatement), besides
the four lattice asserts (:28-29, :33-34), the two conservation asserts (:93-94) and the obstacle
masks of the reference's two scene images.

`boundary="periodic"` has no counterpart in the reference (BASELINE.json configs 2-4 name a
periodic channel); it replaces the stream step's edge rule by a wrap-around (np.roll).
"""
import math

import numpy as np

dtype = np.float32
rtol = 1e-6  # latticeboltzmann.py:18-21 (fp32 branch)

# Constants for D2Q9 (latticeboltzmann.py:24-35)
r2 = 3
w = np.array([1 / 36, 1 / 9, 1 / 36,
              1 / 9, 4 / 9, 1 / 9,
              1 / 36, 1 / 9, 1 / 36], dtype=dtype)
e = np.array([[-1, -1], [0, -1], [1, -1],
              [-1, 0], [0, 0], [1, 0],
              [-1, 1], [0, 1], [1, 1]])
e_f = np.asarray(e, dtype=dtype)

OMEGA = 0.00000000000001  # latticeboltzmann.py:80
p_ambient = 100           # latticeboltzmann.py:81
u_ambient = [0, 0.1]      # latticeboltzmann.py:82


def lattice_asserts():
    """The reference's import-time asserts (latticeboltzmann.py:28-29,33-34)."""
    assert np.all(w == np.flip(w, axis=0))
    assert math.isclose(sum(w), 1, rel_tol=rtol)
    assert np.all(e == -np.flip(e, axis=0))
    assert (np.sum(e, axis=0) == [0, 0]).all()


def getEquilibrium(velocity, density):
    """latticeboltzmann.py:87-89."""
    eu = velocity @ e_f.T
    return np.expand_dims(density, -1) * w * (
        1 + r2 * eu + r2 ** 2 / 2 * np.square(eu)
        - r2 / 2 * np.expand_dims(np.sum(np.square(velocity), axis=-1), -1))


def ambient(p=p_ambient, u=u_ambient):
    """surroundings / insides, latticeboltzmann.py:92-95 (with its two asserts)."""
    s = getEquilibrium(np.array([u], dtype=dtype), np.array([p], dtype=dtype))[0]
    assert np.isclose(sum(s), p, rtol=rtol)
    assert np.isclose(s @ e_f / p, u, rtol=rtol).all()
    return s


def blocked_from_image(path):
    """latticeboltzmann.py:40-41."""
    from PIL import Image
    return np.array(Image.open(path).convert('L')) > 128


def blocked_cylinder(N=400, M=800):
    """latticeboltzmann.jl:25-32,51: np.fromfunction(isBlocked, (N, M))."""
    def isBlocked(y, x):
        return (x - N / 2) ** 2 + (y - N / 2) ** 2 <= (N / 16) ** 2
    return np.fromfunction(isBlocked, (N, M))


def make_omega(blocked, omega_free=OMEGA):
    """latticeboltzmann.py:98, literally: an (N, M, 1) FLOAT64 array (np.array(1) is int64,
    np.array(OMEGA) float64).  `cells *= omega` (:112) therefore multiplies float32 cells by a
    float64 factor: numpy evaluates the product in double and rounds it once to float32."""
    return np.where(np.expand_dims(blocked, -1), np.array(1, ndmin=3),
                    np.array(omega_free, ndmin=3))


def initial_cells(blocked, insides):
    """latticeboltzmann.py:142: every cell (blocked or not) starts at `insides`."""
    return np.where(np.expand_dims(blocked, -1), np.array(insides, ndmin=3, dtype=dtype),
                    np.array(insides, ndmin=3, dtype=dtype)).copy()


def moments(cells):
    """latticeboltzmann.py:180-183.  Returns (u (N,M,2) [row, col], p (N,M))."""
    p = np.sum(cells, axis=2)
    with np.errstate(divide='ignore', invalid='ignore'):
        u = cells @ e_f / np.expand_dims(p, -1)
    np.nan_to_num(u, copy=False)
    return u, p


def collide(cells, u, p, omega):
    """latticeboltzmann.py:107-113 (in place)."""
    equilibrium = getEquilibrium(u, p)
    cells -= equilibrium
    cells *= omega
    cells += equilibrium


def stream(cells, surroundings):
    """latticeboltzmann.py:115-120 (in place).  For dx/dy = -1 the fill slice is `-1:0`, which is
    empty: the last column / last row keep their own post-collision value."""
    N, M = cells.shape[0], cells.shape[1]
    for k, (dy, dx) in enumerate(e):
        cells[max(dy, 0):N + min(dy, 0), max(dx, 0):M + min(dx, 0), k] = \
            cells[max(-dy, 0):N + min(-dy, 0), max(-dx, 0):M + min(-dx, 0), k]
        cells[:, min(dx, 0):max(dx, 0), k] = surroundings[k]
        cells[min(dy, 0):max(dy, 0), :, k] = surroundings[k]


def stream_periodic(cells):
    """No reference counterpart: wrap-around streaming for the periodic-channel configs."""
    for k, (dy, dx) in enumerate(e):
        cells[:, :, k] = np.roll(cells[:, :, k], (dy, dx), axis=(0, 1))


def reflect(cells, blocked):
    """latticeboltzmann.py:122-123 (in place)."""
    cells[blocked] = np.flip(cells[blocked], axis=-1)


def step(cells, blocked, omega, surroundings, boundary="reference"):
    """One timestep in the reference's order (latticeboltzmann.jl:107-120)."""
    u, p = moments(cells)
    collide(cells, u, p, omega)
    if boundary == "reference":
        stream(cells, surroundings)
    elif boundary == "periodic":
        stream_periodic(cells)
    else:
        raise ValueError(boundary)
    reflect(cells, blocked)


def run(cells, blocked, steps, omega_free=OMEGA, surroundings=None, boundary="reference"):
    """Advance `cells` in place by `steps` timesteps; returns cells."""
    if surroundings is None:
        surroundings = ambient()
    omega = make_omega(blocked, omega_free)
    for _ in range(steps):
        step(cells, blocked, omega, surroundings, boundary)
    return cells


def perturbed_cells(N, M, seed=0, amp=0.01, p=p_ambient, u=u_ambient):
    """Bench/parity scene of SURVEY 8(d)(2): equilibrium at (p, u) with a seeded +-amp relative
    perturbation of the density, so fields are neither constant nor compressible."""
    rng = np.random.default_rng(seed)
    rho = (p * (1 + amp * (2 * rng.random((N, M), dtype=np.float32) - 1))).astype(dtype)
    vel = np.broadcast_to(np.array(u, dtype=dtype), (N, M, 2))
    return getEquilibrium(vel, rho).astype(dtype)


def rel_linf(a, b):
    """Global-max-normalised L-inf: max|a-b| / max|b| (hard part 1)."""
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) /
                 np.max(np.abs(b.astype(np.float64))))


# ---------------------------------------------------------------------------------------------
# Kernel-semantics compatibility mode: what the reference's CUDA kernel computes
# where it differs from the numpy path above.  PARITY-UNPINNED: the reference holds no fixture for
# it and the kernel cannot be built or run here (nvcc, an NVIDIA GPU); it is compiled with
# --use_fast_math (approximate division, FMA contraction, flush-to-zero), which no IEEE
# restatement reproduces bit for bit.  This restates its STRUCTURE in IEEE float32, in the
# kernel's own operation order, for INNER_TIMESTEPS = 1 (lb_cuda_kernel.cu:230-365):
#   * density = sum + 0.0001f (:296); velocity by the kernel's sums (:293-300)
#   * equilibrium d*w*(c + 3 eu (1 + 1.5 eu)), c = 1 - 1.5|u|^2 (:323-330)
#   * the row that is collided in loop iteration y is row y+1, but the gate tests blocked[y][x]
#     (:322 against :289): row r is collided iff r >= 1 and not blocked[r-1][x]; row 0 is never
#     collided (:302-306 `continue` in front of the collide)
#   * sources outside the domain are `surr` on all four sides (:215,232-233,286-287); the one below
#     the last row is `surr` passed through the collide (gated by blocked[N-1][x]); edge lanes
#     (x = -1, x = M) are never collided
#   * stream and reflect as the numpy path (:336-357)
# ---------------------------------------------------------------------------------------------
def cuda_kernel_collide(a, omega_free=OMEGA):
    """lb_cuda_kernel.cu:293-300,323-331 on cells a (..., 9), k = 3*i + j = next.d[i][j]."""
    f32 = np.float32
    a = np.asarray(a, dtype=f32)
    s1 = (a[..., 0] + a[..., 1]) + a[..., 2]
    s2 = (a[..., 3] + a[..., 4]) + a[..., 5]
    s3 = (a[..., 6] + a[..., 7]) + a[..., 8]
    d = ((s1 + s2) + s3) + f32(0.0001)
    with np.errstate(divide='ignore', invalid='ignore'):
        uy = (s3 - s1) / d
        ux = (((((a[..., 2] + a[..., 5]) + a[..., 8]) - a[..., 0]) - a[..., 3]) - a[..., 6]) / d
    c = f32(1) - f32(1.5) * (ux * ux + uy * uy)
    om = f32(omega_free)  # `#define OMEGA {OMEGA}f`, latticeboltzmann.py:129
    out = np.empty_like(a)
    for i in range(3):
        for j in range(3):
            eu = f32(i - 1) * uy + f32(j - 1) * ux
            eq = (d * w[3 * i + j]) * (c + (f32(3) * eu) * (f32(1) + f32(1.5) * eu))
            out[..., 3 * i + j] = (a[..., 3 * i + j] - eq) * om + eq
    return out


def cuda_kernel_gate(blocked):
    """Which cells the kernel collides: row r >= 1 with blocked[r-1][x] clear."""
    gate = np.ones(blocked.shape, bool)
    gate[0] = False
    gate[1:] &= ~blocked[:-1]
    return gate


def cuda_kernel_step(cells, blocked, surroundings, omega_free=OMEGA):
    """One launch of the reference kernel with INNER_TIMESTEPS = 1 (structure only, see above):
    returns the new cells."""
    f32 = np.float32
    N, M = blocked.shape
    surr = np.asarray(surroundings, dtype=f32)
    h = np.where(cuda_kernel_gate(blocked)[..., None], cuda_kernel_collide(cells, omega_free), cells)
    P = np.empty((N + 2, M + 2, 9), f32)
    P[:] = surr
    P[1:-1, 1:-1] = h
    P[-1, 1:-1] = np.where(blocked[-1][:, None], surr, cuda_kernel_collide(surr, omega_free))
    out = np.empty((N, M, 9), f32)
    for k, (dy, dx) in enumerate(e):
        out[:, :, k] = P[1 - dy:N + 1 - dy, 1 - dx:M + 1 - dx, k]
    reflect(out, blocked)
    return out


def cuda_kernel_step_loops(cells, blocked, surroundings, omega_free=OMEGA):
    """The same launch as the kernel's own loop nest (one "thread" per column marching down the
    rows with prev / curr / next, lb_cuda_kernel.cu:230-365, INNER_TIMESTEPS = 1), pure Python:
    small grids only.  Pins cuda_kernel_step() against a literal reading of the kernel."""
    f32 = np.float32
    N, M = blocked.shape
    surr = np.asarray(surroundings, dtype=f32)
    new = np.empty((N, M, 9), f32)
    lanes = range(-1, M + 1)  # x = -1 and x = M are edge lanes
    prev = {x: surr.copy() for x in lanes}
    curr = {x: surr.copy() for x in lanes}
    for y in range(-1, N):
        nxt = {}
        for x in lanes:
            edge = x < 0 or x >= M
            n = surr.copy() if (edge or y >= N - 1) else cells[y + 1, x].astype(f32).copy()
            if y != -1 and not edge and not blocked[y, x]:
                n = cuda_kernel_collide(n, omega_free)
            nxt[x] = n
        if y == -1:
            prev, curr = curr, nxt
            continue
        for x in range(M):
            nc = np.empty(9, f32)
            for j, src in enumerate((nxt, curr, prev)):   # d[i][j]: j = 0 next, 1 curr, 2 prev
                nc[0 + j] = src[x + 1][0 + j]              # i = 0: __shfl_down (lane x+1)
                nc[3 + j] = src[x][3 + j]                  # i = 1: own lane
                nc[6 + j] = src[x - 1][6 + j]              # i = 2: __shfl_up (lane x-1)
            if blocked[y, x]:
                nc = nc[::-1].copy()
            new[y, x] = nc
        prev, curr = curr, nxt
    return new


def cuda_kernel_frame(u, p, blocked):
    """The RGB frame of the reference's CUDA kernel (lb_cuda_kernel.cu:27-67 hsv_to_rgb, :308-319):
    h = atan2(uy, ux) + PI with the kernel's `uy` = column (x) and `ux` = row (y) component,
    s = saturate(1000 |u|), v = saturate(density), black where blocked; float32 in the kernel's
    operation order, round-to-nearest * 255.  u (N, M, 2) [row, col], p (N, M) -> uint8 (N, M, 3).
    PARITY-UNPINNED: the reference holds no frame fixture (its videos are not in the tree) and the
    kernel uses --use_fast_math (atan2f / sqrtf approximations): the device frame is compared to
    this with a tolerance of one count per channel."""
    f32 = np.float32
    PI = f32(3.141592653589)
    u0, u1 = u[..., 0], u[..., 1]
    h = (np.arctan2(u1, u0).astype(f32) + PI).astype(f32)
    s = np.clip(f32(1000) * np.sqrt(u0 * u0 + u1 * u1), 0, 1).astype(f32)
    v = np.clip(p, 0, 1).astype(f32)
    h, s, v = (np.where(blocked, f32(0), a) for a in (h, s, v))
    f = (h / (PI / f32(3))).astype(f32)
    hi = np.floor(f)
    f = f - hi
    pp, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    sel = [(hi == 0) | (hi == 6), hi == 1, hi == 2, hi == 3, hi == 4]
    r = np.select(sel, [v, q, pp, pp, t], v)
    g = np.select(sel, [t, v, v, q, pp], pp)
    b = np.select(sel, [pp, pp, t, v, v], q)
    return np.stack([np.rint(f32(255) * c) for c in (r, g, b)], -1).astype(np.uint8)
