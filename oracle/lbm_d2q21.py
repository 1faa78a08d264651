"""CPU oracle (TEST INFRASTRUCTURE, never the product path) for the D2Q21 lattice of the reference's
draft /root/reference/latticeboltzmann.jl.

PARITY UNPINNED, by nature.  The draft is not a runnable program in any language (its first 45 lines
are Julia with prose inside an expression, :38; the rest is the older Python verbatim), and the
reference holds no output of it.  What it does pin, and tests/test_d2q21_cpu.py checks:
  r2 = 3/2                          latticeboltzmann.jl:4
  the 21 weights                    latticeboltzmann.jl:5-11   and their asserts :12-13
  the 21 velocities                 latticeboltzmann.jl:14-20  and their asserts :21-22
Everything else is the reference's numpy algorithm (latticeboltzmann.py, as restated for D2Q9 in
oracle/lbm_numpy.py) with those tables put in, which is what the draft's own body (:35-45, :58-75)
spells: getEquilibrium with the general r2 (latticeboltzmann.py:87-89), moments (:180-183), collide
with the float64 omega (:98, :107-113), stream (:115-120: its slices verbatim, or np.roll for the
periodic form), reflect (:122-123), order latticeboltzmann.jl:107-120.

Two places where "numpy would decide" are decided here instead, because with velocity components
of 2 and 3 the products are no longer exact and BLAS does not define its summation order or its use
of FMA: `eu = velocity @ e_f.T` and `cells @ e_f` are evaluated WITHOUT matmul, as float32
elementwise products (each rounded) added in k order (`_dot_e`, `_momentum`).  The density is
np.sum over the contiguous last axis as in the reference (numpy's pairwise order for 21 terms:
eight running sums, a tree, then the last five terms one by one - `pairwise21` restates it and the
tests check that it IS what np.sum does).
"""
import numpy as np

dtype = np.float32
Q = 21

# latticeboltzmann.jl:4-11
r2 = 3 / 2
w = np.array([1 / 1620,
              1 / 432, 7 / 360, 1 / 432,
              2 / 27, 1 / 12, 2 / 27,
              1 / 1620, 7 / 360, 1 / 12, 91 / 324, 1 / 12, 7 / 360, 1 / 1620,
              2 / 27, 1 / 12, 2 / 27,
              1 / 432, 7 / 360, 1 / 432,
              1 / 1620], dtype=dtype)
# latticeboltzmann.jl:14-20; e[k] = (dy, dx) as latticeboltzmann.py:116 unpacks it
e = np.array([[0, -3],
              [-2, -2], [0, -2], [2, -2],
              [-1, -1], [0, -1], [1, -1],
              [-3, 0], [-2, 0], [-1, 0], [0, 0], [1, 0], [2, 0], [3, 0],
              [-1, 1], [0, 1], [1, 1],
              [-2, 2], [0, 2], [2, 2],
              [0, 3]])
e_f = np.asarray(e, dtype=dtype)

OMEGA = 0.01          # latticeboltzmann.jl:27
p_ambient = 100       # latticeboltzmann.jl:28
u_ambient = [0, 0.3]  # latticeboltzmann.jl:29


def lattice_asserts():
    """latticeboltzmann.jl:12-13, 21-22."""
    assert np.all(w == w[::-1])
    assert np.isclose(float(np.sum(w.astype(np.float64))), 1, rtol=1e-6)
    assert np.all(e == -e[::-1])
    assert (np.sum(e, axis=0) == [0, 0]).all()


def _dot_e(velocity):
    """eu[..., k] = velocity . e[k]: float32 products, each rounded, added row part first."""
    v0 = velocity[..., 0:1].astype(dtype)
    v1 = velocity[..., 1:2].astype(dtype)
    return v0 * e_f[:, 0] + v1 * e_f[:, 1]


def getEquilibrium(velocity, density):
    """latticeboltzmann.py:87-89 with r2 = 3/2: p w (1 + r2 eu + r2^2/2 eu^2 - r2/2 |u|^2)."""
    velocity = np.asarray(velocity, dtype)
    density = np.asarray(density, dtype)
    eu = _dot_e(velocity)
    return np.expand_dims(density, -1) * w * (
        1 + dtype(r2) * eu + dtype(r2 ** 2 / 2) * np.square(eu)
        - dtype(r2 / 2) * np.expand_dims(np.sum(np.square(velocity), axis=-1), -1))


def ambient(p=p_ambient, u=u_ambient):
    return getEquilibrium(np.array([u], dtype=dtype), np.array([p], dtype=dtype))[0]


def pairwise21(a):
    """np.sum over 21 contiguous float32, spelled out (numpy's pairwise_sum for 8 <= n <= 128)."""
    r = [a[..., j] + a[..., 8 + j] for j in range(8)]
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    for i in range(16, 21):
        res = res + a[..., i]
    return res


def _momentum(cells):
    """cells @ e_f without matmul: m_c = sum_k cells[..., k] * e[k][c], products rounded to float32
    and added in k order starting from 0 (terms with e[k][c] == 0 are skipped)."""
    m = np.zeros(cells.shape[:-1] + (2,), dtype)
    for c in range(2):
        acc = np.zeros(cells.shape[:-1], dtype)
        for k in range(Q):
            if e[k][c] != 0:
                acc = acc + cells[..., k] * e_f[k, c]
        m[..., c] = acc
    return m


def moments(cells):
    """latticeboltzmann.py:180-183: p = sum, u = momentum / p, nan_to_num."""
    p = np.sum(cells, axis=2)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        u = _momentum(cells) / np.expand_dims(p, -1)
    np.nan_to_num(u, copy=False)
    return u, p


def make_omega(blocked, omega_free=OMEGA):
    """latticeboltzmann.py:98: float64, 1 at blocked cells."""
    return np.where(np.expand_dims(blocked, -1), np.array(1, ndmin=3), np.array(omega_free, ndmin=3))


def collide(cells, u, p, omega):
    equilibrium = getEquilibrium(u, p)
    with np.errstate(all='ignore'):
        cells -= equilibrium
        cells *= omega
        cells += equilibrium


def stream(cells, surroundings):
    """latticeboltzmann.py:115-120, verbatim, with this lattice's velocities (up to three cells): the
    first dx columns / dy rows take surroundings[k] for dx, dy > 0; the slices for dx, dy < 0
    (`-2:0`, `-3:0`, ...) are empty, so the last |dx| columns / |dy| rows keep their own value."""
    N, M = cells.shape[:2]
    for k, (dy, dx) in enumerate(e):
        cells[max(dy, 0):N + min(dy, 0), max(dx, 0):M + min(dx, 0), k] = \
            cells[max(-dy, 0):N + min(-dy, 0), max(-dx, 0):M + min(-dx, 0), k]
        cells[:, min(dx, 0):max(dx, 0), k] = surroundings[k]
        cells[min(dy, 0):max(dy, 0), :, k] = surroundings[k]


def stream_periodic(cells):
    for k, (dy, dx) in enumerate(e):
        cells[..., k] = np.roll(cells[..., k], (dy, dx), axis=(0, 1))


def reflect(cells, blocked):
    cells[blocked] = np.flip(cells[blocked], axis=-1)


def step(cells, blocked, omega, surroundings=None, boundary="periodic"):
    u, p = moments(cells)
    collide(cells, u, p, omega)
    if boundary == "periodic":
        stream_periodic(cells)
    else:
        stream(cells, surroundings)
    reflect(cells, blocked)


def run(cells, blocked, steps, omega_free=OMEGA, surroundings=None, boundary="periodic"):
    omega = make_omega(blocked, omega_free)
    if boundary != "periodic" and surroundings is None:
        surroundings = ambient()
    for _ in range(steps):
        step(cells, blocked, omega, surroundings, boundary)
    return cells


def blocked_cylinder(N=400, M=800):
    """latticeboltzmann.jl:30-32, 51: (x - N/2)^2 + (y - N/2)^2 <= (N/16)^2."""
    def isBlocked(y, x):
        return (x - N / 2) ** 2 + (y - N / 2) ** 2 <= (N / 16) ** 2
    return np.fromfunction(isBlocked, (N, M))


def perturbed_cells(N, M, seed=0, amp=0.01, p=p_ambient, u=u_ambient):
    rng = np.random.default_rng(seed)
    rho = (p * (1 + amp * (2 * rng.random((N, M)) - 1))).astype(dtype)
    vel = np.broadcast_to(np.asarray(u, dtype), (N, M, 2)) + (amp * (2 * rng.random((N, M, 2)) - 1)).astype(dtype)
    return np.ascontiguousarray(getEquilibrium(vel.astype(dtype), rho), dtype)
