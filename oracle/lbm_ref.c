/* CPU oracle (TEST INFRASTRUCTURE, never the product path): plain-C restatement of the
 * reference's vectorised numpy D2Q9 BGK path, in the same floating-point operation order as
 * numpy evaluates it, so that it agrees with oracle/lbm_numpy.py bit for bit
 * (tests/test_oracle.py checks that).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * Citations are into /root/reference:
 *   constants      latticeboltzmann.py:24-35      getEquilibrium latticeboltzmann.py:87-89
 *   omega          latticeboltzmann.py:98         collide        latticeboltzmann.py:107-113
 *   stream         latticeboltzmann.py:115-120    reflect        latticeboltzmann.py:122-123
 *   moments        latticeboltzmann.py:180-183    loop order     latticeboltzmann.jl:107-120
 *
 * Parity status: field parity is UNPINNED by the reference (it ships no golden fields and cannot
 * run here); constants / equilibrium / masks are pinned by the reference's own asserts and images.
 *
 * numpy evaluation order restated here (measured against numpy 2.2.6 in this container):
 *   np.sum(cells, axis=2)  -> pairwise: ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7)) + a8
 *   cells @ e_f            -> k-sequential multiply-add starting from 0 (products by 0,+-1 exact)
 *   u @ e_f.T              -> u0*e[k][0] + u1*e[k][1]
 *   np.nan_to_num          -> nan -> 0, +-inf -> +-FLT_MAX
 *   cells *= omega         -> omega is float64 (:98): (float)((double)cell * omega), one rounding
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

static const float W[9] = {1.0f / 36, 1.0f / 9, 1.0f / 36, 1.0f / 9, 4.0f / 9,
                           1.0f / 9,  1.0f / 36, 1.0f / 9, 1.0f / 36};
/* e[k] = (dy, dx): e[k][0] shifts rows (axis 0), e[k][1] shifts columns (axis 1). */
static const int E[9][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {0, 0},
                            {1, 0},   {-1, 1}, {0, 1},  {1, 1}};

static inline float nan_to_num(float v) {
    if (isnan(v)) return 0.0f;
    if (isinf(v)) return v > 0 ? FLT_MAX : -FLT_MAX;
    return v;
}

/* latticeboltzmann.py:180-183 for one cell. */
static inline void cell_moments(const float* a, float* p, float* u0, float* u1) {
    float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])) + a[8];
    float m0 = 0.0f, m1 = 0.0f;
    for (int k = 0; k < 9; k++) {
        m0 = m0 + a[k] * (float)E[k][0];
        m1 = m1 + a[k] * (float)E[k][1];
    }
    *p = s;
    *u0 = nan_to_num(m0 / s);
    *u1 = nan_to_num(m1 / s);
}

/* latticeboltzmann.py:87-89 for one cell. */
static inline void cell_equilibrium(float u0, float u1, float p, float* eq) {
    float usq = u0 * u0 + u1 * u1;
    for (int k = 0; k < 9; k++) {
        float eu = u0 * (float)E[k][0] + u1 * (float)E[k][1];
        float br = ((1.0f + 3.0f * eu) + 4.5f * (eu * eu)) - 1.5f * usq;
        eq[k] = (p * W[k]) * br;
    }
}

void lbmref_equilibrium(float u0, float u1, float p, float* out9) {
    cell_equilibrium(u0, u1, p, out9);
}

void lbmref_moments(const float* cells, long ncells, float* rho, float* u) {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncells; c++)
        cell_moments(cells + 9 * c, rho + c, u + 2 * c, u + 2 * c + 1);
}

/* moments + collide (latticeboltzmann.py:180-183,107-113), in place. */
static void collide_all(float* cells, const uint8_t* blocked, long ncells, double omega_free) {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncells; c++) {
        float* a = cells + 9 * c;
        float p, u0, u1, eq[9];
        cell_moments(a, &p, &u0, &u1);
        cell_equilibrium(u0, u1, p, eq);
        /* omega is a float64 array in the reference (:98): `cells *= omega` is a double product
         * rounded once to float32 */
        double om = blocked[c] ? 1.0 : omega_free;
        for (int k = 0; k < 9; k++) {
            float v = a[k] - eq[k];
            v = (float)((double)v * om);
            a[k] = v + eq[k];
        }
    }
}

/* stream + reflect (latticeboltzmann.py:115-123): g -> f, pull form.
 * boundary 0 = reference rule (col 0 / row 0 <- surroundings for dx / dy = +1; last col / last row
 * keep their own value), 1 = periodic wrap (no reference counterpart). */
static void stream_reflect(const float* g, float* f, const uint8_t* blocked, int N, int M,
                           const float* surr, int boundary) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < N; y++) {
        for (int x = 0; x < M; x++) {
            float t[9];
            for (int k = 0; k < 9; k++) {
                int dy = E[k][0], dx = E[k][1];
                int ys = y - dy, xs = x - dx;
                if (boundary == 1) {
                    ys = (ys + N) % N;
                    xs = (xs + M) % M;
                    t[k] = g[((long)ys * M + xs) * 9 + k];
                } else if (ys >= 0 && ys < N && xs >= 0 && xs < M) {
                    t[k] = g[((long)ys * M + xs) * 9 + k];
                } else if ((dy == 1 && y == 0) || (dx == 1 && x == 0)) {
                    t[k] = surr[k];
                } else {
                    t[k] = g[((long)y * M + x) * 9 + k];
                }
            }
            float* o = f + ((long)y * M + x) * 9;
            if (blocked[(long)y * M + x])
                for (int k = 0; k < 9; k++) o[k] = t[8 - k];
            else
                for (int k = 0; k < 9; k++) o[k] = t[k];
        }
    }
}

/* Advance `cells` (N,M,9 float32, k fastest) by nsteps; `tmp` is scratch of the same size.
 * The result is left in `cells`. */
void lbmref_run(float* cells, float* tmp, const uint8_t* blocked, int N, int M, double omega_free,
                const float* surr, int boundary, int nsteps) {
    long ncells = (long)N * M;
    float *a = cells, *b = tmp;
    for (int s = 0; s < nsteps; s++) {
        collide_all(a, blocked, ncells, omega_free);
        stream_reflect(a, b, blocked, N, M, surr, boundary);
        float* t = a; a = b; b = t;
    }
    if (a != cells) memcpy(cells, a, (size_t)ncells * 9 * sizeof(float));
}

/* The two halves, exposed so tests can check the post-collision state the HIP path stores. */
void lbmref_collide(float* cells, const uint8_t* blocked, int N, int M, double omega_free) {
    collide_all(cells, blocked, (long)N * M, omega_free);
}
void lbmref_stream_reflect(const float* g, float* f, const uint8_t* blocked, int N, int M,
                           const float* surr, int boundary) {
    stream_reflect(g, f, blocked, N, M, surr, boundary);
}
