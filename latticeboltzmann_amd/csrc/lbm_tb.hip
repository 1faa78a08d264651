// Temporal blocking: T lattice updates per launch, intermediate time levels staged in LDS.
//
// The reference's INNER_TIMESTEPS (lb_cuda_kernel.cu:213-216,239,288-292) re-reads `newcells` from
// global memory between inner steps with a T-lane halo inside every 32-lane warp, and is racy
// across warps.  This is a different design for CDNA4 (160 KB LDS per CU, 64-lane waves):
//
// A workgroup owns a strip of NB*64 columns (NB AoSoA chunks) and a chunk of Ry rows and marches
// down the rows as a time-skewed pipeline.  In iteration i, stage s (1..T) produces time level s
// of row  q_s(i) = y0 - (T-1) + i - 2(s-1):
//     stage 1     pulls level 0 straight from HBM (the same coalesced 768-byte pieces as the T=1
//                 kernel), software-prefetched two iterations ahead in registers;
//     stage 2..T  pull from a 4-row LDS ring of level s-1;
//     stage s<T   writes its row into the LDS ring of level s, stage T stores to HBM (full lines).
// Waves: per stage NB "main" waves (one per 64-column block) and, for s < T, one "halo" wave
// whose first 2(T-s) lanes recompute the T-s columns on either side of the strip that the next
// level needs.  All waves of a stage share the row (wave-uniform) and differ only in a per-lane
// column, so they run the same code.  All stages run concurrently; one barrier per iteration
// separates an iteration's ring writes from the next iteration's reads (skew 2 per stage =>
// ring depth 4).  Per T updates the strip's cells are read once and written once: HBM bytes per
// cell-update = 72/T (+ halo columns and 3(T-1) warm-up rows per chunk).  Every cell goes through
// the same pull/reflect/collide device functions as the T=1 kernel, so results are bit-identical
// to T single updates.
#include "lbm_internal.h"

namespace lbm {

template <int T, int NB>
struct TB {
    static constexpr int SW = NB * 64;           // strip width
    static constexpr int HP = T - 1;             // halo columns kept per side
    static constexpr int RW = SW + 2 * HP;       // floats per slot row
    static constexpr int ROWF = Q * RW;          // floats per ring row
    static constexpr int LEVF = 4 * ROWF;        // floats per level (ring of 4 rows)
    static constexpr int LDSF = (T - 1) * LEVF;
    static constexpr int NW = T * (NB + 1) - 1;  // waves: NB main per stage + a halo wave for s < T
};

template <int XMODE, int T, int NB, typename ST>
__global__ __launch_bounds__((TB<T, NB>::NW * 64)) void step_tb_kernel(
    const ST* __restrict__ gin, ST* __restrict__ gout, const uint64_t* __restrict__ mask, Geom geo,
    int row0, int nrows, int Ry, int nstrips) {
    using C = TB<T, NB>;
    __shared__ float lds[C::LDSF];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = wg % nstrips, chunk = wg / nstrips;
    const int x0 = strip * C::SW;
    const int y0 = row0 + chunk * Ry;
    const int y1 = min(y0 + Ry, row0 + nrows);
    const int N = geo.N, M = geo.M, G = geo.G;
    const int Wv = min(C::SW, M - x0);
    const bool wrapy = geo.top == YEDGE_WRAP;  // top and bottom wrap together
    const int qlo = geo.top == YEDGE_DOMAIN ? 0 : -(1 << 28);  // rows that exist
    const int qhi = geo.bot == YEDGE_DOMAIN ? N : (1 << 28);

    // ---- role: stage s (wave-uniform) and this lane's column --------------------------------
    const int s = wave / (NB + 1) + 1;
    const int wv = wave % (NB + 1);
    int x;
    bool act;
    if (wv < NB) {
        x = x0 + wv * 64 + lane;
        act = x < M;
    } else {
        const int n = T - s;
        x = lane < n ? x0 - 1 - lane : x0 + Wv + (lane - n);
        act = lane < 2 * n;
        if (XMODE != XMODE_PERIODIC) act = act && x >= 0 && x < M;
    }
    int xn = x;
    if (XMODE == XMODE_PERIODIC) xn = x < 0 ? x + M : (x >= M ? x - M : x);
    if (!act) xn = 0;
    const int cx = x - x0 + C::HP;
    // x edge rule flags (reference mode only); a wave without any keeps the fast paths
    const bool ob_m = XMODE != XMODE_PERIODIC && x + 1 >= M;  // source of dx = -1 outside
    const bool ob_p = XMODE != XMODE_PERIODIC && x - 1 < 0;   // source of dx = +1 outside
    const bool wave_xedge = __any(act && (ob_m || ob_p));

    // per-lane element offsets inside a row of the global layout (stage 1, fast path)
    int xm1 = xn + 1, xp1 = xn - 1;
    if (XMODE == XMODE_PERIODIC) { xm1 = xm1 >= M ? 0 : xm1; xp1 = xp1 < 0 ? M - 1 : xp1; }
    else { xm1 = xm1 >= M ? xn : xm1; xp1 = xp1 < 0 ? xn : xp1; }
    const int off0 = (xn >> 6) * CHUNK + (xn & 63);
    const int offm = (xm1 >> 6) * CHUNK + (xm1 & 63);
    const int offp = (xp1 >> 6) * CHUNK + (xp1 & 63);
    const int mblk = xn >> 6;
    const unsigned mbit = xn & 63;

    // ---- rows: q = this stage's row in the current iteration, qn = its storage row ----------
    int q = y0 - (T - 1) - 2 * (s - 1);
    int qn = q;
    if (wrapy) { qn %= N; if (qn < 0) qn += N; }
    const int rlo = max(y0 - (T - s), qlo), rhi = min(y1 + (T - s), qhi);  // rows this stage makes
    auto norm = [&](int r) {  // storage row of logical row r (|r - [0,N)| < N)
        if (wrapy) { r = r < 0 ? r + N : r; r = r >= N ? r - N : r; }
        return r;
    };
    auto yedge = [&](int r) { return r - 1 < qlo || r + 1 >= qhi; };

    // stage 1: load the 9 pulled populations of logical row r (storage row rn) into dst
    auto load_row = [&](int r, int rn, float* dst) {
        if (wave_xedge || yedge(r)) {
            if (act) pull<XMODE>(gin, geo, rn, xn >> 6, xn & 63, dst);
        } else {
            const ST* r0p = gin + (long long)(rn + G) * geo.rowStride;
            const ST* rup = gin + (long long)(norm(rn - 1) + G) * geo.rowStride;
            const ST* rdp = gin + (long long)(norm(rn + 1) + G) * geo.rowStride;
            if (act) {
#pragma unroll
                for (int k = 0; k < Q; k++) {
                    const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * LANES;
                    const ST* rb = dy == 1 ? rup : (dy == 0 ? r0p : rdp);
                    const int off = dx == 1 ? offp : (dx == 0 ? off0 : offm);
                    dst[k] = ld(rb + off + sl);
                }
            }
        }
    };
    auto load_mask = [&](int rn) -> uint64_t {
        return act ? mask[(long long)(rn + G) * geo.Mb + mblk] : 0ull;
    };
    // stages 2..T: gather the 9 pulled populations of row r from the LDS ring of level s-1
    auto gather = [&](int r, float* a) {
        const float* lev = lds + (s - 2) * C::LEVF + cx;
        const int ro0 = (r & 3) * C::ROWF, rou = ((r - 1) & 3) * C::ROWF, rod = ((r + 1) & 3) * C::ROWF;
        if (wave_xedge || yedge(r)) {
            const bool ob_u = r - 1 < qlo, ob_d = r + 1 >= qhi;
#pragma unroll
            for (int k = 0; k < Q; k++) {
                const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * C::RW;
                const bool oy = dy == 1 ? ob_u : (dy == -1 ? ob_d : false);
                const bool ox = dx == 1 ? ob_p : (dx == -1 ? ob_m : false);
                const int src = (dy == 1 ? rou : (dy == 0 ? ro0 : rod)) - dx;
                const float v = lev[((ox || oy) ? ro0 : src) + sl];
                const bool inlet = (dx == 1 && ox) || (dy == 1 && oy);
                a[k] = inlet ? geo.surr[k] : v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < Q; k++) {
                const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * C::RW;
                a[k] = lev[(dy == 1 ? rou : (dy == 0 ? ro0 : rod)) - dx + sl];
            }
        }
    };

    // ---- prologue: prefetch rows of iterations 0 and 1 (stage 1), mask of iteration 0 ----------
    float pa[Q], pb[Q];
    if (s == 1) {
        if (q >= rlo && q < rhi) load_row(q, qn, pa);
        if (q + 1 >= rlo && q + 1 < rhi) load_row(q + 1, norm(qn + 1), pb);
    }
    uint64_t mw = (q >= rlo && q < rhi) ? load_mask(qn) : 0ull;

    const int niter = (y1 - y0) + 3 * (T - 1);

    auto body = [&](float* pre) {
        const bool go = q >= rlo && q < rhi;  // wave-uniform
        float a[Q];
        if (s == 1) {
#pragma unroll
            for (int k = 0; k < Q; k++) a[k] = pre[k];
            if (q + 2 >= rlo && q + 2 < rhi) load_row(q + 2, norm(norm(qn + 1) + 1), pre);
        } else if (go && act) {
            gather(q, a);
        }
        const uint64_t mcur = mw;
        if (q + 1 >= rlo && q + 1 < rhi) mw = load_mask(norm(qn + 1));
        if (go && act) {
            const bool blocked = (mcur >> mbit) & 1ull;
            reflect(a, blocked);
            collide(a, blocked, geo.omega);
            if (s < T) {
                float* o = lds + (s - 1) * C::LEVF + (q & 3) * C::ROWF + cx;
#pragma unroll
                for (int k = 0; k < Q; k++) o[slot_of(k) * C::RW] = a[k];
            } else {
                ST* o = gout + (long long)(qn + G) * geo.rowStride + off0;
#pragma unroll
                for (int k = 0; k < Q; k++) st_nt(o + slot_of(k) * LANES, a[k]);
            }
        }
        q += 1;
        qn = norm(qn + 1);
    };

    for (int i = 0; i < niter; i += 2) {
        body(pa);
        __syncthreads();
        if (i + 1 < niter) body(pb);
        __syncthreads();
    }
}

template <int XMODE, int T, int NB, typename ST>
static void launch_tb_t(const ST* gin, ST* gout, const uint64_t* mask, const Geom& geo, int row0,
                        int nrows, int Ry, hipStream_t st) {
    using C = TB<T, NB>;
    const int nstrips = (geo.M + C::SW - 1) / C::SW;
    const int nchunks = (nrows + Ry - 1) / Ry;
    hipLaunchKernelGGL((step_tb_kernel<XMODE, T, NB, ST>), dim3((unsigned)(nstrips * nchunks)),
                       dim3(C::NW * 64), 0, st, gin, gout, mask, geo, row0, nrows, Ry, nstrips);
}

template <int T, int NB>
static void launch_tb_T(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                        bool half, int row0, int nrows, int Ry, hipStream_t st) {
    if (xmode == XMODE_PERIODIC) {
        if (half) launch_tb_t<XMODE_PERIODIC, T, NB>((const __half*)gin, (__half*)gout, mask, geo, row0, nrows, Ry, st);
        else launch_tb_t<XMODE_PERIODIC, T, NB>((const float*)gin, (float*)gout, mask, geo, row0, nrows, Ry, st);
    } else {
        if (half) launch_tb_t<XMODE_REFERENCE, T, NB>((const __half*)gin, (__half*)gout, mask, geo, row0, nrows, Ry, st);
        else launch_tb_t<XMODE_REFERENCE, T, NB>((const float*)gin, (float*)gout, mask, geo, row0, nrows, Ry, st);
    }
}

// T updates of rows [row0, row0+nrows) in one launch; input rows [row0-T, row0+nrows+T) must be
// valid (or lie beyond a domain edge).  T in [2, TB_MAX_T].
hipError_t launch_step_tb(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                          bool half, int T, int row0, int nrows, int Ry, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    if (Ry < 1) Ry = 1;
    switch (T) {
        case 2: launch_tb_T<2, 4>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 3: launch_tb_T<3, 4>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 4: launch_tb_T<4, 3>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lbm
