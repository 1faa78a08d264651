// Temporal blocking: T lattice updates per launch, intermediate time levels staged in LDS.
//
// The reference's INNER_TIMESTEPS (lb_cuda_kernel.cu:213-216,239,288-292) re-reads `newcells` from
// global memory between inner steps with a T-lane halo inside every 32-lane warp, and is racy
// across warps.  This is a different design for CDNA4 (160 KB LDS per CU, 64-lane waves):
//
// A workgroup owns a strip of TBW = 256 columns (4 AoSoA chunks) and a chunk of Ry rows and
// marches down the rows as a time-skewed pipeline.  In iteration i, stage s (1..T) produces time
// level s of row  q_s(i) = y0 - (T-1) + i - 2(s-1):
//     stage 1     pulls level 0 straight from HBM (the same coalesced 768-byte pieces as the T=1
//                 kernel), software-prefetched two iterations ahead in registers;
//     stage 2..T  pull from a 4-row LDS ring of level s-1;
//     stage s<T   writes its row into the LDS ring of level s, stage T stores to HBM (full lines).
// One wave per (stage, 64-column block): 4T waves, plus one "halo wave" that recomputes the
// T-s columns on each side of the strip that level s needs (T(T-1) cells per iteration).  All
// stages run concurrently; one barrier per iteration separates an iteration's ring writes from
// the next iteration's reads (skew 2 per stage => ring depth 4).  Per T updates the strip's cells
// are read once and written once: HBM bytes per cell-update = 72/T (+ halo columns and 3(T-1)
// warm-up rows per chunk).  Every cell goes through the same pull/reflect/collide device functions
// as the T=1 kernel, so results are bit-identical to T single updates.
#include "lbm_internal.h"

namespace lbm {

constexpr int TBW = 256;

template <int T>
struct TB {
    static constexpr int HP = T - 1;           // halo columns kept per side
    static constexpr int RW = TBW + 2 * HP;    // floats per slot row
    static constexpr int ROWF = Q * RW;        // floats per ring row
    static constexpr int LEVF = 4 * ROWF;      // floats per level (ring of 4 rows)
    static constexpr int LDSF = (T - 1) * LEVF;
    static constexpr int NROLES = 4 * T + 1;
    static constexpr int NW = NROLES <= 16 ? NROLES : (NROLES + 1) / 2;
    static constexpr int RPW = (NROLES + NW - 1) / NW;  // roles per wave
};

template <int XMODE, int T, typename ST>
__global__ __launch_bounds__(TB<T>::NW * 64) void step_tb_kernel(
    const ST* __restrict__ gin, ST* __restrict__ gout, const uint64_t* __restrict__ mask, Geom geo,
    int row0, int nrows, int Ry, int nstrips) {
    using C = TB<T>;
    __shared__ float lds[C::LDSF];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = wg % nstrips, chunk = wg / nstrips;
    const int x0 = strip * TBW;
    const int y0 = row0 + chunk * Ry;
    const int y1 = min(y0 + Ry, row0 + nrows);
    const int N = geo.N, M = geo.M;
    const int Wv = min(TBW, M - x0);
    const bool wrapy = geo.top == YEDGE_WRAP;  // top and bottom wrap together
    const int qlo = geo.top == YEDGE_DOMAIN ? 0 : -(1 << 28);   // rows that exist
    const int qhi = geo.bot == YEDGE_DOMAIN ? N : (1 << 28);

    // per-role cell descriptors (compile-time indexed)
    int rs[C::RPW], rx[C::RPW], rxn[C::RPW], rq[C::RPW], rqn[C::RPW];
    bool ract[C::RPW];
    float pa[C::RPW][Q], pb[C::RPW][Q];
#pragma unroll
    for (int r = 0; r < C::RPW; r++) {
        const int role = wave + r * C::NW;
        int s = 1, x = 0;
        bool act = false;
        if (role < 4 * T) {
            s = role / 4 + 1;
            x = x0 + (role & 3) * 64 + lane;
            act = x < M;
        } else if (role == 4 * T) {
            int idx = lane;
#pragma unroll
            for (int ss = 1; ss < T; ss++) {
                const int n = T - ss;
                if (idx >= 0 && idx < 2 * n) {
                    s = ss;
                    x = idx < n ? x0 - 1 - idx : x0 + Wv + (idx - n);
                    act = true;
                }
                idx -= 2 * n;
            }
            if (XMODE != XMODE_PERIODIC) act = act && x >= 0 && x < M;
        }
        int xn = x;
        if (XMODE == XMODE_PERIODIC) {
            xn = x < 0 ? x + M : (x >= M ? x - M : x);
        }
        rs[r] = s; rx[r] = x; rxn[r] = xn; ract[r] = act;
        rq[r] = y0 - (T - 1) - 2 * (s - 1);  // row of iteration 0
        int qn = rq[r];
        if (wrapy) { qn %= N; if (qn < 0) qn += N; }
        rqn[r] = qn;
    }

    auto row_active = [&](int s, int q) {
        return q >= y0 - (T - s) && q < y1 + (T - s) && q >= qlo && q < qhi;
    };
    // stage-1 prefetch of the row `ahead` iterations from now
    auto prefetch = [&](int r, int ahead, float* dst) {
        int q = rq[r] + ahead, qn = rqn[r] + ahead;
        if (wrapy) { if (qn >= N) qn -= N; if (qn >= N) qn -= N; }
        if (ract[r] && rs[r] == 1 && row_active(1, q)) pull<XMODE>(gin, geo, qn, rxn[r] >> 6, rxn[r] & 63, dst);
    };
#pragma unroll
    for (int r = 0; r < C::RPW; r++) { prefetch(r, 0, pa[r]); prefetch(r, 1, pb[r]); }

    const int niter = (y1 - y0) + 3 * (T - 1);

    auto body = [&](int r, float* pre) {
        const int s = rs[r], x = rx[r], xn = rxn[r], q = rq[r], qn = rqn[r];
        const bool go = ract[r] && row_active(s, q);
        float a[Q];
        if (go) {
            if (s == 1) {
#pragma unroll
                for (int k = 0; k < Q; k++) a[k] = pre[k];
            } else {
                const float* lev = lds + (s - 2) * C::LEVF;
                const int cx = x - x0 + C::HP;
                const bool ob_m = XMODE != XMODE_PERIODIC && x + 1 >= M;  // source for dx = -1
                const bool ob_p = XMODE != XMODE_PERIODIC && x - 1 < 0;   // source for dx = +1
                const bool ob_u = q - 1 < qlo, ob_d = q + 1 >= qhi;       // sources for dy = +1 / -1
                const int own = (q & 3) * C::ROWF + cx;
#pragma unroll
                for (int k = 0; k < Q; k++) {
                    const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * C::RW;
                    const bool oy = dy == 1 ? ob_u : (dy == -1 ? ob_d : false);
                    const bool ox = dx == 1 ? ob_p : (dx == -1 ? ob_m : false);
                    const int src = ((q - dy) & 3) * C::ROWF + (cx - dx);
                    const float v = lev[((ox || oy) ? own : src) + sl];
                    const bool inlet = (dx == 1 && ox) || (dy == 1 && oy);
                    a[k] = inlet ? geo.surr[k] : v;
                }
            }
        }
        // next prefetch for stage-1 lanes (row of iteration +2)
        if (s == 1) prefetch(r, 2, pre);
        if (go) {
            const uint64_t mw = mask[(long long)(qn + geo.G) * geo.Mb + (xn >> 6)];
            const bool blocked = (mw >> (xn & 63)) & 1ull;
            reflect(a, blocked);
            collide(a, blocked, geo.omega);
            if (s < T) {
                float* o = lds + (s - 1) * C::LEVF + (q & 3) * C::ROWF + (x - x0 + C::HP);
#pragma unroll
                for (int k = 0; k < Q; k++) o[slot_of(k) * C::RW] = a[k];
            } else {
                ST* o = gout + (long long)(qn + geo.G) * geo.rowStride + (long long)(xn >> 6) * CHUNK + (xn & 63);
#pragma unroll
                for (int k = 0; k < Q; k++) st_nt(o + slot_of(k) * LANES, a[k]);
            }
        }
        rq[r] = q + 1;
        int qn1 = qn + 1;
        if (wrapy && qn1 >= N) qn1 -= N;
        rqn[r] = qn1;
    };

    for (int i = 0; i < niter; i += 2) {
#pragma unroll
        for (int r = 0; r < C::RPW; r++) body(r, pa[r]);
        __syncthreads();
        if (i + 1 < niter) {
#pragma unroll
            for (int r = 0; r < C::RPW; r++) body(r, pb[r]);
        }
        __syncthreads();
    }
}

template <int XMODE, int T, typename ST>
static void launch_tb_t(const ST* gin, ST* gout, const uint64_t* mask, const Geom& geo, int row0,
                        int nrows, int Ry, hipStream_t st) {
    using C = TB<T>;
    const int nstrips = (geo.M + TBW - 1) / TBW;
    const int nchunks = (nrows + Ry - 1) / Ry;
    hipLaunchKernelGGL((step_tb_kernel<XMODE, T, ST>), dim3((unsigned)(nstrips * nchunks)),
                       dim3(C::NW * 64), 0, st, gin, gout, mask, geo, row0, nrows, Ry, nstrips);
}

template <int T>
static void launch_tb_T(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                        bool half, int row0, int nrows, int Ry, hipStream_t st) {
    if (xmode == XMODE_PERIODIC) {
        if (half) launch_tb_t<XMODE_PERIODIC, T>((const __half*)gin, (__half*)gout, mask, geo, row0, nrows, Ry, st);
        else launch_tb_t<XMODE_PERIODIC, T>((const float*)gin, (float*)gout, mask, geo, row0, nrows, Ry, st);
    } else {
        if (half) launch_tb_t<XMODE_REFERENCE, T>((const __half*)gin, (__half*)gout, mask, geo, row0, nrows, Ry, st);
        else launch_tb_t<XMODE_REFERENCE, T>((const float*)gin, (float*)gout, mask, geo, row0, nrows, Ry, st);
    }
}

// T updates of rows [row0, row0+nrows) in one launch; input rows [row0-T, row0+nrows+T) must be
// valid (or lie beyond a domain edge).  T in [2, TB_MAX_T].
hipError_t launch_step_tb(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                          bool half, int T, int row0, int nrows, int Ry, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    if (Ry < 1) Ry = 1;
    switch (T) {
        case 2: launch_tb_T<2>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 3: launch_tb_T<3>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 4: launch_tb_T<4>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lbm
