// Temporal blocking: T lattice updates per launch, intermediate time levels staged in LDS.
//
// The reference's INNER_TIMESTEPS (lb_cuda_kernel.cu:213-216,239,288-292) re-reads `newcells` from
// global memory between inner steps with a T-lane halo inside every 32-lane warp, and is racy
// across warps.  This is a different design for CDNA4 (160 KB LDS per CU, 64-lane waves):
//
// A workgroup owns a WINDOW of W = NB*64 columns and a chunk of Ry rows and marches down the rows
// as a time-skewed pipeline.  In iteration i, stage s (1..T) produces time level s of row
// q_s(i) = y0 - (T-1) + i - 2(s-1), for ALL W columns of the window:
//     stage 1     pulls level 0 straight from HBM (the same coalesced 768-byte pieces as the T=1
//                 kernel), software-prefetched two iterations ahead in registers;
//     stage 2..T  pull from a 4-row LDS ring of level s-1;
//     stage s<T   writes its row into the LDS ring of level s, stage T stores to HBM.
// One wave per (stage, 64-column block): T*NB waves, all running the same code.  Level 1 is
// right in every window column; each further level loses one column at either window edge
// (its neighbour lies outside the window), so only the inner W - 2*MARG columns (MARG = 8 >= T-1)
// are stored: windows of neighbouring workgroups overlap by 16 columns (halo recompute, 6 %).
// Output strips are 16 columns (64 bytes per population) aligned.  All stages run concurrently;
// one barrier per iteration separates an iteration's ring writes from the next iteration's reads
// (skew 2 per stage => ring depth 4).  Per T updates the cells are read once (1.07x with the
// overlap, mostly L2 hits) and written once: HBM bytes per cell-update = 72/T (+ 3(T-1) warm-up
// rows per chunk).  Every cell goes through the same pull/reflect/collide device functions as
// the T=1 kernel, so results are bit-identical to T single updates.
//
// All loads in the row loops are unconditional and straight-line (rows outside a stage's range are
// clamped to a valid row instead of skipped), so that hipcc can use counted s_waitcnt vmcnt(N) and
// the stage-1 prefetch really stays in flight across two iterations and barriers.  Only stage 1
// touches the bit-packed mask; it hands each cell's blocked flag to the later stages through LDS.
#include <type_traits>

#include "lbm_internal.h"

#ifndef LBM_TB_PF
#define LBM_TB_PF 2
#endif
#ifndef LBM_TB_MARG
#define LBM_TB_MARG 8
#endif
#ifndef LBM_TB_STORE
#define LBM_TB_STORE st
#endif

namespace lbm {

constexpr int TB_PF = LBM_TB_PF;  // stage-1 rows in flight
constexpr int TB_MARG = LBM_TB_MARG;  // window columns on either side that are computed but not stored

template <int T, int NB>
struct TB {
    static_assert(T - 1 <= TB_MARG, "margin too small for this depth");
    static_assert(NB % 2 == 0, "a lane owns two 64-column blocks");
    static constexpr int W = NB * 64;            // window width (columns computed per stage)
    static constexpr int SW = W - 2 * TB_MARG;   // strip width (columns stored)
    static constexpr int RW = W + 2;             // floats per slot row (+1 pad either side)
    static constexpr int ROWF = Q * RW;          // floats per ring row
    static constexpr int LEVF = 4 * ROWF;        // floats per level (ring of 4 rows)
    static constexpr int LDSF = (T - 1) * LEVF;
    static constexpr int NW = T * NB / 2;        // waves: one per (stage, pair of blocks)
    static constexpr int FROWS = 16;             // ring depth of the blocked-flag rows (> 2(T-1)+1)
};

// Every lane owns TWO cells of its stage's row, 64 columns apart (blocks 2j and 2j+1 of the
// window), held as the .x / .y halves of float2 registers, so that collide runs on packed fp32
// instructions (the kernel is VALU-bound, not HBM-bound).
template <int XMODE, int T, int NB, int PF, typename ST>
__global__ __launch_bounds__((TB<T, NB>::NW * 64)) void step_tb_kernel(
    const ST* __restrict__ gin, ST* __restrict__ gout, const uint64_t* __restrict__ mask, Geom geo,
    int row0, int nrows, int Ry, int nstrips) {
    using C = TB<T, NB>;
    __shared__ float lds[C::LDSF];
    __shared__ uint8_t flags[C::FROWS * C::RW];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = wg % nstrips, chunk = wg / nstrips;
    const int xo = strip * C::SW;  // first stored column
    const int y0 = row0 + chunk * Ry;
    const int y1 = min(y0 + Ry, row0 + nrows);
    const int N = geo.N, M = geo.M, G = geo.G;
    const bool wrapy = geo.top == YEDGE_WRAP;  // top and bottom wrap together
    const int qlo = geo.top == YEDGE_DOMAIN ? 0 : -(1 << 28);  // rows that exist
    const int qhi = geo.bot == YEDGE_DOMAIN ? N : (1 << 28);
    const long long RS = geo.rowStride;
    const float omega = geo.omega;
    const float su0 = geo.surr[0], su1 = geo.surr[1], su2 = geo.surr[2], su3 = geo.surr[3],
                su4 = geo.surr[4], su5 = geo.surr[5], su6 = geo.surr[6], su7 = geo.surr[7],
                su8 = geo.surr[8];
    auto surr = [&](int k) {  // k is a compile-time constant after unrolling
        return k == 0 ? su0 : k == 1 ? su1 : k == 2 ? su2 : k == 3 ? su3 : k == 4 ? su4
             : k == 5 ? su5 : k == 6 ? su6 : k == 7 ? su7 : su8;
    };

    // ---- role: stage s (wave-uniform) and this lane's two columns (c = 0, 1) --------------------
    const int s = wave / (NB / 2) + 1;
    const int j = wave % (NB / 2);
    const int cx = 1 + j * 128 + lane;  // ring-row index of cell 0; cell 1 is 64 further
    int xs[2], off0[2], offm[2], offp[2];
    unsigned mbit[2];
    int mblk[2];
    bool stored[2], ob_m[2], ob_p[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int x = xo - TB_MARG + j * 128 + c * 64 + lane;  // logical column
        int xn;                                                // storage column
        if (XMODE == XMODE_PERIODIC) { xn = x % M; if (xn < 0) xn += M; }
        else xn = min(max(x, 0), M - 1);  // outside the domain: a valid address, never stored
        stored[c] = x >= xo && x < xo + C::SW && x < M;
        // x edge rule (reference mode only): source of dx = -1 / +1 outside the domain
        ob_m[c] = XMODE != XMODE_PERIODIC && x + 1 >= M;
        ob_p[c] = XMODE != XMODE_PERIODIC && x - 1 < 0;
        int xm1 = xn + 1, xp1 = xn - 1;
        if (XMODE == XMODE_PERIODIC) { xm1 = xm1 >= M ? 0 : xm1; xp1 = xp1 < 0 ? M - 1 : xp1; }
        else { xm1 = xm1 >= M ? xn : xm1; xp1 = xp1 < 0 ? xn : xp1; }
        xs[c] = x;
        off0[c] = (xn >> 6) * CHUNK + (xn & 63);
        offm[c] = (xm1 >> 6) * CHUNK + (xm1 & 63);
        offp[c] = (xp1 >> 6) * CHUNK + (xp1 & 63);
        mblk[c] = xn >> 6;
        mbit[c] = xn & 63;
    }
    const bool wave_xedge = __any(ob_m[0] || ob_p[0] || ob_m[1] || ob_p[1]);

    // ---- rows ----------------------------------------------------------------------------------
    const int qstart = y0 - (T - 1) - 2 * (s - 1);                        // row of iteration 0
    const int rlo = max(y0 - (T - s), qlo), rhi = min(y1 + (T - s), qhi);  // rows this stage makes
    const int niter = (y1 - y0) + 3 * (T - 1);
    const int nit = (niter + PF - 1) / PF * PF;  // every wave runs the same number of barriers
    auto wrap_row = [&](int r) {  // storage row of a logical row in [-N, 2N)
        if (wrapy) { r = r < 0 ? r + N : r; r = r >= N ? r - N : r; }
        return r;
    };
    auto clampr = [&](int r) { return min(max(r, rlo), rhi - 1); };

    // stage 1: issue the 2 x 9 pulled loads of logical row r (clamped into this stage's range) and
    // the two mask words.  Pure loads from always-valid addresses (a source outside the domain is
    // replaced by the own cell); the inlet rule is applied when the row is consumed (inlet_fix),
    // so that no load sits behind a branch and the vmcnt bookkeeping stays exact.
    auto load_row = [&](auto xedge, int r, f2* dst, uint64_t* mw) {
        r = clampr(r);
        const int rn = wrap_row(r);
        const bool ob_u = r - 1 < qlo, ob_d = r + 1 >= qhi;
        const int rnu = ob_u ? rn : wrap_row(r - 1), rnd = ob_d ? rn : wrap_row(r + 1);
        const ST* p0 = gin + (long long)(rn + G) * RS;
        const ST* pu = gin + (long long)(rnu + G) * RS;
        const ST* pd = gin + (long long)(rnd + G) * RS;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * LANES;
            const ST* rb = dy == 1 ? pu : (dy == 0 ? p0 : pd);
            const bool oy = dy == 1 ? ob_u : (dy == -1 ? ob_d : false);
            float v[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                // a source row outside the domain falls back to the own cell: own row AND column
                const int off = oy ? off0[c] : (dx == 1 ? offp[c] : (dx == 0 ? off0[c] : offm[c]));
                if (decltype(xedge)::value) {
                    const bool ox = dx == 1 ? ob_p[c] : (dx == -1 ? ob_m[c] : false);
                    const ST* pp = ox ? (p0 + off0[c]) : (rb + off);
                    v[c] = ld(pp + sl);
                } else {
                    v[c] = ld(rb + off + sl);
                }
            }
            dst[k] = f2{v[0], v[1]};
        }
        const uint64_t* mrow = mask + (long long)(rn + G) * geo.Mb;
        mw[0] = mrow[mblk[0]];
        mw[1] = mrow[mblk[1]];
    };
    auto inlet_fix = [&](auto xedge, int r, f2* a) {
        if (XMODE == XMODE_PERIODIC) return;  // no domain edge in x, and y wraps or is a seam
        const bool ob_u = r - 1 < qlo;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dy = dy_of(k), dx = dx_of(k);
            if (decltype(xedge)::value && dx == 1)
                a[k] = f2{ob_p[0] ? surr(k) : a[k].x, ob_p[1] ? surr(k) : a[k].y};
            if (dy == 1) a[k] = ob_u ? f2{surr(k), surr(k)} : a[k];
        }
    };
    // stages 2..T: gather the pulled populations of logical row r from the ring of level s-1
    auto gather = [&](auto xedge, int r, f2* a) {
        const float* lev = lds + (s - 2) * C::LEVF + cx;
        const bool ob_u = r - 1 < qlo, ob_d = r + 1 >= qhi;
        const int ro0 = (r & 3) * C::ROWF;
        const int rou = ob_u ? ro0 : ((r - 1) & 3) * C::ROWF;
        const int rod = ob_d ? ro0 : ((r + 1) & 3) * C::ROWF;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * C::RW;
            const int ro = dy == 1 ? rou : (dy == 0 ? ro0 : rod);
            const bool oy = dy == 1 ? ob_u : (dy == -1 ? ob_d : false);
            const int dxe = oy ? 0 : dx;  // own-cell fallback: own row AND own column
            float v[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (decltype(xedge)::value) {
                    const bool ox = dx == 1 ? ob_p[c] : (dx == -1 ? ob_m[c] : false);
                    v[c] = lev[(ox ? ro0 : ro) - (ox ? 0 : dxe) + sl + 64 * c];
                    if (dx == 1 && ox) v[c] = surr(k);
                } else {
                    v[c] = lev[ro - dxe + sl + 64 * c];
                }
                if (XMODE != XMODE_PERIODIC && dy == 1) v[c] = oy ? surr(k) : v[c];
            }
            a[k] = f2{v[0], v[1]};
        }
    };
    auto to_lds = [&](int r, const f2* a) {
        float* o = lds + (s - 1) * C::LEVF + (r & 3) * C::ROWF + cx;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            o[slot_of(k) * C::RW] = a[k].x;
            o[slot_of(k) * C::RW + 64] = a[k].y;
        }
    };

    auto run = [&](auto xedge) {
        int q = qstart;
        if (s == 1) {
            // PF rows in flight: row q+PF is requested while row q is consumed; the mask words
            // travel with their row (vmcnt retires in order)
            f2 pre[PF][Q];
            uint64_t pmw[PF][2];
#pragma unroll
            for (int d = 0; d < PF; d++) load_row(xedge, q + d, pre[d], pmw[d]);
            auto it = [&](f2* pr, uint64_t* mw) {
                f2 a[Q];
#pragma unroll
                for (int k = 0; k < Q; k++) a[k] = pr[k];
                inlet_fix(xedge, clampr(q), a);
                const bool b0 = (mw[0] >> mbit[0]) & 1ull, b1 = (mw[1] >> mbit[1]) & 1ull;
                load_row(xedge, q + PF, pr, mw);
                uint8_t* fl = flags + (q & (C::FROWS - 1)) * C::RW + cx;
                fl[0] = b0;
                fl[64] = b1;
                reflect2(a, b0, b1);
                collide2(a, b0, b1, omega);
                to_lds(q, a);
                q += 1;
                __syncthreads();
            };
            for (int i = 0; i < nit; i += PF) {
#pragma unroll
                for (int d = 0; d < PF; d++) it(pre[d], pmw[d]);
            }
        } else {
            for (int i = 0; i < nit; i++) {
                f2 a[Q];
                const uint8_t* fl = flags + (q & (C::FROWS - 1)) * C::RW + cx;
                const bool b0 = fl[0] != 0, b1 = fl[64] != 0;
                gather(xedge, q, a);
                reflect2(a, b0, b1);
                collide2(a, b0, b1, omega);
                if (s < T) {
                    to_lds(q, a);
                } else if (q >= rlo && q < rhi) {
                    ST* orow = gout + (long long)(wrap_row(q) + G) * RS;
                    if (stored[0]) {
#pragma unroll
                        for (int k = 0; k < Q; k++) LBM_TB_STORE(orow + off0[0] + slot_of(k) * LANES, a[k].x);
                    }
                    if (stored[1]) {
#pragma unroll
                        for (int k = 0; k < Q; k++) LBM_TB_STORE(orow + off0[1] + slot_of(k) * LANES, a[k].y);
                    }
                }
                q += 1;
                __syncthreads();
            }
        }
    };
    if (wave_xedge) run(std::true_type{});
    else run(std::false_type{});
}

template <int XMODE, int T, int NB, typename ST>
static void launch_tb_t(const ST* gin, ST* gout, const uint64_t* mask, const Geom& geo, int row0,
                        int nrows, int Ry, hipStream_t st) {
    using C = TB<T, NB>;
    const int nstrips = (geo.M + C::SW - 1) / C::SW;
    const int nchunks = (nrows + Ry - 1) / Ry;
    hipLaunchKernelGGL((step_tb_kernel<XMODE, T, NB, TB_PF, ST>), dim3((unsigned)(nstrips * nchunks)),
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
        case 4: launch_tb_T<4, 4>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lbm
