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

namespace lbm {

constexpr int TB_MARG = 8;  // window columns on either side that are computed but not stored

template <int T, int NB>
struct TB {
    static_assert(T - 1 <= TB_MARG, "margin too small for this depth");
    static constexpr int W = NB * 64;            // window width (columns computed per stage)
    static constexpr int SW = W - 2 * TB_MARG;   // strip width (columns stored)
    static constexpr int RW = W + 2;             // floats per slot row (+1 pad either side)
    static constexpr int ROWF = Q * RW;          // floats per ring row
    static constexpr int LEVF = 4 * ROWF;        // floats per level (ring of 4 rows)
    static constexpr int LDSF = (T - 1) * LEVF;
    static constexpr int NW = T * NB;            // waves
    static constexpr int FROWS = 16;             // ring depth of the blocked-flag rows (> 2(T-1)+1)
};

template <int XMODE, int T, int NB, typename ST>
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

    // ---- role: stage s (wave-uniform) and this lane's column --------------------------------
    const int s = wave / NB + 1;
    const int j = wave % NB;
    const int cx = 1 + j * 64 + lane;              // column index inside a ring row
    const int x = xo - TB_MARG + j * 64 + lane;    // logical column
    int xn;                                        // storage column
    if (XMODE == XMODE_PERIODIC) { xn = x % M; if (xn < 0) xn += M; }
    else xn = min(max(x, 0), M - 1);               // outside the domain: a valid address, unused
    const bool stored = x >= xo && x < xo + C::SW && x < M;
    // x edge rule (reference mode only): source of dx = -1 / +1 outside the domain
    const bool ob_m = XMODE != XMODE_PERIODIC && x + 1 >= M;
    const bool ob_p = XMODE != XMODE_PERIODIC && x - 1 < 0;
    const bool wave_xedge = __any(ob_m || ob_p);

    // per-lane element offsets inside a row of the global layout (stage 1)
    int xm1 = xn + 1, xp1 = xn - 1;
    if (XMODE == XMODE_PERIODIC) { xm1 = xm1 >= M ? 0 : xm1; xp1 = xp1 < 0 ? M - 1 : xp1; }
    else { xm1 = xm1 >= M ? xn : xm1; xp1 = xp1 < 0 ? xn : xp1; }
    const int off0 = (xn >> 6) * CHUNK + (xn & 63);
    const int offm = (xm1 >> 6) * CHUNK + (xm1 & 63);
    const int offp = (xp1 >> 6) * CHUNK + (xp1 & 63);
    const unsigned mbit = xn & 63;

    // ---- rows ----------------------------------------------------------------------------------
    const int qstart = y0 - (T - 1) - 2 * (s - 1);                        // row of iteration 0
    const int rlo = max(y0 - (T - s), qlo), rhi = min(y1 + (T - s), qhi);  // rows this stage makes
    const int niter = (y1 - y0) + 3 * (T - 1);
    auto wrap_row = [&](int r) {  // storage row of a logical row in [-N, 2N)
        if (wrapy) { r = r < 0 ? r + N : r; r = r >= N ? r - N : r; }
        return r;
    };
    auto clampr = [&](int r) { return min(max(r, rlo), rhi - 1); };
    auto load_mask = [&](int rn) -> uint64_t {
        return mask[(long long)(rn + G) * geo.Mb + (xn >> 6)];
    };

    // stage 1: issue the 9 pulled loads of logical row r (clamped into this stage's range)
    auto load_row = [&](auto xedge, int r, float* dst) {
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
            // a source row outside the domain falls back to the own cell: own row AND own column
            const int off = oy ? off0 : (dx == 1 ? offp : (dx == 0 ? off0 : offm));
            float v;
            if (decltype(xedge)::value) {
                const bool ox = dx == 1 ? ob_p : (dx == -1 ? ob_m : false);
                const ST* pp = ox ? (p0 + off0) : (rb + off);
                v = ld(pp + sl);
                if ((dx == 1 && ox)) v = surr(k);
            } else {
                v = ld(rb + off + sl);
            }
            if (dy == 1) v = oy ? surr(k) : v;
            dst[k] = v;
        }
    };
    // stages 2..T: gather the 9 pulled populations of logical row r from the ring of level s-1
    auto gather = [&](auto xedge, int r, float* a) {
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
            float v;
            if (decltype(xedge)::value) {
                const bool ox = dx == 1 ? ob_p : (dx == -1 ? ob_m : false);
                v = lev[(ox ? ro0 : ro) - (ox ? 0 : dxe) + sl];
                if (dx == 1 && ox) v = surr(k);
            } else {
                v = lev[ro - dxe + sl];
            }
            if (dy == 1) v = oy ? surr(k) : v;
            a[k] = v;
        }
    };
    auto to_lds = [&](int r, const float* a) {
        float* o = lds + (s - 1) * C::LEVF + (r & 3) * C::ROWF + cx;
#pragma unroll
        for (int k = 0; k < Q; k++) o[slot_of(k) * C::RW] = a[k];
    };

    auto run = [&](auto xedge) {
        int q = qstart;
        if (s == 1) {
            float pa[Q], pb[Q];
            load_row(xedge, q, pa);
            load_row(xedge, q + 1, pb);
            uint64_t mw = load_mask(wrap_row(clampr(q)));
            auto it = [&](float* pre) {
                float a[Q];
#pragma unroll
                for (int k = 0; k < Q; k++) a[k] = pre[k];
                load_row(xedge, q + 2, pre);
                const bool blocked = (mw >> mbit) & 1ull;
                mw = load_mask(wrap_row(clampr(q + 1)));
                flags[(q & (C::FROWS - 1)) * C::RW + cx] = blocked;
                reflect(a, blocked);
                collide(a, blocked, omega);
                to_lds(q, a);
                q += 1;
                __syncthreads();
            };
            for (int i = 0; i < niter; i += 2) { it(pa); it(pb); }
        } else {
            for (int i = 0; i < niter + (niter & 1); i++) {
                float a[Q];
                const bool blocked = flags[(q & (C::FROWS - 1)) * C::RW + cx] != 0;
                gather(xedge, q, a);
                reflect(a, blocked);
                collide(a, blocked, omega);
                if (s < T) {
                    to_lds(q, a);
                } else if (q >= rlo && q < rhi && stored) {
                    ST* o = gout + (long long)(wrap_row(q) + G) * RS + off0;
#pragma unroll
                    for (int k = 0; k < Q; k++) st_nt(o + slot_of(k) * LANES, a[k]);
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
        case 4: launch_tb_T<4, 4>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lbm
