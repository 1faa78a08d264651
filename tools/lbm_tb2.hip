// NOT BUILT, NOT SHIPPED (moved out of the product library in round 3).  The two-levels-per-wave variant
// of the LDS temporal-blocking kernel from round 2: bit-identical to lbm_tb.hip but slower (T = 6:
// 276-277 against 318 GLUPS) and without that kernel's later improvements (sink-style collide2_to,
// immediate-only ring offsets, scaled quotients, ballots).  Kept as a record of the design
// (profiles/r02_tb_experiments.md); to revive it, put it back beside lbm_tb.hip, add it to SRC in
// latticeboltzmann_amd/csrc/Makefile and re-declare launch_step_tb2 / tb2_has_depth in lbm_internal.h.
// Temporal blocking, second design: TWO time levels per wave and row.
//
// lbm_tb.hip gives every time level its own waves and passes every level through LDS: T-1 ring
// levels, one barrier per row and level, 72 B per cell and level across the LDS.  Its SQ counters
// (profiles/r01_tb_sq_counters.md) show VALU 66 % busy and the LDS phases of all waves falling
// together.  Here a wave ("stage" X = 1 .. ceil(T/2)) computes level 2X-1 of row r exactly as
// before -- stage 1 from HBM (register prefetch), later stages from the LDS ring of level 2X-2 --
// and then, in the same iteration and without touching LDS, level 2X of row r-1 from the three rows
// of level 2X-1 it holds in registers: the dy = -1 populations of the row just computed, the dy = 0
// ones of the row before and the dy = +1 ones of the row before that (a 9-register ring per cell),
// shifted in x by a DPP wave shift (v_mov_b32_dpp wave_shr:1 / wave_shl:1).  Only the even levels
// cross LDS: (T-1)/2 ring levels instead of T-1, half the barriers per update, 2.5x less LDS
// traffic at T = 6, and the LDS freed buys windows twice as wide.
//
// A DPP shift cannot reach the neighbouring wave, so the even level is wrong in lanes 0 and 63 of
// every 64-lane block.  Blocks therefore overlap by two columns (block b covers window columns
// [62 b, 62 b + 64)): the valid lanes 1..62 of consecutive blocks tile the window without a gap,
// at 3 % redundant arithmetic.  Odd levels are right in all 64 lanes (LDS holds the neighbours).
// Rows: stage X runs 3 rows behind stage X-1 (it needs row r+1 of level 2X-2 one barrier old).
// Everything else -- ring layout, phantom row and side ring of the reference edge rule, chunk /
// strip decomposition, XCD remap, prefetch -- is lbm_tb.hip's.  Every cell still goes through the
// same pull / reflect / collide arithmetic: bit-identical to T single updates.
#include <atomic>
#include <type_traits>

#include "lbm_tb_common.h"

#ifndef LBM_TB2_PF
#define LBM_TB2_PF 2
#endif

namespace lbm {

template <int T, int NB, int CPL, int MARG>
struct TB2 {
    static_assert(T - 1 <= MARG, "margin too small for this depth");
    static_assert(NB % CPL == 0, "a lane owns CPL 64-lane blocks");
    static constexpr int BS = 62;                // block stride: blocks overlap by two columns
    static constexpr int DA = 2;                 // ring depth of the dy = -1 slots
    static constexpr int W = BS * NB + 2;        // window width (columns computed per stage)
    static constexpr int SW = W - 2 * MARG;      // strip width (columns stored)
    static constexpr int RW = W + 2;             // floats per slot row (+1 pad either side)
    static constexpr int A_ROW = 3 * RW * 4;
    static constexpr int B_ROW = 6 * RW * 4;
    static constexpr int B_OFF = DA * A_ROW;
    static constexpr int F_OFF = B_OFF + 4 * B_ROW;
    static constexpr int E_OFF = (F_OFF + 4 * RW + 3) / 4 * 4;
    static constexpr int LEVB = (E_OFF + 16 + 15) / 16 * 16;
    static constexpr int NS = (T + 1) / 2;       // stages
    static constexpr int NL = (T - 1) / 2;       // ring levels in LDS (levels 2, 4, ... < T)
    static constexpr int LDSB = NL > 0 ? NL * LEVB : 16;
    static constexpr int NW = NS * NB / CPL;
    static constexpr int slot_off(int sl, int ph, int age) {
        return sl < 3 ? ((ph - age) & (DA - 1)) * A_ROW + sl * RW * 4
                      : B_OFF + ((ph - age) & 3) * B_ROW + (sl - 3) * RW * 4;
    }
    static constexpr int flag_off(int ph, int age) { return F_OFF + ((ph - age) & 3) * RW; }
    static constexpr int edge_off(int ph, int age) { return E_OFF + ((ph - age) & 3) * 4; }
};

// value of the lane dx to the left (dx = +1: from lane-1) or right (dx = -1: from lane+1); the
// lane that has no such neighbour keeps its own value (an ordinary number: its result is unused)
template <int DX>
__device__ __forceinline__ float lane_shift(float v) {
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(b, b, DX == 1 ? 0x138 : 0x130, 0xf, 0xf, false));
}
template <int DX>
__device__ __forceinline__ f2 lane_shift(f2 v) { return f2{lane_shift<DX>(v.x), lane_shift<DX>(v.y)}; }

template <int XMODE, int T, int NB, int CPL, int MARG, typename ST>
__global__ __launch_bounds__((TB2<T, NB, CPL, MARG>::NW * 64)) void step_tb2_kernel(
    const ST* __restrict__ gin, ST* __restrict__ gout, const uint64_t* __restrict__ mask, Geom geo,
    int row0, int nrows, int Ry, int nstrips) {
    using C = TB2<T, NB, CPL, MARG>;
    using L = LaneV<CPL>;
    using V = typename L::V;
    constexpr int PF = LBM_TB2_PF;
    constexpr int BS = C::BS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[C::LDSB];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = wg % nstrips;
    const int chunk = wg / nstrips;
    const int xo = strip * C::SW;
    const int y0 = row0 + chunk * Ry;
    const int y1 = min(y0 + Ry, row0 + nrows);
    const int N = geo.N, M = geo.M, G = geo.G;
    const bool wrapy = geo.top == YEDGE_WRAP;
    const int qlo = geo.top == YEDGE_DOMAIN ? 0 : -(1 << 28);
    const int qhi = geo.bot == YEDGE_DOMAIN ? N : (1 << 28);
    const long long RS = geo.rowStride;
    const Omega omega = omega_of(geo);
    const float su0 = geo.surr[0], su1 = geo.surr[1], su2 = geo.surr[2], su3 = geo.surr[3],
                su4 = geo.surr[4], su5 = geo.surr[5], su6 = geo.surr[6], su7 = geo.surr[7],
                su8 = geo.surr[8];
    auto surr = [&](int k) {
        return k == 0 ? su0 : k == 1 ? su1 : k == 2 ? su2 : k == 3 ? su3 : k == 4 ? su4
             : k == 5 ? su5 : k == 6 ? su6 : k == 7 ? su7 : su8;
    };

    // ---- role -----------------------------------------------------------------------------------
    constexpr int WPS = NB / CPL;
    const int X = wave / WPS + 1;       // stage: levels l1 = 2X-1 and l2 = 2X
    const int j = wave % WPS;
    const int l1 = 2 * X - 1, l2 = 2 * X;
    const bool two = l2 <= T;           // false: last stage of an odd T only makes level l1
    const bool writes_lds = l2 < T;     // level l2 is read by stage X+1
    const int cx = 1 + BS * (j * CPL) + lane;  // ring-row index of cell 0; cell c is BS*c further
    const bool lane_ok = lane >= 1 && lane <= 62;  // lanes whose even level is right
    int off0[CPL], offm[CPL], offp[CPL], mblk[CPL];
    unsigned mbit[CPL];
    bool stored[CPL], ob_m[CPL], ob_p[CPL], last_col[CPL], care1[CPL], care2[CPL];
    bool any_xedge = false;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int wc = BS * (j * CPL + c) + lane;  // window column
        const int x = xo - MARG + wc;              // logical column
        int xn;
        if (XMODE == XMODE_PERIODIC) { xn = x % M; if (xn < 0) xn += M; }
        else xn = min(max(x, 0), M - 1);
        stored[c] = lane_ok && x >= xo && x < xo + C::SW && x < M;
        care1[c] = wc >= l1 - 1 && wc < C::W - (l1 - 1);
        care2[c] = lane_ok && wc >= l2 - 1 && wc < C::W - (l2 - 1);
        ob_m[c] = XMODE != XMODE_PERIODIC && x + 1 >= M;
        ob_p[c] = XMODE != XMODE_PERIODIC && x - 1 < 0;
        last_col[c] = XMODE != XMODE_PERIODIC && x == M - 1;
        any_xedge = any_xedge || ob_m[c] || ob_p[c];
        int xm1 = xn + 1, xp1 = xn - 1;
        if (XMODE == XMODE_PERIODIC) { xm1 = xm1 >= M ? 0 : xm1; xp1 = xp1 < 0 ? M - 1 : xp1; }
        else { xm1 = xm1 >= M ? xn : xm1; xp1 = xp1 < 0 ? xn : xp1; }
        off0[c] = (xn >> 6) * CHUNK + (xn & 63);
        offm[c] = (xm1 >> 6) * CHUNK + (xm1 & 63);
        offp[c] = (xp1 >> 6) * CHUNK + (xp1 & 63);
        mblk[c] = xn >> 6;
        mbit[c] = xn & 63;
    }
    const bool wave_xedge = __any(any_xedge);

    // ---- rows -----------------------------------------------------------------------------------
    // iteration i: level l1 of row q = qstart + i, level l2 of row q - 1
    const int qstart = y0 - (T - 1) - 3 * (X - 1);
    const int rlo1 = max(y0 - (T - l1), qlo), rhi1 = min(y1 + (T - l1), qhi);
    const int rlo2 = max(y0 - (T - l2), qlo), rhi2 = min(y1 + (T - l2), qhi);
    const int niter = (y1 - y0) + (T - 1) + 3 * (C::NS - 1) + (T % 2 == 0 ? 1 : 0);
    const int nit = (niter + 3) & ~3;
    auto wrap_row = [&](int r) {
        if (wrapy) { r = r < 0 ? r + N : r; r = r >= N ? r - N : r; }
        return r;
    };
    auto clampr = [&](int r) { return min(max(r, rlo1), rhi1 - 1); };  // stage 1 loads (level 1)
    auto live1 = [&](int r) { return r >= rlo1 && r < rhi1; };
    auto live2 = [&](int r) { return two && r >= rlo2 && r < rhi2; };

    // LDS views: ring level m (0-based) holds time level 2(m+1)
    unsigned char* const wr_base = lds + (X - 1) * C::LEVB + cx * 4;
    const unsigned char* const rd_base = lds + (X - 2) * C::LEVB + cx * 4;
    unsigned char* const wr_flag = lds + (X - 1) * C::LEVB + cx;
    const unsigned char* const rd_flag = lds + (X - 2) * C::LEVB + cx;
    unsigned char* const wr_lev = lds + (X - 1) * C::LEVB;
    const unsigned char* const rd_lev = lds + (X - 2) * C::LEVB;

    auto load_row = [&](auto xedge, int r, V* dst, uint64_t* mw) {
        r = clampr(r);
        const int rn = wrap_row(r);
        const bool ob_u = r - 1 < qlo, ob_d = XMODE != XMODE_PERIODIC && r + 1 >= qhi;
        const int rnu = ob_u ? rn : wrap_row(r - 1), rnd = ob_d ? rn : wrap_row(r + 1);
        const ST* p0 = gin + (long long)(rn + G) * RS;
        const ST* pu = gin + (long long)(rnu + G) * RS;
        const ST* pd = gin + (long long)(rnd + G) * RS;
        int o0[CPL], om[CPL], op[CPL], omd[CPL], opd[CPL];
        bool xm[CPL], xp[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int a0 = off0[c], am = offm[c], ap = offp[c];
            o0[c] = a0; om[c] = am; op[c] = ap;
            omd[c] = ob_d ? a0 : am;
            opd[c] = ob_d ? a0 : ap;
            xm[c] = ob_m[c]; xp[c] = ob_p[c];
        }
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * LANES;
            const ST* rb = dy == 1 ? pu : (dy == 0 ? p0 : pd);
            float v[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int off = dx == 0 ? o0[c] : (dy == -1 ? (dx == 1 ? opd[c] : omd[c])
                                                             : (dx == 1 ? op[c] : om[c]));
                const ST* pp = rb + off;
                if (decltype(xedge)::value && dx != 0) {
                    const ST* own = p0 + o0[c];
                    pp = (dx == 1 ? xp[c] : xm[c]) ? own : pp;
                }
                v[c] = ld(pp + sl);
            }
            dst[k] = L::make(v);
        }
        const uint64_t* mrow = mask + (long long)(rn + G) * geo.Mb;
#pragma unroll
        for (int c = 0; c < CPL; c++) mw[c] = mrow[mblk[c]];
    };
    // inlet of the reference edge rule: column 0 takes surr for dx = +1, row qlo for dy = +1
    auto inlet_fix = [&](auto xedge, int r, V* a) {
        if (XMODE == XMODE_PERIODIC) return;
        const bool ob_u = r - 1 < qlo;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dx = dx_of(k);
            float v[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                v[c] = L::get(a[k], c);
                if (decltype(xedge)::value && dx == 1) v[c] = ob_p[c] ? surr(k) : v[c];
            }
            a[k] = L::make(v);
        }
        if (ob_u) {
#pragma unroll
            for (int k = 0; k < Q; k++) {
                if (dy_of(k) != 1) continue;
                float v[CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++) v[c] = surr(k);
                a[k] = L::make(v);
            }
        }
    };
    auto gather = [&](auto xedge, auto ph, int r, V* a, bool* blocked) {
        constexpr int PH = decltype(ph)::value;
#pragma unroll
        for (int c = 0; c < CPL; c++) blocked[c] = rd_flag[C::flag_off(PH, 2) + BS * c] != 0;
        const bool ob_u = XMODE != XMODE_PERIODIC && r - 1 < qlo;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k);
            const int src = C::slot_off(sl, PH, dy == 1 ? 3 : (dy == 0 ? 2 : 1)) - 4 * dx;
            float v[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                v[c] = *reinterpret_cast<const float*>(rd_base + src + 4 * BS * c);
                if (decltype(xedge)::value && dx == -1) {
                    const float own = sl == 0
                        ? *reinterpret_cast<const float*>(rd_lev + C::edge_off(PH, 2))
                        : *reinterpret_cast<const float*>(rd_base + C::slot_off(sl, PH, 2) + 4 * BS * c);
                    v[c] = ob_m[c] ? own : v[c];
                }
                if (decltype(xedge)::value && dx == 1) v[c] = ob_p[c] ? surr(k) : v[c];
            }
            a[k] = L::make(v);
        }
        if (ob_u) {
#pragma unroll
            for (int k = 0; k < Q; k++) {
                if (dy_of(k) != 1) continue;
                float v[CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++) v[c] = surr(k);
                a[k] = L::make(v);
            }
        }
    };
    auto ring_value = [](float v) {
        if (std::is_same<ST, __half>::value) return __half2float(__float2half_rn(v));
        return v;
    };
    // level l2 of row q into the ring (lanes 1..62 only: the others belong to the neighbours)
    auto to_lds = [&](auto xedge, auto ph, int q, const V* a, const bool* blocked) {
        constexpr int PH = decltype(ph)::value;
        if (!lane_ok) return;
        if (XMODE != XMODE_PERIODIC && q == qhi) {  // phantom row, see lbm_tb_common.h
#pragma unroll
            for (int sl = 0; sl < 3; sl++) {
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const float v = *reinterpret_cast<const float*>(
                        wr_base + C::slot_off(sl, PH, 1) + 4 * (sl - 1) + 4 * BS * c);
                    *reinterpret_cast<float*>(wr_base + C::slot_off(sl, PH, 0) + 4 * BS * c) = v;
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < Q; k++) {
#pragma unroll
            for (int c = 0; c < CPL; c++)
                *reinterpret_cast<float*>(wr_base + C::slot_off(slot_of(k), PH, 0) + 4 * BS * c) = ring_value(L::get(a[k], c));
        }
#pragma unroll
        for (int c = 0; c < CPL; c++) wr_flag[C::flag_off(PH, 0) + BS * c] = blocked[c];
        if (decltype(xedge)::value) {
#pragma unroll
            for (int k = 0; k < Q; k++) {
                if (slot_of(k) != 0) continue;
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    if (last_col[c]) *reinterpret_cast<float*>(wr_lev + C::edge_off(PH, 0)) = ring_value(L::get(a[k], c));
            }
        }
    };
    auto to_hbm = [&](int r, const V* a) {
        ST* orow = gout + (long long)(wrap_row(r) + G) * RS;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            if (stored[c]) {
#pragma unroll
                for (int k = 0; k < Q; k++) st(orow + off0[c] + slot_of(k) * LANES, L::get(a[k], c));
            }
        }
    };

    auto run = [&](auto xedge) {
        int q = qstart;
        // register history of level l1 (post-collision), indexed by iteration PHASE (i & 3) so
        // that nothing is ever copied: the row loop is unrolled by four and every index below is
        // a compile-time constant.  Row q-1 sits at phase PH-1, row q-2 at PH-2.  The values are
        // rounded like a ring / HBM hand-over (fp16 storage).
        V h_dym[4][3], h_dy0[4][3], h_dyp[4][3];  // dy = -1 (k=0,3,6; reference edge rule only), 0 (1,4,7), +1 (2,5,8)
        bool h_blk[4][CPL];
#pragma unroll
        for (int p = 0; p < 4; p++) {
#pragma unroll
            for (int n = 0; n < 3; n++) {
                float one[CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++) one[c] = 1.0f;
                h_dym[p][n] = h_dy0[p][n] = h_dyp[p][n] = L::make(one);
            }
#pragma unroll
            for (int c = 0; c < CPL; c++) h_blk[p][c] = false;
        }

        // the fused level: level l2 of row q-1 from level l1 rows q (a), q-1, q-2 (ring)
        auto second = [&](auto ph, const V* a) {
            constexpr int P1 = (decltype(ph)::value + 3) & 3, P2 = (decltype(ph)::value + 2) & 3;
            const V* r_dym = h_dym[P1];
            const V* r_dy0 = h_dy0[P1];
            const V* r_dyp = h_dyp[P1];
            const V* rr_dyp = h_dyp[P2];
            const bool* r_blk = h_blk[P1];
            const int r2 = q - 1;
            if (live2(r2)) {
                V t[Q];
                // dy = -1 (k = 0, 3, 6) from row q, dy = 0 (1, 4, 7) from q-1, dy = +1 (2, 5, 8) from q-2
                t[0] = lane_shift<-1>(a[0]);      t[3] = a[3];      t[6] = lane_shift<1>(a[6]);
                t[1] = lane_shift<-1>(r_dy0[0]);  t[4] = r_dy0[1];  t[7] = lane_shift<1>(r_dy0[2]);
                t[2] = lane_shift<-1>(rr_dyp[0]); t[5] = rr_dyp[1]; t[8] = lane_shift<1>(rr_dyp[2]);
                if (XMODE != XMODE_PERIODIC) {
                    if (decltype(xedge)::value) {
                        // last column: dx = -1 keeps the own value (own row r2 = q-1, own column);
                        // first column: dx = +1 takes surr
                        float v[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; c++) v[c] = ob_m[c] ? L::get(r_dym[0], c) : L::get(t[0], c);
                        t[0] = L::make(v);
#pragma unroll
                        for (int c = 0; c < CPL; c++) v[c] = ob_m[c] ? L::get(r_dy0[0], c) : L::get(t[1], c);
                        t[1] = L::make(v);
#pragma unroll
                        for (int c = 0; c < CPL; c++) v[c] = ob_m[c] ? L::get(r_dyp[0], c) : L::get(t[2], c);
                        t[2] = L::make(v);
#pragma unroll
                        for (int k = 6; k < Q; k++) {
#pragma unroll
                            for (int c = 0; c < CPL; c++) v[c] = ob_p[c] ? surr(k) : L::get(t[k], c);
                            t[k] = L::make(v);
                        }
                    }
                    if (r2 + 1 >= qhi) {  // last domain row: dy = -1 keeps the own value (rare, uniform)
                        t[0] = r_dym[0]; t[3] = r_dym[1]; t[6] = r_dym[2];
                        if (decltype(xedge)::value) {  // ... but column 0 still takes surr for dx = +1
                            float v[CPL];
#pragma unroll
                            for (int c = 0; c < CPL; c++) v[c] = ob_p[c] ? surr(6) : L::get(t[6], c);
                            t[6] = L::make(v);
                        }
                    }
                    if (r2 - 1 < qlo) {   // first domain row: dy = +1 takes surr (rare, uniform)
#pragma unroll
                        for (int k = 0; k < Q; k++) {
                            if (dy_of(k) != 1) continue;
                            float v[CPL];
#pragma unroll
                            for (int c = 0; c < CPL; c++) v[c] = surr(k);
                            t[k] = L::make(v);
                        }
                    }
                }
                reflect_v(t, r_blk);
                collide_v(t, r_blk, omega, care2);
                if (writes_lds) to_lds(xedge, ph, r2, t, r_blk);
                else to_hbm(r2, t);
            } else if (XMODE != XMODE_PERIODIC && writes_lds && r2 == qhi) {
                to_lds(xedge, ph, r2, a, r_blk);  // phantom row (a is not read)
            }
        };
        auto remember = [&](auto ph, const V* a, const bool* blocked) {
            constexpr int P0 = decltype(ph)::value;
            auto rv = [&](V v) {
                float w[CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++) w[c] = ring_value(L::get(v, c));
                return L::make(w);
            };
            if (XMODE != XMODE_PERIODIC) { h_dym[P0][0] = rv(a[0]); h_dym[P0][1] = rv(a[3]); h_dym[P0][2] = rv(a[6]); }
            h_dy0[P0][0] = rv(a[1]); h_dy0[P0][1] = rv(a[4]); h_dy0[P0][2] = rv(a[7]);
            h_dyp[P0][0] = rv(a[2]); h_dyp[P0][1] = rv(a[5]); h_dyp[P0][2] = rv(a[8]);
#pragma unroll
            for (int c = 0; c < CPL; c++) h_blk[P0][c] = blocked[c];
        };

        if (X == 1) {
            V pre[PF][Q];
            uint64_t pmw[PF][CPL];
#pragma unroll
            for (int d = 0; d < PF; d++) load_row(xedge, q + d, pre[d], pmw[d]);
            auto it = [&](auto ph, V* pr, uint64_t* mw) {
                V a[Q];
                bool blocked[CPL];
#pragma unroll
                for (int k = 0; k < Q; k++) a[k] = pr[k];
#pragma unroll
                for (int c = 0; c < CPL; c++) blocked[c] = (mw[c] >> mbit[c]) & 1ull;
                load_row(xedge, q + PF, pr, mw);
                const bool lv = live1(q);
                if (lv) {
                    inlet_fix(xedge, q, a);
                    reflect_v(a, blocked);
                    collide_v(a, blocked, omega, care1);
                }
                if (two) {
                    // the second level reads a[] of row q through the same rounding as a hand-over
                    if (lv && std::is_same<ST, __half>::value) {
#pragma unroll
                        for (int k = 0; k < Q; k++) {
                            float w[CPL];
#pragma unroll
                            for (int c = 0; c < CPL; c++) w[c] = ring_value(L::get(a[k], c));
                            a[k] = L::make(w);
                        }
                    }
                    second(ph, a);
                    remember(ph, a, blocked);  // unconditional: a dead history entry frees its registers
                } else if (lv) {
                    to_hbm(q, a);
                }
                q += 1;
                if (C::NL > 0) __syncthreads();
            };
            for (int i = 0; i < nit; i += 4) {
                it(std::integral_constant<int, 0>{}, pre[0], pmw[0]);
                it(std::integral_constant<int, 1>{}, pre[1 % PF], pmw[1 % PF]);
                it(std::integral_constant<int, 2>{}, pre[2 % PF], pmw[2 % PF]);
                it(std::integral_constant<int, 3>{}, pre[3 % PF], pmw[3 % PF]);
            }
        } else {
            auto it = [&](auto ph) {
                V a[Q];
                bool blocked[CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++) blocked[c] = false;
                const bool lv = live1(q);
                if (lv) {
                    gather(xedge, ph, q, a, blocked);
                    reflect_v(a, blocked);
                    collide_v(a, blocked, omega, care1);
                    if (two && std::is_same<ST, __half>::value) {
#pragma unroll
                        for (int k = 0; k < Q; k++) {
                            float w[CPL];
#pragma unroll
                            for (int c = 0; c < CPL; c++) w[c] = ring_value(L::get(a[k], c));
                            a[k] = L::make(w);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < Q; k++) a[k] = h_dy0[0][0];  // any ordinary value: never consumed
                }
                if (two) {
                    second(ph, a);
                    remember(ph, a, blocked);  // unconditional: a dead history entry frees its registers
                } else if (lv) {
                    to_hbm(q, a);
                }
                q += 1;
                __syncthreads();
            };
            for (int i = 0; i < nit; i += 4) {
                it(std::integral_constant<int, 0>{});
                it(std::integral_constant<int, 1>{});
                it(std::integral_constant<int, 2>{});
                it(std::integral_constant<int, 3>{});
            }
        }
    };

    // ring columns nobody writes (the pads, and window columns 0 and W-1 of an even level): give
    // them an ordinary value once, so that the edge lanes reading them stay on the fast paths
    if (writes_lds && j == 0) {
        constexpr int NSR = C::DA * 3 + 4 * 6;
        for (int n = lane; n < 4 * NSR; n += 64) {
            const int sr = n >> 2, col = n & 3;
            const int off = sr < C::DA * 3 ? (sr / 3) * C::A_ROW + (sr % 3) * C::RW * 4
                                           : C::B_OFF + ((sr - C::DA * 3) / 6) * C::B_ROW + ((sr - C::DA * 3) % 6) * C::RW * 4;
            const int idx = col < 2 ? col : C::RW - 4 + col;
            *reinterpret_cast<float*>(lds + (X - 1) * C::LEVB + off + idx * 4) = 1.0f;
        }
    }
    const int need_hi = min(xo + C::SW, M) + MARG;
    if (xo - MARG + BS * (j * CPL) >= need_hi) {
        if (C::NL > 0 || X > 1)
            for (int i = 0; i < nit; i++) __syncthreads();
        return;
    }
    if (wave_xedge) run(std::true_type{});
    else run(std::false_type{});
}

static int auto_chunk_rows2(int nrows, int nstrips, int extra, int capacity) {
    long long best_cost = -1;
    int best = nrows < 4 ? (nrows > 0 ? nrows : 1) : 4;
    for (int ry = 512; ry >= 4; ry -= 4) {
        if (ry > nrows && ry != 4) continue;
        const int r = ry < nrows ? ry : nrows;
        const long long wgs = (long long)nstrips * ((nrows + r - 1) / r);
        const long long rounds = (wgs + capacity - 1) / capacity;
        const long long cost = rounds * (r + extra);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
    }
    return best;
}

template <int XMODE, int T, int NB, int CPL, int MARG, typename ST>
static void launch_tb2_t(const ST* gin, ST* gout, const uint64_t* mask, const Geom& geo, int row0,
                         int nrows, int Ry, hipStream_t st) {
    using C = TB2<T, NB, CPL, MARG>;
    const int nstrips = (geo.M + C::SW - 1) / C::SW;
    if (Ry <= 0) {
        static std::atomic<int> capacity_cache{0};
        int capacity = capacity_cache.load(std::memory_order_relaxed);
        if (capacity == 0) {
            int dev = 0, cus = 256, per_cu = 1;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                    &per_cu, step_tb2_kernel<XMODE, T, NB, CPL, MARG, ST>, C::NW * 64, 0) != hipSuccess || per_cu < 1)
                per_cu = 1;
            capacity = cus * per_cu;
            capacity_cache.store(capacity, std::memory_order_relaxed);
        }
        Ry = auto_chunk_rows2(nrows, nstrips, (T - 1) + 3 * (C::NS - 1) + 1, capacity);
    }
    const int nchunks = (nrows + Ry - 1) / Ry;
    hipLaunchKernelGGL((step_tb2_kernel<XMODE, T, NB, CPL, MARG, ST>), dim3((unsigned)(nstrips * nchunks)),
                       dim3(C::NW * 64), 0, st, gin, gout, mask, geo, row0, nrows, Ry, nstrips);
}

template <int T, int NB, int CPL, int MARG>
static void launch_tb2_T(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                         bool half, int row0, int nrows, int Ry, hipStream_t st) {
    if (xmode == XMODE_PERIODIC) {
        if (half) launch_tb2_t<XMODE_PERIODIC, T, NB, CPL, MARG>((const __half*)gin, (__half*)gout, mask, geo, row0, nrows, Ry, st);
        else launch_tb2_t<XMODE_PERIODIC, T, NB, CPL, MARG>((const float*)gin, (float*)gout, mask, geo, row0, nrows, Ry, st);
    } else {
        if (half) launch_tb2_t<XMODE_REFERENCE, T, NB, CPL, MARG>((const __half*)gin, (__half*)gout, mask, geo, row0, nrows, Ry, st);
        else launch_tb2_t<XMODE_REFERENCE, T, NB, CPL, MARG>((const float*)gin, (float*)gout, mask, geo, row0, nrows, Ry, st);
    }
}

#ifndef LBM_TB2_NB4
#define LBM_TB2_NB4 12
#endif
#ifndef LBM_TB2_NB6
#define LBM_TB2_NB6 8
#endif

// T updates of rows [row0, row0+nrows) in one launch with the two-levels-per-wave kernel.
// Returns hipErrorNotSupported for depths it is not instantiated for.
hipError_t launch_step_tb2(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                           bool half, int T, int row0, int nrows, int Ry, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    switch (T) {
        case 4: launch_tb2_T<4, LBM_TB2_NB4, 2, 8>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 6: launch_tb2_T<6, LBM_TB2_NB6, 2, 8>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

bool tb2_has_depth(int T) { return T == 4 || T == 6; }

}  // namespace lbm
