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
//     stage 2..T  pull from the LDS rings of level s-1 (struct TB below: 2 rows for the dy = -1
//                 populations, 4 for the others and the blocked flags);
//     stage s<T   writes its row into the LDS rings of level s, stage T stores to HBM.
// One wave per (stage, 64*CPL columns): T*NB/CPL waves, all running the same code; with CPL = 2 a
// lane owns two cells 64 columns apart and collide runs on packed fp32.  Level 1 is right in
// every window column; each further level loses one column at either window edge (its neighbour
// lies outside the window), so only the inner W - 2*MARG columns (MARG >= T-1, a multiple of 8)
// are stored: windows of neighbouring workgroups overlap by 2*MARG columns (halo recompute,
// 4-8 %).  All stages run concurrently; one barrier per iteration separates an iteration's ring
// writes from the next iteration's reads (skew 2 per stage).  Per T updates the cells are read
// once (1.04-1.07x with the overlap, mostly L2 hits) and written once: HBM bytes per cell-update =
// 72/T (+ 3(T-1) warm-up rows per chunk).  Every cell goes through the same pull/reflect/collide
// device functions as the T=1 kernel, so results are bit-identical to T single updates.
//
// All loads in the row loops are unconditional and straight-line (rows outside a stage's range are
// clamped to a valid row instead of skipped), and nothing else in the stage-1 loop touches vector
// memory, so that hipcc can use counted s_waitcnt vmcnt(N) and the prefetch really stays in
// flight across two iterations and barriers.  Only stage 1 touches the bit-packed mask; it hands
// each cell's blocked flag to the later stages through LDS.
#include <atomic>
#include <type_traits>

#include "lbm_internal.h"

// Window width (NB 64-column blocks), cells per
// lane and margin for each launch depth, from sweeps on MI355X at 16384 x 16384.  GLUPS, periodic /
// reference edges, round 1: T=2 143 / 146, T=3 213 / 210, T=4 249 / 237, T=5 241 / 222, T=6 274 / 253;
// end of round 2 (the common-case collide of lbm_device.h, immediate-only LDS addressing, ...):
// T=2 143, T=3 215, T=4 276 / 266, T=5 279, T=6 318 / 300 (profiles/r02_tb_experiments.md).
#ifndef LBM_TB_NB2
#define LBM_TB_NB2 12
#define LBM_TB_CPL2 2
#endif
#ifndef LBM_TB_NB3
#define LBM_TB_NB3 10  // round 3: 218.4 against 214.5 GLUPS with NB = 8 (158 KB of LDS, 15 waves; NB = 6: 209.5)
#define LBM_TB_CPL3 2
#endif
#ifndef LBM_TB_NB4
#define LBM_TB_NB4 6
#define LBM_TB_CPL4 2
#endif
#ifndef LBM_TB_NB5
#define LBM_TB_NB5 4
#define LBM_TB_CPL5 2
#endif
#ifndef LBM_TB_NB6
#define LBM_TB_NB6 4
#define LBM_TB_CPL6 2
#endif
#ifndef LBM_TB_MARG2
#define LBM_TB_MARG2 16
#endif
#ifndef LBM_TB_MARG3
#define LBM_TB_MARG3 16
#endif
#ifndef LBM_TB_MARG4
#define LBM_TB_MARG4 8
#endif
#ifndef LBM_TB_MARG5
#define LBM_TB_MARG5 8
#endif
#ifndef LBM_TB_MARG6
#define LBM_TB_MARG6 8
#endif
#ifndef LBM_TB_NODEAD
#define LBM_TB_NODEAD 0
#endif
#ifndef LBM_TB_PF
#define LBM_TB_PF 2
#endif
#ifndef LBM_TB_STORE
#define LBM_TB_STORE st
#endif
#ifndef LBM_TB_LOAD
#define LBM_TB_LOAD ld
#endif
#ifndef LBM_TB_REMAP
#define LBM_TB_REMAP 1
#endif
// Workgroup -> XCD mapping at depth 4 with fp32 storage (the depth that sits on the HBM ceiling of
// this access pattern): plain round-robin, i.e. all eight XCDs march down the SAME band of rows,
// is 2.8 % faster there than one band per XCD (same-box A/B: 276 against 268 GLUPS; T = 2: -3.6 %,
// T = 3: 0, T = 5 / 6: -1..2 %, fp16: -0.5 %).
#ifndef LBM_TB_REMAP4
#define LBM_TB_REMAP4 0
#endif
#ifndef LBM_TB_PADINIT
#define LBM_TB_PADINIT 1
#endif

// 1: write-through streaming stores and line-aligned windows where the rule of TBStore allows them
#ifndef LBM_TB_STREAM_STORE
#define LBM_TB_STREAM_STORE 1
#endif

#include "lbm_tb_common.h"

namespace lbm {

// How stage T writes its rows.  The T = 1 kernel gained 2 % from `nt sc1` stores (write-through at agent
// scope, st_stream in lbm_device.h); here that only pays when every wave-wide store covers whole
// 128-byte lines - written through, partial lines are poison (T = 4: 275 -> 148 GLUPS).  The windows
// start MARG columns left of their strips, 32 or 64 bytes into a line, so with fp32 storage stage T
// works on columns SHIFTED by dsh in [-MARG, MARG] against the other stages (it reads the rings of level
// T-1 at any column it likes; the columns it thereby leaves out at one window edge and adds beyond the
// other are outside the strip): dsh = the shift that puts its wave-wide stores on line boundaries.
// Strips start on multiples of 16 columns, so this is +-8 or 0 / 16; only the line at a strip boundary
// that is not a line boundary (SW is an odd multiple of 16 columns at T >= 4: every second boundary)
// is still written in two halves.  fp16 storage (a line is 64 columns) keeps plain stores.
// Same box, 16384 x 16384, GLUPS: see profiles/r03_tb_hbm.md section 3d.
template <int MARG, typename ST>
struct TBStore {
    static constexpr bool LINES = LBM_TB_STREAM_STORE && sizeof(ST) == 4;
    static constexpr int SH = LBM_TB_SH;
    // first stored column xo -> shift of stage T's columns
    static __device__ __forceinline__ int shift(int xo) {
        if (!LINES) return 0;
        int r = (MARG - xo) & 31;
        if (r > MARG) r -= 32;
        return r >= -MARG ? r : 0;  // (xo not a multiple of 16: LBM_TB_SH experiments)
    }
};

// cell c of a lane's nine populations to its place in the output row, streaming stores (st_stream)
template <int CPL, int K = 0, typename ST, typename V>
__device__ __forceinline__ void store_cell(ST* p, const V* out, int c) {
    if constexpr (K < Q) {
        st_stream<slot_of(K) * LANES * (int)sizeof(ST)>(p, LaneV<CPL>::get(out[K], c));
        store_cell<CPL, K + 1>(p, out, c);
    }
}

// Ring rows are addressed by ITERATION, not by row: in iteration i a stage writes its row into
// ring row i mod depth of its level, and the next stage (two rows behind) finds rows q+1, q, q-1
// in the ring rows written one, two and three iterations earlier.  The row loop is unrolled by four, so every LDS address is a
// per-lane base plus a compile-time offset.  A cell's blocked flag travels through the rings as
// a tenth, one-byte "slot": only stage 1 reads the bit-packed mask.
template <int XMODE, int T, int NB, int CPL, int MARG, typename ST>
__global__ __launch_bounds__((TB<T, NB, CPL, MARG>::NW * 64)) void step_tb_kernel(
    const ST* __restrict__ gin, ST* __restrict__ gout, const uint64_t* __restrict__ mask, Geom geo,
    int row0, int nrows, int Ry, int nstrips) {
    using C = TB<T, NB, CPL, MARG>;
    using L = LaneV<CPL>;
    using V = typename L::V;
    constexpr int PF = LBM_TB_PF;  // stage-1 rows in flight (1, 2 or 4: divides the unroll)
    __shared__ __attribute__((aligned(16))) unsigned char lds[C::LDSB];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    constexpr int REMAP = (T == 4 && std::is_same<ST, float>::value) ? LBM_TB_REMAP4 : LBM_TB_REMAP;
    const unsigned wg = REMAP == 1 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int nchunks_ = gridDim.x / nstrips;
    const int strip = REMAP == 2 ? wg / nchunks_ : wg % nstrips;
    const int chunk = REMAP == 2 ? wg % nchunks_ : wg / nstrips;
    const int xo = strip * C::SW - TBStore<MARG, ST>::SH;  // first stored column (strip 0: clipped at column 0)
    const int y0 = row0 + chunk * Ry;
    const int y1 = min(y0 + Ry, row0 + nrows);
    const int N = geo.N, M = geo.M, G = geo.G;
    const bool wrapy = geo.top == YEDGE_WRAP;  // top and bottom wrap together
    const int qlo = geo.top == YEDGE_DOMAIN ? 0 : -(1 << 28);  // rows that exist
    const int qhi = geo.bot == YEDGE_DOMAIN ? N : (1 << 28);
    const long long RS = geo.rowStride;
    const Omega omega = omega_of(geo);
    const float su0 = geo.surr[0], su1 = geo.surr[1], su2 = geo.surr[2], su3 = geo.surr[3],
                su4 = geo.surr[4], su5 = geo.surr[5], su6 = geo.surr[6], su7 = geo.surr[7],
                su8 = geo.surr[8];
    auto surr = [&](int k) {  // k is a compile-time constant after unrolling
        return k == 0 ? su0 : k == 1 ? su1 : k == 2 ? su2 : k == 3 ? su3 : k == 4 ? su4
             : k == 5 ? su5 : k == 6 ? su6 : k == 7 ? su7 : su8;
    };

    // ---- role: stage s (wave-uniform) and this lane's CPL columns ------------------------------
    constexpr int WPS = NB / CPL;  // waves per stage
    const int s = wave / WPS + 1;
    const int j = wave % WPS;
    const int dsh = s == T ? TBStore<MARG, ST>::shift(xo) : 0;  // stage T: stores on line boundaries (TBStore)
    const int cx = C::PADC + j * (64 * CPL) + lane + dsh;  // ring-row index of cell 0; cell c is 64*c further
    int off0[CPL];
    unsigned bo0[CPL], bom[CPL], bop[CPL];  // byte offsets of the own / dx = -1 / dx = +1 source column in a row
    unsigned mwoff[CPL];                    // this cell's 32-bit mask word (byte offset in a mask row)
    bool stored[CPL], ob_m[CPL], ob_p[CPL], last_col[CPL], care[CPL];
    lanemask caremask[CPL];
    unsigned mbit[CPL];  // this cell's bit in its 32-bit mask word
    bool any_xedge = false;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int x = xo - MARG + j * (64 * CPL) + c * 64 + lane + dsh;  // logical column
        int xn;                                                       // storage column
        if (XMODE == XMODE_PERIODIC) { xn = x % M; if (xn < 0) xn += M; }
        else xn = min(max(x, 0), M - 1);  // outside the domain: a valid address, never stored
        stored[c] = x >= xo && x >= 0 && x < xo + C::SW && x < M;
        // level s is only right s-1 columns inside the window (see the header): the columns
        // beyond compute on stale ring contents and nobody reads what they produce
        const int wc = j * (64 * CPL) + c * 64 + lane + dsh;
        care[c] = wc >= s - 1 && wc < C::W - (s - 1);
        caremask[c] = ballot(care[c]);
        // x edge rule (reference mode only): source of dx = -1 / +1 outside the domain
        ob_m[c] = XMODE != XMODE_PERIODIC && x + 1 >= M;
        ob_p[c] = XMODE != XMODE_PERIODIC && x - 1 < 0;
        last_col[c] = XMODE != XMODE_PERIODIC && x == M - 1;
        any_xedge = any_xedge || ob_m[c] || ob_p[c];
        int xm1 = xn + 1, xp1 = xn - 1;
        if (XMODE == XMODE_PERIODIC) { xm1 = xm1 >= M ? 0 : xm1; xp1 = xp1 < 0 ? M - 1 : xp1; }
        else { xm1 = xm1 >= M ? xn : xm1; xp1 = xp1 < 0 ? xn : xp1; }
        off0[c] = (xn >> 6) * CHUNK + (xn & 63);
        bo0[c] = (unsigned)off0[c] * (unsigned)sizeof(ST);
        bom[c] = (unsigned)((xm1 >> 6) * CHUNK + (xm1 & 63)) * (unsigned)sizeof(ST);
        bop[c] = (unsigned)((xp1 >> 6) * CHUNK + (xp1 & 63)) * (unsigned)sizeof(ST);
        mwoff[c] = (unsigned)(xn >> 5) * 4u;
        mbit[c] = xn & 31;
    }
    const bool wave_xedge = __any(any_xedge);

    // ---- rows ----------------------------------------------------------------------------------
    const int qstart = y0 - (T - 1) - 2 * (s - 1);                        // row of iteration 0
    const int rlo = max(y0 - (T - s), qlo), rhi = min(y1 + (T - s), qhi);  // rows this stage makes
    const int niter = (y1 - y0) + 3 * (T - 1);
    const int nit = (niter + 3) & ~3;  // unrolled by 4; every wave runs the same number of barriers
    auto wrap_row = [&](int r) {  // storage row of a logical row in [-N, 2N)
        if (wrapy) { r = r < 0 ? r + N : r; r = r >= N ? r - N : r; }
        return r;
    };
    auto clampr = [&](int r) { return min(max(r, rlo), rhi - 1); };

    // LDS views (layout: struct TB).  Level l (1..T-1) lives at lds + (l-1)*LEVB.
    unsigned char* const wr_lev = lds + C::FRONT + (s - 1) * C::LEVB;  // this stage's level (s < T)
    const unsigned char* const rd_lev = lds + C::FRONT + (s - 2) * C::LEVB;  // level s-1 (s > 1)
    unsigned char* const wr_base = wr_lev + cx * 4;
    const unsigned char* const rd_base = rd_lev + cx * 4;
    unsigned char* const wr_flag = wr_lev + cx;
    const unsigned char* const rd_flag = rd_lev + cx;

    // stage 1: issue the CPL x 9 pulled loads of logical row r (clamped into this stage's range)
    // and the mask words.  Pure loads from always-valid addresses (a source outside the domain is
    // replaced by the own cell); the inlet rule is applied when the row is consumed (inlet_fix),
    // so that no load sits behind a branch and the vmcnt bookkeeping stays exact.
    // Addressing: the three row bases are wave-uniform (SGPR pairs) and a lane carries 32-bit byte
    // offsets, so every load is `global_load_dword v, voffset, s[base] offset:slot` without any
    // per-row address arithmetic on the vector unit (as in pull(), lbm_device.h).
    auto load_row = [&](auto xedge, int r, V* dst, uint32_t* mw) {
        r = clampr(r);
        const int rn = wrap_row(r);
        const bool ob_u = r - 1 < qlo, ob_d = XMODE != XMODE_PERIODIC && r + 1 >= qhi;
        const int rnu = ob_u ? rn : wrap_row(r - 1), rnd = ob_d ? rn : wrap_row(r + 1);
        const char* p0 = reinterpret_cast<const char*>(gin + (long long)(rn + G) * RS);
        const char* pu = reinterpret_cast<const char*>(gin + (long long)(rnu + G) * RS);
        const char* pd = reinterpret_cast<const char*>(gin + (long long)(rnd + G) * RS);
        // column offsets as plain values (never select between array elements behind a branch:
        // that turns the arrays into scratch memory).  Below the last domain row the dy = -1
        // source is the own cell: own row (pd = p0) AND own column.
        unsigned o0[CPL], om[CPL], op[CPL], omd[CPL], opd[CPL];
        bool xm[CPL], xp[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            unsigned a0 = bo0[c], am = bom[c], ap = bop[c];
            // keep the offsets 32-bit values of THIS iteration: hoisted out of the loop they would
            // be zero-extended once and every load would pay a 64-bit vector add for its address
            asm volatile("" : "+v"(a0), "+v"(am), "+v"(ap));
            o0[c] = a0; om[c] = am; op[c] = ap;
            omd[c] = ob_d ? a0 : am;
            opd[c] = ob_d ? a0 : ap;
            xm[c] = ob_m[c]; xp[c] = ob_p[c];
        }
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dy = dy_of(k), dx = dx_of(k);
            const size_t sl = (size_t)slot_of(k) * LANES * sizeof(ST);
            const char* rb = dy == 1 ? pu : (dy == 0 ? p0 : pd);
            float v[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const unsigned off = dx == 0 ? o0[c] : (dy == -1 ? (dx == 1 ? opd[c] : omd[c])
                                                                  : (dx == 1 ? op[c] : om[c]));
                const ST* pp = reinterpret_cast<const ST*>(rb + sl + (size_t)off);
                if (decltype(xedge)::value && dx != 0) {
                    // a source column outside the domain: the own cell (any valid address does
                    // for dx = +1, which inlet_fix overwrites)
                    const ST* own = reinterpret_cast<const ST*>(p0 + sl + (size_t)o0[c]);
                    pp = (dx == 1 ? xp[c] : xm[c]) ? own : pp;
                }
                v[c] = LBM_TB_LOAD(pp);
            }
            dst[k] = L::make(v);
        }
        const char* mrow = reinterpret_cast<const char*>(mask + (long long)(rn + G) * geo.Mb);
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            unsigned mo = mwoff[c];
            asm volatile("" : "+v"(mo));
            mw[c] = *reinterpret_cast<const uint32_t*>(mrow + (size_t)mo);
        }
    };
    auto inlet_fix = [&](auto xedge, int r, V* a) {
        if (XMODE == XMODE_PERIODIC) return;  // no domain edge in x, and y wraps or is a seam
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
        // top inlet: a wave-uniform, rare branch (row 0 only) rather than selects in every row, so
        // that the ambient values need no registers in the row loop (a spilled scalar would be
        // reloaded there by a vmem load whose in-order vmcnt wait drains the stage-1 prefetch)
        if (ob_u) {
            asm volatile("; top inlet");  // keeps hipcc from turning the branch into selects in every row
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
    // stages 2..T, iteration phase PH = i & 3: gather the pulled populations of row r and the
    // blocked flags from the ring of level s-1
    auto gather = [&](auto xedge, auto ph, int r, V* a, Flags<CPL>& blocked) {
        constexpr int PH = decltype(ph)::value;
        // the previous stage wrote row r+1 one, row r two and row r-1 three iterations ago (below
        // the last domain row: the phantom row, which holds this row's own dy = -1 values)
        const bool ob_u = XMODE != XMODE_PERIODIC && r - 1 < qlo;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k);
            const int src = C::slot_off(sl, PH, dy == 1 ? 3 : (dy == 0 ? 2 : 1)) - 4 * dx;
            float v[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                v[c] = *reinterpret_cast<const float*>(rd_base + src + 256 * c);
                if (decltype(xedge)::value && dx == -1) {
                    // source column M is outside the domain: the own cell's value (own row AND
                    // column), two iterations old; slot 0 has left its 2-row ring, see edge_off
                    const float own = sl == 0
                        ? *reinterpret_cast<const float*>(rd_lev + C::edge_off(PH, 2))
                        : *reinterpret_cast<const float*>(rd_base + C::slot_off(sl, PH, 2) + 256 * c);
                    v[c] = ob_m[c] ? own : v[c];
                }
                if (decltype(xedge)::value && dx == 1) v[c] = ob_p[c] ? surr(k) : v[c];
            }
            a[k] = L::make(v);
        }
        // the flags last: their first use (reflect's wave vote) would otherwise make hipcc wait
        // for them before it has issued the population reads
#pragma unroll
        for (int c = 0; c < CPL; c++) blocked.set(c, rd_flag[C::flag_off(PH, 2) + 64 * c]);
        if (ob_u) {  // top inlet, wave-uniform and rare (see inlet_fix)
            asm volatile("; top inlet");  // keeps hipcc from turning the branch into selects in every row
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
    // With fp16 storage every intermediate time level is rounded through half on its way into the
    // ring, exactly as a store to / load from the fp16 field between two single updates would
    // (the reference rounds `newcells` through half on every inner timestep too,
    // lb_cuda_kernel.cu:292,361): a depth-T launch equals T single updates bit for bit for both
    // storage types, so fp16 results do not depend on launch depth, tuning or slab count.
    auto ring_value = [](float v) {
        if (std::is_same<ST, __half>::value) return __half2float(__float2half_rn(v));
        return v;
    };
    // the same for a lane's value(s): two cells go through one packed conversion (v_cvt_pk_f16_f32,
    // round to nearest even like __float2half_rn)
    auto ring_round = [&](V v) {
        if (!std::is_same<ST, __half>::value) return v;
        if constexpr (CPL == 2) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            return __builtin_convertvector(__builtin_convertvector(v, h2), V);
        } else {
            float x[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) x[c] = ring_value(L::get(v, c));
            return L::make(x);
        }
    };
    auto to_lds = [&](auto xedge, auto ph, int q, const V* a, const Flags<CPL>& blocked) {
        constexpr int PH = decltype(ph)::value;
        if (XMODE != XMODE_PERIODIC && q == qhi) {
            // phantom row below the last domain row (wave-uniform, once per chunk at most): the
            // next level pulls its dy = -1 populations of row N-1 from here with the usual
            // column shift and must find row N-1's own values
#pragma unroll
            for (int sl = 0; sl < 3; sl++) {
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const float v = *reinterpret_cast<const float*>(
                        wr_base + C::slot_off(sl, PH, 1) + 4 * (sl - 1) + 256 * c);
                    *reinterpret_cast<float*>(wr_base + C::slot_off(sl, PH, 0) + 256 * c) = v;
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const V rv = ring_round(a[k]);
#pragma unroll
            for (int c = 0; c < CPL; c++)
                *reinterpret_cast<float*>(wr_base + C::slot_off(slot_of(k), PH, 0) + 256 * c) = L::get(rv, c);
        }
#pragma unroll
        for (int c = 0; c < CPL; c++) wr_flag[C::flag_off(PH, 0) + 64 * c] = (unsigned char)blocked.raw[c];
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

    // does this stage produce row r?  Also true for the phantom row below the last domain row
    // (to_lds); wave-uniform
    auto live = [&](int r) {
        return (r >= rlo && r < rhi) || (XMODE != XMODE_PERIODIC && s < T && r == qhi);
    };
    auto run = [&](auto xedge) {
        int q = qstart;
        if (s == 1) {
            // PF rows in flight: row q+PF is requested while row q is consumed; the mask words
            // travel with their row (vmcnt retires in order)
            V pre[PF][Q];
            uint32_t pmw[PF][CPL];
#pragma unroll
            for (int d = 0; d < PF; d++) load_row(xedge, q + d, pre[d], pmw[d]);
            auto it = [&](auto ph, V* pr, uint32_t* mw) {
                V a[Q];
                Flags<CPL> blocked;
#pragma unroll
                for (int k = 0; k < Q; k++) a[k] = pr[k];
#pragma unroll
                for (int c = 0; c < CPL; c++) blocked.set(c, __builtin_amdgcn_ubfe(mw[c], mbit[c], 1u));
                load_row(xedge, q + PF, pr, mw);
                // rows outside this stage's range (pipeline fill and drain: 3(T-1) of the
                // iterations) are nobody's input: skip their arithmetic, keep loads and barrier
                if (live(q)) {
                    inlet_fix(xedge, clampr(q), a);
                    reflect_v(a, blocked);
                    collide_to(a, blocked, omega, care, caremask,
                               [&](const V* out) { to_lds(xedge, ph, q, out, blocked); });
                }
                q += 1;
                __syncthreads();
            };
            for (int i = 0; i < nit; i += 4) {
                it(std::integral_constant<int, 0>{}, pre[0], pmw[0]);
                it(std::integral_constant<int, 1>{}, pre[1 % PF], pmw[1 % PF]);
                it(std::integral_constant<int, 2>{}, pre[2 % PF], pmw[2 % PF]);
                it(std::integral_constant<int, 3>{}, pre[3 % PF], pmw[3 % PF]);
            }
        } else {
            ST* orow = gout + (long long)(wrap_row(clampr(q)) + G) * RS;
            auto it = [&](auto ph) {
                if (live(q)) {
                    V a[Q];
                    Flags<CPL> blocked;
                    gather(xedge, ph, q, a, blocked);
                    reflect_v(a, blocked);
                    collide_to(a, blocked, omega, care, caremask, [&](const V* out) {
                        if (s < T) {
                            to_lds(xedge, ph, q, out, blocked);
                        } else {
#pragma unroll
                            for (int c = 0; c < CPL; c++) {
                                if (stored[c]) {
                                    if constexpr (TBStore<MARG, ST>::LINES) {
                                        store_cell<CPL>(orow + off0[c], out, c);
                                    } else {
#pragma unroll
                                        for (int k = 0; k < Q; k++)
                                            LBM_TB_STORE(orow + off0[c] + slot_of(k) * LANES, L::get(out[k], c));
                                    }
                                }
                            }
                        }
                    });
                    if (s == T) {
                        // next row of the output; a wrapped domain restarts at row 0
                        orow += RS;
                        if (wrapy && wrap_row(q) == N - 1) orow = gout + (long long)G * RS;
                    }
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
    // Only the older layout with pad columns (LBM_TB_PADC = 1; the default has none, lbm_tb_common.h):
    // the pad columns either side of a ring row are never written, yet the window-edge lanes of the
    // next level read them (their results are nobody's input).  Give them an ordinary value once:
    // stale LDS there could hold anything, and one non-finite or zero-density lane sends its whole
    // wave down the exact-division path (and, for tiny omega, the general relaxation) in every row.
    if (LBM_TB_PADINIT && C::PADC && s < T && j == 0 && lane < 2 * (C::DA * 3 + 4 * 6)) {
        const int n = lane >> 1, side = lane & 1;
        const int off = n < C::DA * 3 ? (n / 3) * C::A_ROW + (n % 3) * C::RW * 4
                                      : C::B_OFF + ((n - C::DA * 3) / 6) * C::B_ROW + ((n - C::DA * 3) % 6) * C::RW * 4;
        *reinterpret_cast<float*>(wr_lev + off + side * (C::RW - 1) * 4) = 1.0f;
    }
    // A wave whose columns all lie beyond what the strip needs (the last strip of a row is
    // usually narrower than the window) only keeps the barriers company: no loads, no arithmetic.
    const int need_hi = min(xo + C::SW, M) + MARG;  // logical columns at or beyond: nobody's input
    if (!LBM_TB_NODEAD && xo - MARG + j * (64 * CPL) + dsh >= need_hi) {
        for (int i = 0; i < nit; i++) __syncthreads();
        return;
    }
    if (wave_xedge) run(std::true_type{});
    else run(std::false_type{});
}

// Rows per workgroup chunk.  A workgroup runs Ry + 3(T-1) iterations, and the grid is executed
// in rounds of (CUs x resident workgroups per CU): pick the Ry in [4, 512] that minimises
// rounds x iterations (tail effect and pipeline warm-up against each other; small grids get
// short chunks so that every CU has a workgroup).
static int auto_chunk_rows(int nrows, int nstrips, int T, int capacity) {
    long long best_cost = -1;
    int best = nrows < 4 ? (nrows > 0 ? nrows : 1) : 4;
    for (int ry = 512; ry >= 4; ry -= 4) {
        if (ry > nrows && ry != 4) continue;
        const int r = ry < nrows ? ry : nrows;
        const long long wgs = (long long)nstrips * ((nrows + r - 1) / r);
        const long long rounds = (wgs + capacity - 1) / capacity;
        const long long cost = rounds * (r + 3 * (T - 1));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
    }
    return best;
}

template <int XMODE, int T, int NB, int CPL, int MARG, typename ST>
static void launch_tb_t(const ST* gin, ST* gout, const uint64_t* mask, const Geom& geo, int row0,
                        int nrows, int Ry, hipStream_t st) {
    using C = TB<T, NB, CPL, MARG>;
    const int nstrips = (geo.M + TBStore<MARG, ST>::SH + C::SW - 1) / C::SW;
    if (Ry <= 0) {
        // per process and instantiation: all devices of a node are the same part (a benign race:
        // concurrent first calls compute the same value)
        static std::atomic<int> capacity_cache{0};
        int capacity = capacity_cache.load(std::memory_order_relaxed);
        if (capacity == 0) {
            int dev = 0, cus = 256, per_cu = 1;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                    &per_cu, step_tb_kernel<XMODE, T, NB, CPL, MARG, ST>, C::NW * 64, 0) != hipSuccess || per_cu < 1)
                per_cu = 1;
            capacity = cus * per_cu;
            capacity_cache.store(capacity, std::memory_order_relaxed);
        }
        Ry = auto_chunk_rows(nrows, nstrips, T, capacity);
    }
    const int nchunks = (nrows + Ry - 1) / Ry;
    hipLaunchKernelGGL((step_tb_kernel<XMODE, T, NB, CPL, MARG, ST>), dim3((unsigned)(nstrips * nchunks)),
                       dim3(C::NW * 64), 0, st, gin, gout, mask, geo, row0, nrows, Ry, nstrips);
}

template <int T, int NB, int CPL, int MARG>
static void launch_tb_T(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                        bool half, int row0, int nrows, int Ry, hipStream_t st) {
    if (xmode == XMODE_PERIODIC) {
        if (half) launch_tb_t<XMODE_PERIODIC, T, NB, CPL, MARG>((const __half*)gin, (__half*)gout, mask, geo, row0, nrows, Ry, st);
        else launch_tb_t<XMODE_PERIODIC, T, NB, CPL, MARG>((const float*)gin, (float*)gout, mask, geo, row0, nrows, Ry, st);
    } else {
        if (half) launch_tb_t<XMODE_REFERENCE, T, NB, CPL, MARG>((const __half*)gin, (__half*)gout, mask, geo, row0, nrows, Ry, st);
        else launch_tb_t<XMODE_REFERENCE, T, NB, CPL, MARG>((const float*)gin, (float*)gout, mask, geo, row0, nrows, Ry, st);
    }
}

// T updates of rows [row0, row0+nrows) in one launch; input rows [row0-T, row0+nrows+T) must be
// valid (or lie beyond a domain edge).  T in [2, TB_MAX_T].
hipError_t launch_step_tb(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                          bool half, int T, int row0, int nrows, int Ry, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    switch (T) {
        case 2: launch_tb_T<2, LBM_TB_NB2, LBM_TB_CPL2, LBM_TB_MARG2>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 3: launch_tb_T<3, LBM_TB_NB3, LBM_TB_CPL3, LBM_TB_MARG3>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 4: launch_tb_T<4, LBM_TB_NB4, LBM_TB_CPL4, LBM_TB_MARG4>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 5: launch_tb_T<5, LBM_TB_NB5, LBM_TB_CPL5, LBM_TB_MARG5>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        case 6: launch_tb_T<6, LBM_TB_NB6, LBM_TB_CPL6, LBM_TB_MARG6>(gin, gout, mask, geo, xmode, half, row0, nrows, Ry, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lbm
