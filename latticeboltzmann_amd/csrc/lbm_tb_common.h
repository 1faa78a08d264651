// Shared pieces of the LDS temporal-blocking kernels (lbm_tb.hip: one level per wave and row;
// tools/lbm_tb2.hip: two levels per wave and row, not built).
#pragma once
#include <type_traits>

#ifndef LBM_TB_PADC
#define LBM_TB_PADC 0
#endif

#include "lbm_internal.h"

namespace lbm {



// LDS image of one time level.  Each stored population is read by exactly one cell of the next
// level: the dy = -1 populations (slots 0-2) of a row one iteration after they were written, the
// dy = 0 ones (slots 3-5) two, the dy = +1 ones (slots 6-8) three iterations after.  So slots 0-2
// need a ring of only DA = 2 rows, the others (and the flag bytes) 4: 124 instead of 148 bytes per
// column and level.  The reference edge rule ("the last row / column keeps its own value",
// latticeboltzmann.py:115-120) would read a cell's own slot 0-2 value two iterations late; instead
//   - the stage that has just written the last domain row N-1 writes, one iteration later, a
//     phantom row N whose slot 0-2 entries are row N-1's shifted by dx, so that the ordinary pull
//     of row N-1 finds its own values there;
//   - the lane of the last column M-1 also drops its slot-0 value into a 4-entry side ring
//     (E_OFF), from which the x-edge code path takes it two iterations later.
template <int T, int NB, int CPL, int MARG>
struct TB {
    static_assert(T - 1 <= MARG, "margin too small for this depth");
    static_assert(NB % CPL == 0, "a lane owns CPL 64-column blocks");
    static constexpr int DA = 2;                 // ring depth of the dy = -1 slots
    static constexpr int W = NB * 64;            // window width (columns computed per stage)
    static constexpr int SW = W - 2 * MARG;      // strip width (columns stored)
    // Strip s stores columns [s*SW - SH, (s+1)*SW - SH) (clipped to the domain); SH = LBM_TB_SH, a
    // development knob that stays 0 (TBStore in lbm_tb.hip).  Strips must start on multiples of 16 fp32 columns = 64 bytes, so that
    // no 64-byte write request is shared by two workgroups (two XCDs): with MARG = 8 the WINDOW then
    // starts 32 bytes into a request and every wave-wide store is five L1->L2 write requests instead of
    // four (TCP_TCC_WRITE_REQ 1.22x the ideal at T = 4), but those halves meet in one L2 and cost nothing
    // - shifting the strips so that the windows are aligned instead (SH = 8) costs 17 % at T = 4 (275 ->
    // 227 GLUPS), any other shift 7-8 %, 2-6 % at T = 2 / 3 (profiles/r03_tb_hbm.md, round 3).
    // (What does get shifted, since the end of round 3, is the set of columns the LAST stage works on.)
#ifndef LBM_TB_SH
#define LBM_TB_SH 0
#endif
    // A slot row holds exactly the W window columns (a multiple of 64 floats = 256 bytes) and a
    // level is padded to a multiple of 256 bytes, so every (slot, ring row) offset fits the 256-byte
    // granular offset fields of ds_read2st64_b32 / ds_write2st64_b32: one address register per
    // x-shift serves all nine slots in all four phases, and the row loop has no LDS address
    // arithmetic at all.  The window-edge lanes (whose results are nobody's input, see `care`) read
    // their missing neighbour column from the adjacent slot row instead of a pad column: ordinary
    // populations; FRONT bytes in front of level 1 give its first slot row the columns -1 ... -MARG-1
    // (stage T may work on columns shifted by up to MARG, TBStore in lbm_tb.hip).
    static constexpr int PADC = LBM_TB_PADC;     // pad columns either side of a slot row (0; 1 = the older layout)
    static constexpr int RW = W + 2 * PADC;      // floats per slot row
    static constexpr int FRONT = 4 * MARG + 16;
    static constexpr int A_ROW = 3 * RW * 4;     // bytes per ring row of slots 0-2
    static constexpr int B_ROW = 6 * RW * 4;     // bytes per ring row of slots 3-8
    static constexpr int B_OFF = DA * A_ROW;
    static constexpr int F_OFF = B_OFF + 4 * B_ROW;          // flag bytes, ring of 4 rows of RW bytes
    static constexpr int E_OFF = (F_OFF + 4 * RW + 3) / 4 * 4;  // last column's slot 0, ring of 4 floats
    static constexpr int LEVB = PADC ? (E_OFF + 16 + 15) / 16 * 16 : (E_OFF + 16 + 255) / 256 * 256;  // bytes per level
    static constexpr int LDSB = FRONT + (T - 1) * LEVB + 16;
    static_assert(LDSB <= 160 * 1024, "LDS image too large");
    static constexpr int NW = T * NB / CPL;      // waves: one per (stage, CPL blocks)
    // byte offset of population slot sl in the ring row written `age` iterations before phase ph
    static constexpr int slot_off(int sl, int ph, int age) {
        return sl < 3 ? ((ph - age) & (DA - 1)) * A_ROW + sl * RW * 4
                      : B_OFF + ((ph - age) & 3) * B_ROW + (sl - 3) * RW * 4;
    }
    static constexpr int flag_off(int ph, int age) { return F_OFF + ((ph - age) & 3) * RW; }
    static constexpr int edge_off(int ph, int age) { return E_OFF + ((ph - age) & 3) * 4; }
};

// value type of a lane: one cell (float) or two cells 64 columns apart (packed float2)
template <int CPL> struct LaneV;
template <> struct LaneV<1> {
    using V = float;
    static __device__ __forceinline__ V make(const float* v) { return v[0]; }
    static __device__ __forceinline__ float get(V v, int) { return v; }
};
template <> struct LaneV<2> {
    using V = f2;
    static __device__ __forceinline__ V make(const float* v) { return f2{v[0], v[1]}; }
    static __device__ __forceinline__ float get(V v, int c) { return c == 0 ? v.x : v.y; }
};
// blocked flags of a lane's cells: the byte that travels through the rings (0 / nonzero), the
// lane predicate and its ballot (taken where the predicate is computed, see reflect2)
template <int CPL> struct Flags {
    unsigned raw[CPL];
    bool b[CPL];
    lanemask m[CPL];
    __device__ __forceinline__ void set(int c, unsigned r) { raw[c] = r; b[c] = r != 0; m[c] = ballot(r != 0); }
};
__device__ __forceinline__ void reflect_v(float* a, const bool* b) { reflect(a, b[0]); }
__device__ __forceinline__ void reflect_v(f2* a, const bool* b) { reflect2(a, b[0], b[1]); }
__device__ __forceinline__ void reflect_v(float* a, const Flags<1>& f) { reflect(a, f.b[0]); }
__device__ __forceinline__ void reflect_v(f2* a, const Flags<2>& f) { reflect2(a, f.b[0], f.b[1], f.m[0], f.m[1]); }
__device__ __forceinline__ void collide_v(float* a, const bool* b, const Omega& om, const bool* care) { collide(a, b[0], om, care[0]); }
__device__ __forceinline__ void collide_v(f2* a, const bool* b, const Omega& om, const bool* care) { collide2(a, b[0], b[1], om, care[0], care[1]); }
// collide and hand the result to sink(const V*): see collide2_to (lbm_device.h); cm = ballot(care)
template <class Sink>
__device__ __forceinline__ void collide_to(float* a, const Flags<1>& f, const Omega& om, const bool* care, const lanemask*, Sink sink) {
    collide(a, f.b[0], om, care[0]);
    sink(a);
}
template <class Sink>
__device__ __forceinline__ void collide_to(f2* a, const Flags<2>& f, const Omega& om, const bool*, const lanemask* cm, Sink sink) {
    collide2_to(a, f.b[0], f.b[1], f.m[0], f.m[1], om, cm[0], cm[1], sink);
}


}  // namespace lbm
