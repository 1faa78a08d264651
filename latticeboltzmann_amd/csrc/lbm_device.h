// Device-side building blocks of the D2Q9 BGK update for gfx950 (wave64).
//
// HBM layout (private to the library).  The distribution field is AoSoA with the wavefront as
// the inner dimension:
//     field[row + G][xb][slot][lane]      row in [-G, N+G), xb in [0, Mb), slot in [0,9), lane in [0,64)
// x = 64*xb + lane, G = ghost rows.  One (row, xb) "chunk" is 9*64 elements, contiguous; every
// wave-level access to one slot is one aligned 256-byte (fp32) line pair.  Slot order is NOT the
// reference's k order: slot(k) = 3*(dy+1) + (dx+1) with (dy, dx) = e[k] = (k%3-1, k/3-1)
// (latticeboltzmann.py:30-32), so the three populations a destination row pulls from one source
// row are adjacent (768 contiguous bytes per source row).
// The obstacle mask is bit-packed: mask[row + G][xb] is a 64-bit word, bit `lane` = blocked.
//
// What is stored is the POST-COLLISION state g = collide(f) (see DESIGN.md "collide-once"): one
// update is  g' = collide(reflect(pull-stream(g))),  numpy's order being
// moments, collide, stream, reflect (latticeboltzmann.jl:107-120).
//
// Floating point: every expression below is written in the operation order numpy uses for the
// reference's vectorised path (restated in oracle/lbm_ref.c) and the library is compiled with
// -ffp-contract=off, so fp32 results are bit-identical to the oracle, not merely within 1e-5.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace lbm {

constexpr int LANES = 64;
constexpr int Q = 9;
constexpr int CHUNK = Q * LANES;  // elements per (row, xb) chunk

enum : int { XMODE_REFERENCE = 0, XMODE_PERIODIC = 1, XMODE_CUDA = 2 };
enum : int { YEDGE_DOMAIN = 0, YEDGE_WRAP = 1, YEDGE_GHOST = 2 };

struct Geom {
    int N;           // owned rows
    int M;           // columns
    int Mb;          // ceil(M / 64)
    int G;           // ghost rows above and below
    long long rowStride;  // elements per row = Mb * CHUNK
    int top;         // YEDGE_* for y-1 at y == 0
    int bot;         // YEDGE_* for y+1 at y == N-1
    double omega;    // OMEGA (latticeboltzmann.py:80), a Python float = double
    float omega_f;   // (float)omega
    int omega_f32_exact;  // omega == (double)omega_f: the fp32 product below is the same number
    int omega_tiny;       // |omega| <= 2^-34: see relax_is_identity()
    float surr[9];   // surroundings, reference k order
};

// The relaxation factor of the free cells as collide() wants it (wave-uniform, lives in SGPRs).
struct Omega {
    double d;
    float f;
    bool f32_exact;
    bool tiny;
};
__device__ __forceinline__ Omega omega_of(const Geom& geo) {
    return Omega{geo.omega, geo.omega_f, geo.omega_f32_exact != 0, geo.omega_tiny != 0};
}
// Wave votes on the lane mask itself (compare -> SGPR pair -> s_cmp): HIP's __any / __all go
// through an integer per lane (v_cndmask + v_cmp) first.
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0; }

constexpr double OMEGA_TINY_MAX = 1.0 / 17179869184.0;  // 2^-34 = 5.8e-11 (the reference's OMEGA is 1e-14)

// For |omega| <= 2^-34 -- the reference's own setting, OMEGA = 1e-14 (latticeboltzmann.py:80), is
// "f <- eq" in all but name -- the relaxation  a_k <- fl(fl(fl(a_k - eq_k) * omega) + eq_k)  returns
// eq_k EXACTLY for every free cell in an ordinary state, and the 27 subtract / multiply / add (45
// with the float64 product) per cell can be skipped without changing a bit.  Ordinary means, per
// cell: (1) every population has a clear sign bit, (2) 2^-100 <= p <= 2^100, (3) c = 1.5|u|^2 <= 1/8.
// Proof.  (1) makes the pairwise fp32 sum monotone: p >= a_k for every k, and with (2) the moments
// are finite.  (3) bounds |e.u| <= |u0| + |u1| <= 0.4083, so every bracket
// b_k = ((1 +- 3 e.u) + 4.5 (e.u)^2) - c lies in [0.3749, 2.975] (real minimum 1/2 - 1/8 at
// e.u = -1/3, a few 1e-7 of rounding), hence 0.010413 p <= eq_k <= 0.44445 p: normal, positive.
// Both a_k and eq_k are non-negative, so |fl(a_k - eq_k)| <= max(a_k, eq_k) <= p, and the product
// w (double, rounded to float; or the plain fp32 product) has |w| <= p * 2^-34 * (1 + 2^-23) + 2^-149
// < 2^-26 * 0.010413 p <= 2^-26 eq_k < ulp(eq_k) / 4.  A float whose neighbours are at least
// ulp/2 away (ulp/4 towards zero at a power of two) absorbs such a w: fl(w + eq_k) = eq_k.  QED
// The test is wave-uniform (a vote of all lanes); blocked cells (omega = 1: (a - eq) + eq, the product by 1 is
// exact) are handled lane by lane inside the shortcut; a wave with a free cell
// outside (1)-(3) -- zero density, negative or non-finite populations, |u| > 0.29 -- takes the
// general path, so results are bit-identical for every input.
// care = false: a lane whose result nobody reads (window-edge columns of the LDS kernel, which
// compute on stale ring contents) must not veto the wave.
__device__ __forceinline__ bool relax_is_identity(const float a[Q], float p, float c, bool blocked,
                                                  bool care = true) {
    unsigned sg = __float_as_uint(a[0]);
#pragma unroll
    for (int k = 1; k < Q; k++) sg |= __float_as_uint(a[k]);
    // a blocked cell does not veto its wave: its relaxation factor is 1 and the caller computes
    // (a - eq) + eq for it (exact in either product: v * 1 = v)
    // (bitwise, not short-circuit: see relax_is_identity2)
    const bool ok = !care | blocked | (((int)sg >= 0) & (p >= 7.888609052210118e-31f /* 2^-100 */) &
                    (p <= 1.2676506002282294e30f /* 2^100 */) & (c <= 0.125f));
    return wave_all(ok);
}
// `cells *= omega` (latticeboltzmann.py:112).  The reference's omega is a FLOAT64 array
// (np.where(blocked, np.array(1), np.array(OMEGA)), :98), so numpy multiplies the float32 cell by
// it in double and rounds once to float32.  When omega is exactly representable in float32 the
// double product of two 24-bit significands is exact (48 bits) and that single rounding is the
// fp32 multiply's own: the plain v_mul_f32 gives the same bits (blocked cells: omega = 1).
__device__ __forceinline__ float mul_omega(float v, bool blocked, const Omega& om) {
    if (om.f32_exact) return v * (blocked ? 1.0f : om.f);
    return (float)((double)v * (blocked ? 1.0 : om.d));
}

__host__ __device__ constexpr int slot_of(int k) { return 3 * (k % 3) + k / 3; }
__host__ __device__ constexpr int dy_of(int k) { return k % 3 - 1; }
__host__ __device__ constexpr int dx_of(int k) { return k / 3 - 1; }

// --- storage element <-> float ---------------------------------------------------------------
__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(__half* p, float v) { *p = __float2half_rn(v); }
__device__ __forceinline__ float ld_nt(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float ld_nt(const __half* p) {
    return __half2float(__ushort_as_half(__builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(p))));
}
// streaming (non-temporal) store: the new state is not re-read before the next launch
__device__ __forceinline__ void st_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_nt(__half* p, float v) {
    __builtin_nontemporal_store(__half_as_ushort(__float2half_rn(v)), reinterpret_cast<unsigned short*>(p));
}
// The stores of the new state: non-temporal AND agent scope (`nt sc1`).  With sc1 the L2 writes the
// line through to memory as it arrives instead of keeping it dirty until something evicts it, so the
// writes reach the DRAM in the order the sweep produces them.  Measured on the T = 1 kernel, same box,
// 16384 x 16384 (gpurun_out/r3_ab_stpolicy.log): plain 84.2 GLUPS, sc0 84.2, nt 88.1 (what
// __builtin_nontemporal_store gives), sc1 89.4, sc0 sc1 89.3, nt sc0 sc1 89.7, nt sc1 89.85 - the step
// from 0.793 to 0.809 of the 8 TB/s peak.  hipcc has no builtin that sets both bits (an agent-scope
// atomic store gives sc1 alone), hence the inline assembly; a store needs no s_waitcnt bookkeeping of
// the compiler's (nothing reads it back in the kernel), and stronger scope than needed is always safe.
// BYTE_OFF: the instruction's immediate offset (13 bits signed).
#ifndef LBM_ST_POLICY
#define LBM_ST_POLICY "nt sc1"
#endif
// ... to wave_base (wave-uniform: an SGPR pair) + lane_byte_off + BYTE_OFF
template <int BYTE_OFF>
__device__ __forceinline__ void st_stream(float* wave_base, unsigned lane_byte_off, float v) {
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 4096, "immediate offset of a global store");
    asm volatile("global_store_dword %0, %1, %2 offset:%3 " LBM_ST_POLICY
                 :: "v"(lane_byte_off), "v"(v), "s"(wave_base), "n"(BYTE_OFF) : "memory");
}
template <int BYTE_OFF>
__device__ __forceinline__ void st_stream(__half* wave_base, unsigned lane_byte_off, float v) {
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 4096, "immediate offset of a global store");
    const unsigned h = __half_as_ushort(__float2half_rn(v));
    asm volatile("global_store_short %0, %1, %2 offset:%3 " LBM_ST_POLICY
                 :: "v"(lane_byte_off), "v"(h), "s"(wave_base), "n"(BYTE_OFF) : "memory");
}
// ... to a per-lane address + BYTE_OFF
template <int BYTE_OFF>
__device__ __forceinline__ void st_stream(float* p, float v) {
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 4096, "immediate offset of a global store");
    asm volatile("global_store_dword %0, %1, off offset:%2 " LBM_ST_POLICY :: "v"(p), "v"(v), "n"(BYTE_OFF) : "memory");
}
template <int BYTE_OFF>
__device__ __forceinline__ void st_stream(__half* p, float v) {
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 4096, "immediate offset of a global store");
    const unsigned h = __half_as_ushort(__float2half_rn(v));
    asm volatile("global_store_short %0, %1, off offset:%2 " LBM_ST_POLICY :: "v"(p), "v"(h), "n"(BYTE_OFF) : "memory");
}

// --- pull-stream (latticeboltzmann.py:115-120 in pull form) ------------------------------------
// t[k] = g_k(y - dy, x - dx); where that source is outside the domain:
//   reference: surroundings[k] if (dx == +1 and x == 0) or (dy == +1 and y == 0), else the
//              cell's own g_k(y, x)  (the reference's empty `-1:0` slices);
//   periodic : wrapped source.
// Seams (YEDGE_GHOST) read the ghost rows.  `y` must be wave-uniform.
// NT: non-temporal loads (every stored value is read by exactly one cell per update: nothing to
// keep in the caches).
// Addressing: the three row bases are wave-uniform pointers (SGPR pairs) and a lane only carries
// 32-bit element offsets, so every load is `global_load_dword v, voffset, s[base] offset:slot`
// and a wave reaches its first load after a few dozen instructions.  All nine loads are issued
// before the first use (two loops), so one s_waitcnt covers them.
template <int XMODE, typename T, bool NT = false>
__device__ __forceinline__ void pull(const T* __restrict__ g, const Geom& geo, int y, int xb,
                                     int lane, float t[Q]) {
    const int M = geo.M;
    int x = xb * LANES + lane;
    x = x < M ? x : M - 1;  // padded lanes (x >= M) compute on a valid address; never stored
    int xm = x + 1;         // source column for dx = -1
    int xp = x - 1;         // source column for dx = +1
    bool ob_m = xm >= M, ob_p = xp < 0;
    if (XMODE == XMODE_PERIODIC) {
        xm = ob_m ? 0 : xm;
        xp = ob_p ? M - 1 : xp;
        ob_m = ob_p = false;
    }
    const unsigned off0 = (unsigned)((x >> 6) * CHUNK + (x & 63));
    const unsigned offm = ob_m ? off0 : (unsigned)((xm >> 6) * CHUNK + (xm & 63));
    const unsigned offp = ob_p ? off0 : (unsigned)((xp >> 6) * CHUNK + (xp & 63));

    int yu = y - 1, yd = y + 1;  // source rows for dy = +1 / dy = -1
    bool ob_u = false, ob_d = false;
    if (y == 0 && geo.top != YEDGE_GHOST) {
        if (geo.top == YEDGE_WRAP) yu = geo.N - 1; else { ob_u = true; yu = y; }
    }
    if (y == geo.N - 1 && geo.bot != YEDGE_GHOST) {
        if (geo.bot == YEDGE_WRAP) yd = 0; else { ob_d = true; yd = y; }
    }
    // wave-uniform row bases; a source row outside the domain is the own row (yu / yd = y above)
    const T* r0 = g + (long long)(y + geo.G) * geo.rowStride;
    const T* ru = g + (long long)(yu + geo.G) * geo.rowStride;
    const T* rd = g + (long long)(yd + geo.G) * geo.rowStride;
    // ... and then the own column too: the source is the cell's own value
    const unsigned offm_u = ob_u ? off0 : offm, offp_u = ob_u ? off0 : offp;
    const unsigned offm_d = ob_d ? off0 : offm, offp_d = ob_d ? off0 : offp;

    float v[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * LANES;
        const T* rb = dy == 1 ? ru : (dy == 0 ? r0 : rd);
        const unsigned off = dx == 0 ? off0
                           : dy == 1 ? (dx == 1 ? offp_u : offm_u)
                           : dy == -1 ? (dx == 1 ? offp_d : offm_d)
                                      : (dx == 1 ? offp : offm);
        const T* p = reinterpret_cast<const T*>(reinterpret_cast<const char*>(rb + sl) +
                                                (size_t)(off * (unsigned)sizeof(T)));
        v[k] = NT ? ld_nt(p) : ld(p);
    }
    // Reference edge rule at the last column: the source column M of dx = -1 does not exist and
    // the cell keeps its OWN value, own row included -- but the row bases above are wave-uniform.
    // Only the wave that holds column M-1 pays for it: two more loads from the own row (dy = 0
    // reads the own row anyway), chosen by the one lane concerned.
    float own_m[3] = {0.0f, 0.0f, 0.0f};
    if (XMODE != XMODE_PERIODIC && __any(ob_m)) {
#pragma unroll
        for (int k = 0; k < Q; k++) {
            if (dx_of(k) != -1 || dy_of(k) == 0) continue;
            own_m[dy_of(k) + 1] = ld(r0 + slot_of(k) * LANES + off0);
        }
    }
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int dy = dy_of(k), dx = dx_of(k);
        float w = v[k];
        if (XMODE != XMODE_PERIODIC && dx == -1 && dy != 0) w = ob_m ? own_m[dy + 1] : w;
        // hipcc would fold `inlet ? surr[k] : load` into ONE plain load from a selected (flat)
        // pointer, which drops the non-temporal hint and puts the loads behind branches; an
        // empty asm on the value keeps the load a load
        if (NT) asm("" : "+v"(w));
        const bool inlet = (dx == 1 && ob_p) || (dy == 1 && ob_u);
        t[k] = inlet ? geo.surr[k] : w;
    }
}

// --- reflect (latticeboltzmann.py:122-123): k <-> 8-k at blocked cells ------------------------
// Obstacles are sparse (< 1 % of the reference's scenes): a wave without a blocked lane skips
// the eight selects.
__device__ __forceinline__ void reflect(float t[Q], bool blocked) {
    if (wave_any(blocked)) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float a = t[k], b = t[8 - k];
            t[k] = blocked ? b : a;
            t[8 - k] = blocked ? a : b;
        }
    }
}

// --- moments (latticeboltzmann.py:180-183), numpy's evaluation order ---------------------------
// np.sum over 9 contiguous floats is pairwise; `cells @ e_f` accumulates k-sequentially and its
// products by 0 / +-1 are exact, so they are written as adds and subtracts.  u0 = row (y)
// component, u1 = column (x) component.
__device__ __forceinline__ float nan_to_num(float v) {
    v = (v != v) ? 0.0f : v;
    return (__builtin_fabsf(v) == __builtin_inff()) ? __builtin_copysignf(3.402823466e+38f, v) : v;
}
__device__ __forceinline__ void moments(const float a[Q], float& p, float& u0, float& u1) {
    p = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + a[8];
    float m0 = 0.0f - a[0];
    m0 = m0 + a[2]; m0 = m0 - a[3]; m0 = m0 + a[5]; m0 = m0 - a[6]; m0 = m0 + a[8];
    float m1 = 0.0f - a[0];
    m1 = m1 - a[1]; m1 = m1 - a[2]; m1 = m1 + a[6]; m1 = m1 + a[7]; m1 = m1 + a[8];
    u0 = m0 / p;
    u1 = m1 / p;
    // np.nan_to_num only acts on non-finite values (p == 0 or a blown-up cell): rare, so the
    // fix-up is a wave-uniform branch instead of ten selects per cell.
    const float inf = __builtin_inff();
    if (wave_any(!(__builtin_fabsf(u0) < inf) || !(__builtin_fabsf(u1) < inf))) {
        u0 = nan_to_num(u0);
        u1 = nan_to_num(u1);
    }
}

// --- equilibrium (latticeboltzmann.py:87-89) ---------------------------------------------------
// eq_k = (p * w_k) * (((1 + 3 eu) + 4.5 eu^2) - 1.5 |u|^2),  eu = u0*dy + u1*dx.
__device__ __forceinline__ float equilibrium(float u0, float u1, float p, float eq[Q]) {
    const float W_CORNER = 1.0f / 36, W_AXIS = 1.0f / 9, W_REST = 4.0f / 9;
    const float c = 1.5f * (u0 * u0 + u1 * u1);
    const float pc = p * W_CORNER, pa = p * W_AXIS, pr = p * W_REST;
    const float sp = u0 + u1, sm = u0 - u1;
    // k and 8-k have opposite eu: share eu^2.
    // k=0: eu = -(u0+u1), k=8: +(u0+u1)
    { float q = 4.5f * (sp * sp), h = 3.0f * sp;
      eq[8] = pc * (((1.0f + h) + q) - c); eq[0] = pc * (((1.0f - h) + q) - c); }
    // k=1: eu = -u1 (dy=0, dx=-1), k=7: +u1
    { float q = 4.5f * (u1 * u1), h = 3.0f * u1;
      eq[7] = pa * (((1.0f + h) + q) - c); eq[1] = pa * (((1.0f - h) + q) - c); }
    // k=2: eu = u0 - u1 (dy=+1, dx=-1), k=6: -(u0-u1)
    { float q = 4.5f * (sm * sm), h = 3.0f * sm;
      eq[2] = pc * (((1.0f + h) + q) - c); eq[6] = pc * (((1.0f - h) + q) - c); }
    // k=3: eu = -u0 (dy=-1, dx=0), k=5: +u0
    { float q = 4.5f * (u0 * u0), h = 3.0f * u0;
      eq[5] = pa * (((1.0f + h) + q) - c); eq[3] = pa * (((1.0f - h) + q) - c); }
    // k=4: eu = 0:  ((1 + 0) + 0) - c
    eq[4] = pr * (1.0f - c);
    return c;
}

// --- collide (latticeboltzmann.py:107-113 with omega of :98) -----------------------------------
__device__ __forceinline__ void collide(float a[Q], bool blocked, const Omega& om, bool care = true) {
    float p, u0, u1, eq[Q];
    moments(a, p, u0, u1);
    const float c = equilibrium(u0, u1, p, eq);
    if (om.tiny && relax_is_identity(a, p, c, blocked, care)) {  // wave-uniform
        if (wave_any(blocked)) {  // obstacles are sparse: most waves skip this
#pragma unroll
            for (int k = 0; k < Q; k++) {
                const float v = a[k] - eq[k];
                a[k] = blocked ? v + eq[k] : eq[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < Q; k++) a[k] = eq[k];
        }
        return;
    }
    if (om.f32_exact) {  // wave-uniform
        const float o = blocked ? 1.0f : om.f;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            float v = a[k] - eq[k];
            v = v * o;
            a[k] = v + eq[k];
        }
    } else {
        const double o = blocked ? 1.0 : om.d;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            float v = a[k] - eq[k];
            v = (float)((double)v * o);
            a[k] = v + eq[k];
        }
    }
}

// --- two cells per lane, packed fp32 ------------------------------------------------------------
// gfx950 issues v_pk_add_f32 / v_pk_mul_f32 (two fp32 results per lane) at nearly the rate of the
// scalar forms (tools/valu_rate.hip), so the VALU-bound temporal-blocking kernel gives every lane
// two cells and runs collide on float2 vectors.  Each component goes through exactly the same
// IEEE operations in the same order as the scalar functions above: bit-identical results.
typedef float f2 __attribute__((ext_vector_type(2)));

// bm0 / bm1: ballot(b0) / ballot(b1), taken by the caller right where b0 / b1 are computed (a
// ballot of a bool that was defined in another basic block costs a v_cndmask + v_cmp round trip)
__device__ __forceinline__ void reflect2(f2 t[Q], bool b0, bool b1, lanemask bm0, lanemask bm1) {
    if ((bm0 | bm1) != 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const f2 a = t[k], b = t[8 - k];
            t[k] = f2{b0 ? b.x : a.x, b1 ? b.y : a.y};
            t[8 - k] = f2{b0 ? a.x : b.x, b1 ? a.y : b.y};
        }
    }
}
__device__ __forceinline__ void reflect2(f2 t[Q], bool b0, bool b1) { reflect2(t, b0, b1, ballot(b0), ballot(b1)); }

__device__ __forceinline__ void moments2(const f2 a[Q], f2& p, f2& u0, f2& u1) {
    p = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + a[8];
    f2 m0 = 0.0f - a[0];
    m0 = m0 + a[2]; m0 = m0 - a[3]; m0 = m0 + a[5]; m0 = m0 - a[6]; m0 = m0 + a[8];
    f2 m1 = 0.0f - a[0];
    m1 = m1 - a[1]; m1 = m1 - a[2]; m1 = m1 + a[6]; m1 = m1 + a[7]; m1 = m1 + a[8];
    // Division.  `m / p` expands to v_div_scale x2, v_rcp, 7 FMAs, v_div_fmas, v_div_fixup per
    // quotient, all scalar: 44 instructions for the four quotients of a lane.  The scale / fixup
    // parts only act on extreme operands, so the common case runs the same refinement unscaled,
    // on packed FMAs, with ONE reciprocal per cell (14 instructions).  It is bit-identical to the
    // IEEE quotient whenever v_div_scale would not have scaled: p, m0, m1 have frexp exponents in
    // [-80, 40] (0 for a zero operand) and |u| < 2^40 -- i.e. denominator and reciprocal normal,
    // quotient normal, numerator not tiny, exponent difference < 96.  Any lane outside that range
    // (or non-finite) sends its wave down the exact path, which also applies np.nan_to_num.
    // (A -0 numerator gives +0 instead of -0 on the fast path; no later operation can see it.)
    const f2 r0 = f2{__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
    const f2 one = f2{1.0f, 1.0f};
    f2 e = __builtin_elementwise_fma(-p, r0, one);
    const f2 r = __builtin_elementwise_fma(e, r0, r0);
    f2 q = m0 * r;
    e = __builtin_elementwise_fma(-p, q, m0);
    q = __builtin_elementwise_fma(e, r, q);
    e = __builtin_elementwise_fma(-p, q, m0);
    u0 = __builtin_elementwise_fma(e, r, q);
    q = m1 * r;
    e = __builtin_elementwise_fma(-p, q, m1);
    q = __builtin_elementwise_fma(e, r, q);
    e = __builtin_elementwise_fma(-p, q, m1);
    u1 = __builtin_elementwise_fma(e, r, q);
    const int ex0 = __builtin_amdgcn_frexp_expf(p.x), ex1 = __builtin_amdgcn_frexp_expf(m0.x),
              ex2 = __builtin_amdgcn_frexp_expf(m1.x), ey0 = __builtin_amdgcn_frexp_expf(p.y),
              ey1 = __builtin_amdgcn_frexp_expf(m0.y), ey2 = __builtin_amdgcn_frexp_expf(m1.y);
    const int emin = min(min(min(ex0, ex1), ex2), min(min(ey0, ey1), ey2));
    const int emax = max(max(max(ex0, ex1), ex2), max(max(ey0, ey1), ey2));
    const float big = 1099511627776.0f;  // 2^40
    const bool odd = emin < -80 || emax > 40 ||
                     !(__builtin_fabsf(u0.x) < big) || !(__builtin_fabsf(u0.y) < big) ||
                     !(__builtin_fabsf(u1.x) < big) || !(__builtin_fabsf(u1.y) < big);
    if (wave_any(odd)) {
        u0 = m0 / p;  // IEEE divisions
        u1 = m1 / p;
        u0 = f2{nan_to_num(u0.x), nan_to_num(u0.y)};
        u1 = f2{nan_to_num(u1.x), nan_to_num(u1.y)};
    }
}

__device__ __forceinline__ f2 equilibrium2(f2 u0, f2 u1, f2 p, f2 eq[Q]) {
    const float W_CORNER = 1.0f / 36, W_AXIS = 1.0f / 9, W_REST = 4.0f / 9;
    const f2 c = 1.5f * (u0 * u0 + u1 * u1);
    const f2 pc = p * W_CORNER, pa = p * W_AXIS, pr = p * W_REST;
    const f2 sp = u0 + u1, sm = u0 - u1;
    { const f2 q = 4.5f * (sp * sp), h = 3.0f * sp;
      eq[8] = pc * (((1.0f + h) + q) - c); eq[0] = pc * (((1.0f - h) + q) - c); }
    { const f2 q = 4.5f * (u1 * u1), h = 3.0f * u1;
      eq[7] = pa * (((1.0f + h) + q) - c); eq[1] = pa * (((1.0f - h) + q) - c); }
    { const f2 q = 4.5f * (sm * sm), h = 3.0f * sm;
      eq[2] = pc * (((1.0f + h) + q) - c); eq[6] = pc * (((1.0f - h) + q) - c); }
    { const f2 q = 4.5f * (u0 * u0), h = 3.0f * u0;
      eq[5] = pa * (((1.0f + h) + q) - c); eq[3] = pa * (((1.0f - h) + q) - c); }
    eq[4] = pr * (1.0f - c);
    return c;
}

// relax_is_identity() for the two cells of a lane.  bm = ballot(blocked), cm = ballot(care), per cell.
// Combined on the lane masks (one SGPR pair per compare, scalar and / or): no lane-divergent
// branch (short-circuit evaluation would make hipcc structurize the whole collide: flow blocks,
// register copies on the common path) and no per-lane integer for the vote.
__device__ __forceinline__ bool relax_is_identity2(const f2 a[Q], f2 p, f2 c, lanemask bm0, lanemask bm1,
                                                   lanemask cm0, lanemask cm1) {
    unsigned sx = __float_as_uint(a[0].x), sy = __float_as_uint(a[0].y);
#pragma unroll
    for (int k = 1; k < Q; k++) { sx |= __float_as_uint(a[k].x); sy |= __float_as_uint(a[k].y); }
    const float lo = 7.888609052210118e-31f, hi = 1.2676506002282294e30f;
    const lanemask good0 = ballot((int)sx >= 0) & ballot(p.x >= lo) & ballot(p.x <= hi) & ballot(c.x <= 0.125f);
    const lanemask good1 = ballot((int)sy >= 0) & ballot(p.y >= lo) & ballot(p.y <= hi) & ballot(c.y <= 0.125f);
    const lanemask ok0 = ~cm0 | bm0 | good0;
    const lanemask ok1 = ~cm1 | bm1 | good1;
    return (~(ok0 & ok1) & ballot(true)) == 0;
}

// Moments for the tiny-omega common path: the same pairwise / sequential sums as moments2(), and
// the two quotients by the refinement v_div_scale / v_div_fmas run when they scale the NUMERATOR by
// 2^64 -- unconditionally, with no test of the operands' exponents.  Valid (bit-identical to the
// IEEE quotient wherever a bit of it can matter) under the conditions ordinary2() checks afterwards:
// every population non-negative and 2^-60 <= p <= 2^39.  Then |m| <= 5 max a_k <= 5p < 2^42, the
// scaled numerator n = m 2^64 is exact and either zero or in [2^-85, 2^106], r = 1/p (one
// Newton step on v_rcp_f32) is normal, q0 = n r >= 2^-125 is normal, and every residual
// fma(-p, q, n) is a multiple of 2^(E_n - 47) >= 2^-132: no underflow anywhere, so q2 = RN(n / p)
// as in the hardware's own scaled expansion.  u = q2 2^-64 is exact unless |u| < 2^-126; there a
// second rounding can move the last denormal bit, which nothing downstream can see: with
// |u0| < 2^-126 the sums u0 +- u1 equal the IEEE ones unless |u1| < 2^-124 too, and then all of
// 3 e.u, 4.5 (e.u)^2 and c vanish against the 1 they are added to (|e.u| < 2^-123).
__device__ __forceinline__ void moments2_scaled(const f2 a[Q], f2& p, f2& u0, f2& u1) {
    p = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + a[8];
    f2 m0 = 0.0f - a[0];
    m0 = m0 + a[2]; m0 = m0 - a[3]; m0 = m0 + a[5]; m0 = m0 - a[6]; m0 = m0 + a[8];
    f2 m1 = 0.0f - a[0];
    m1 = m1 - a[1]; m1 = m1 - a[2]; m1 = m1 + a[6]; m1 = m1 + a[7]; m1 = m1 + a[8];
    const f2 r0 = f2{__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
    const f2 one = f2{1.0f, 1.0f};
    const float up = 18446744073709551616.0f /* 2^64 */, down = 5.421010862427522e-20f /* 2^-64 */;
    f2 e = __builtin_elementwise_fma(-p, r0, one);
    const f2 r = __builtin_elementwise_fma(e, r0, r0);
    f2 n = m0 * up;
    f2 q = n * r;
    e = __builtin_elementwise_fma(-p, q, n);
    q = __builtin_elementwise_fma(e, r, q);
    e = __builtin_elementwise_fma(-p, q, n);
    u0 = __builtin_elementwise_fma(e, r, q) * down;
    n = m1 * up;
    q = n * r;
    e = __builtin_elementwise_fma(-p, q, n);
    q = __builtin_elementwise_fma(e, r, q);
    e = __builtin_elementwise_fma(-p, q, n);
    u1 = __builtin_elementwise_fma(e, r, q) * down;
}
// The conditions of moments2_scaled() and of relax_is_identity() in one vote over the lanes that
// matter (cm = ballot(care)): sign bits clear, 2^-60 <= p <= 2^39, c <= 1/8.
__device__ __forceinline__ bool ordinary2(const f2 a[Q], f2 p, f2 c, lanemask cm0, lanemask cm1) {
    unsigned sx = __float_as_uint(a[0].x), sy = __float_as_uint(a[0].y);
#pragma unroll
    for (int k = 1; k < Q; k++) { sx |= __float_as_uint(a[k].x); sy |= __float_as_uint(a[k].y); }
    const float lo = 8.673617379884035e-19f /* 2^-60 */, hi = 549755813888.0f /* 2^39 */;
    const lanemask good0 = ballot((int)sx >= 0) & ballot(p.x >= lo) & ballot(p.x <= hi) & ballot(c.x <= 0.125f);
    const lanemask good1 = ballot((int)sy >= 0) & ballot(p.y >= lo) & ballot(p.y <= hi) & ballot(c.y <= 0.125f);
    return (~((~cm0 | good0) & (~cm1 | good1)) & ballot(true)) == 0;
}

// The result is handed to `sink(const f2 out[Q])` instead of being returned: the common case (tiny
// omega, ordinary cells, no obstacle in the wave) passes eq[] on as it is, every other case
// corrects eq[] inside ONE rare, wave-uniform branch and calls the sink from there.  Returned
// through a[] the two paths would join in a phi, and hipcc places that phi's nine register-pair
// copies on the common path (18 of ~150 vector instructions per row of the LDS kernel).
template <class Sink>
__device__ __forceinline__ void collide2_to(const f2 a[Q], bool b0, bool b1, lanemask bm0, lanemask bm1,
                                            const Omega& om, lanemask cm0, lanemask cm1, Sink sink) {
    const bool obstacle = (bm0 | bm1) != 0;
    // The common case first and on its own: tiny omega, ordinary cells (blocked ones included, so
    // that their moments are exact too).  Free cells: eq is the result.  Blocked cells (rare: a
    // wave-uniform branch) relax with omega = 1, written as fma(a - eq, o, eq) with o = 1 (blocked)
    // or 0 (free): the product is exact, so the fma rounds once like the reference's separate
    // multiply and add, and o = 0 returns eq (a - eq is finite for an ordinary cell).
    if (om.tiny) {
        f2 p, u0, u1, eq[Q];
        moments2_scaled(a, p, u0, u1);
        const f2 c = equilibrium2(u0, u1, p, eq);
        if (__builtin_expect(ordinary2(a, p, c, cm0, cm1), 1)) {
            if (obstacle) {
                const f2 o = f2{b0 ? 1.0f : 0.0f, b1 ? 1.0f : 0.0f};
#pragma unroll
                for (int k = 0; k < Q; k++) eq[k] = __builtin_elementwise_fma(a[k] - eq[k], o, eq[k]);
                asm volatile("; collide2_to: common path, obstacle in the wave");
                sink(eq);
                asm volatile("; collide2_to: common path, obstacle in the wave end");
                return;
            }
            asm volatile("; collide2_to: common path");
            sink(eq);
            asm volatile("; collide2_to: common path end");
            return;
        }
    }
    // Everything else, for every input
    f2 p, u0, u1, eq[Q];
    moments2(a, p, u0, u1);
    const f2 c = equilibrium2(u0, u1, p, eq);
    // both votes evaluated unconditionally and joined bitwise: one wave-uniform branch
    const bool fast = om.tiny & relax_is_identity2(a, p, c, bm0, bm1, cm0, cm1);
    if (__builtin_expect(!fast | obstacle, 0)) {
        if (fast) {
            // only blocked cells relax, with omega = 1.  fma(a - eq, o, eq) with o = 1 (blocked) or
            // 0 (free): the product by 0 or 1 is exact, so the fma rounds exactly once, like the
            // reference's separate multiply and add, and o = 0 returns eq (a - eq is finite here)
            const f2 o = f2{b0 ? 1.0f : 0.0f, b1 ? 1.0f : 0.0f};
#pragma unroll
            for (int k = 0; k < Q; k++) eq[k] = __builtin_elementwise_fma(a[k] - eq[k], o, eq[k]);
        } else if (om.f32_exact) {  // wave-uniform
            const f2 o = f2{b0 ? 1.0f : om.f, b1 ? 1.0f : om.f};
#pragma unroll
            for (int k = 0; k < Q; k++) {
                f2 v = a[k] - eq[k];
                v = v * o;
                eq[k] = v + eq[k];
            }
        } else {  // the reference's float64 omega: double product, one rounding (no packed f64 exists)
            const double o0 = b0 ? 1.0 : om.d, o1 = b1 ? 1.0 : om.d;
#pragma unroll
            for (int k = 0; k < Q; k++) {
                f2 v = a[k] - eq[k];
                v = f2{(float)((double)v.x * o0), (float)((double)v.y * o1)};
                eq[k] = v + eq[k];
            }
        }
        asm volatile("; collide2_to: general path");  // (the markers keep the sink copies from being merged again)
        sink(eq);
        asm volatile("; collide2_to: general path end");
    } else {
        asm volatile("; collide2_to: tiny omega, extreme exponents");
        sink(eq);
        asm volatile("; collide2_to: tiny omega, extreme exponents end");
    }
}
__device__ __forceinline__ void collide2(f2 a[Q], bool b0, bool b1, const Omega& om,
                                         bool care0 = true, bool care1 = true) {
    collide2_to(a, b0, b1, ballot(b0), ballot(b1), om, ballot(care0), ballot(care1), [&](const f2* out) {
#pragma unroll
        for (int k = 0; k < Q; k++) a[k] = out[k];
    });
}

// --- one cell per lane: the same common path for the T = 1 kernel ---------------------------------
// moments2_scaled() / ordinary2() / collide2_to() for a single cell per lane (see there for the
// conditions and the proof).
__device__ __forceinline__ void moments_scaled(const float a[Q], float& p, float& u0, float& u1) {
    p = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + a[8];
    float m0 = 0.0f - a[0];
    m0 = m0 + a[2]; m0 = m0 - a[3]; m0 = m0 + a[5]; m0 = m0 - a[6]; m0 = m0 + a[8];
    float m1 = 0.0f - a[0];
    m1 = m1 - a[1]; m1 = m1 - a[2]; m1 = m1 + a[6]; m1 = m1 + a[7]; m1 = m1 + a[8];
    const float r0 = __builtin_amdgcn_rcpf(p);
    const float up = 18446744073709551616.0f /* 2^64 */, down = 5.421010862427522e-20f /* 2^-64 */;
    float e = __builtin_fmaf(-p, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    float n = m0 * up;
    float q = n * r;
    e = __builtin_fmaf(-p, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-p, q, n);
    u0 = __builtin_fmaf(e, r, q) * down;
    n = m1 * up;
    q = n * r;
    e = __builtin_fmaf(-p, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-p, q, n);
    u1 = __builtin_fmaf(e, r, q) * down;
}
__device__ __forceinline__ bool ordinary(const float a[Q], float p, float c) {
    unsigned sg = __float_as_uint(a[0]);
#pragma unroll
    for (int k = 1; k < Q; k++) sg |= __float_as_uint(a[k]);
    const float lo = 8.673617379884035e-19f /* 2^-60 */, hi = 549755813888.0f /* 2^39 */;
    const lanemask good = ballot((int)sg >= 0) & ballot(p >= lo) & ballot(p <= hi) & ballot(c <= 0.125f);
    return (~good & ballot(true)) == 0;
}
// collide of one cell per lane (a[] already reflected), result handed to sink(const float out[Q]).
// bm = ballot(blocked), taken where `blocked` is computed.
template <class Sink>
__device__ __forceinline__ void collide_to(const float a[Q], bool blocked, lanemask bm, const Omega& om, Sink sink) {
    if (om.tiny) {
        float p, u0, u1, eq[Q];
        moments_scaled(a, p, u0, u1);
        const float c = equilibrium(u0, u1, p, eq);
        if (__builtin_expect(ordinary(a, p, c), 1)) {
            if (bm != 0) {  // blocked cells relax with omega = 1: one fma, see collide2_to
                const float o = blocked ? 1.0f : 0.0f;
#pragma unroll
                for (int k = 0; k < Q; k++) eq[k] = __builtin_fmaf(a[k] - eq[k], o, eq[k]);
                asm volatile("; collide_to: common path, obstacle in the wave");
                sink(eq);
                asm volatile("; collide_to: common path, obstacle in the wave end");
                return;
            }
            asm volatile("; collide_to: common path");
            sink(eq);
            asm volatile("; collide_to: common path end");
            return;
        }
    }
    float b[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) b[k] = a[k];
    collide(b, blocked, om);
    asm volatile("; collide_to: general path");
    sink(b);
    asm volatile("; collide_to: general path end");
}

// --- kernel-semantics compatibility mode (XMODE_CUDA) -----------------------------
// What the reference's CUDA kernel computes where it differs from its numpy path, restated in IEEE
// fp32 in the kernel's operation order (oracle: oracle/lbm_numpy.py cuda_kernel_step; PARITY-
// UNPINNED: the reference holds no fixture for it and is built with --use_fast_math):
// density = sum + 0.0001f, its own velocity sums and equilibrium form (lb_cuda_kernel.cu:293-300,
// 323-331); a cell of row r is collided iff r >= 1 and blocked[r-1][x] is clear (:322 tests row y
// for the cell of row y+1; :302-306 skips row 0); all four sides of the domain supply `surr`,
// the side below the last row `surr` passed through the (gated) collide (:215,232-233,286-287).
// a[] is in the reference's k order, k = 3*i + j = next.d[i][j].
__device__ __forceinline__ void collide_cuda(float a[Q], float omega_f) {
    const float s1 = (a[0] + a[1]) + a[2], s2 = (a[3] + a[4]) + a[5], s3 = (a[6] + a[7]) + a[8];
    const float d = ((s1 + s2) + s3) + 0.0001f;
    const float uy = (s3 - s1) / d;
    const float ux = (((((a[2] + a[5]) + a[8]) - a[0]) - a[3]) - a[6]) / d;
    const float c = 1.0f - 1.5f * (ux * ux + uy * uy);
    const float W[3] = {1.0f / 36, 1.0f / 9, 4.0f / 9};
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float eu = (float)(i - 1) * uy + (float)(j - 1) * ux;
            const float wk = W[(i == 1) + (j == 1)];
            const float eq = (d * wk) * (c + (3.0f * eu) * (1.0f + 1.5f * eu));
            a[3 * i + j] = (a[3 * i + j] - eq) * omega_f + eq;
        }
    }
}

// Pull for XMODE_CUDA: a source outside the domain is surr[k]; below the last row it is surr
// passed through the collide unless the cell above it (row N-1 of the source column) is blocked.
template <typename T>
__device__ __forceinline__ void pull_cuda(const T* __restrict__ g, const uint64_t* __restrict__ mask,
                                          const Geom& geo, int y, int xb, int lane, float t[Q]) {
    const int N = geo.N, M = geo.M;
    int x = xb * LANES + lane;
    x = x < M ? x : M - 1;
    float csurr[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) csurr[k] = geo.surr[k];
    if (y == N - 1) collide_cuda(csurr, geo.omega_f);  // wave-uniform, last row only
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int dy = dy_of(k), dx = dx_of(k);
        const int ys = y - dy, xs = x - dx;
        const bool xin = xs >= 0 && xs < M, yin = ys >= 0 && ys < N;
        const int yc = yin ? ys : y, xc = xin ? xs : x;  // a valid address in any case
        const float v = ld(g + (long long)(yc + geo.G) * geo.rowStride + (xc >> 6) * CHUNK + (xc & 63) +
                           slot_of(k) * LANES);
        float out = geo.surr[k];
        if (dy == -1) {  // the only direction that can look below the last row
            const bool above_blocked = (mask[(long long)(N - 1 + geo.G) * geo.Mb + (xc >> 6)] >> (xc & 63)) & 1ull;
            out = (ys == N && xin && !above_blocked) ? csurr[k] : out;
        }
        t[k] = (xin && yin) ? v : out;
    }
}
// Is the cell (y, x) collided in XMODE_CUDA?
__device__ __forceinline__ bool cuda_gate(const uint64_t* __restrict__ mask, const Geom& geo, int y,
                                          int xb, int lane) {
    if (y < 1) return false;
    return !((mask[(long long)(y - 1 + geo.G) * geo.Mb + xb] >> lane) & 1ull);
}

// Exact division of a tile index n < 2^31 by a run-time divisor d (blocks per row) without the
// ~300-instruction 64-bit software division a wave would otherwise run before it can issue its
// first load: q = (n * m) >> s with m = ceil(2^s / d), s = 31 + ceil(log2 d) (Granlund &
// Montgomery); m < 2^33, so the product fits 64 bits.  All operands are wave-uniform (SALU).
struct FastDiv {
    unsigned long long m;
    int s;
    int d;
};
__host__ inline FastDiv make_fastdiv(int d) {
    int l = 0;
    while ((1ll << l) < d) l++;
    const int s = 31 + l;
    const unsigned long long m = (unsigned long long)((((unsigned __int128)1 << s) + (unsigned)d - 1) / (unsigned)d);
    return FastDiv{m, s, d};
}
__device__ __forceinline__ void fastdivmod(unsigned n, const FastDiv& f, int& q, int& r) {
    q = (int)(((unsigned long long)n * f.m) >> f.s);
    r = (int)n - q * f.d;
}

// Bijective XCD-aware remap of a linear workgroup id: workgroups b, b+8, b+16, ... share an XCD
// (round-robin dispatch), so give each XCD one contiguous band of the tile space and let
// consecutive workgroups of that XCD walk it in order (neighbouring tiles then share an L2).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u, idx = b >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace lbm
