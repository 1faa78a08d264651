// D2Q21 (scoped): the lattice of the reference's draft latticeboltzmann.jl:3-22 as a
// second compile-time lattice for the ONE-update kernel.  Periodic or reference edges, fp32 storage, single
// domain, no LDS temporal blocking, no rendering.  PARITY UNPINNED by the reference (the draft is not
// a runnable program; only its weight and velocity tables and their four asserts are pins); the
// semantics are those of oracle/lbm_d2q21.py, which the kernels reproduce bit for bit.
//
// Layout: the D2Q9 one with 21 slots,  field[row][xb][slot][lane],  a (row, xb) chunk = 21 * 64
// floats = 5376 bytes.  Slots are ordered by dy, so the populations a destination row pulls from
// one source row are adjacent: 1 / 3 / 3 / 7 / 3 / 3 / 1 slots from rows y+3 .. y-3.
// One update is g' = collide(reflect(pull(g))) as for D2Q9 (collide-once):
// one wave per 64-cell tile, 21 non-temporal loads (lane-shifted by up to 3 columns, from 7 rows),
// ~230 VALU, 21 non-temporal 256-byte stores.  Algorithmic HBM bytes: 2 * 21 * 4 = 168 per update.
#include <algorithm>
#include <cstdlib>

#include "lbm_internal.h"

namespace lbm {
namespace q21 {

constexpr int Q = 21;
constexpr int CH = Q * LANES;  // elements per chunk

// latticeboltzmann.jl:14-20, e[k] = (dy, dx)
__host__ __device__ constexpr int dy_of(int k) {
    constexpr int t[Q] = {0, -2, 0, 2, -1, 0, 1, -3, -2, -1, 0, 1, 2, 3, -1, 0, 1, -2, 0, 2, 0};
    return t[k];
}
__host__ __device__ constexpr int dx_of(int k) {
    constexpr int t[Q] = {-3, -2, -2, -2, -1, -1, -1, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 3};
    return t[k];
}
// slots ordered by dy (then k): dy=-3: k7 | -2: k1 k8 k17 | -1: k4 k9 k14 | 0: k0 k2 k5 k10 k15 k18 k20
// | 1: k6 k11 k16 | 2: k3 k12 k19 | 3: k13
__host__ __device__ constexpr int slot_of(int k) {
    constexpr int t[Q] = {7, 1, 8, 17, 4, 9, 14, 0, 2, 5, 10, 15, 18, 20, 6, 11, 16, 3, 12, 19, 13};
    return t[k];
}
// latticeboltzmann.jl:5-11 as float32 (the double quotient rounded once, like np.array(..., float32))
__device__ __forceinline__ float w_of(int k) {
    constexpr float a = (float)(1.0 / 1620), b = (float)(1.0 / 432), c = (float)(7.0 / 360),
                    d = (float)(2.0 / 27), f = (float)(1.0 / 12), g = (float)(91.0 / 324);
    constexpr float t[Q] = {a, b, c, b, d, f, d, a, c, f, g, f, c, a, d, f, d, b, c, b, a};
    return t[k];
}

struct Surr { float v[Q]; };  // surroundings, reference k order (lbm_set_surroundings)

// --- pull ----------------------------------------------------------------------------------------
// t[k] = g_k(y - dy, x - dx).  Periodic: wrapped.  Reference (latticeboltzmann.py:115-120 with this
// lattice's velocities, which reach up to three cells): where that source is outside the domain,
// surroundings[k] if (dx > 0 and x < dx) or (dy > 0 and y < dy), else the cell's OWN g_k(y, x) (the
// reference's empty negative slices: the last |dx| columns / |dy| rows keep their value).  Only
// waves within three cells of an edge pay for the rule (EDGE, wave-uniform).
template <int XMODE, bool EDGE>
__device__ __forceinline__ void pull_impl(const float* __restrict__ g, const Geom& geo, const Surr& su, int y,
                                          int xb, int lane, float t[Q]) {
    const int N = geo.N, M = geo.M;
    int x = xb * LANES + lane;
    x = x < M ? x : M - 1;
    unsigned off[7];  // element offset of source column x - dx inside a row, dx = -3 .. 3 (index dx + 3)
    bool cob[7];      // that column does not exist (reference edges)
#pragma unroll
    for (int d = -3; d <= 3; d++) {
        int xs = x - d;
        cob[d + 3] = false;
        if (XMODE == XMODE_PERIODIC) {
            xs = xs < 0 ? xs + M : xs;  // M >= 4 (checked by the API)
            xs = xs >= M ? xs - M : xs;
        } else if (EDGE) {
            cob[d + 3] = xs < 0 || xs >= M;
            xs = cob[d + 3] ? x : xs;
        }
        off[d + 3] = (unsigned)((xs >> 6) * CH + (xs & 63));
    }
    const float* rb[7];  // wave-uniform base of source row y - dy, dy = -3 .. 3
    bool rob[7];
#pragma unroll
    for (int d = -3; d <= 3; d++) {
        int ys = y - d;
        rob[d + 3] = false;
        if (XMODE == XMODE_PERIODIC) {
            ys = ys < 0 ? ys + N : ys;
            ys = ys >= N ? ys - N : ys;
        } else if (EDGE) {
            rob[d + 3] = ys < 0 || ys >= N;
            ys = rob[d + 3] ? y : ys;
        }
        rb[d + 3] = g + (long long)ys * geo.rowStride;
    }
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int dy = dy_of(k), dx = dx_of(k);
        const float* p = reinterpret_cast<const float*>(
            reinterpret_cast<const char*>(rb[dy + 3] + slot_of(k) * LANES) + (size_t)(off[dx + 3] * 4u));
        if (XMODE != XMODE_PERIODIC && EDGE) {
            // own value = own row AND own column, whichever of the two was missing
            const float* own = reinterpret_cast<const float*>(
                reinterpret_cast<const char*>(rb[3] + slot_of(k) * LANES) + (size_t)(off[3] * 4u));
            p = (rob[dy + 3] || cob[dx + 3]) ? own : p;
            const float v = *p;
            const bool inlet = (dx > 0 && x < dx) || (dy > 0 && y < dy);
            t[k] = inlet ? su.v[k] : v;
        } else {
            t[k] = __builtin_nontemporal_load(p);
        }
    }
}
template <int XMODE>
__device__ __forceinline__ void pull(const float* __restrict__ g, const Geom& geo, const Surr& su, int y, int xb,
                                     int lane, float t[Q]) {
    if (XMODE != XMODE_PERIODIC) {
        const int x = xb * LANES + lane;
        const bool edge = y < 3 || y >= geo.N - 3 || wave_any(x < 3 || x >= geo.M - 3);
        if (edge) {
            pull_impl<XMODE, true>(g, geo, su, y, xb, lane, t);
            return;
        }
    }
    pull_impl<XMODE, false>(g, geo, su, y, xb, lane, t);
}

__device__ __forceinline__ void reflect(float t[Q], bool blocked) {
    if (wave_any(blocked)) {
#pragma unroll
        for (int k = 0; k < Q / 2; k++) {
            const float a = t[k], b = t[Q - 1 - k];
            t[k] = blocked ? b : a;
            t[Q - 1 - k] = blocked ? a : b;
        }
    }
}

// oracle/lbm_d2q21.py moments(): numpy's pairwise sum of 21 terms, momentum sums in k order with
// rounded products, IEEE division, nan_to_num
__device__ __forceinline__ void moments(const float a[Q], float& p, float& u0, float& u1) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = a[j] + a[8 + j];
    p = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int i = 16; i < Q; i++) p = p + a[i];
    float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
    for (int k = 0; k < Q; k++) {
        if (dy_of(k) != 0) m0 = m0 + a[k] * (float)dy_of(k);
        if (dx_of(k) != 0) m1 = m1 + a[k] * (float)dx_of(k);
    }
    u0 = m0 / p;
    u1 = m1 / p;
    const float inf = __builtin_inff();
    if (wave_any(!(__builtin_fabsf(u0) < inf) || !(__builtin_fabsf(u1) < inf))) {
        u0 = nan_to_num(u0);
        u1 = nan_to_num(u1);
    }
}

// oracle/lbm_d2q21.py getEquilibrium + collide: eq_k = (p w_k) (((1 + 1.5 eu) + 1.125 eu^2) - 0.75 |u|^2),
// eu = u0 dy + u1 dx (products rounded, a zero component skipped: it adds +-0);
// a <- fl(fl(fl(a - eq) * omega) + eq) with the product in double (omega is float64, blocked: 1)
__device__ __forceinline__ void collide(float a[Q], bool blocked, double omega) {
    float p, u0, u1;
    moments(a, p, u0, u1);
    const float c = 0.75f * (u0 * u0 + u1 * u1);
    const double o = blocked ? 1.0 : omega;
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int dy = dy_of(k), dx = dx_of(k);
        float eu;
        if (dy != 0 && dx != 0) eu = u0 * (float)dy + u1 * (float)dx;
        else if (dy != 0) eu = u0 * (float)dy;
        else if (dx != 0) eu = u1 * (float)dx;
        else eu = 0.0f;
        const float eq = (p * w_of(k)) * (((1.0f + 1.5f * eu) + 1.125f * (eu * eu)) - c);
        float v = a[k] - eq;
        v = (float)((double)v * o);
        a[k] = v + eq;
    }
}

// --- kernels -------------------------------------------------------------------------------------
// a lane's 21 populations into its chunk (wave-uniform address): streaming stores as in the D2Q9 kernel
// (st_stream, lbm_device.h); the chunk is 5376 bytes, beyond a store's immediate offset, hence two bases
template <int K = 0>
__device__ __forceinline__ void store_chunk(float* chunk, int lane, const float* v) {
    if constexpr (K < Q) {
        constexpr int off = slot_of(K) * LANES * (int)sizeof(float), upper = 16 * LANES * (int)sizeof(float);
        if constexpr (off < upper) st_stream<off>(chunk, (unsigned)lane * 4u, v[K]);
        else st_stream<off - upper>(chunk + 16 * LANES, (unsigned)lane * 4u, v[K]);
        store_chunk<K + 1>(chunk, lane, v);
    }
}

template <int XMODE>
__global__ __launch_bounds__(64) void step_kernel(const float* __restrict__ gin, float* __restrict__ gout,
                                                  const uint64_t* __restrict__ mask, Geom geo, FastDiv by_mb,
                                                  Surr su) {
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x;
    int y, xb;
    fastdivmod(tile, by_mb, y, xb);
    float a[Q];
    const bool blocked = (mask[(long long)y * geo.Mb + xb] >> lane) & 1ull;
    pull<XMODE>(gin, geo, su, y, xb, lane, a);
    reflect(a, blocked);
    collide(a, blocked, geo.omega);
    if (xb * LANES + lane < geo.M) store_chunk(gout + (long long)y * geo.rowStride + (long long)xb * CH, lane, a);
}

__global__ __launch_bounds__(256) void collide_kernel(float* __restrict__ f, const uint64_t* __restrict__ mask,
                                                      Geom geo) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)geo.N * geo.Mb) return;
    const int y = (int)(tile / geo.Mb), xb = (int)(tile % geo.Mb);
    if (xb * LANES + lane >= geo.M) return;
    float* c = f + (long long)y * geo.rowStride + (long long)xb * CH + lane;
    float a[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) a[k] = c[slot_of(k) * LANES];
    const bool blocked = (mask[(long long)y * geo.Mb + xb] >> lane) & 1ull;
    collide(a, blocked, geo.omega);
#pragma unroll
    for (int k = 0; k < Q; k++) c[slot_of(k) * LANES] = a[k];
}

// numpy-visible state f over a window (see observe_kernel in lbm_kernels.hip)
template <int XMODE, bool PULLED>
__global__ __launch_bounds__(256) void observe_kernel(const float* __restrict__ g, const uint64_t* __restrict__ mask,
                                                      Geom geo, int row0, int nrows, int col0, int ncols,
                                                      float* __restrict__ cells, float* __restrict__ rho,
                                                      float* __restrict__ u, Surr su) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int xb0 = col0 >> 6;
    const int nxb = ((col0 + ncols + 63) >> 6) - xb0;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)nrows * nxb) return;
    const int yr = (int)(tile / nxb), y = row0 + yr, xb = xb0 + (int)(tile % nxb);
    const int x = xb * LANES + lane;
    float a[Q];
    if (PULLED) {
        const bool blocked = (mask[(long long)y * geo.Mb + xb] >> lane) & 1ull;
        pull<XMODE>(g, geo, su, y, xb, lane, a);
        reflect(a, blocked);
    } else {
        const int xc = x < geo.M ? x : geo.M - 1;
        const float* c = g + (long long)y * geo.rowStride + (long long)(xc >> 6) * CH + (xc & 63);
#pragma unroll
        for (int k = 0; k < Q; k++) a[k] = c[slot_of(k) * LANES];
    }
    if (x < col0 || x >= col0 + ncols) return;
    const long long o = (long long)yr * ncols + (x - col0);
    if (cells) {
#pragma unroll
        for (int k = 0; k < Q; k++) cells[o * Q + k] = a[k];
    }
    if (rho || u) {
        float p, u0, u1;
        moments(a, p, u0, u1);
        if (rho) rho[o] = p;
        if (u) { u[o * 2] = u0; u[o * 2 + 1] = u1; }
    }
}

__global__ __launch_bounds__(256) void import_cells_kernel(float* __restrict__ f, Geom geo,
                                                           const float* __restrict__ cells, int row0, int nrows) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)nrows * geo.Mb) return;
    const int yr = (int)(tile / geo.Mb), xb = (int)(tile % geo.Mb);
    const int x = xb * LANES + lane;
    float* c = f + (long long)(row0 + yr) * geo.rowStride + (long long)xb * CH + lane;
    const bool in = x < geo.M;
    const float* src = cells + ((long long)yr * geo.M + (in ? x : 0)) * Q;
#pragma unroll
    for (int k = 0; k < Q; k++) c[slot_of(k) * LANES] = in ? src[k] : 0.0f;
}

// f = equilibrium(rho (1 + amp hash), u), the same hash as the D2Q9 initialiser
__device__ __forceinline__ float hash_pm1(uint64_t seed, uint64_t a, uint64_t b) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a * 0x100000001B3ull + b + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 8388608.0f) - 1.0f;
}
__global__ __launch_bounds__(256) void init_equilibrium_kernel(float* __restrict__ f, Geom geo, float rho, float u0,
                                                               float u1, float amp, uint64_t seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)geo.N * geo.Mb) return;
    const int y = (int)(tile / geo.Mb), xb = (int)(tile % geo.Mb);
    const int x = xb * LANES + lane;
    float p = rho;
    if (amp != 0.0f) p = rho * (1.0f + amp * hash_pm1(seed, (uint64_t)y, (uint64_t)x));
    const float c = 0.75f * (u0 * u0 + u1 * u1);
    float* o = f + (long long)y * geo.rowStride + (long long)xb * CH + lane;
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const float eu = u0 * (float)dy_of(k) + u1 * (float)dx_of(k);
        const float eq = (p * w_of(k)) * (((1.0f + 1.5f * eu) + 1.125f * (eu * eu)) - c);
        o[slot_of(k) * LANES] = x < geo.M ? eq : 0.0f;
    }
}

}  // namespace q21

// --- launchers -----------------------------------------------------------------------------------
static inline unsigned q21_wgs(long long tiles) { return (unsigned)((tiles + 3) / 4); }

static q21::Surr surr_of(const float* surr21) {
    q21::Surr su;
    for (int k = 0; k < q21::Q; k++) su.v[k] = surr21[k];
    return su;
}

hipError_t launch_q21_step(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                           const float* surr21, hipStream_t st) {
    const long long tiles = (long long)geo.N * geo.Mb;
    if (tiles <= 0) return hipSuccess;
    if (tiles >= (1ll << 26)) return hipErrorInvalidValue;  // 2^32 threads per launch (4.3e9 cells = 360 GB: cannot exist)
    // twice the bytes per wave of the D2Q9 kernel: the same residency cap in bytes in flight is 8 waves
    // per CU; 18 KB each (tools: 16 / 12 / 8 per CU measured at bring-up)
    static const unsigned lds = [] {
        const char* e = getenv("LBM_Q21_LDS_CAP");
        return e ? (unsigned)atoi(e) : 18u << 10;
    }();
    if (xmode == XMODE_PERIODIC)
        hipLaunchKernelGGL(q21::step_kernel<XMODE_PERIODIC>, dim3((unsigned)tiles), dim3(64), lds, st, (const float*)gin,
                           (float*)gout, mask, geo, make_fastdiv(geo.Mb), surr_of(surr21));
    else
        hipLaunchKernelGGL(q21::step_kernel<XMODE_REFERENCE>, dim3((unsigned)tiles), dim3(64), lds, st, (const float*)gin,
                           (float*)gout, mask, geo, make_fastdiv(geo.Mb), surr_of(surr21));
    return hipGetLastError();
}

hipError_t launch_q21_collide(void* f, const uint64_t* mask, const Geom& geo, hipStream_t st) {
    hipLaunchKernelGGL(q21::collide_kernel, dim3(q21_wgs((long long)geo.N * geo.Mb)), dim3(256), 0, st, (float*)f,
                       mask, geo);
    return hipGetLastError();
}

hipError_t launch_q21_observe(const void* g, const uint64_t* mask, const Geom& geo, int xmode, const float* surr21,
                              bool pulled, int row0, int nrows, int col0, int ncols, float* cells, float* rho,
                              float* u, hipStream_t st) {
    if (nrows <= 0 || ncols <= 0) return hipSuccess;
    const int nxb = ((col0 + ncols + 63) >> 6) - (col0 >> 6);
    const unsigned nwg = q21_wgs((long long)nrows * nxb);
    const q21::Surr su = surr_of(surr21);
#define Q21_OBS(XM, PU)                                                                                            \
    hipLaunchKernelGGL((q21::observe_kernel<XM, PU>), dim3(nwg), dim3(256), 0, st, (const float*)g, mask, geo, row0, \
                       nrows, col0, ncols, cells, rho, u, su)
    if (xmode == XMODE_PERIODIC) { if (pulled) Q21_OBS(XMODE_PERIODIC, true); else Q21_OBS(XMODE_PERIODIC, false); }
    else { if (pulled) Q21_OBS(XMODE_REFERENCE, true); else Q21_OBS(XMODE_REFERENCE, false); }
#undef Q21_OBS
    return hipGetLastError();
}

hipError_t launch_q21_import_cells(void* f, const Geom& geo, const float* cells, int row0, int nrows, hipStream_t st) {
    const unsigned nwg = q21_wgs((long long)nrows * geo.Mb);
    if (nwg == 0) return hipSuccess;
    hipLaunchKernelGGL(q21::import_cells_kernel, dim3(nwg), dim3(256), 0, st, (float*)f, geo, cells, row0, nrows);
    return hipGetLastError();
}

hipError_t launch_q21_init_equilibrium(void* f, const Geom& geo, float rho, float u0, float u1, float amp,
                                       uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(q21::init_equilibrium_kernel, dim3(q21_wgs((long long)geo.N * geo.Mb)), dim3(256), 0, st,
                       (float*)f, geo, rho, u0, u1, amp, seed);
    return hipGetLastError();
}

}  // namespace lbm
