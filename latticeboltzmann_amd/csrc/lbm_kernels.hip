// gfx950 kernels of the D2Q9 BGK path and their launchers.  Layout and numerics: lbm_device.h.
#include <algorithm>
#include <cstdlib>

#include "lbm_internal.h"

#ifndef LBM_T1_NT_LOADS
#define LBM_T1_NT_LOADS true
#endif

namespace lbm {

// ---------------------------------------------------------------------------------------------
// The hot kernel, one update (INNER_TIMESTEPS = 1):  g' = collide(reflect(pull(g))).
// Replaces fcs<> (lb_cuda_kernel.cu:209-370).  One wavefront per (row, xb) tile of 64 cells:
// 9 wave-wide loads (3 x 768 contiguous bytes from rows y-1, y, y+1; the dx = +-1 ones shifted by
// one lane, crossing into the neighbouring chunk for one lane), ~110 VALU, 9 aligned 256-byte
// write-through stores (`nt sc1`, st_stream) that fill one contiguous 2304-byte chunk.  No LDS, no barrier: nothing is reused.
// Algorithmic HBM bytes per cell-update: 36 read + 36 written (latticeboltzmann.py:176).
// ---------------------------------------------------------------------------------------------
// a lane's nine populations into its chunk (wave-uniform address), streaming stores (st_stream, lbm_device.h)
template <int K = 0, typename T>
__device__ __forceinline__ void store_chunk(T* chunk, int lane, const float* v) {
    if constexpr (K < Q) {
        st_stream<slot_of(K) * LANES * (int)sizeof(T)>(chunk, (unsigned)lane * (unsigned)sizeof(T), v[K]);
        store_chunk<K + 1>(chunk, lane, v);
    }
}

template <int XMODE, typename T>
__global__ __launch_bounds__(64) void step_kernel(const T* __restrict__ gin, T* __restrict__ gout,
                                                   const uint64_t* __restrict__ mask, Geom geo,
                                                   int row0, int nrows, FastDiv by_mb) {
    const unsigned nwg = gridDim.x;
    const unsigned wg = xcd_remap(blockIdx.x, nwg);
    const int lane = threadIdx.x;
    const unsigned tile = wg;  // one wave per workgroup, one tile per wave (launch_step_t); nrows * Mb < 2^31
    if (tile >= (unsigned)nrows * (unsigned)geo.Mb) return;
    int yr, xb;
    fastdivmod(tile, by_mb, yr, xb);
    const int y = row0 + yr;

    float a[Q];
    const uint64_t mrow = mask[(long long)(y + geo.G) * geo.Mb + xb];
    const bool blocked = (mrow >> lane) & 1ull;
    if (XMODE == XMODE_CUDA) {  // kernel-semantics compatibility mode (never the default)
        pull_cuda(gin, mask, geo, y, xb, lane, a);
        reflect(a, blocked);
        float b[Q];
#pragma unroll
        for (int k = 0; k < Q; k++) b[k] = a[k];
        collide_cuda(b, geo.omega_f);
        const bool gate = cuda_gate(mask, geo, y, xb, lane);
#pragma unroll
        for (int k = 0; k < Q; k++) a[k] = gate ? b[k] : a[k];
    } else {
        pull<XMODE, T, LBM_T1_NT_LOADS>(gin, geo, y, xb, lane, a);
        const lanemask bm = ballot(blocked);
        reflect(a, blocked);
        collide_to(a, blocked, bm, omega_of(geo), [&](const float* out) {
            if (xb * LANES + lane < geo.M) store_chunk(gout + (long long)(y + geo.G) * geo.rowStride + (long long)xb * CHUNK, lane, out);
        });
        return;
    }

    if (xb * LANES + lane < geo.M) store_chunk(gout + (long long)(y + geo.G) * geo.rowStride + (long long)xb * CHUNK, lane, a);
}

// f -> g = collide(f) in place over rows [row0, row0+nrows): the first half of the first update
// after an upload (latticeboltzmann.py:180-183,107-113).
template <typename T, bool CUDA_SEMANTICS = false>
__global__ __launch_bounds__(256) void collide_kernel(T* __restrict__ f,
                                                      const uint64_t* __restrict__ mask, Geom geo,
                                                      int row0, int nrows) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)nrows * geo.Mb) return;
    const int y = row0 + (int)(tile / geo.Mb);
    const int xb = (int)(tile % geo.Mb);
    if (xb * LANES + lane >= geo.M) return;
    T* c = f + (long long)(y + geo.G) * geo.rowStride + (long long)xb * CHUNK + lane;
    float a[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) a[k] = ld(c + slot_of(k) * LANES);
    const bool blocked = (mask[(long long)(y + geo.G) * geo.Mb + xb] >> lane) & 1ull;
    if (CUDA_SEMANTICS) {
        if (!cuda_gate(mask, geo, y, xb, lane)) return;
        collide_cuda(a, geo.omega_f);
    } else {
        collide(a, blocked, omega_of(geo));
    }
#pragma unroll
    for (int k = 0; k < Q; k++) st(c + slot_of(k) * LANES, a[k]);
}

// HSV -> RGB as lb_cuda_kernel.cu:27-67 (h in radians [0, 2 pi], round-to-nearest * 255).
__device__ __forceinline__ void hsv_to_rgb(float h, float s, float v, uint8_t* rgb) {
    const float PI = 3.141592653589f;
    float f = h / (PI / 3);
    const float hi = floorf(f);
    f = f - hi;
    const float p = v * (1 - s), q = v * (1 - s * f), t = v * (1 - s * (1 - f));
    float r, g, b;
    if (hi == 0.0f || hi == 6.0f) { r = v; g = t; b = p; }
    else if (hi == 1.0f) { r = q; g = v; b = p; }
    else if (hi == 2.0f) { r = p; g = v; b = t; }
    else if (hi == 3.0f) { r = p; g = q; b = v; }
    else if (hi == 4.0f) { r = t; g = p; b = v; }
    else { r = v; g = p; b = q; }
    rgb[0] = (uint8_t)__float2uint_rn(255.0f * r);
    rgb[1] = (uint8_t)__float2uint_rn(255.0f * g);
    rgb[2] = (uint8_t)__float2uint_rn(255.0f * b);
}

// Observation of the numpy-visible state f over a window: f = reflect(pull(g)) when the buffer
// holds g (PULLED), or the buffer itself right after an upload.  Writes, where non-null, dense
// host-layout windows: cells (nrows, ncols, 9) reference k order, rho (nrows, ncols),
// u (nrows, ncols, 2) [row, col], rgb (nrows, ncols, 3).
template <int XMODE, bool PULLED, typename T>
__global__ __launch_bounds__(256) void observe_kernel(const T* __restrict__ g,
                                                      const uint64_t* __restrict__ mask, Geom geo,
                                                      int row0, int nrows, int col0, int ncols,
                                                      float* __restrict__ cells,
                                                      float* __restrict__ rho, float* __restrict__ u,
                                                      uint8_t* __restrict__ rgb) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int xb0 = col0 >> 6;
    const int nxb = ((col0 + ncols + 63) >> 6) - xb0;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)nrows * nxb) return;
    const int yr = (int)(tile / nxb);
    const int y = row0 + yr;
    const int xb = xb0 + (int)(tile % nxb);
    const int x = xb * LANES + lane;

    float a[Q];
    const bool blocked = (mask[(long long)(y + geo.G) * geo.Mb + xb] >> lane) & 1ull;
    if (PULLED) {
        if (XMODE == XMODE_CUDA) pull_cuda(g, mask, geo, y, xb, lane, a);
        else pull<XMODE>(g, geo, y, xb, lane, a);
        reflect(a, blocked);
    } else {
        const int xc = x < geo.M ? x : geo.M - 1;
        const T* c = g + (long long)(y + geo.G) * geo.rowStride + (long long)(xc >> 6) * CHUNK + (xc & 63);
#pragma unroll
        for (int k = 0; k < Q; k++) a[k] = ld(c + slot_of(k) * LANES);
    }
    if (x < col0 || x >= col0 + ncols) return;
    const long long o = (long long)yr * ncols + (x - col0);
    if (cells) {
#pragma unroll
        for (int k = 0; k < Q; k++) cells[o * 9 + k] = a[k];
    }
    if (rho || u || rgb) {
        float p, u0, u1;
        moments(a, p, u0, u1);
        if (rho) rho[o] = p;
        if (u) { u[o * 2] = u0; u[o * 2 + 1] = u1; }
        if (rgb) {
            // lb_cuda_kernel.cu:308-319: its `uy` is the column (x) component, `ux` the row one.
            const float PI = 3.141592653589f;
            float h = atan2f(u1, u0) + PI;
            float s = __saturatef(1000 * sqrtf(u0 * u0 + u1 * u1));
            float v = __saturatef(p);
            if (blocked) h = s = v = 0;
            hsv_to_rgb(h, s, v, rgb + o * 3);
        }
    }
}

// Host-layout cells (nrows, M, 9), reference k order -> device layout, rows [row0, row0+nrows).
template <typename T>
__global__ __launch_bounds__(256) void import_cells_kernel(T* __restrict__ f, Geom geo,
                                                           const float* __restrict__ cells, int row0,
                                                           int nrows) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)nrows * geo.Mb) return;
    const int yr = (int)(tile / geo.Mb);
    const int xb = (int)(tile % geo.Mb);
    const int x = xb * LANES + lane;
    T* c = f + (long long)(row0 + yr + geo.G) * geo.rowStride + (long long)xb * CHUNK + lane;
    const bool in = x < geo.M;
    const float* src = cells + ((long long)yr * geo.M + (in ? x : 0)) * 9;
#pragma unroll
    for (int k = 0; k < Q; k++) st(c + slot_of(k) * LANES, in ? src[k] : 0.0f);
}

// uint8 (nrows, M) -> bit-packed mask words, one lane-ballot per (row, xb).
__global__ __launch_bounds__(256) void import_mask_kernel(uint64_t* __restrict__ mask, Geom geo,
                                                          const uint8_t* __restrict__ blocked,
                                                          int row0, int nrows) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)nrows * geo.Mb) return;
    const int yr = (int)(tile / geo.Mb);
    const int xb = (int)(tile % geo.Mb);
    const int x = xb * LANES + lane;
    const bool b = x < geo.M && blocked[(long long)yr * geo.M + x] != 0;
    const uint64_t word = __ballot(b);
    if (lane == 0) mask[(long long)(row0 + yr + geo.G) * geo.Mb + xb] = word;
}

__global__ __launch_bounds__(256) void export_mask_kernel(const uint64_t* __restrict__ mask, Geom geo,
                                                          uint8_t* __restrict__ blocked, int row0,
                                                          int nrows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nrows * geo.M) return;
    const int yr = (int)(i / geo.M), x = (int)(i % geo.M);
    blocked[i] = (mask[(long long)(row0 + yr + geo.G) * geo.Mb + (x >> 6)] >> (x & 63)) & 1ull;
}

// splitmix64-style hash -> float in [-1, 1)
__device__ __forceinline__ float hash_pm1(uint64_t seed, uint64_t a, uint64_t b) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a * 0x100000001B3ull + b + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 8388608.0f) - 1.0f;  // 24 bits -> [0, 2) - 1
}

// f = equilibrium(rho * (1 + amp * hash), u) on every allocated row (ghost rows included).
template <typename T>
__global__ __launch_bounds__(256) void init_equilibrium_kernel(T* __restrict__ f, Geom geo, float rho,
                                                               float u0, float u1, float amp,
                                                               uint64_t seed, long long row_offset,
                                                               long long wrap_rows) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const int rows = geo.N + 2 * geo.G;
    if (tile >= (long long)rows * geo.Mb) return;
    const int r = (int)(tile / geo.Mb);
    const int xb = (int)(tile % geo.Mb);
    const int x = xb * LANES + lane;
    long long gy = row_offset + (r - geo.G);
    if (wrap_rows > 0) gy = ((gy % wrap_rows) + wrap_rows) % wrap_rows;
    float p = rho;
    if (amp != 0.0f) p = rho * (1.0f + amp * hash_pm1(seed, (uint64_t)gy, (uint64_t)x));
    float eq[Q];
    equilibrium(u0, u1, p, eq);
    T* c = f + (long long)r * geo.rowStride + (long long)xb * CHUNK + lane;
    const bool in = x < geo.M;
#pragma unroll
    for (int k = 0; k < Q; k++) st(c + slot_of(k) * LANES, in ? eq[k] : 0.0f);
}

// Seeded random discs: disc d has centre (hash, hash) in the global grid; blocked if inside any.
__global__ __launch_bounds__(256) void init_discs_kernel(uint64_t* __restrict__ mask, Geom geo,
                                                         int ndiscs, int radius, uint64_t seed,
                                                         long long row_offset, long long global_rows) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const int rows = geo.N + 2 * geo.G;
    if (tile >= (long long)rows * geo.Mb) return;
    const int r = (int)(tile / geo.Mb);
    const int xb = (int)(tile % geo.Mb);
    const int x = xb * LANES + lane;
    long long gy = row_offset + (r - geo.G);
    gy = ((gy % global_rows) + global_rows) % global_rows;
    bool b = false;
    for (int d = 0; d < ndiscs; d++) {
        const float cy = (hash_pm1(seed, 2 * d, 77) * 0.5f + 0.5f) * (float)global_rows;
        const float cx = (hash_pm1(seed, 2 * d + 1, 77) * 0.5f + 0.5f) * (float)geo.M;
        const float ddy = (float)gy - cy, ddx = (float)x - cx;
        b = b || (ddy * ddy + ddx * ddx <= (float)radius * (float)radius);
    }
    b = b && x < geo.M;
    const uint64_t word = __ballot(b);
    if (lane == 0) mask[(long long)r * geo.Mb + xb] = word;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline unsigned tiles_to_wgs(long long tiles) { return (unsigned)((tiles + 3) / 4); }

// Resident waves per CU of step_kernel.  The kernel needs few VGPRs and no LDS, so a CU would
// take 32 waves (74 KB of loads in flight); the memory system delivers most with about half of
// that in flight (tools/kbench.hip, 16384^2 fp32, four-wave workgroups: 8 WG/CU 5.98 TB/s, 6: 6.06,
// 5: 6.18, 4: 6.19, 3: 4.83; a pure copy in the kernel's own access shape 6.03 -> 6.30, which is what
// a float4 streaming copy reaches).  The cap is imposed the portable way: an unused dynamic-LDS
// request per workgroup.  Workgroups are ONE wave: a four-wave workgroup holds its LDS, i.e. its
// place under the cap, until its last wave has finished, so fewer than sixteen waves are at work on
// average; with single-wave workgroups the cap is exact (same-box: 87.1 -> 87.8 GLUPS).
// LDS is handed out in 1280-byte granules, so the steps are 16 waves per CU (up to 10240 bytes each),
// 14 (up to 11520), 12 (up to 12800).  Re-swept in round 3 with the `nt sc1` stores
// (gpurun_out/r3_cap_sweep*.log, GLUPS): uncapped 83.7, 16 waves 89.8-89.9, 14 waves 90.7-90.8, 12 waves
// 87.2 with the tiny-omega common path; with the general path (float64 omega product: OMEGA = 0.05,
// obstacles) 16 waves 90.3, 14 waves 88.9 - more arithmetic per byte wants more waves.
// fp16 storage has half the bytes per wave in flight and twice the arithmetic per byte: it wants
// all 32 waves (uncapped 177.6 GLUPS, 16 waves 118).
static unsigned step_lds_cap(bool half, bool tiny_omega) {
    static const int env = [] {
        const char* e = getenv("LBM_T1_LDS_CAP");  // development override, bytes per workgroup
        return e ? atoi(e) : -1;
    }();
    if (env >= 0) return (unsigned)env;
    if (half) return 0u;
    return tiny_omega ? 11u << 10 : 10u << 10;
}

template <typename T>
static hipError_t launch_step_t(const T* gin, T* gout, const uint64_t* mask, const Geom& geo,
                                int xmode, int row0, int nrows, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    // HIP refuses a launch of 2^32 threads or more, i.e. 2^26 single-wave workgroups (4.3e9 cells: an
    // fp16 grid of that size fits the 288 GB): such a range goes in several launches, split by rows
    // (rows of one update are independent).
    const long long max_tiles = (1ll << 26) - 1;
    if (geo.Mb > max_tiles) return hipErrorInvalidValue;  // one row alone: M > 4.3e9
    if ((long long)nrows * geo.Mb > max_tiles) {
        const int rows_per = (int)(max_tiles / geo.Mb);
        for (int r = 0; r < nrows; r += rows_per) {
            const hipError_t e = launch_step_t(gin, gout, mask, geo, xmode, row0 + r, std::min(rows_per, nrows - r), st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    const unsigned nwg = (unsigned)((long long)nrows * geo.Mb);  // one single-wave workgroup per tile
    const unsigned lds = step_lds_cap(sizeof(T) == 2, geo.omega_tiny != 0);
    const FastDiv by_mb = make_fastdiv(geo.Mb);
    if (xmode == XMODE_PERIODIC)
        hipLaunchKernelGGL((step_kernel<XMODE_PERIODIC, T>), dim3(nwg), dim3(64), lds, st, gin, gout,
                           mask, geo, row0, nrows, by_mb);
    else if (xmode == XMODE_CUDA)
        hipLaunchKernelGGL((step_kernel<XMODE_CUDA, T>), dim3(nwg), dim3(64), 0, st, gin, gout,
                           mask, geo, row0, nrows, by_mb);
    else
        hipLaunchKernelGGL((step_kernel<XMODE_REFERENCE, T>), dim3(nwg), dim3(64), lds, st, gin, gout,
                           mask, geo, row0, nrows, by_mb);
    return hipGetLastError();
}

hipError_t launch_step(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                       bool half, int row0, int nrows, hipStream_t st) {
    return half ? launch_step_t((const __half*)gin, (__half*)gout, mask, geo, xmode, row0, nrows, st)
                : launch_step_t((const float*)gin, (float*)gout, mask, geo, xmode, row0, nrows, st);
}

hipError_t launch_collide(void* f, const uint64_t* mask, const Geom& geo, int xmode, bool half,
                          int row0, int nrows, hipStream_t st) {
    const unsigned nwg = tiles_to_wgs((long long)nrows * geo.Mb);
    if (nwg == 0) return hipSuccess;
    if (xmode == XMODE_CUDA) {
        if (half)
            hipLaunchKernelGGL((collide_kernel<__half, true>), dim3(nwg), dim3(256), 0, st, (__half*)f, mask,
                               geo, row0, nrows);
        else
            hipLaunchKernelGGL((collide_kernel<float, true>), dim3(nwg), dim3(256), 0, st, (float*)f, mask,
                               geo, row0, nrows);
        return hipGetLastError();
    }
    if (half)
        hipLaunchKernelGGL(collide_kernel<__half>, dim3(nwg), dim3(256), 0, st, (__half*)f, mask, geo,
                           row0, nrows);
    else
        hipLaunchKernelGGL(collide_kernel<float>, dim3(nwg), dim3(256), 0, st, (float*)f, mask, geo,
                           row0, nrows);
    return hipGetLastError();
}

template <int XMODE, bool PULLED, typename T>
static void launch_observe_t(const T* g, const uint64_t* mask, const Geom& geo, int row0, int nrows,
                             int col0, int ncols, float* cells, float* rho, float* u, uint8_t* rgb,
                             hipStream_t st) {
    const int nxb = ((col0 + ncols + 63) >> 6) - (col0 >> 6);
    const unsigned nwg = tiles_to_wgs((long long)nrows * nxb);
    hipLaunchKernelGGL((observe_kernel<XMODE, PULLED, T>), dim3(nwg), dim3(256), 0, st, g, mask, geo,
                       row0, nrows, col0, ncols, cells, rho, u, rgb);
}

hipError_t launch_observe(const void* g, const uint64_t* mask, const Geom& geo, int xmode, bool half,
                          bool pulled, int row0, int nrows, int col0, int ncols, float* cells,
                          float* rho, float* u, uint8_t* rgb, hipStream_t st) {
    if (nrows <= 0 || ncols <= 0) return hipSuccess;
#define LBM_OBS(XM, PU)                                                                          \
    do {                                                                                         \
        if (half) launch_observe_t<XM, PU>((const __half*)g, mask, geo, row0, nrows, col0, ncols, \
                                           cells, rho, u, rgb, st);                              \
        else launch_observe_t<XM, PU>((const float*)g, mask, geo, row0, nrows, col0, ncols, cells, \
                                      rho, u, rgb, st);                                          \
    } while (0)
    if (xmode == XMODE_PERIODIC) { if (pulled) LBM_OBS(XMODE_PERIODIC, true); else LBM_OBS(XMODE_PERIODIC, false); }
    else if (xmode == XMODE_CUDA) { if (pulled) LBM_OBS(XMODE_CUDA, true); else LBM_OBS(XMODE_CUDA, false); }
    else { if (pulled) LBM_OBS(XMODE_REFERENCE, true); else LBM_OBS(XMODE_REFERENCE, false); }
#undef LBM_OBS
    return hipGetLastError();
}

hipError_t launch_import_cells(void* f, const Geom& geo, bool half, const float* cells, int row0,
                               int nrows, hipStream_t st) {
    const unsigned nwg = tiles_to_wgs((long long)nrows * geo.Mb);
    if (nwg == 0) return hipSuccess;
    if (half)
        hipLaunchKernelGGL(import_cells_kernel<__half>, dim3(nwg), dim3(256), 0, st, (__half*)f, geo,
                           cells, row0, nrows);
    else
        hipLaunchKernelGGL(import_cells_kernel<float>, dim3(nwg), dim3(256), 0, st, (float*)f, geo,
                           cells, row0, nrows);
    return hipGetLastError();
}

hipError_t launch_import_mask(uint64_t* mask, const Geom& geo, const uint8_t* blocked, int row0,
                              int nrows, hipStream_t st) {
    const unsigned nwg = tiles_to_wgs((long long)nrows * geo.Mb);
    if (nwg == 0) return hipSuccess;
    hipLaunchKernelGGL(import_mask_kernel, dim3(nwg), dim3(256), 0, st, mask, geo, blocked, row0, nrows);
    return hipGetLastError();
}

hipError_t launch_export_mask(const uint64_t* mask, const Geom& geo, uint8_t* blocked, int row0,
                              int nrows, hipStream_t st) {
    const long long n = (long long)nrows * geo.M;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(export_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mask,
                       geo, blocked, row0, nrows);
    return hipGetLastError();
}

hipError_t launch_init_equilibrium(void* f, const Geom& geo, bool half, float rho, float u0, float u1,
                                   float amp, uint64_t seed, long long row_offset, long long wrap_rows,
                                   hipStream_t st) {
    const unsigned nwg = tiles_to_wgs((long long)(geo.N + 2 * geo.G) * geo.Mb);
    if (half)
        hipLaunchKernelGGL(init_equilibrium_kernel<__half>, dim3(nwg), dim3(256), 0, st, (__half*)f,
                           geo, rho, u0, u1, amp, seed, row_offset, wrap_rows);
    else
        hipLaunchKernelGGL(init_equilibrium_kernel<float>, dim3(nwg), dim3(256), 0, st, (float*)f, geo,
                           rho, u0, u1, amp, seed, row_offset, wrap_rows);
    return hipGetLastError();
}

hipError_t launch_init_discs(uint64_t* mask, const Geom& geo, int ndiscs, int radius, uint64_t seed,
                             long long row_offset, long long global_rows, hipStream_t st) {
    const unsigned nwg = tiles_to_wgs((long long)(geo.N + 2 * geo.G) * geo.Mb);
    hipLaunchKernelGGL(init_discs_kernel, dim3(nwg), dim3(256), 0, st, mask, geo, ndiscs, radius, seed,
                       row_offset, global_rows);
    return hipGetLastError();
}

}  // namespace lbm
