// Development micro-benchmark: variants of the T=1 fused kernel timed interleaved in ONE process
// (cdna_hip_programming.md 5.4 rule 24).  Not part of the product; build: make -C tools kbench.
// Scene: periodic, obstacle-free mask (all zero), perturbed equilibrium.  Every variant computes
// the same update; a checksum of a few cells is printed so that a broken variant shows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../latticeboltzmann_amd/csrc/lbm_device.h"

using namespace lbm;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { F_COPY9 = 8192, F_COPY9_ROWS = 16384, F_NOCOMPUTE = 32768, F_DPP = 2048, F_DPP_VEDGE = 4096, F_NT_ST = 1, F_NT_LD = 2, F_NO_REMAP = 4, F_ROWS4 = 8, F_PERSIST = 16, F_TWO = 32, F_PK = 64, F_PKROW = 128, F_ROWXCD = 256, F_ROWXCD2 = 512, F_GROUPED = 1024 };

template <int FLAGS>
__device__ __forceinline__ float ldx(const float* p) {
    if (FLAGS & F_NT_LD) return __builtin_nontemporal_load(p);
    return *p;
}
template <int FLAGS>
__device__ __forceinline__ void stx(float* p, float v) {
    if (FLAGS & F_NT_ST) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// periodic pull, fp32
template <int FLAGS>
__device__ __forceinline__ void pullp(const float* __restrict__ g, const Geom& geo, int y, int xb, int lane, float t[Q]) {
    const int M = geo.M;
    int x = xb * LANES + lane;
    int xm = x + 1; xm = xm >= M ? 0 : xm;
    int xp = x - 1; xp = xp < 0 ? M - 1 : xp;
    // F_GROUPED: row = [dy group (3)][xb][3 slots][64 lanes] instead of [xb][9 slots][64 lanes]: the
    // three populations a destination row pulls from one source row are contiguous across blocks
    const int BS = (FLAGS & F_GROUPED) ? 3 * LANES : CHUNK;
    const int off0 = xb * BS + lane;
    const int offm = (xm >> 6) * BS + (xm & 63);
    const int offp = (xp >> 6) * BS + (xp & 63);
    int yu = y == 0 ? geo.N - 1 : y - 1;
    int yd = y == geo.N - 1 ? 0 : y + 1;
    const float* r0 = g + (long long)y * geo.rowStride;
    const float* ru = g + (long long)yu * geo.rowStride;
    const float* rd = g + (long long)yd * geo.rowStride;
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int dy = dy_of(k), dx = dx_of(k);
        const int sl = (FLAGS & F_GROUPED) ? (slot_of(k) / 3) * (geo.Mb * 3 * LANES) + (slot_of(k) % 3) * LANES
                                           : slot_of(k) * LANES;
        const float* rb = dy == 1 ? ru : (dy == 0 ? r0 : rd);
        const int off = dx == 1 ? offp : (dx == 0 ? off0 : offm);
        t[k] = ldx<FLAGS>(rb + off + sl);
    }
}

// Aligned variant (SURVEY section 7 / VERDICT r1 item 3): all nine loads are the aligned 256-byte
// slot planes of the own block; dx = +-1 come from a DPP wave shift by one lane, and the one lane
// that falls off the wave takes ONE dword of the neighbouring block (wave-uniform address: a
// scalar load; F_DPP_VEDGE: a single-lane vector load instead).
template <int FLAGS>
__device__ __forceinline__ void pullp_dpp(const float* __restrict__ g, const Geom& geo, int y, int xb, int lane, float t[Q]) {
    const int xbm = xb + 1 == geo.Mb ? 0 : xb + 1;   // block of x+1 for lane 63
    const int xbp = xb == 0 ? geo.Mb - 1 : xb - 1;   // block of x-1 for lane 0
    const int yu = y == 0 ? geo.N - 1 : y - 1;
    const int yd = y == geo.N - 1 ? 0 : y + 1;
    const float* r0 = g + (long long)y * geo.rowStride;
    const float* ru = g + (long long)yu * geo.rowStride;
    const float* rd = g + (long long)yd * geo.rowStride;
    float edge[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int dy = dy_of(k), dx = dx_of(k), sl = slot_of(k) * LANES;
        const float* rb = dy == 1 ? ru : (dy == 0 ? r0 : rd);
        if (dx != 0) {
            const float* pe = rb + (dx == 1 ? xbp * CHUNK + 63 : xbm * CHUNK) + sl;
            if (FLAGS & F_DPP_VEDGE) {
                edge[k] = 0.0f;
                if (lane == (dx == 1 ? 0 : 63)) edge[k] = ldx<FLAGS>(pe);
            } else {
                edge[k] = *pe;  // uniform address
            }
        }
        t[k] = ldx<FLAGS>(rb + xb * CHUNK + lane + sl);
    }
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int dx = dx_of(k);
        if (dx == 1)   // source lane-1: wave_shr:1, lane 0 keeps `old` = edge
            t[k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge[k]), __float_as_int(t[k]), 0x138, 0xf, 0xf, false));
        else if (dx == -1)  // source lane+1: wave_shl:1, lane 63 keeps edge
            t[k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge[k]), __float_as_int(t[k]), 0x130, 0xf, 0xf, false));
    }
}

template <int FLAGS>
__device__ __forceinline__ void do_tile(const float* __restrict__ gin, float* __restrict__ gout,
                                        const uint64_t* __restrict__ mask, const Geom& geo, int y, int xb, int lane) {
    float a[Q];
    if (FLAGS & (F_COPY9 | F_COPY9_ROWS)) {
        // yardsticks in the kernel's own shape: nine aligned dword loads of the own block (from the
        // own row, or from rows y-1 / y / y+1 as the pull does), no arithmetic, nine nt stores
        const int yu = y == 0 ? geo.N - 1 : y - 1, yd = y == geo.N - 1 ? 0 : y + 1;
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dy = dy_of(k);
            const int ys = (FLAGS & F_COPY9_ROWS) ? (dy == 1 ? yu : (dy == 0 ? y : yd)) : y;
            a[k] = ldx<FLAGS>(gin + (long long)ys * geo.rowStride + (long long)xb * CHUNK + lane + slot_of(k) * LANES);
        }
        float* o = gout + (long long)y * geo.rowStride + (long long)xb * CHUNK + lane;
#pragma unroll
        for (int k = 0; k < Q; k++) stx<FLAGS>(o + slot_of(k) * LANES, a[k]);
        return;
    }
    if (FLAGS & F_NOCOMPUTE) {
        // the real shifted pull, stored without reflect / collide
        pullp<FLAGS>(gin, geo, y, xb, lane, a);
        float* o = gout + (long long)y * geo.rowStride + (long long)xb * CHUNK + lane;
#pragma unroll
        for (int k = 0; k < Q; k++) stx<FLAGS>(o + slot_of(k) * LANES, a[k]);
        return;
    }
    if (FLAGS & (F_DPP | F_DPP_VEDGE)) pullp_dpp<FLAGS>(gin, geo, y, xb, lane, a);
    else pullp<FLAGS>(gin, geo, y, xb, lane, a);
    const uint64_t mrow = mask[(long long)y * geo.Mb + xb];
    const bool blocked = (mrow >> lane) & 1ull;
    reflect(a, blocked);
    collide(a, blocked, omega_of(geo));
    if (FLAGS & F_GROUPED) {
        float* o = gout + (long long)y * geo.rowStride + (long long)xb * (3 * LANES) + lane;
#pragma unroll
        for (int k = 0; k < Q; k++)
            stx<FLAGS>(o + (slot_of(k) / 3) * (geo.Mb * 3 * LANES) + (slot_of(k) % 3) * LANES, a[k]);
        return;
    }
    float* o = gout + (long long)y * geo.rowStride + (long long)xb * CHUNK + lane;
#pragma unroll
    for (int k = 0; k < Q; k++) stx<FLAGS>(o + slot_of(k) * LANES, a[k]);
}

template <int FLAGS>
__global__ __launch_bounds__(256) void kvar(const float* __restrict__ gin, float* __restrict__ gout,
                                            const uint64_t* __restrict__ mask, Geom geo) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long ntiles = (long long)geo.N * geo.Mb;
    if (FLAGS & F_PERSIST) {
        // each workgroup owns a contiguous span of tiles
        const long long per = (ntiles / 4 + gridDim.x - 1) / gridDim.x;  // wg-tiles (4 tiles each)
        const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
        long long t0 = (long long)wg * per, t1 = std::min<long long>(t0 + per, ntiles / 4);
        for (long long t = t0; t < t1; t++) {
            const long long tile = t * 4 + wave;
            do_tile<FLAGS>(gin, gout, mask, geo, (int)(tile / geo.Mb), (int)(tile % geo.Mb), lane);
        }
        return;
    }
    const unsigned wg = (FLAGS & F_NO_REMAP) ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    if (FLAGS & (F_PK | F_PKROW)) {
        // two tiles per wave on packed fp32: F_PK = blocks xb, xb+1 of one row; F_PKROW = rows y, y+1
        const long long t = (long long)wg * 4 + wave;
        int y0, y1, xb0, xb1;
        if (FLAGS & F_PK) { y0 = y1 = (int)(t / (geo.Mb / 2)); xb0 = 2 * (int)(t % (geo.Mb / 2)); xb1 = xb0 + 1; }
        else { y0 = 2 * (int)(t / geo.Mb); y1 = y0 + 1; xb0 = xb1 = (int)(t % geo.Mb); }
        float a[Q], b[Q];
        pullp<FLAGS>(gin, geo, y0, xb0, lane, a);
        pullp<FLAGS>(gin, geo, y1, xb1, lane, b);
        const bool bl0 = (mask[(long long)y0 * geo.Mb + xb0] >> lane) & 1ull;
        const bool bl1 = (mask[(long long)y1 * geo.Mb + xb1] >> lane) & 1ull;
        f2 v[Q];
#pragma unroll
        for (int k = 0; k < Q; k++) v[k] = f2{a[k], b[k]};
        reflect2(v, bl0, bl1);
        collide2(v, bl0, bl1, omega_of(geo));
        float* o0 = gout + (long long)y0 * geo.rowStride + (long long)xb0 * CHUNK + lane;
        float* o1 = gout + (long long)y1 * geo.rowStride + (long long)xb1 * CHUNK + lane;
#pragma unroll
        for (int k = 0; k < Q; k++) stx<FLAGS>(o0 + slot_of(k) * LANES, v[k].x);
#pragma unroll
        for (int k = 0; k < Q; k++) stx<FLAGS>(o1 + slot_of(k) * LANES, v[k].y);
        return;
    }
    if (FLAGS & F_TWO) {
        // a wave does two vertically adjacent tiles (18 loads in flight)
        const long long t = (long long)wg * 4 + wave;  // index over (N/2) x Mb
        const int y = 2 * (int)(t / geo.Mb), xb = (int)(t % geo.Mb);
        float a[Q], b[Q];
        pullp<FLAGS>(gin, geo, y, xb, lane, a);
        pullp<FLAGS>(gin, geo, y + 1, xb, lane, b);
        const bool bl0 = (mask[(long long)y * geo.Mb + xb] >> lane) & 1ull;
        const bool bl1 = (mask[(long long)(y + 1) * geo.Mb + xb] >> lane) & 1ull;
        reflect(a, bl0); collide(a, bl0, omega_of(geo));
        reflect(b, bl1); collide(b, bl1, omega_of(geo));
        float* o = gout + (long long)y * geo.rowStride + (long long)xb * CHUNK + lane;
#pragma unroll
        for (int k = 0; k < Q; k++) stx<FLAGS>(o + slot_of(k) * LANES, a[k]);
        o += geo.rowStride;
#pragma unroll
        for (int k = 0; k < Q; k++) stx<FLAGS>(o + slot_of(k) * LANES, b[k]);
        return;
    }
    int y, xb;
    if (FLAGS & (F_ROWXCD | F_ROWXCD2)) {
        // XCD x takes whole rows (F_ROWXCD: rows = x mod 8; F_ROWXCD2: pairs of rows), so that all
        // XCDs sweep the same few rows at a time: a globally sequential footprint
        const unsigned b = blockIdx.x, xcd = b & 7u, idx = b >> 3;
        const unsigned wpr = geo.Mb / 4;  // workgroups per row
        const unsigned RPX = (FLAGS & F_ROWXCD2) ? 2 : 1;
        const unsigned grp = idx / (wpr * RPX), rem = idx % (wpr * RPX);
        y = (int)((grp * 8 + xcd) * RPX + rem / wpr);
        xb = (int)(rem % wpr) * 4 + wave;
    } else if (FLAGS & F_ROWS4) {
        // workgroup = 4 rows x 1 xb
        const long long t = wg;  // over (N/4) x Mb
        y = 4 * (int)(t / geo.Mb) + wave; xb = (int)(t % geo.Mb);
    } else {
        const long long tile = (long long)wg * 4 + wave;
        y = (int)(tile / geo.Mb); xb = (int)(tile % geo.Mb);
    }
    do_tile<FLAGS>(gin, gout, mask, geo, y, xb, lane);
}

// Marching variant: a wave owns one 64-column block and marches down R rows.  Every row is loaded
// ONCE, whole (nine aligned planes of one 2304-byte block = the "same row" copy shape), shifted in
// x by DPP with one scalar edge dword per shifted plane, and kept in a register history until the
// three destination rows that pull from it have been computed.  P rows are in flight.
template <int R, int P>
__global__ __launch_bounds__(256) void kmarch(const float* __restrict__ gin, float* __restrict__ gout,
                                              const uint64_t* __restrict__ mask, Geom geo) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int wpr = geo.Mb / 4;                 // workgroups per row of blocks
    const int xb = (int)(wg % wpr) * 4 + wave;
    const int y0 = (int)(wg / wpr) * R;
    const int xbm = xb + 1 == geo.Mb ? 0 : xb + 1, xbp = xb == 0 ? geo.Mb - 1 : xb - 1;
    const Omega om = omega_of(geo);
    auto rowp = [&](int y) { y = y < 0 ? y + geo.N : (y >= geo.N ? y - geo.N : y); return gin + (long long)y * geo.rowStride; };
    // one row: 9 aligned values + 6 edge values (uniform addresses -> scalar loads)
    auto load = [&](int y, float* L, float* E) {
        const float* r = rowp(y);
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dx = dx_of(k), sl = slot_of(k) * LANES;
            L[k] = __builtin_nontemporal_load(r + xb * CHUNK + lane + sl);
            if (dx == 1) E[k] = r[xbp * CHUNK + 63 + sl];
            else if (dx == -1) E[k] = r[xbm * CHUNK + sl];
            else E[k] = 0.0f;
        }
    };
    auto shift = [&](float* L, const float* E) {
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const int dx = dx_of(k);
            if (dx == 1) L[k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(E[k]), __float_as_int(L[k]), 0x138, 0xf, 0xf, false));
            else if (dx == -1) L[k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(E[k]), __float_as_int(L[k]), 0x130, 0xf, 0xf, false));
        }
    };
    float pre[P][Q], pree[P][Q];
    float hm1[Q], h0[Q];  // shifted rows y-1 and y
    {
        float e[Q];
        load(y0 - 1, hm1, e); shift(hm1, e);
        load(y0, h0, e); shift(h0, e);
    }
#pragma unroll
    for (int d = 0; d < P; d++) load(y0 + 1 + d, pre[d], pree[d]);
#pragma unroll 1
    for (int i = 0; i < R; i += P) {
#pragma unroll
        for (int d = 0; d < P; d++) {
            const int y = y0 + i + d;
            float hp1[Q];
#pragma unroll
            for (int k = 0; k < Q; k++) hp1[k] = pre[d][k];
            float e[Q];
#pragma unroll
            for (int k = 0; k < Q; k++) e[k] = pree[d][k];
            if (i + d + P < R + 0) load(y + 1 + P, pre[d], pree[d]);
            shift(hp1, e);
            float a[Q];
#pragma unroll
            for (int k = 0; k < Q; k++) {
                const int dy = dy_of(k);
                a[k] = dy == -1 ? hp1[k] : (dy == 0 ? h0[k] : hm1[k]);
            }
            const bool blocked = (mask[(long long)y * geo.Mb + xb] >> lane) & 1ull;
            reflect(a, blocked);
            collide(a, blocked, om);
            float* o = gout + (long long)y * geo.rowStride + (long long)xb * CHUNK + lane;
#pragma unroll
            for (int k = 0; k < Q; k++) __builtin_nontemporal_store(a[k], o + slot_of(k) * LANES);
#pragma unroll
            for (int k = 0; k < Q; k++) { hm1[k] = h0[k]; h0[k] = hp1[k]; }
        }
    }
}

typedef float v4f __attribute__((ext_vector_type(4)));
// pure streaming copies of the same 9.66 GB buffer: what the memory system gives a kernel that does
// nothing else (the yardstick for the fused kernel's 72 B/update)
__global__ __launch_bounds__(256) void copy4(const v4f* __restrict__ in, v4f* __restrict__ out, long long n4) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(in[i], out + i);
}
__global__ __launch_bounds__(256) void copy4nt(const v4f* __restrict__ in, v4f* __restrict__ out, long long n4) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
__global__ __launch_bounds__(256) void copy1(const float* __restrict__ in, float* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(in[i], out + i);
}
__global__ __launch_bounds__(256) void copy4p(const v4f* __restrict__ in, v4f* __restrict__ out, long long n4) {
    // persistent grid-stride
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        __builtin_nontemporal_store(in[i], out + i);
}

// U float4 loads in flight per thread, then U stores (bursts of reads, then of writes)
template <int U, int NT>
__global__ __launch_bounds__(256) void copy4u(const v4f* __restrict__ in, v4f* __restrict__ out, long long n4) {
    const long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    v4f v[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
        const long long i = base + j * 256;
        if (NT & 2) v[j] = __builtin_nontemporal_load(in + (i < n4 ? i : 0));
        else v[j] = in[i < n4 ? i : 0];
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
        const long long i = base + j * 256;
        if (i < n4) { if (NT & 1) __builtin_nontemporal_store(v[j], out + i); else out[i] = v[j]; }
    }
}
// read-only and write-only sweeps: the two halves of the mix
__global__ __launch_bounds__(256) void read4(const v4f* __restrict__ in, float* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    v4f v = i < n4 ? in[i] : v4f{0, 0, 0, 0};
    if (v.x == 12345.678f && v.y == -1.0f) out[0] = v.z;  // never true: keeps the load
}
__global__ __launch_bounds__(256) void write4(v4f* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(v4f{1.0f, 2.0f, 3.0f, (float)threadIdx.x}, out + i);
}

__global__ void initk(float* f, Geom geo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)geo.N * geo.rowStride;
    if (i >= n) return;
    const int slot = (int)((i / 64) % 9);
    const float base[9] = {2.03f, 10.9f, 3.69f, 8.11f, 43.8f, 14.8f, 2.03f, 10.9f, 3.69f};
    unsigned h = (unsigned)(i * 2654435761u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    f[i] = base[slot] * (1.0f + 0.01f * ((h & 0xffff) / 32768.0f - 1.0f));
}

__global__ void diffk(const unsigned* x, const unsigned* y, long long n, unsigned long long* cnt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && x[i] != y[i]) atomicAdd(cnt, 1ull);
}

struct Var { const char* name; void (*fn)(const float*, float*, const uint64_t*, Geom); int div; bool persist; };

template <int FLAGS> static Var mk(const char* n) {
    return Var{n, kvar<FLAGS>, (FLAGS & (F_TWO | F_PK | F_PKROW)) ? 2 : 1, (FLAGS & F_PERSIST) != 0};
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16384, M = argc > 2 ? atoi(argv[2]) : 16384;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5, reps = 10;
    Geom geo{};
    geo.N = N; geo.M = M; geo.Mb = M / 64; geo.G = 0; const int padc = argc > 4 ? atoi(argv[4]) : 0;  // extra (unused) chunks per row: row stride experiment
    geo.rowStride = (long long)(geo.Mb + padc) * CHUNK;
    geo.top = geo.bot = YEDGE_WRAP; geo.omega = 1e-14; geo.omega_f = 1e-14f; geo.omega_f32_exact = 0; geo.omega_tiny = getenv("KB_TINY") ? 1 : 0;
    const size_t bytes = (size_t)N * geo.rowStride * 4;
    float *a, *b; uint64_t* mask;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&mask, (size_t)N * geo.Mb * 8)); CHECK(hipMemset(mask, 0, (size_t)N * geo.Mb * 8));
    hipLaunchKernelGGL(initk, dim3((unsigned)(((long long)N * geo.rowStride + 255) / 256)), dim3(256), 0, 0, a, geo);
    CHECK(hipDeviceSynchronize());
    std::vector<Var> vars = {
        mk<0>("base"), mk<F_NT_ST>("nt_store"), mk<F_NT_ST | F_NT_LD>("nt_both"), mk<F_NT_LD>("nt_load"),
        mk<F_NO_REMAP>("no_remap"), mk<F_ROWS4>("rows4"), mk<F_ROWS4 | F_NT_ST>("rows4_ntst"),
        mk<F_TWO | F_NT_ST>("two_rows_ntst"), mk<F_PK | F_NT_ST>("pk_cols_ntst"), mk<F_PKROW | F_NT_ST>("pk_rows_ntst"),
        mk<F_PK>("pk_cols"), mk<F_GROUPED | F_NT_ST>("grouped_ntst"), mk<F_GROUPED>("grouped"), mk<F_ROWXCD | F_NT_ST>("rowxcd_ntst"), mk<F_ROWXCD2 | F_NT_ST>("rowxcd2_ntst"),
        mk<F_NT_ST | F_NT_LD | F_ROWXCD2>("nt_both_rowxcd2"), mk<F_NT_ST | F_NT_LD | F_ROWXCD>("nt_both_rowxcd"),
        mk<F_NT_ST | F_NT_LD | F_TWO>("nt_both_two_rows"), mk<F_NT_ST | F_NT_LD | F_ROWS4>("nt_both_rows4"), mk<F_NT_ST | F_NT_LD | F_DPP>("nt_both_dpp"),
        mk<F_COPY9 | F_NT_ST | F_NT_LD>("copy9_nt_both"), mk<F_COPY9_ROWS | F_NT_ST | F_NT_LD>("copy9_3rows_nt_both"),
        mk<F_COPY9 | F_NT_ST>("copy9_same_row"), mk<F_COPY9_ROWS | F_NT_ST>("copy9_three_rows"), mk<F_NOCOMPUTE | F_NT_ST>("pull_only"),
        mk<F_DPP | F_NT_ST>("dpp_sedge_ntst"), mk<F_DPP_VEDGE | F_NT_ST>("dpp_vedge_ntst"), mk<F_DPP>("dpp_sedge"),
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long long ntiles = (long long)N * geo.Mb;
    {   // every variant must produce the bits of the base variant (layout-changing ones excepted)
        float* ref; unsigned long long* cnt;
        CHECK(hipMalloc(&ref, bytes)); CHECK(hipMalloc(&cnt, 8));
        hipLaunchKernelGGL(vars[0].fn, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, (const float*)a, ref, (const uint64_t*)mask, geo);
        for (size_t v = 1; v < vars.size(); v++) {
            if (strstr(vars[v].name, "grouped") || strstr(vars[v].name, "copy9") || strstr(vars[v].name, "pull_only")) continue;
            unsigned grid = vars[v].persist ? 256 * 8 : (unsigned)(ntiles / 4 / vars[v].div);
            CHECK(hipMemset(cnt, 0, 8));
            hipLaunchKernelGGL(vars[v].fn, dim3(grid), dim3(256), 0, 0, (const float*)a, b, (const uint64_t*)mask, geo);
            const long long n = (long long)N * geo.rowStride;
            hipLaunchKernelGGL(diffk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const unsigned*)ref, (const unsigned*)b, n, cnt);
            unsigned long long h = 0; CHECK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
            if (h) printf("VARIANT %s DIFFERS from base in %llu values\n", vars[v].name, h);
        }
        CHECK(hipFree(ref)); CHECK(hipFree(cnt));
    }
    std::vector<std::vector<float>> ms(vars.size());
    const unsigned dyn_lds = argc > 6 ? (unsigned)atoi(argv[6]) : 0;  // residency cap for every variant
    for (int r = 0; r < rounds + 1; r++) {
        for (size_t v = 0; v < vars.size(); v++) {
            unsigned grid = vars[v].persist ? 256 * 8 : (unsigned)(ntiles / 4 / vars[v].div);
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; i++) {
                hipLaunchKernelGGL(vars[v].fn, dim3(grid), dim3(256), dyn_lds, 0, (const float*)a, b, (const uint64_t*)mask, geo);
                std::swap(a, b);
            }
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float t; CHECK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms[v].push_back(t / reps);
        }
    }
    {
        const long long n = (long long)N * geo.rowStride;
        for (int mode = 0; mode < 10; mode++) {
            std::vector<float> t;
            for (int r = 0; r < rounds + 1; r++) {
                CHECK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; i++) {
                    if (mode == 0) hipLaunchKernelGGL(copy4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, (const v4f*)a, (v4f*)b, n / 4);
                    else if (mode == 1) hipLaunchKernelGGL(copy1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const float*)a, b, n);
                    else if (mode == 2) hipLaunchKernelGGL(copy4p, dim3(256 * 16), dim3(256), 0, 0, (const v4f*)a, (v4f*)b, n / 4);
                    else if (mode == 3) CHECK(hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, 0));
                    else if (mode == 4) hipLaunchKernelGGL((copy4u<2, 1>), dim3((unsigned)((n / 4 + 511) / 512)), dim3(256), 0, 0, (const v4f*)a, (v4f*)b, n / 4);
                    else if (mode == 5) hipLaunchKernelGGL((copy4u<4, 1>), dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, 0, (const v4f*)a, (v4f*)b, n / 4);
                    else if (mode == 6) hipLaunchKernelGGL((copy4u<8, 1>), dim3((unsigned)((n / 4 + 2047) / 2048)), dim3(256), 0, 0, (const v4f*)a, (v4f*)b, n / 4);
                    else if (mode == 7) hipLaunchKernelGGL((copy4u<4, 3>), dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, 0, (const v4f*)a, (v4f*)b, n / 4);
                    else if (mode == 8) hipLaunchKernelGGL(read4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, (const v4f*)a, b, n / 4);
                    else hipLaunchKernelGGL(write4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, (v4f*)b, n / 4);
                }
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms1; CHECK(hipEventElapsedTime(&ms1, e0, e1));
                if (r > 0) t.push_back(ms1 / reps);
            }
            std::sort(t.begin(), t.end());
            const char* nm[10] = {"copy float4", "copy float", "copy float4 persistent", "hipMemcpy D2D",
                                  "copy float4 x2", "copy float4 x4", "copy float4 x8", "copy float4 x4 nt-load",
                                  "read-only float4", "write-only float4"};
            const double vol = mode >= 8 ? 1.0 : 2.0;
            printf("%-24s median %.3f ms  %.3f TB/s\n", nm[mode], t[t.size() / 2], vol * n * 4 / t[t.size() / 2] / 1e9);
        }
    }
    if (argc > 5) {
        // relative placement of the two ping-pong buffers: does an offset between them change the
        // bandwidth of the fused kernel (HBM channel / bank aliasing of the read and write streams)?
        float* big; const size_t slack = 64ull << 20;
        CHECK(hipMalloc(&big, 2 * bytes + 2 * slack));
        printf("buffers: a=%p b=%p (separate hipMallocs), big=%p\n", (void*)a, (void*)b, (void*)big);
        const size_t offs[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1048576, 2097152 + 4096, 8388608 + 768,
                               33554432 + 2304, 589824, 589824 * 2, 589824 * 3 + 768};
        for (size_t oi = 0; oi < sizeof offs / sizeof offs[0]; oi++) {
            float* pa = big;
            float* pb = (float*)((char*)big + bytes + offs[oi]);
            hipLaunchKernelGGL(initk, dim3((unsigned)(((long long)N * geo.rowStride + 255) / 256)), dim3(256), 0, 0, pa, geo);
            std::vector<float> t;
            for (int r = 0; r < rounds + 1; r++) {
                CHECK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; i++) {
                    hipLaunchKernelGGL(kvar<F_NT_ST>, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, (const float*)pa, pb, (const uint64_t*)mask, geo);
                    std::swap(pa, pb);
                }
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms1; CHECK(hipEventElapsedTime(&ms1, e0, e1));
                if (r > 0) t.push_back(ms1 / reps);
            }
            std::sort(t.begin(), t.end());
            printf("offset %10zu B: %.3f ms  %.3f TB/s\n", offs[oi], t[t.size() / 2], (double)N * M * 72 / t[t.size() / 2] / 1e9);
        }
        CHECK(hipFree(big));
    }
    {   // occupancy sweep: dynamic LDS caps the resident workgroups per CU (160 KB / lds): how many
        // bytes in flight per CU does the memory system like best?
        const int ldsz[] = {0, 20 << 10, 23 << 10, 26 << 10, 32 << 10, 40 << 10, 53 << 10, 80 << 10};
        for (int which = 0; which < 4; which++) {
            for (size_t li = 0; li < sizeof ldsz / sizeof ldsz[0]; li++) {
                std::vector<float> t;
                for (int r = 0; r < rounds + 1; r++) {
                    CHECK(hipEventRecord(e0, 0));
                    for (int i = 0; i < reps; i++) {
                        if (which == 0) hipLaunchKernelGGL(kvar<F_NT_ST>, dim3((unsigned)(ntiles / 4)), dim3(256), ldsz[li], 0, (const float*)a, b, (const uint64_t*)mask, geo);
                        else if (which == 2) hipLaunchKernelGGL((kvar<F_NT_ST | F_NT_LD>), dim3((unsigned)(ntiles / 4)), dim3(256), ldsz[li], 0, (const float*)a, b, (const uint64_t*)mask, geo);
                        else if (which == 3) hipLaunchKernelGGL((kvar<F_NT_ST | F_NT_LD | F_ROWXCD2>), dim3((unsigned)(ntiles / 4)), dim3(256), ldsz[li], 0, (const float*)a, b, (const uint64_t*)mask, geo);
                        else hipLaunchKernelGGL((kvar<F_COPY9 | F_NT_ST>), dim3((unsigned)(ntiles / 4)), dim3(256), ldsz[li], 0, (const float*)a, b, (const uint64_t*)mask, geo);
                        std::swap(a, b);
                    }
                    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                    float ms1; CHECK(hipEventElapsedTime(&ms1, e0, e1));
                    if (r > 0) t.push_back(ms1 / reps);
                }
                std::sort(t.begin(), t.end());
                printf("%s dynLDS %6d B (<= %d WG/CU): %.3f ms  %.3f TB/s\n", which == 0 ? "nt_store" : which == 1 ? "copy9   " : which == 2 ? "nt_both " : "nt_both_rowxcd2", ldsz[li],
                       ldsz[li] ? std::min(8, (160 << 10) / ldsz[li]) : 8, t[t.size() / 2], (double)N * M * 72 / t[t.size() / 2] / 1e9);
            }
        }
    }
    {   // the float4 streaming copy with non-temporal loads as well, under the same residency caps
        const long long n = (long long)N * geo.rowStride;
        const int ldsz[] = {0, 23 << 10, 32 << 10, 40 << 10, 53 << 10, 80 << 10};
        for (size_t li = 0; li < sizeof ldsz / sizeof ldsz[0]; li++) {
            std::vector<float> t;
            for (int r = 0; r < rounds + 1; r++) {
                CHECK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; i++)
                    hipLaunchKernelGGL(copy4nt, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), ldsz[li], 0, (const v4f*)a, (v4f*)b, n / 4);
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms1; CHECK(hipEventElapsedTime(&ms1, e0, e1));
                if (r > 0) t.push_back(ms1 / reps);
            }
            std::sort(t.begin(), t.end());
            printf("copy float4 nt-load nt-store dynLDS %6d B: %.3f ms  %.3f TB/s\n", ldsz[li], t[t.size() / 2], 2.0 * n * 4 / t[t.size() / 2] / 1e9);
        }
    }
    {   // marching kernels: check against base, then time under residency caps
        float* ref; unsigned long long* cnt;
        CHECK(hipMalloc(&ref, bytes)); CHECK(hipMalloc(&cnt, 8));
        hipLaunchKernelGGL(vars[0].fn, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, (const float*)a, ref, (const uint64_t*)mask, geo);
        struct MV { const char* name; void (*fn)(const float*, float*, const uint64_t*, Geom); int R; };
        // (deeper prefetch, P = 2 / 4, and R = 32 / 128 were timed too: 4.7-5.1 TB/s, no better)
        const MV mv[] = {{"march R=64 P=1", kmarch<64, 1>, 64}};
        const long long n = (long long)N * geo.rowStride;
        for (const MV& m : mv) {
            const unsigned grid = (unsigned)((long long)(geo.Mb / 4) * (N / m.R));
            CHECK(hipMemset(cnt, 0, 8));
            hipLaunchKernelGGL(m.fn, dim3(grid), dim3(256), 0, 0, (const float*)a, b, (const uint64_t*)mask, geo);
            hipLaunchKernelGGL(diffk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const unsigned*)ref, (const unsigned*)b, n, cnt);
            unsigned long long h = 0; CHECK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
            if (h) printf("VARIANT %s DIFFERS from base in %llu values\n", m.name, h);
            const int ldsz[] = {0, 20 << 10, 26 << 10, 36 << 10, 53 << 10, 80 << 10};
            for (size_t li = 0; li < sizeof ldsz / sizeof ldsz[0]; li++) {
                std::vector<float> t;
                for (int r = 0; r < rounds + 1; r++) {
                    CHECK(hipEventRecord(e0, 0));
                    for (int i = 0; i < reps; i++) {
                        hipLaunchKernelGGL(m.fn, dim3(grid), dim3(256), ldsz[li], 0, (const float*)a, b, (const uint64_t*)mask, geo);
                        std::swap(a, b);
                    }
                    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                    float ms1; CHECK(hipEventElapsedTime(&ms1, e0, e1));
                    if (r > 0) t.push_back(ms1 / reps);
                }
                std::sort(t.begin(), t.end());
                printf("%-16s dynLDS %6d B: %.3f ms  %.3f TB/s\n", m.name, ldsz[li], t[t.size() / 2], (double)N * M * 72 / t[t.size() / 2] / 1e9);
            }
        }
        CHECK(hipFree(ref)); CHECK(hipFree(cnt));
    }
    float chk[4]; CHECK(hipMemcpy(chk, a + 12345 * 64, sizeof chk, hipMemcpyDeviceToHost));
    printf("check %g %g %g %g\n", chk[0], chk[1], chk[2], chk[3]);
    const double cells = (double)N * M;
    for (size_t v = 0; v < vars.size(); v++) {
        std::sort(ms[v].begin(), ms[v].end());
        float med = ms[v][ms[v].size() / 2], mn = ms[v][0];
        printf("%-16s median %.3f ms  min %.3f ms  %.2f GLUPS  %.3f TB/s\n", vars[v].name, med, mn,
               cells / med / 1e6, cells * 72 / med / 1e9);
    }
    return 0;
}
