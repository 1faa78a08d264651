// Development micro-benchmark: what makes a streaming copy slow on MI355X -- long-lived waves, each
// wave owning a contiguous stream, or bursts?  All variants move the same 2 x 9.66 GB in 2304-byte
// chunks (nine 256-byte dword-per-lane loads and stores per chunk, non-temporal), 256 threads per
// workgroup, residency capped by an unused dynamic-LDS request.
//   short      one chunk per wave, grid = all chunks (the T=1 kernel's shape)
//   long/sweep persistent waves; in step i the grid's waves cover one contiguous span, then advance
//   long/own   persistent waves; each wave copies its own contiguous run of chunks
//   long/wg    persistent waves; each WORKGROUP owns a contiguous run, its waves interleaved
//   PF         chunks loaded ahead of the one being stored (register prefetch)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { SHORT = 0, SWEEP = 1, OWN = 2, WG = 3, SPLIT = 4, LOCK = 5, ROT = 6 };  // ROT: sweep, but which wave takes which chunk of the span changes every step  // LOCK: sweep with a soft (bounded-spin) grid-wide rendezvous every SYNC_EVERY steps  // SPLIT: sweep, odd waves only load, even waves only store

__device__ unsigned g_arrived;
template <int MODE, int PF, int IO = 0, int EXTRA = 0>  // IO: 0 copy, 1 read only, 2 write only; EXTRA: per-step extras (1 s_sleep, 2 L1 invalidate, 3 rotating priority, 4 wait for the stores)
__global__ __launch_bounds__(256) void kstream(const float* __restrict__ in, float* __restrict__ out,
                                               long long nchunks, int iters, int sync_every, unsigned arrive_base) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    auto chunk_of = [&](int it) -> long long {
        if (MODE == SHORT) return gw;
        if (MODE == SWEEP || MODE == LOCK) return (long long)it * nw + gw;
        if (MODE == ROT) return (long long)it * nw + (gw + (long long)it * 1237) % nw;
        if (MODE == SPLIT) return (long long)it * (nw / 2) + gw / 2;
        if (MODE == OWN) return gw * iters + it;
        return ((long long)blockIdx.x * iters + it) * 4 + (threadIdx.x >> 6);  // WG: 4 chunks per step
    };
    float v[PF + 1][9];
    auto load = [&](int it, float* d) {
        if (IO == 2 || (MODE == SPLIT && !(gw & 1))) {  // a storing wave: nothing to load
#pragma unroll
            for (int k = 0; k < 9; k++) d[k] = (float)(it + k);
            return;
        }
        long long c = chunk_of(it);
        c = c < nchunks ? c : nchunks - 1;
        const float* p = in + c * 576 + lane;
#pragma unroll
        for (int k = 0; k < 9; k++) d[k] = __builtin_nontemporal_load(p + 64 * k);
    };
    auto store = [&](int it, const float* d) {
        const long long c = chunk_of(it);
        if (c >= nchunks) return;
        if (IO == 1 || (MODE == SPLIT && (gw & 1))) {  // a loading wave: consume the values, store (almost) never
            float s = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) s += d[k];
            if (s == 1.2345e-30f) out[c * 576 + lane] = s;
            return;
        }
        float* p = out + c * 576 + lane;
#pragma unroll
        for (int k = 0; k < 9; k++) __builtin_nontemporal_store(d[k], p + 64 * k);
    };
    if (EXTRA == 5) {  // de-phase the waves of a CU once: a pseudo-random start delay of 0..15 x ~500 cycles
        const int phase = (int)(((blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u) >> 28);
        for (int i = 0; i < phase; i++) __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int d = 0; d < PF; d++) load(d, v[d]);
    // PF+1 register sets, rotated by unrolling the loop PF+1 times
    for (int it = 0; it < iters; it += PF + 1) {
        if (MODE == LOCK && it > 0 && it % sync_every == 0) {
            // never a hang: a workgroup that has waited 20000 polls goes on alone
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&g_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = arrive_base + (unsigned)(it / sync_every) * gridDim.x;
                for (int spin = 0; spin < 20000; spin++)
                    if ((int)(__hip_atomic_load(&g_arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) break;
            }
            __syncthreads();
        }
        if (EXTRA == 1) __builtin_amdgcn_s_sleep(2);
        if (EXTRA == 2) asm volatile("buffer_inv sc1");
        if (EXTRA == 3) { if (it & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
        if (EXTRA == 4) __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int u = 0; u <= PF; u++) {
            if (it + u < iters) {
                if (it + u + PF < iters + 0) load(it + u + PF, v[(u + PF) % (PF + 1)]);
                store(it + u, v[u]);
            }
        }
    }
}

// four chunks per wave: dword-per-lane loads (the shape the lattice kernels need: one column per
// lane), transposed through LDS, stored as nine 1-KB float4 instructions.  WIDE_LD: loads wide too.
template <int WIDE_LD>
__global__ __launch_bounds__(256) void kwide(const float* __restrict__ in, float* __restrict__ out, long long nchunks) {
    __shared__ float lds[4][2304];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c0 = ((long long)blockIdx.x * 4 + wave) * 4;
    if (c0 >= nchunks) return;
    const float* p = in + c0 * 576;
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 w[9];
    if (WIDE_LD) {
#pragma unroll
        for (int j = 0; j < 9; j++) w[j] = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p) + j * 64 + lane);
    } else {
        float v[36];
#pragma unroll
        for (int j = 0; j < 36; j++) v[j] = __builtin_nontemporal_load(p + 64 * j + lane);
#pragma unroll
        for (int j = 0; j < 36; j++) lds[wave][64 * j + lane] = v[j];
        // (one wave reads what it wrote itself: no barrier)
#pragma unroll
        for (int j = 0; j < 9; j++) w[j] = reinterpret_cast<const v4*>(lds[wave])[j * 64 + lane];
    }
    v4* o = reinterpret_cast<v4*>(out + c0 * 576);
#pragma unroll
    for (int j = 0; j < 9; j++) __builtin_nontemporal_store(w[j], o + j * 64 + lane);
}
template <int WIDE_LD>
static int run_wide(const char* name, const float* a, float* b, long long nchunks, int rounds) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((nchunks + 15) / 16);
    std::vector<float> t;
    for (int r = 0; r < rounds + 1; r++) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kwide<WIDE_LD>, dim3(grid), dim3(256), 0, 0, a, b, nchunks);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms / 5);
    }
    std::sort(t.begin(), t.end());
    printf("%-40s: %.3f ms  %.3f TB/s\n", name, t[t.size() / 2], 2.0 * nchunks * 2304 / t[t.size() / 2] / 1e9);
    fflush(stdout);
    return 0;
}
// DYN: persistent waves that take their chunks from a queue (one counter per XCD band, G chunks per
// grab) instead of a fixed assignment: chunk order = the order in which waves become free, which is
// what the hardware dispatcher gives one-chunk waves.  If the lifetime effect were waves drifting
// apart (a front spread over ever more DRAM pages), this would run like `short`.
__device__ unsigned g_next[8 * 32];  // one counter per XCD, 128 bytes apart
template <int G>
__global__ __launch_bounds__(256) void kdyn(const float* __restrict__ in, float* __restrict__ out, long long nchunks) {
    const int lane = threadIdx.x & 63;
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned band = (unsigned)(nchunks / 8), lo = xcd * band;  // nchunks is a multiple of 8 here
    auto grab = [&]() -> unsigned {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(&g_next[xcd * 32], (unsigned)G);
        return __builtin_amdgcn_readfirstlane(v);
    };
    unsigned cur = grab();
    while (cur < band) {  // the counter only grows: every wave gets here with cur >= band in the end
        const unsigned nxt = grab();
#pragma unroll
        for (int g = 0; g < G; g++) {
            const unsigned c = cur + g;
            if (c < band) {
                const float* p = in + (long long)(lo + c) * 576 + lane;
                float v[9];
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] = __builtin_nontemporal_load(p + 64 * k);
                float* o = out + (long long)(lo + c) * 576 + lane;
#pragma unroll
                for (int k = 0; k < 9; k++) __builtin_nontemporal_store(v[k], o + 64 * k);
            }
        }
        cur = nxt;
    }
}
template <int G>
static int run_dyn(const char* name, const float* a, float* b, long long nchunks, int wg_per_cu, unsigned lds, int rounds) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    void* cnt; CHECK(hipGetSymbolAddress(&cnt, HIP_SYMBOL(g_next)));
    const unsigned grid = 256u * wg_per_cu;
    std::vector<float> t;
    for (int r = 0; r < rounds + 1; r++) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; i++) {
            CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * 8 * 32, 0));
            hipLaunchKernelGGL(kdyn<G>, dim3(grid), dim3(256), lds, 0, a, b, nchunks);
        }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms / 5);
    }
    std::sort(t.begin(), t.end());
    printf("%-22s G=%d  %d WG/CU (LDS %6u): %.3f ms  %.3f TB/s\n", name, G, wg_per_cu, lds, t[t.size() / 2],
           2.0 * nchunks * 2304 / t[t.size() / 2] / 1e9);
    fflush(stdout);
    return 0;
}
// MARCH: long-lived waves that walk DOWN the rows of a 16384-column field like stage 1 of the LDS
// kernel (a wave owns one 64-column block, R rows).  WHOLE = 0: per destination row the three
// 768-byte pieces the pull takes from rows y+1, y, y-1 (each chunk is read in three visits, an iteration
// apart); WHOLE = 1: one whole 2304-byte chunk of row y+1 per iteration (what a register history of two
// rows would make of it).  Both store one chunk per iteration.
template <int WHOLE>
__global__ __launch_bounds__(256) void kmarchcopy(const float* __restrict__ in, float* __restrict__ out, int nrows,
                                                  int R) {
    const int lane = threadIdx.x & 63;
    const int nb = 256;  // blocks per row
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int xb = (int)(w % nb), y0 = (int)(w / nb) * R;
    const long long RS = (long long)nb * 576;
    auto row = [&](int y) { y = y < 0 ? y + nrows : (y >= nrows ? y - nrows : y); return in + (long long)y * RS + (long long)xb * 576 + lane; };
    float v[9];
    for (int i = 0; i < R; i++) {
        const int y = y0 + i;
        if (WHOLE) {
            const float* p = row(y + 1);
#pragma unroll
            for (int k = 0; k < 9; k++) v[k] = __builtin_nontemporal_load(p + 64 * k);
        } else {
            const float* pd = row(y + 1), *p0 = row(y), *pu = row(y - 1);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                v[k] = __builtin_nontemporal_load(pd + 64 * k);
                v[3 + k] = __builtin_nontemporal_load(p0 + 64 * (3 + k));
                v[6 + k] = __builtin_nontemporal_load(pu + 64 * (6 + k));
            }
        }
        float* o = out + (long long)y * RS + (long long)xb * 576 + lane;
#pragma unroll
        for (int k = 0; k < 9; k++) __builtin_nontemporal_store(v[k], o + 64 * k);
    }
}
template <int WHOLE>
static int run_march(const char* name, const float* a, float* b, int R, unsigned lds, int rounds) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nrows = 16384;
    const unsigned grid = (unsigned)(256 / 4 * (nrows / R));
    std::vector<float> t;
    for (int r = 0; r < rounds + 1; r++) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kmarchcopy<WHOLE>, dim3(grid), dim3(256), lds, 0, a, b, nrows, R);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms / 5);
    }
    std::sort(t.begin(), t.end());
    printf("%-34s R=%4d (LDS %6u): %.3f ms  %.3f TB/s\n", name, R, lds, t[t.size() / 2],
           2.0 * 16384 * 256 * 2304 / t[t.size() / 2] / 1e9);
    fflush(stdout);
    return 0;
}
static unsigned g_base = 0;
template <int MODE, int PF, int IO = 0, int EXTRA = 0>
static int run(const char* name, const float* a, float* b, long long nchunks, int wg_per_cu, unsigned lds, int rounds, int sync_every = 8) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned grid; int iters;
    if (MODE == SHORT) { grid = (unsigned)((nchunks + 3) / 4); iters = 1; }
    else if (MODE == SPLIT) { grid = 256u * wg_per_cu; iters = (int)((nchunks + grid * 2ll - 1) / (grid * 2ll)); }
    else if (wg_per_cu < 0) { iters = -wg_per_cu; grid = (unsigned)((nchunks + 4ll * iters - 1) / (4ll * iters)); }  // medium-lived waves
    else { grid = 256u * wg_per_cu; iters = (int)((nchunks + grid * 4ll - 1) / (grid * 4ll)); }
    std::vector<float> t;
    for (int r = 0; r < rounds + 1; r++) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; i++) {
            hipLaunchKernelGGL((kstream<MODE, PF, IO, EXTRA>), dim3(grid), dim3(256), lds, 0, a, b, nchunks, iters, sync_every, g_base);
            if (MODE == LOCK) g_base += (unsigned)((iters - 1) / sync_every) * grid;  // arrivals this launch adds (iters not a multiple of PF+1 aside)
        }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms / 5);
    }
    std::sort(t.begin(), t.end());
    printf("%-22s PF=%d  %d WG/CU (LDS %6u): %.3f ms  %.3f TB/s\n", name, PF, wg_per_cu, lds, t[t.size() / 2],
           (IO ? 1.0 : 2.0) * nchunks * 2304 / t[t.size() / 2] / 1e9);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const long long nchunks = 16384ll * 256;  // 16384 x 16384 cells
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    float *a, *b;
    CHECK(hipMalloc(&a, nchunks * 2304)); CHECK(hipMalloc(&b, nchunks * 2304));
    CHECK(hipMemset(a, 0, nchunks * 2304));
    const struct { int wg; unsigned lds; } caps[] = {{4, 36u << 10}, {6, 26u << 10}, {3, 53u << 10}, {2, 80u << 10}};
    if (argc > 2 && !strcmp(argv[2], "pmc")) {  // the two ends of the lifetime effect only (for rocprofv3 --pmc passes)
        run<SHORT, 0>("short", a, b, nchunks, 4, 36u << 10, rounds);
        run<SWEEP, 0>("long/sweep", a, b, nchunks, 4, 36u << 10, rounds);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "march")) {  // three pieces from three rows against one whole chunk, marching waves
        for (int R : {64, 256, 1024})
            for (unsigned l : {36u << 10, 53u << 10}) {
                run_march<0>("march, 3 pieces of 768 B", a, b, R, l, rounds);
                run_march<1>("march, 1 whole chunk", a, b, R, l, rounds);
            }
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "dyn")) {  // queue-fed persistent waves against both ends
        run<SHORT, 0>("short", a, b, nchunks, 4, 36u << 10, rounds);
        run<SWEEP, 0>("long/sweep", a, b, nchunks, 4, 36u << 10, rounds);
        run<SWEEP, 0, 0, 5>("long/sweep dephased", a, b, nchunks, 4, 36u << 10, rounds);
        run<SWEEP, 0, 2, 0>("long/sweep write-only", a, b, nchunks, 4, 36u << 10, rounds);
        run<SWEEP, 0, 2, 5>("long/sweep write-only dephased", a, b, nchunks, 4, 36u << 10, rounds);
        run_dyn<2>("long/queue", a, b, nchunks, 4, 36u << 10, rounds);
        run_dyn<4>("long/queue", a, b, nchunks, 4, 36u << 10, rounds);
        run_dyn<16>("long/queue", a, b, nchunks, 4, 36u << 10, rounds);
        run_dyn<1>("long/queue", a, b, nchunks, 6, 26u << 10, rounds);
        run_dyn<4>("long/queue", a, b, nchunks, 6, 26u << 10, rounds);
        run<SHORT, 0>("short", a, b, nchunks, 4, 36u << 10, rounds);
        return 0;
    }
    if (argc > 2)  // a soft grid-wide rendezvous every k steps: 0.3-3 TB/s, the rendezvous itself is what one measures
        for (int k : {1, 2, 4, 8, 32}) run<LOCK, 0>("long/sweep lockstep", a, b, nchunks, 4, 36u << 10, rounds, k);
    for (int wg : {1, 2, 3}) {
        const unsigned l = wg == 1 ? 100u << 10 : wg == 2 ? 60u << 10 : 50u << 10;
        run<SWEEP, 0>("long/sweep", a, b, nchunks, wg, l, rounds);
        run<SWEEP, 1>("long/sweep", a, b, nchunks, wg, l, rounds);
        run<SWEEP, 3>("long/sweep", a, b, nchunks, wg, l, rounds);
        run<OWN, 1>("long/own", a, b, nchunks, wg, l, rounds);
    }
    run<SWEEP, 0, 0, 0>("long/sweep", a, b, nchunks, 4, 36u << 10, rounds);
    run<SWEEP, 0, 0, 1>("long/sweep + s_sleep", a, b, nchunks, 4, 36u << 10, rounds);
    run<SWEEP, 0, 0, 2>("long/sweep + wbinvl1", a, b, nchunks, 4, 36u << 10, rounds);
    run<SWEEP, 0, 0, 3>("long/sweep + setprio", a, b, nchunks, 4, 36u << 10, rounds);
    run<SWEEP, 0, 0, 4>("long/sweep + waitcnt0", a, b, nchunks, 4, 36u << 10, rounds);
    run_wide<0>("4 chunks/wave, dword loads, float4 stores", a, b, nchunks, rounds);
    run_wide<1>("4 chunks/wave, float4 loads and stores", a, b, nchunks, rounds);
    run<SHORT, 0>("short", a, b, nchunks, 4, 36u << 10, rounds);
    run<SHORT, 0, 1>("short read-only", a, b, nchunks, 4, 36u << 10, rounds);
    run<SWEEP, 0, 1>("long/sweep read-only", a, b, nchunks, 4, 36u << 10, rounds);
    run<SWEEP, 3, 1>("long/sweep read-only", a, b, nchunks, 4, 36u << 10, rounds);
    run<SHORT, 0, 2>("short write-only", a, b, nchunks, 4, 36u << 10, rounds);
    run<SWEEP, 0, 2>("long/sweep write-only", a, b, nchunks, 4, 36u << 10, rounds);
    run<ROT, 0>("long/sweep rotated", a, b, nchunks, 4, 36u << 10, rounds);
    run<ROT, 1>("long/sweep rotated", a, b, nchunks, 4, 36u << 10, rounds);
    for (int it : {2, 4, 8, 16, 64}) run<SWEEP, 0>("sweep, iters per wave:", a, b, nchunks, -it, 36u << 10, rounds);
    for (int it : {2, 8, 64}) run<OWN, 0>("own, iters per wave:", a, b, nchunks, -it, 36u << 10, rounds);
    for (int wg : {4, 6, 8}) { run<SPLIT, 0>("long/split", a, b, nchunks, wg, wg == 4 ? 36u << 10 : wg == 6 ? 26u << 10 : 0, rounds); run<SPLIT, 1>("long/split", a, b, nchunks, wg, wg == 4 ? 36u << 10 : wg == 6 ? 26u << 10 : 0, rounds); run<SPLIT, 3>("long/split", a, b, nchunks, wg, wg == 4 ? 36u << 10 : wg == 6 ? 26u << 10 : 0, rounds); }
    for (auto c : caps) {
        if (c.wg != 4) continue;
        run<SHORT, 0>("short", a, b, nchunks, c.wg, c.lds, rounds);
        run<SWEEP, 0>("long/sweep", a, b, nchunks, c.wg, c.lds, rounds);
        run<OWN, 0>("long/own", a, b, nchunks, c.wg, c.lds, rounds);
        run<WG, 0>("long/wg", a, b, nchunks, c.wg, c.lds, rounds);
        run<SWEEP, 1>("long/sweep", a, b, nchunks, c.wg, c.lds, rounds);
        run<OWN, 1>("long/own", a, b, nchunks, c.wg, c.lds, rounds);
        run<OWN, 3>("long/own", a, b, nchunks, c.wg, c.lds, rounds);
        run<WG, 3>("long/wg", a, b, nchunks, c.wg, c.lds, rounds);
    }
    return 0;
}
