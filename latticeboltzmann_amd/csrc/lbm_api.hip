// C-ABI host side of liblbm_hip.so (declarations and reference citations: include/lbm_hip.h).
// No torch types, no exceptions across the boundary, no CPU fallback.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/lbm_hip.h"
#include "lbm_internal.h"

using namespace lbm;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) (void)hipGetLastError(); /* un-stick it for later launch checks */ \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? LBM_ERR_NOMEM : LBM_ERR_HIP,            \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct lbm_solver {
    lbm_config cfg;
    Geom geo;
    int xmode;
    bool q21;     // LBM_LATTICE_D2Q21 (lbm_d2q21.hip): fp32, one update per launch
    float surr21[Q21];  // its surroundings (reference edges)
    int Q;        // populations per cell: 9 or 21
    bool half;
    int T;
    int tb_max;   // deepest temporal-blocking launch (1 = never use the LDS kernel, 0 = automatic)
    int tb_rows;  // rows per workgroup chunk of the LDS kernel
    hipStream_t own_stream, stream;
    void* buf[2];
    size_t buf_bytes;
    int cur;
    uint64_t* mask;
    size_t mask_bytes;
    bool pulled;       // buf[cur] holds g = collide(f_{steps-1}); false: holds f_steps itself
    int64_t steps;
    int ghost_valid;   // valid ghost rows beyond each seam
    bool open_update;  // lbm_advance_rows(finish = 0) has written part of the next state ...
    int open_depth;    // ... which lies this many updates ahead
    int n_cov;         // owned-row intervals [a, b) the open update has written so far (merged,
    int cov[16][2];    // sorted); finish is refused until they cover [0, N)
    void* staging;     // device staging for host-layout windows
    size_t staging_bytes;
    uint8_t* frame_dev;
    void* scratch;     // this device's copy of the static scratch array (never freed)
    hipEvent_t ev0, ev1;
};

static const size_t STAGING_BYTES = 256ull << 20;

// --- static device scratch of the allocation-free entry point --------------------------------
// lbm_fused_collide_stream_display_noalloc renders its frame into this array instead of a
// hipMalloc'ed staging buffer.  One copy per GPU, held until the process ends.  In host code the
// array's NAME is only a handle for hipGetSymbolAddress: the device address is looked up once per
// device (in lbm_create, with that device current) and kept in g_scratch_ptr / lbm_solver::scratch;
// kernels and copies only ever see that pointer.
static __device__ __align__(256) unsigned char scratch[LBM_NOALLOC_FRAME_MAX_BYTES];
static const int MAX_DEVICES = 64;
static void* g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mu;

static hipError_t scratch_for_device(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // the current device's copy
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

// Makes the solver's device current for the duration of one API call and puts the calling
// thread's previous device back afterwards: a torch process that steps a solver on device d keeps
// its own current device.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int device) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        if (prev == device) return hipSuccess;
        e = hipSetDevice(device);
        switched = e == hipSuccess && prev >= 0;
        return e;
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
};

static int check(lbm_solver* s, DeviceScope& scope) {
    if (!s) return fail(LBM_ERR_INVALID, "null solver handle");
    HIPCHK(scope.enter(s->cfg.device));
    // HIP's last error is sticky and thread-wide: one left behind by other code in the process
    // (torch, another library) must not be reported by this call's launch checks
    (void)hipGetLastError();
    return LBM_OK;
}
#define ENTER(s)                          \
    DeviceScope device_scope_;            \
    do {                                  \
        int rc_ = check(s, device_scope_); \
        if (rc_ != LBM_OK) return rc_;    \
    } while (0)

#define NOT_FOR_Q21(s, what)                                                                   \
    do {                                                                                       \
        if ((s)->q21) return fail(LBM_ERR_UNSUPPORTED, what " is not available for LBM_LATTICE_D2Q21"); \
    } while (0)

static size_t elem_size(const lbm_solver* s) { return s->half ? 2 : 4; }
static bool seam_top(const lbm_solver* s) { return s->geo.top == YEDGE_GHOST; }
static bool seam_bot(const lbm_solver* s) { return s->geo.bot == YEDGE_GHOST; }

// Unused 2304-byte blocks at the end of every row of the field buffers (never read or written).
// LBM_ROW_PAD in the environment overrides it (development sweeps).
static int row_pad_blocks(int Mb) {
    static const int env = [] {
        const char* e = getenv("LBM_ROW_PAD");
        return e ? atoi(e) : -1;
    }();
    if (env >= 0) return env;
    (void)Mb;
    return 0;
}

extern "C" {

int lbm_abi_version(void) { return LBM_ABI_VERSION; }
const char* lbm_last_error(void) { return g_err; }

int lbm_device_count(int* count) {
    if (!count) return fail(LBM_ERR_INVALID, "count is null");
    *count = 0;
    HIPCHK(hipGetDeviceCount(count));
    return LBM_OK;
}

int lbm_create(const lbm_config* cfg, lbm_solver** out) {
    if (!cfg || !out) return fail(LBM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(lbm_config))
        return fail(LBM_ERR_INVALID, "lbm_config.struct_size %u != %zu", cfg->struct_size,
                    sizeof(lbm_config));
    if (cfg->abi_version != LBM_ABI_VERSION)
        return fail(LBM_ERR_INVALID, "lbm_config.abi_version %d != %d: the caller was built against another "
                                     "lbm_hip.h (omega is a double since ABI 2)", cfg->abi_version, LBM_ABI_VERSION);
    if (cfg->N < 1 || cfg->M < 2) return fail(LBM_ERR_INVALID, "need N >= 1 and M >= 2");
    if (cfg->inner_timesteps < 1 || cfg->inner_timesteps > 64)
        return fail(LBM_ERR_INVALID, "inner_timesteps must be in [1, 64]");
    if (cfg->boundary != LBM_BOUNDARY_REFERENCE && cfg->boundary != LBM_BOUNDARY_PERIODIC &&
        cfg->boundary != LBM_BOUNDARY_CUDA_KERNEL)
        return fail(LBM_ERR_INVALID, "unknown boundary %d", cfg->boundary);
    if (cfg->boundary == LBM_BOUNDARY_CUDA_KERNEL &&
        (cfg->halo != 0 || cfg->top_edge != LBM_YEDGE_DOMAIN || cfg->bottom_edge != LBM_YEDGE_DOMAIN))
        return fail(LBM_ERR_UNSUPPORTED, "LBM_BOUNDARY_CUDA_KERNEL is single-domain only (no halo, no seams)");
    if (cfg->storage != LBM_STORE_F32 && cfg->storage != LBM_STORE_F16)
        return fail(LBM_ERR_INVALID, "unknown storage %d", cfg->storage);
    if (cfg->halo < 0 || cfg->halo > 64) return fail(LBM_ERR_INVALID, "halo must be in [0, 64]");
    if (cfg->lattice != LBM_LATTICE_D2Q9 && cfg->lattice != LBM_LATTICE_D2Q21)
        return fail(LBM_ERR_INVALID, "unknown lattice %d", cfg->lattice);
    if (cfg->lattice == LBM_LATTICE_D2Q21) {
        if (cfg->boundary == LBM_BOUNDARY_CUDA_KERNEL || cfg->storage != LBM_STORE_F32 || cfg->halo != 0 ||
            cfg->top_edge != LBM_YEDGE_DOMAIN || cfg->bottom_edge != LBM_YEDGE_DOMAIN)
            return fail(LBM_ERR_UNSUPPORTED, "LBM_LATTICE_D2Q21 is periodic or reference edges, fp32, single-domain only");
        if (cfg->N < 4 || cfg->M < 4) return fail(LBM_ERR_INVALID, "LBM_LATTICE_D2Q21 needs N >= 4 and M >= 4");
    }
    const bool st = cfg->top_edge == LBM_YEDGE_SEAM, sb = cfg->bottom_edge == LBM_YEDGE_SEAM;
    if ((st || sb) && cfg->halo < 1) return fail(LBM_ERR_INVALID, "a seam edge needs halo >= 1");
    if ((st || sb) && cfg->N < cfg->halo) return fail(LBM_ERR_INVALID, "N must be >= halo");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(LBM_ERR_INVALID, "device %d out of range (%d HIP devices)", cfg->device, ndev);
    DeviceScope device_scope_;
    HIPCHK(device_scope_.enter(cfg->device));

    lbm_solver* s = new (std::nothrow) lbm_solver();
    if (!s) return fail(LBM_ERR_NOMEM, "host allocation failed");
    memset(s, 0, sizeof *s);
    s->cfg = *cfg;
    s->xmode = cfg->boundary == LBM_BOUNDARY_PERIODIC ? XMODE_PERIODIC
             : cfg->boundary == LBM_BOUNDARY_CUDA_KERNEL ? XMODE_CUDA : XMODE_REFERENCE;
    s->half = cfg->storage == LBM_STORE_F16;
    s->q21 = cfg->lattice == LBM_LATTICE_D2Q21;
    s->Q = s->q21 ? Q21 : Q;
    s->T = cfg->inner_timesteps;
    s->tb_max = 0;  // automatic (depth_cap)
    s->tb_rows = 0;  // 0 = chosen per launch (auto_chunk_rows)
    Geom& g = s->geo;
    g.N = cfg->N;
    g.M = cfg->M;
    g.Mb = (cfg->M + LANES - 1) / LANES;
    g.G = cfg->halo;
    g.rowStride = s->q21 ? (long long)g.Mb * Q21 * LANES : (long long)(g.Mb + row_pad_blocks(g.Mb)) * CHUNK;
    const int dom = cfg->boundary == LBM_BOUNDARY_PERIODIC ? YEDGE_WRAP : YEDGE_DOMAIN;
    g.top = st ? YEDGE_GHOST : dom;
    g.bot = sb ? YEDGE_GHOST : dom;
    g.omega = cfg->omega;
    g.omega_f = (float)cfg->omega;
    g.omega_f32_exact = (double)g.omega_f == cfg->omega;
    g.omega_tiny = cfg->omega <= OMEGA_TINY_MAX && cfg->omega >= -OMEGA_TINY_MAX;  // false for NaN
    const size_t rows = (size_t)g.N + 2 * (size_t)g.G;
    s->buf_bytes = rows * (size_t)g.rowStride * elem_size(s);
    s->mask_bytes = rows * (size_t)g.Mb * sizeof(uint64_t);
    s->staging_bytes = STAGING_BYTES;

    int rc = LBM_OK;
    auto bail = [&](hipError_t e, const char* what) {
        (void)hipGetLastError();  // HIP keeps the last error until it is read: do not leave it behind
        rc = fail(e == hipErrorOutOfMemory ? LBM_ERR_NOMEM : LBM_ERR_HIP, "%s failed: %s", what,
                  hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking)) != hipSuccess) bail(e, "hipStreamCreate");
    if (rc == LBM_OK && (e = hipMalloc(&s->buf[0], s->buf_bytes)) != hipSuccess) bail(e, "hipMalloc(cells)");
    if (rc == LBM_OK && (e = hipMalloc(&s->buf[1], s->buf_bytes)) != hipSuccess) bail(e, "hipMalloc(newcells)");
    if (rc == LBM_OK && (e = hipMalloc((void**)&s->mask, s->mask_bytes)) != hipSuccess) bail(e, "hipMalloc(mask)");
    if (rc == LBM_OK && (e = hipMalloc(&s->staging, s->staging_bytes)) != hipSuccess) bail(e, "hipMalloc(staging)");
    if (rc == LBM_OK && (e = scratch_for_device(cfg->device, &s->scratch)) != hipSuccess) bail(e, "hipGetSymbolAddress(scratch)");
    if (rc == LBM_OK && (e = hipEventCreate(&s->ev0)) != hipSuccess) bail(e, "hipEventCreate");
    if (rc == LBM_OK && (e = hipEventCreate(&s->ev1)) != hipSuccess) bail(e, "hipEventCreate");
    // On the solver's own (non-blocking) stream: a hipMemset on the null stream returns before it
    // has run and is NOT ordered against this stream, so it could wipe what the first upload wrote.
    if (rc == LBM_OK && (e = hipMemsetAsync(s->buf[0], 0, s->buf_bytes, s->own_stream)) != hipSuccess) bail(e, "hipMemset");
    if (rc == LBM_OK && (e = hipMemsetAsync(s->buf[1], 0, s->buf_bytes, s->own_stream)) != hipSuccess) bail(e, "hipMemset");
    if (rc == LBM_OK && (e = hipMemsetAsync(s->mask, 0, s->mask_bytes, s->own_stream)) != hipSuccess) bail(e, "hipMemset");
    if (rc == LBM_OK && (e = hipStreamSynchronize(s->own_stream)) != hipSuccess) bail(e, "hipStreamSynchronize");
    if (rc != LBM_OK) {
        lbm_destroy(s);
        return rc;
    }
    s->stream = s->own_stream;
    *out = s;
    return LBM_OK;
}

int lbm_destroy(lbm_solver* s) {
    if (!s) return LBM_OK;
    DeviceScope device_scope_;
    (void)device_scope_.enter(s->cfg.device);
    if (s->own_stream) (void)hipStreamSynchronize(s->own_stream);
    for (int i = 0; i < 2; i++)
        if (s->buf[i]) (void)hipFree(s->buf[i]);
    if (s->mask) (void)hipFree(s->mask);
    if (s->staging) (void)hipFree(s->staging);
    if (s->frame_dev) (void)hipFree(s->frame_dev);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    delete s;
    return LBM_OK;
}

int lbm_set_stream(lbm_solver* s, void* hip_stream) {
    ENTER(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
    return LBM_OK;
}

int lbm_set_surroundings(lbm_solver* s, const float* surroundings) {
    ENTER(s);
    if (!surroundings) return fail(LBM_ERR_INVALID, "surroundings is null");
    if (s->q21) {  // 21 values for this lattice
        memcpy(s->surr21, surroundings, Q21 * sizeof(float));
        return LBM_OK;
    }
    memcpy(s->geo.surr, surroundings, 9 * sizeof(float));
    return LBM_OK;
}

static int check_rows(lbm_solver* s, int row0, int nrows, bool ghosts_ok) {
    const int lo = ghosts_ok ? -s->geo.G : 0, hi = s->geo.N + (ghosts_ok ? s->geo.G : 0);
    if (nrows < 0 || row0 < lo || (long long)row0 + nrows > hi)
        return fail(LBM_ERR_INVALID, "rows [%d, %d) outside [%d, %d)", row0, row0 + nrows, lo, hi);
    return LBM_OK;
}

int lbm_upload_blocked(lbm_solver* s, const uint8_t* blocked, int row0, int nrows) {
    ENTER(s);
    if (!blocked) return fail(LBM_ERR_INVALID, "blocked is null");
    int rc = check_rows(s, row0, nrows, true);
    if (rc) return rc;
    const size_t row_bytes = (size_t)s->geo.M;
    const int chunk = (int)std::max<size_t>(1, s->staging_bytes / row_bytes);
    for (int r = 0; r < nrows; r += chunk) {
        const int n = std::min(chunk, nrows - r);
        HIPCHK(hipMemcpyAsync(s->staging, blocked + (size_t)r * row_bytes, (size_t)n * row_bytes,
                              hipMemcpyHostToDevice, s->stream));
        HIPCHK(launch_import_mask(s->mask, s->geo, (const uint8_t*)s->staging, row0 + r, n, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return LBM_OK;
}

int lbm_upload_cells(lbm_solver* s, const float* cells, int row0, int nrows) {
    ENTER(s);
    if (!cells) return fail(LBM_ERR_INVALID, "cells is null");
    int rc = check_rows(s, row0, nrows, true);
    if (rc) return rc;
    s->open_update = false;
    if (s->pulled) {  // a new initial condition: back to the freshly-uploaded state
        // the buffer holds post-collision populations now; a partial upload would mix the two
        if (row0 > 0 || row0 + nrows < s->geo.N)
            return fail(LBM_ERR_INVALID, "after stepping, an upload must cover all %d rows (got [%d, %d))",
                        s->geo.N, row0, row0 + nrows);
        s->pulled = false;
        s->steps = 0;
    }
    const size_t row_bytes = (size_t)s->geo.M * s->Q * sizeof(float);
    const int chunk = (int)std::max<size_t>(1, s->staging_bytes / row_bytes);
    if (row_bytes > s->staging_bytes) return fail(LBM_ERR_UNSUPPORTED, "row too long for staging");
    for (int r = 0; r < nrows; r += chunk) {
        const int n = std::min(chunk, nrows - r);
        HIPCHK(hipMemcpyAsync(s->staging, (const char*)cells + (size_t)r * row_bytes,
                              (size_t)n * row_bytes, hipMemcpyHostToDevice, s->stream));
        if (s->q21)
            HIPCHK(launch_q21_import_cells(s->buf[s->cur], s->geo, (const float*)s->staging, row0 + r, n, s->stream));
        else
        HIPCHK(launch_import_cells(s->buf[s->cur], s->geo, s->half, (const float*)s->staging, row0 + r,
                                   n, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return LBM_OK;
}

int lbm_init_equilibrium(lbm_solver* s, float rho, float u_row, float u_col, float amp, uint64_t seed,
                         int64_t row_offset, int64_t global_rows) {
    ENTER(s);
    s->pulled = false;
    s->open_update = false;
    s->steps = 0;
    s->ghost_valid = s->geo.G;
    if (s->q21) {
        HIPCHK(launch_q21_init_equilibrium(s->buf[s->cur], s->geo, rho, u_row, u_col, amp, seed, s->stream));
        return LBM_OK;
    }
    HIPCHK(launch_init_equilibrium(s->buf[s->cur], s->geo, s->half, rho, u_row, u_col, amp, seed,
                                   row_offset, global_rows, s->stream));
    return LBM_OK;
}

int lbm_init_blocked_discs(lbm_solver* s, int ndiscs, int radius, uint64_t seed, int64_t row_offset,
                           int64_t global_rows) {
    ENTER(s);
    if (ndiscs < 0 || ndiscs > 4096 || radius < 0) return fail(LBM_ERR_INVALID, "bad disc parameters");
    if (global_rows <= 0) global_rows = s->geo.N;
    HIPCHK(launch_init_discs(s->mask, s->geo, ndiscs, radius, seed, row_offset, global_rows, s->stream));
    return LBM_OK;
}

int lbm_halo_mark_filled(lbm_solver* s) {
    ENTER(s);
    s->ghost_valid = s->geo.G;
    return LBM_OK;
}

int lbm_halo_valid(lbm_solver* s, int* rows) {
    ENTER(s);
    if (!rows) return fail(LBM_ERR_INVALID, "rows is null");
    *rows = (seam_top(s) || seam_bot(s)) ? s->ghost_valid : 1 << 30;
    return LBM_OK;
}

static hipError_t launch_deep(lbm_solver* s, int t, int r0, int nrows) {
    return launch_step_tb(s->buf[s->cur], s->buf[s->cur ^ 1], s->mask, s->geo, s->xmode, s->half, t, r0, nrows,
                          s->tb_rows, s->stream);
}

// Deepest launch in use.  Automatic: TB_DEFAULT_T, but the plain kernel for grids below 2^17 cells,
// where both are bound by launch latency alike (128 x 256: 8.3 GLUPS either way; from 400 x 800 on
// the LDS kernel wins: 62 against 56 GLUPS).

static int depth_cap(const lbm_solver* s) {
    if (s->xmode == XMODE_CUDA) return 1;  // the compatibility mode only exists in the plain kernel
    if (s->tb_max > 0) return std::min(s->tb_max, TB_MAX_T);
    return (long long)s->geo.N * s->geo.M < TB_MIN_CELLS ? 1 : TB_DEFAULT_T;
}

// Advance the numpy-visible state by `total` updates.  Right after an upload the buffer holds f:
// the first update only collides in place (its stream + reflect half is deferred into the next
// kernel or into the observers).  Afterwards updates go in balanced launches of depth t <= max_depth:
// t = 1 is the plain fused kernel, t > 1 the LDS temporal-blocking kernel (lbm_tb.hip).  At a
// seam each update consumes one ghost row.
static int advance(lbm_solver* s, long long total, int max_depth) {
    const Geom& g = s->geo;
    if (total > 0 && s->open_update)
        return fail(LBM_ERR_INVALID, "an update opened by lbm_step_rows(finish = 0) is unfinished: "
                                     "finish it (finish != 0) before stepping further");
    const bool seam = seam_top(s) || seam_bot(s);
    while (total > 0 && s->q21) {  // D2Q21: collide-only first update, then one update per launch
        if (!s->pulled) {
            HIPCHK(launch_q21_collide(s->buf[s->cur], s->mask, g, s->stream));
            s->pulled = true;
        } else {
            HIPCHK(launch_q21_step(s->buf[s->cur], s->buf[s->cur ^ 1], s->mask, g, s->xmode, s->surr21, s->stream));
            s->cur ^= 1;
        }
        s->steps++;
        total--;
    }
    while (total > 0) {
        if (!s->pulled) {
            HIPCHK(launch_collide(s->buf[s->cur], s->mask, g, s->xmode, s->half, -g.G, g.N + 2 * g.G, s->stream));
            s->pulled = true;
            s->steps++;
            total--;
            continue;
        }
        if (seam && s->ghost_valid < 1)
            return fail(LBM_ERR_INVALID, "ghost rows exhausted: exchange halos (lbm_halo_region) and "
                                         "call lbm_halo_mark_filled before stepping further");
        // balanced launches: 7 updates at depth cap 6 go as 4 + 3, not 6 + 1 (a depth-1 launch
        // costs more than half a depth-6 one)
        const long long nl = (total + max_depth - 1) / max_depth;
        int t = (int)((total + nl - 1) / nl);
        // ... except that depth 5 is the slow one (ten waves on four SIMDs: per update slower than depth 4,
        // DESIGN section 5): when more launches follow anyway, 10 goes as 6 + 4 and 15 as 6 + 6 + 3, not
        // as 5 + 5 (+ 5).  The number of launches stays the same.
        if (t == 5 && total > 5 && max_depth >= 6) t = 6;
        if (seam) t = std::min(t, s->ghost_valid);
        if (s->xmode == XMODE_CUDA) t = 1;
        if (g.M < 8 || g.N < 16) t = 1;  // tiny domains: rows wrap at most once in the strip pipeline
        const int ext = seam ? s->ghost_valid - t : 0;
        const int r0 = seam_top(s) ? -ext : 0;
        const int r1 = g.N + (seam_bot(s) ? ext : 0);
        if (t == 1)
            HIPCHK(launch_step(s->buf[s->cur], s->buf[s->cur ^ 1], s->mask, g, s->xmode, s->half, r0,
                               r1 - r0, s->stream));
        else
            HIPCHK(launch_deep(s, t, r0, r1 - r0));
        s->cur ^= 1;
        if (seam) s->ghost_valid -= t;
        s->steps += t;
        total -= t;
    }
    return LBM_OK;
}

// One outer iteration = inner_timesteps updates, in ceil(T / depth_cap) balanced launches (advance();
// e.g. T = 7 at cap 6 -> 4 + 3).
static int outer_iteration(lbm_solver* s) {
    int rem = s->T;
    if (!s->pulled) {
        int rc = advance(s, 1, 1);
        if (rc) return rc;
        rem--;
    }
    if (rem <= 0) return LBM_OK;
    return advance(s, rem, depth_cap(s));
}

int lbm_fused_collide_stream(lbm_solver* s, int n_outer) {
    ENTER(s);
    if (n_outer < 0) return fail(LBM_ERR_INVALID, "n_outer < 0");
    for (int i = 0; i < n_outer; i++) {
        int rc = outer_iteration(s);
        if (rc) return rc;
    }
    return LBM_OK;
}

int lbm_advance(lbm_solver* s, int n_updates) {
    ENTER(s);
    if (n_updates < 0) return fail(LBM_ERR_INVALID, "n_updates < 0");
    return advance(s, n_updates, depth_cap(s));
}

int lbm_set_tuning(lbm_solver* s, int tb_max_depth, int tb_rows) {
    ENTER(s);
    NOT_FOR_Q21(s, "lbm_set_tuning");
    if (tb_max_depth < 0 || tb_max_depth > TB_MAX_T || tb_rows < 0)
        return fail(LBM_ERR_INVALID, "tb_max_depth must be in [0, %d] and tb_rows >= 0", TB_MAX_T);
    s->tb_max = tb_max_depth;
    s->tb_rows = tb_rows;
    return LBM_OK;
}

int lbm_advance_rows(lbm_solver* s, int row0, int nrows, int depth, int finish) {
    ENTER(s);
    NOT_FOR_Q21(s, "lbm_step_rows / lbm_advance_rows");
    int rc = check_rows(s, row0, nrows, true);
    if (rc) return rc;
    if (!s->pulled)
        return fail(LBM_ERR_INVALID, "lbm_step_rows / lbm_advance_rows need a stepped state (call "
                                     "lbm_fused_collide_stream once after an upload)");
    const Geom& g = s->geo;
    if (depth < 1 || depth > TB_MAX_T) return fail(LBM_ERR_INVALID, "depth must be in [1, %d]", TB_MAX_T);
    if (depth > 1 && (g.M < 8 || g.N < 16))
        return fail(LBM_ERR_UNSUPPORTED, "grids below 16 x 8 only take depth 1");
    if (depth > 1 && s->xmode == XMODE_CUDA)
        return fail(LBM_ERR_UNSUPPORTED, "LBM_BOUNDARY_CUDA_KERNEL only takes depth 1");
    if (s->open_update && depth != s->open_depth)
        return fail(LBM_ERR_INVALID, "the open update has depth %d, not %d", s->open_depth, depth);
    // Rows that may be computed: owned rows, plus -- at a seam only -- the ghost rows whose
    // inputs [row-depth, row+depth] are still valid.  Beyond a domain (or wrap) edge there is
    // nothing to compute and nothing allocated to read.
    const int lo = seam_top(s) ? -(s->ghost_valid - depth) : 0;
    const int hi = g.N + (seam_bot(s) ? s->ghost_valid - depth : 0);
    if (nrows > 0 && (row0 < lo || row0 + nrows > hi))
        return fail(LBM_ERR_INVALID, "rows [%d, %d) at depth %d: only [%d, %d) can be computed "
                    "(%d valid ghost rows at a seam, none beyond a domain edge)",
                    row0, row0 + nrows, depth, lo, hi, s->ghost_valid);
    // owned rows this open update will have written after this call
    int ncov = s->open_update ? s->n_cov : 0;
    int cov[17][2];
    memcpy(cov, s->cov, sizeof s->cov);
    {
        const int a = std::max(row0, 0), b = std::min(row0 + nrows, g.N);
        if (a < b) {
            int m = 0, na = a, nb = b;  // merge [a, b) into the sorted interval list
            int out[17][2];
            bool placed = false;
            for (int i = 0; i < ncov; i++) {
                if (cov[i][1] < na) { out[m][0] = cov[i][0]; out[m][1] = cov[i][1]; m++; }
                else if (cov[i][0] > nb) {
                    if (!placed) { out[m][0] = na; out[m][1] = nb; m++; placed = true; }
                    out[m][0] = cov[i][0]; out[m][1] = cov[i][1]; m++;
                } else { na = std::min(na, cov[i][0]); nb = std::max(nb, cov[i][1]); }
            }
            if (!placed) { out[m][0] = na; out[m][1] = nb; m++; }
            if (m > 16) return fail(LBM_ERR_UNSUPPORTED, "more than 16 disjoint row ranges in one open update");
            memcpy(cov, out, sizeof(int) * 2 * m);
            ncov = m;
        }
    }
    if (finish && !(ncov == 1 && cov[0][0] == 0 && cov[0][1] == g.N))
        return fail(LBM_ERR_INVALID, "finish: the row ranges of this update do not cover all %d owned rows "
                    "(uncovered rows would keep stale data)", g.N);
    if (depth == 1)
        HIPCHK(launch_step(s->buf[s->cur], s->buf[s->cur ^ 1], s->mask, g, s->xmode, s->half, row0, nrows,
                           s->stream));
    else
        HIPCHK(launch_deep(s, depth, row0, nrows));
    s->open_update = !finish;
    s->open_depth = depth;
    s->n_cov = finish ? 0 : ncov;
    memcpy(s->cov, cov, sizeof s->cov);
    if (finish) {
        s->cur ^= 1;
        s->steps += depth;
        if (seam_top(s) || seam_bot(s)) s->ghost_valid = std::max(0, s->ghost_valid - depth);
    }
    return LBM_OK;
}

int lbm_step_rows(lbm_solver* s, int row0, int nrows, int finish) {
    return lbm_advance_rows(s, row0, nrows, 1, finish);
}

int lbm_fused_collide_stream_timed(lbm_solver* s, int n_outer, float* elapsed_ms) {
    ENTER(s);
    if (!elapsed_ms) return fail(LBM_ERR_INVALID, "elapsed_ms is null");
    HIPCHK(hipEventRecord(s->ev0, s->stream));
    int rc = lbm_fused_collide_stream(s, n_outer);
    if (rc) return rc;
    HIPCHK(hipEventRecord(s->ev1, s->stream));
    HIPCHK(hipEventSynchronize(s->ev1));
    HIPCHK(hipEventElapsedTime(elapsed_ms, s->ev0, s->ev1));
    return LBM_OK;
}

int lbm_sync(lbm_solver* s) {
    ENTER(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    return LBM_OK;
}

int lbm_steps_done(lbm_solver* s, int64_t* steps) {
    ENTER(s);
    if (!steps) return fail(LBM_ERR_INVALID, "steps is null");
    *steps = s->steps;
    return LBM_OK;
}

static int check_window(lbm_solver* s, int row0, int nrows, int col0, int ncols) {
    int rc = check_rows(s, row0, nrows, false);
    if (rc) return rc;
    if (ncols < 0 || col0 < 0 || (long long)col0 + ncols > s->geo.M)
        return fail(LBM_ERR_INVALID, "columns [%d, %d) outside [0, %d)", col0, col0 + ncols, s->geo.M);
    if (s->pulled && (seam_top(s) || seam_bot(s)) && s->ghost_valid < 1)
        return fail(LBM_ERR_INVALID, "ghost rows exhausted: exchange halos before reading back");
    return LBM_OK;
}

// Observe a window chunk by chunk through the device staging buffer.
static int observe(lbm_solver* s, int row0, int nrows, int col0, int ncols, float* cells, float* rho,
                   float* u, uint8_t* rgb) {
    if (nrows == 0 || ncols == 0) return LBM_OK;
    const size_t cell_b = (size_t)s->Q * sizeof(float);
    const size_t per_row = (size_t)ncols * ((cells ? cell_b : 0) + (rho ? 4 : 0) + (u ? 8 : 0) + (rgb ? 4 : 0));
    if (per_row == 0) return LBM_OK;
    if (per_row > s->staging_bytes) return fail(LBM_ERR_UNSUPPORTED, "row too long for staging");
    const int chunk = (int)std::max<size_t>(1, s->staging_bytes / per_row);
    for (int r = 0; r < nrows; r += chunk) {
        const int n = std::min(chunk, nrows - r);
        char* p = (char*)s->staging;
        float* dc = nullptr; float* dr = nullptr; float* du = nullptr; uint8_t* dg = nullptr;
        const size_t cells_n = (size_t)n * ncols;
        if (cells) { dc = (float*)p; p += cells_n * cell_b; }
        if (rho) { dr = (float*)p; p += cells_n * 4; }
        if (u) { du = (float*)p; p += cells_n * 8; }
        if (rgb) { dg = (uint8_t*)p; }
        if (s->q21)
            HIPCHK(launch_q21_observe(s->buf[s->cur], s->mask, s->geo, s->xmode, s->surr21, s->pulled, row0 + r, n, col0,
                                      ncols, dc, dr, du, s->stream));
        else
        HIPCHK(launch_observe(s->buf[s->cur], s->mask, s->geo, s->xmode, s->half, s->pulled, row0 + r, n,
                              col0, ncols, dc, dr, du, dg, s->stream));
        const size_t o = (size_t)r * ncols;
        if (cells) HIPCHK(hipMemcpyAsync(cells + o * s->Q, dc, cells_n * cell_b, hipMemcpyDeviceToHost, s->stream));
        if (rho) HIPCHK(hipMemcpyAsync(rho + o, dr, cells_n * 4, hipMemcpyDeviceToHost, s->stream));
        if (u) HIPCHK(hipMemcpyAsync(u + o * 2, du, cells_n * 8, hipMemcpyDeviceToHost, s->stream));
        if (rgb) HIPCHK(hipMemcpyAsync(rgb + o * 3, dg, cells_n * 3, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return LBM_OK;
}

int lbm_read_cells(lbm_solver* s, float* cells, int row0, int nrows, int col0, int ncols) {
    ENTER(s);
    if (!cells) return fail(LBM_ERR_INVALID, "cells is null");
    int rc = check_window(s, row0, nrows, col0, ncols);
    if (rc) return rc;
    return observe(s, row0, nrows, col0, ncols, cells, nullptr, nullptr, nullptr);
}

int lbm_read_moments(lbm_solver* s, float* rho, float* u, int row0, int nrows, int col0, int ncols) {
    ENTER(s);
    if (!rho && !u) return fail(LBM_ERR_INVALID, "rho and u are both null");
    int rc = check_window(s, row0, nrows, col0, ncols);
    if (rc) return rc;
    return observe(s, row0, nrows, col0, ncols, nullptr, rho, u, nullptr);
}

int lbm_read_blocked(lbm_solver* s, uint8_t* blocked, int row0, int nrows) {
    ENTER(s);
    if (!blocked) return fail(LBM_ERR_INVALID, "blocked is null");
    int rc = check_rows(s, row0, nrows, true);
    if (rc) return rc;
    const size_t row_bytes = (size_t)s->geo.M;
    const int chunk = (int)std::max<size_t>(1, s->staging_bytes / row_bytes);
    for (int r = 0; r < nrows; r += chunk) {
        const int n = std::min(chunk, nrows - r);
        HIPCHK(launch_export_mask(s->mask, s->geo, (uint8_t*)s->staging, row0 + r, n, s->stream));
        HIPCHK(hipMemcpyAsync(blocked + (size_t)r * row_bytes, s->staging, (size_t)n * row_bytes,
                              hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return LBM_OK;
}

int lbm_render(lbm_solver* s, uint8_t* rgb, int row0, int nrows) {
    ENTER(s);
    NOT_FOR_Q21(s, "lbm_render");
    if (!rgb) return fail(LBM_ERR_INVALID, "rgb is null");
    int rc = check_window(s, row0, nrows, 0, s->geo.M);
    if (rc) return rc;
    return observe(s, row0, nrows, 0, s->geo.M, nullptr, nullptr, nullptr, rgb);
}

int lbm_host_alloc(size_t bytes, void** host_ptr) {
    if (!host_ptr || bytes == 0) return fail(LBM_ERR_INVALID, "bad argument");
    *host_ptr = nullptr;
    HIPCHK(hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return LBM_OK;
}

int lbm_host_free(void* host_ptr) {
    if (!host_ptr) return LBM_OK;
    HIPCHK(hipHostFree(host_ptr));
    return LBM_OK;
}

// A frame is rendered from the state AFTER the outer iteration, which at a seam needs one valid
// ghost row then: checked before the step, so that an error return leaves the state untouched.
static int check_frame_after_step(lbm_solver* s) {
    if (seam_top(s) || seam_bot(s)) {
        const int consumed = s->pulled ? s->T : s->T - 1;  // the first update after an upload only collides
        if (s->ghost_valid - consumed < 1)
            return fail(LBM_ERR_INVALID, "a frame after this outer iteration needs %d valid ghost rows, %d are "
                                         "left: exchange halos first", consumed + 1, s->ghost_valid);
    }
    return LBM_OK;
}

int lbm_fused_collide_stream_display(lbm_solver* s, uint8_t* frame) {
    ENTER(s);
    if (frame) NOT_FOR_Q21(s, "a rendered frame");
    const size_t bytes = (size_t)s->geo.N * s->geo.M * 3;
    if (frame) {
        int rc = check_frame_after_step(s);
        if (rc) return rc;
        if (!s->frame_dev) HIPCHK(hipMalloc((void**)&s->frame_dev, bytes));
    }
    int rc = lbm_fused_collide_stream(s, 1);
    if (rc || !frame) return rc;
    HIPCHK(launch_observe(s->buf[s->cur], s->mask, s->geo, s->xmode, s->half, s->pulled, 0, s->geo.N, 0,
                          s->geo.M, nullptr, nullptr, nullptr, s->frame_dev, s->stream));
    HIPCHK(hipMemcpyAsync(frame, s->frame_dev, bytes, hipMemcpyDefault, s->stream));  // host or device frame
    return LBM_OK;
}

// The same step + frame without any allocation: the frame is rendered into the per-device static
// scratch (looked up in lbm_create) and copied from there.  Nothing here calls hipMalloc / hipFree /
// hipHostMalloc or any routine that does; a pageable `frame` is rejected because the runtime would
// stage such a copy through buffers of its own.
int lbm_fused_collide_stream_display_noalloc(lbm_solver* s, uint8_t* frame) {
    ENTER(s);
    NOT_FOR_Q21(s, "lbm_fused_collide_stream_display_noalloc");
    const size_t bytes = (size_t)s->geo.N * s->geo.M * 3;
    if (frame) {
        if (bytes > (size_t)LBM_NOALLOC_FRAME_MAX_BYTES)
            return fail(LBM_ERR_UNSUPPORTED,
                        "a %d x %d frame is %zu bytes, more than the %u-byte static scratch of "
                        "lbm_fused_collide_stream_display_noalloc (use lbm_fused_collide_stream_display)",
                        s->geo.N, s->geo.M, bytes, (unsigned)LBM_NOALLOC_FRAME_MAX_BYTES);
        if (!s->scratch) return fail(LBM_ERR_HIP, "the static scratch was not resolved on device %d", s->cfg.device);
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        const hipError_t pe = hipPointerGetAttributes(&at, frame);
        if (pe != hipSuccess) (void)hipGetLastError();
        if (pe != hipSuccess || at.type == hipMemoryTypeUnregistered)
            return fail(LBM_ERR_INVALID, "frame must be page-locked host memory (lbm_host_alloc) or device "
                                         "memory: a pageable destination makes the runtime stage the copy");
        int rc = check_frame_after_step(s);  // every check that can fail comes before the step
        if (rc) return rc;
    }
    int rc = lbm_fused_collide_stream(s, 1);
    if (rc || !frame) return rc;
    HIPCHK(launch_observe(s->buf[s->cur], s->mask, s->geo, s->xmode, s->half, s->pulled, 0, s->geo.N, 0,
                          s->geo.M, nullptr, nullptr, nullptr, (uint8_t*)s->scratch, s->stream));
    // hipMemcpyDefault: `frame` may be page-locked host memory or device memory
    HIPCHK(hipMemcpyAsync(frame, s->scratch, bytes, hipMemcpyDefault, s->stream));
    return LBM_OK;
}

int lbm_halo_region(lbm_solver* s, int side, int recv, int next, void** device_ptr, size_t* bytes) {
    ENTER(s);
    NOT_FOR_Q21(s, "lbm_halo_region");
    if (!device_ptr || !bytes || side < 0 || side > 1) return fail(LBM_ERR_INVALID, "bad argument");
    const int b = next ? (s->cur ^ 1) : s->cur;
    const Geom& g = s->geo;
    if (g.G < 1) return fail(LBM_ERR_INVALID, "solver has no halo");
    // rows in allocation coordinates (row + G)
    long long r;
    if (side == 0) r = recv ? 0 : g.G;
    else r = recv ? g.G + g.N : g.N;  // owned rows [N-G, N) -> allocation rows [N, N+G)
    *device_ptr = (char*)s->buf[b] + (size_t)r * g.rowStride * elem_size(s);
    *bytes = (size_t)g.G * g.rowStride * elem_size(s);
    return LBM_OK;
}

int lbm_mask_halo_region(lbm_solver* s, int side, int recv, void** device_ptr, size_t* bytes) {
    ENTER(s);
    if (!device_ptr || !bytes || side < 0 || side > 1) return fail(LBM_ERR_INVALID, "bad argument");
    const Geom& g = s->geo;
    if (g.G < 1) return fail(LBM_ERR_INVALID, "solver has no halo");
    long long r;
    if (side == 0) r = recv ? 0 : g.G;
    else r = recv ? g.G + g.N : g.N;
    *device_ptr = (char*)s->mask + (size_t)r * g.Mb * sizeof(uint64_t);
    *bytes = (size_t)g.G * g.Mb * sizeof(uint64_t);
    return LBM_OK;
}

}  // extern "C"
