/* lbm_hip.h -- C-ABI of liblbm_hip.so, the MI355X (gfx950) D2Q9 BGK lattice-Boltzmann path.
 *
 * This is the drop-in boundary for the reference's GPU host layer.  In the reference
 * (/root/reference) the boundary is pycuda: latticeboltzmann.py:125-140 JIT-compiles
 * lb_cuda_kernel.cu with N, M, OMEGA, INNER_TIMESTEPS baked in as #defines, :147-152 uploads
 * cells / blocked / surroundings with drv.to_device, :187-196 launches
 * fused_collide_stream_display(newcells, frame|0, cells, blocked, surroundings)
 * (lb_cuda_kernel.cu:371-385) once per outer iteration and :202 swaps the two cell buffers on the
 * Python side.  Here the same steps are plain C calls on an opaque solver handle; geometry and
 * physics are run-time parameters, the ping-pong and the device layout (AoSoA-64, bit-packed mask)
 * are private to the library.
 *
 * Conventions (all from the reference's numpy path, latticeboltzmann.py:24-35,115-123):
 *   cells      float32 (N, M, 9), C order, k fastest;  k = 3*i + j,  e[k] = (dy, dx) = (j-1, i-1)
 *   blocked    uint8 / numpy bool (N, M), nonzero = obstacle
 *   u          float32 (N, M, 2) = [row (y) component, column (x) component]
 *   rho        float32 (N, M)
 * Every function returns 0 on success and a nonzero lbm_status otherwise; lbm_last_error() gives
 * the message of the calling thread's last failure.  No exception, abort or exit crosses this ABI.
 * A handle is not thread-safe; different handles may be used from different threads.
 * Every call makes the solver's device current while it runs and restores the calling thread's
 * previous current device before it returns.
 * There is no CPU fallback: without a HIP device every call that needs one fails.
 */
#ifndef LBM_HIP_H
#define LBM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_ABI_VERSION 3 /* 2: lbm_config.omega is a double; 3: lbm_config.abi_version, checked by lbm_create */

typedef enum {
    LBM_OK = 0,
    LBM_ERR_INVALID = 1,   /* bad argument / bad state */
    LBM_ERR_HIP = 2,       /* a HIP runtime call failed */
    LBM_ERR_NOMEM = 3,
    LBM_ERR_UNSUPPORTED = 4
} lbm_status;

/* Edge rule of the stream step. */
typedef enum {
    LBM_BOUNDARY_REFERENCE = 0, /* latticeboltzmann.py:115-120: column 0 / row 0 take surroundings[k]
                                   for dx / dy = +1; last column / last row keep their own value */
    LBM_BOUNDARY_PERIODIC = 1,  /* wrap-around in x and y (BASELINE.json configs 2-4; no reference
                                   counterpart) */
    LBM_BOUNDARY_CUDA_KERNEL = 2 /* compatibility mode, never the default: the semantics of the
                                   reference's CUDA kernel where they differ from its numpy path
                                   (lb_cuda_kernel.cu:215,232-233,286-306,322-331): all four sides
                                   supply surroundings, density = sum + 0.0001f, the collide of row
                                   r is gated by blocked[r-1][x], row 0 is never collided.  IEEE
                                   fp32 in the kernel's operation order (the kernel itself is built
                                   with --use_fast_math): PARITY-UNPINNED, no reference fixture
                                   exists.  Single domain, one update per launch only; omega is
                                   used as a float (`#define OMEGA ...f`). */
} lbm_boundary;

/* How a y-edge of THIS solver's row range is treated (y-slab decomposition). */
typedef enum {
    LBM_YEDGE_DOMAIN = 0, /* a true domain edge: `boundary` applies */
    LBM_YEDGE_SEAM = 1    /* a slab seam: `halo` ghost rows beyond it are filled by the caller
                             (lbm_halo_*), no edge rule */
} lbm_yedge;

/* Velocity set.  Everything in this header describes D2Q9 unless it says otherwise. */
typedef enum {
    LBM_LATTICE_D2Q9 = 0,  /* latticeboltzmann.py:24-35 */
    LBM_LATTICE_D2Q21 = 1  /* the draft's lattice, latticeboltzmann.jl:3-22 (r2 = 3/2, 21 weights and
                              velocities (dy, dx) with |d| <= 3; opposite of k is 20-k).  Scoped: one
                              update per launch, LBM_BOUNDARY_PERIODIC or LBM_BOUNDARY_REFERENCE (the
                              slices of latticeboltzmann.py:115-120 with these velocities: the first
                              dx columns / dy rows take surroundings[k], the last |dx| / |dy| keep
                              their own value), LBM_STORE_F32, single domain.  cells are float32
                              (N, M, 21) and lbm_set_surroundings takes 21 values; N, M >= 4.
                              Supported calls: create / set_surroundings /
                              destroy / set_stream, upload_blocked / upload_cells / read_blocked,
                              init_equilibrium (row_offset / global_rows ignored), init_blocked_discs,
                              fused_collide_stream[_timed], advance, sync, steps_done, read_cells,
                              read_moments; the rest return LBM_ERR_UNSUPPORTED.  PARITY UNPINNED: the
                              draft is not a runnable program and the reference holds no output of it;
                              results are those of oracle/lbm_d2q21.py (the reference's numpy algorithm
                              with the draft's tables), bit for bit. */
} lbm_lattice;

typedef enum {
    LBM_STORE_F32 = 0,
    LBM_STORE_F16 = 1 /* fp16 storage / fp32 compute, lb_cuda_kernel.cu:152-179 (half_enable).
                         Every time level is rounded through fp16, also between the updates of one
                         temporal-blocking launch (as the reference rounds newcells on every inner
                         timestep, lb_cuda_kernel.cu:292,361): results depend on the number of
                         updates only, not on launch depth, tuning, grid size or slab count. */
} lbm_storage;

/* Replaces the #define block latticeboltzmann.py:126-132 and the launch geometry :187. */
typedef struct {
    uint32_t struct_size;     /* = sizeof(lbm_config), for ABI evolution */
    int32_t device;           /* HIP device ordinal */
    int32_t N;                /* rows owned by this solver */
    int32_t M;                /* columns */
    double omega;             /* OMEGA, latticeboltzmann.py:80, a Python float: the reference's omega
                                 array (:98; blocked cells use 1) is float64 and `cells *= omega`
                                 (:112) is a double product rounded once to float32.  The device
                                 does the same (plain fp32 multiply when omega is a float32 value,
                                 which gives the same bits). */
    int32_t inner_timesteps;  /* INNER_TIMESTEPS, latticeboltzmann.py:56: updates per outer step */
    int32_t boundary;         /* lbm_boundary */
    int32_t storage;          /* lbm_storage */
    int32_t halo;             /* ghost rows kept above and below (0 for a single domain) */
    int32_t top_edge;         /* lbm_yedge for row 0 */
    int32_t bottom_edge;      /* lbm_yedge for row N-1 */
    int32_t abi_version;      /* = LBM_ABI_VERSION of the header the caller was built against.
                                 sizeof(lbm_config) did not change when omega became a double (ABI 2),
                                 so struct_size alone cannot tell an ABI-1 caller apart: lbm_create
                                 rejects any other value here (an older caller leaves 0) */
    int32_t lattice;          /* lbm_lattice (0 = D2Q9) */
    int32_t reserved[2];
} lbm_config;

typedef struct lbm_solver lbm_solver;

/* --- library ------------------------------------------------------------------------------- */
int lbm_abi_version(void);
const char* lbm_last_error(void);
int lbm_device_count(int* count);

/* --- lifetime: replaces SourceModule/get_function/prepare (latticeboltzmann.py:134-140) and the
 * device allocations (:147-152). ------------------------------------------------------------- */
int lbm_create(const lbm_config* cfg, lbm_solver** out);
int lbm_destroy(lbm_solver* s);
/* Run on a caller-owned hipStream_t (e.g. torch.cuda.Stream().cuda_stream) instead of the solver's
 * own stream; NULL restores the solver's stream.  Replaces the stream argument of
 * prepared_async_call (latticeboltzmann.py:187). */
int lbm_set_stream(lbm_solver* s, void* hip_stream);

/* --- upload: replaces drv.to_device(cells / blocked / surroundings), latticeboltzmann.py:147-150.
 * Host pointers; the library copies, the caller keeps ownership.  Rows [row0, row0+nrows) in
 * solver-local coordinates; ghost rows are addressed as -halo..-1 and N..N+halo-1.  Before the
 * first step cells may be uploaded in pieces; once the solver has stepped, an upload is a new
 * initial condition and must cover all N rows (it resets the step count). -------------------- */
int lbm_set_surroundings(lbm_solver* s, const float* surroundings /* 9 values; 21 for LBM_LATTICE_D2Q21 */);
int lbm_upload_blocked(lbm_solver* s, const uint8_t* blocked, int row0, int nrows);
int lbm_upload_cells(lbm_solver* s, const float* cells, int row0, int nrows);
/* Device-side initial condition for grids too large to stage through the host: every cell
 * (ghost rows included) = equilibrium(rho * (1 + amp * r), u) with r in [-1, 1) a hash of
 * (seed, global row = (row_offset + y) mod global_rows, column); global_rows = 0 means no wrap.
 * amp = 0 reproduces latticeboltzmann.py:142. */
int lbm_init_equilibrium(lbm_solver* s, float rho, float u_row, float u_col, float amp,
                         uint64_t seed, int64_t row_offset, int64_t global_rows);
/* Device-side obstacle scene: seeded random discs (bench config 3); clears the mask if ndiscs=0. */
int lbm_init_blocked_discs(lbm_solver* s, int ndiscs, int radius, uint64_t seed, int64_t row_offset,
                           int64_t global_rows);

/* --- the hot path: replaces fused_collide_stream / fused_collide_stream_display
 * (lb_cuda_kernel.cu:371-385) and the buffer swap (latticeboltzmann.py:202).  Asynchronous. --- */
/* n_outer outer iterations of inner_timesteps updates each. */
int lbm_fused_collide_stream(lbm_solver* s, int n_outer);
/* Exactly n_updates lattice updates, independent of inner_timesteps, in the deepest launches the
 * tuning and (at slab seams) the valid ghost rows allow.  No reference counterpart: the slab
 * schedule needs update counts that are not multiples of INNER_TIMESTEPS. */
int lbm_advance(lbm_solver* s, int n_updates);
/* One outer iteration, then (if frame != NULL) render the new state into the caller's host
 * buffer uint8 (N, M, 3) as lb_cuda_kernel.cu:27-67,308-319 does; frame == NULL is the plain step,
 * as in latticeboltzmann.py:195-196.  The frame is complete after lbm_sync. */
int lbm_fused_collide_stream_display(lbm_solver* s, uint8_t* frame);
/* The same call for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.
 * lbm_fused_collide_stream_display hipMallocs a device staging frame on its first call; this
 * variant renders into a static device array inside the library instead (one copy of
 * LBM_NOALLOC_FRAME_MAX_BYTES per GPU, resolved in lbm_create, held until the process ends) and
 * calls no hipMalloc / hipFree / allocating routine.  Limits: N*M*3 <= LBM_NOALLOC_FRAME_MAX_BYTES
 * (otherwise LBM_ERR_UNSUPPORTED); `frame` must be page-locked host memory (lbm_host_alloc) or
 * device memory (a pageable one is rejected: LBM_ERR_INVALID); NULL = plain step.  Every check that
 * can fail runs BEFORE the step: an error return leaves the state untouched.  All solvers on one GPU share that GPU's array: calls whose frames are in flight at the
 * same time on one GPU (different solvers / streams, no lbm_sync between) race; calls on one
 * solver are ordered by its stream.  Results are identical to lbm_fused_collide_stream_display. */
#define LBM_NOALLOC_FRAME_MAX_BYTES (32u << 20)
int lbm_fused_collide_stream_display_noalloc(lbm_solver* s, uint8_t* frame);
/* Tuning of the INNER_TIMESTEPS > 1 path (replaces the reference's per-GPU presets,
 * latticeboltzmann.py:55-71): deepest temporal-blocking launch (1..6; inner_timesteps is split
 * into balanced launches no deeper than this; 1 = plain kernel only; 0 = automatic, the default:
 * 6, but 1 for grids below 2^17 cells, which are launch-latency-bound either way) and rows per workgroup
 * chunk (0 = chosen by the library per launch). */
int lbm_set_tuning(lbm_solver* s, int tb_max_depth, int tb_rows);
/* One update restricted to rows [row0, row0+nrows) (may extend into ghost rows), written into the
 * other cell buffer.  finish != 0 ends the update (buffer swap, step count, one ghost row
 * consumed); finish == 0 leaves it open so that further row ranges of the same update can follow:
 * the building block of the overlapped slab schedule (edge rows first, interior while the halos
 * travel).  While an update is open, the other stepping calls are rejected. */
int lbm_step_rows(lbm_solver* s, int row0, int nrows, int finish);
/* The same with `depth` (1..6) updates in one launch: rows [row0, row0+nrows) jump `depth` time
 * levels into the other buffer (LDS kernel for depth > 1); their inputs, rows
 * [row0-depth, row0+nrows+depth), must be owned rows, valid ghost rows or beyond a domain edge.
 * Rows beyond a DOMAIN edge are never computable (ghost rows exist at seams only).
 * All row ranges of one open update use the same depth; finish != 0 consumes `depth` ghost rows
 * and is refused unless the ranges of the update cover all N owned rows.
 * With it the last `depth` updates of a halo cycle overlap with the exchange as well. */
int lbm_advance_rows(lbm_solver* s, int row0, int nrows, int depth, int finish);
/* Same as lbm_fused_collide_stream, bracketed by HIP events on the solver's stream; blocks and
 * returns the elapsed device time.  For bench.py's roofline figure. */
int lbm_fused_collide_stream_timed(lbm_solver* s, int n_outer, float* elapsed_ms);
/* Replaces stream.synchronize() (latticeboltzmann.py:165,214-215). */
int lbm_sync(lbm_solver* s);
int lbm_steps_done(lbm_solver* s, int64_t* steps);

/* --- readback (new capability named by the north star: density / velocity for rendering; the
 * reference only reads RGB frames back, latticeboltzmann.py:191).  Synchronous.  Rows
 * [row0,row0+nrows) x columns [col0,col0+ncols); outputs are dense windows in the conventions
 * above.  Of rho and u one may be NULL. ------------------------------------------------------ */
int lbm_read_cells(lbm_solver* s, float* cells, int row0, int nrows, int col0, int ncols);
int lbm_read_moments(lbm_solver* s, float* rho, float* u, int row0, int nrows, int col0, int ncols);
int lbm_read_blocked(lbm_solver* s, uint8_t* blocked, int row0, int nrows);
/* Render rows [row0,row0+nrows) as RGB uint8 (nrows, M, 3), lb_cuda_kernel.cu:308-319. */
int lbm_render(lbm_solver* s, uint8_t* rgb, int row0, int nrows);
/* Page-locked host memory (replaces drv.pagelocked_empty for frame1 / frame2,
 * latticeboltzmann.py:158-159), so that the frame copy of lbm_fused_collide_stream_display is a
 * true asynchronous DMA and stepping continues under it. */
int lbm_host_alloc(size_t bytes, void** host_ptr);
int lbm_host_free(void* host_ptr);

/* --- y-slab halo exchange (no reference code exists, only README.md:12).  The
 * library exposes device pointers; the exchange itself (RCCL send/recv) belongs to the caller.
 * side 0 = top, 1 = bottom.  `send` = the first/last `halo` owned rows of the current state,
 * `recv` = the ghost rows beyond that edge; next != 0 addresses the buffer the open update is
 * writing (see lbm_step_rows) instead of the current one.  Each region is contiguous, *bytes long. */
int lbm_halo_region(lbm_solver* s, int side, int recv, int next, void** device_ptr, size_t* bytes);
int lbm_mask_halo_region(lbm_solver* s, int side, int recv, void** device_ptr, size_t* bytes);
/* Each update consumes one ghost row per seam (the deep halo is recomputed, not re-exchanged):
 * lbm_halo_valid reports how many updates remain before the next exchange is due, and
 * lbm_halo_mark_filled tells the solver that the caller has refilled all `halo` ghost rows. */
int lbm_halo_valid(lbm_solver* s, int* rows);
int lbm_halo_mark_filled(lbm_solver* s);

#ifdef __cplusplus
}
#endif
#endif /* LBM_HIP_H */
