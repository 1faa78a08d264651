// Launcher declarations shared by lbm_kernels.hip (device code) and lbm_api.hip (C-ABI host code).
#pragma once
#include "lbm_device.h"

namespace lbm {

hipError_t launch_step(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                       bool half, int row0, int nrows, hipStream_t st);
constexpr int TB_MAX_T = 6;      // deepest launch the LDS kernel is instantiated for
constexpr int TB_DEFAULT_T = 6;  // deepest launch used unless lbm_set_tuning says otherwise
constexpr long long TB_MIN_CELLS = 1 << 17;  // smaller grids: plain kernel (automatic tuning only)
hipError_t launch_step_tb(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                          bool half, int T, int row0, int nrows, int Ry, hipStream_t st);
hipError_t launch_collide(void* f, const uint64_t* mask, const Geom& geo, int xmode, bool half,
                          int row0, int nrows, hipStream_t st);
hipError_t launch_observe(const void* g, const uint64_t* mask, const Geom& geo, int xmode, bool half,
                          bool pulled, int row0, int nrows, int col0, int ncols, float* cells,
                          float* rho, float* u, uint8_t* rgb, hipStream_t st);
hipError_t launch_import_cells(void* f, const Geom& geo, bool half, const float* cells, int row0,
                               int nrows, hipStream_t st);
hipError_t launch_import_mask(uint64_t* mask, const Geom& geo, const uint8_t* blocked, int row0,
                              int nrows, hipStream_t st);
hipError_t launch_export_mask(const uint64_t* mask, const Geom& geo, uint8_t* blocked, int row0,
                              int nrows, hipStream_t st);
hipError_t launch_init_equilibrium(void* f, const Geom& geo, bool half, float rho, float u0, float u1,
                                   float amp, uint64_t seed, long long row_offset, long long wrap_rows,
                                   hipStream_t st);
hipError_t launch_init_discs(uint64_t* mask, const Geom& geo, int ndiscs, int radius, uint64_t seed,
                             long long row_offset, long long global_rows, hipStream_t st);

// D2Q21 (lbm_d2q21.hip): 21 slots per chunk, periodic or reference edges, fp32, one update per launch
constexpr int Q21 = 21;
hipError_t launch_q21_step(const void* gin, void* gout, const uint64_t* mask, const Geom& geo, int xmode,
                           const float* surr21, hipStream_t st);
hipError_t launch_q21_collide(void* f, const uint64_t* mask, const Geom& geo, hipStream_t st);
hipError_t launch_q21_observe(const void* g, const uint64_t* mask, const Geom& geo, int xmode, const float* surr21,
                              bool pulled, int row0, int nrows, int col0, int ncols, float* cells, float* rho,
                              float* u, hipStream_t st);
hipError_t launch_q21_import_cells(void* f, const Geom& geo, const float* cells, int row0, int nrows, hipStream_t st);
hipError_t launch_q21_init_equilibrium(void* f, const Geom& geo, float rho, float u0, float u1, float amp,
                                       uint64_t seed, hipStream_t st);

}  // namespace lbm
