/* A plain C99 host of liblbm_hip.so: the drop-in boundary used the way a compiled caller (or a cgo /
 * JNI / N-API shim) would use it - no Python, no torch, no C++.  Replaces what
 * /root/reference/latticeboltzmann.py:125-159,186-206 does through pycuda.
 *   abi_host <N> <M> <omega> <steps> <boundary> <in: surroundings+blocked+cells> <out: cells+rho+u>
 * Reads float32 surroundings[9], uint8 blocked[N*M], float32 cells[N*M*9]; runs `steps` outer
 * iterations; writes float32 cells[N*M*9], rho[N*M], u[N*M*2].  Exit status = first lbm_status. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lbm_hip.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != LBM_OK) {                                               \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lbm_last_error()); \
            return rc_;                                                    \
        }                                                                  \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 8) return 64;
    const int N = atoi(argv[1]), M = atoi(argv[2]), steps = atoi(argv[4]);
    const double omega = atof(argv[3]);
    const size_t cells_n = (size_t)N * M * 9;
    float surr[9];
    uint8_t* blocked = malloc((size_t)N * M);
    float* cells = malloc(cells_n * sizeof(float));
    float* rho = malloc((size_t)N * M * sizeof(float));
    float* u = malloc((size_t)N * M * 2 * sizeof(float));
    FILE* f = fopen(argv[6], "rb");
    if (!f || !blocked || !cells || !rho || !u) return 65;
    if (fread(surr, sizeof(float), 9, f) != 9 || fread(blocked, 1, (size_t)N * M, f) != (size_t)N * M ||
        fread(cells, sizeof(float), cells_n, f) != cells_n)
        return 66;
    fclose(f);

    if (lbm_abi_version() != LBM_ABI_VERSION) return 67;
    lbm_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.abi_version = LBM_ABI_VERSION;
    cfg.device = 0;
    cfg.N = N;
    cfg.M = M;
    cfg.omega = omega;
    cfg.inner_timesteps = 1;
    cfg.boundary = strcmp(argv[5], "periodic") ? LBM_BOUNDARY_REFERENCE : LBM_BOUNDARY_PERIODIC;
    cfg.storage = LBM_STORE_F32;
    lbm_solver* s = NULL;
    CHECK(lbm_create(&cfg, &s));
    CHECK(lbm_set_surroundings(s, surr));
    CHECK(lbm_upload_blocked(s, blocked, 0, N));
    CHECK(lbm_upload_cells(s, cells, 0, N));
    CHECK(lbm_fused_collide_stream(s, steps));
    CHECK(lbm_sync(s));
    int64_t done = 0;
    CHECK(lbm_steps_done(s, &done));
    if (done != steps) return 68;
    CHECK(lbm_read_cells(s, cells, 0, N, 0, M));
    CHECK(lbm_read_moments(s, rho, u, 0, N, 0, M));
    /* error behaviour: a bad argument is a status and a message, never an abort */
    if (lbm_read_cells(s, cells, 0, N + 1, 0, M) != LBM_ERR_INVALID || !strstr(lbm_last_error(), "rows")) return 69;
    CHECK(lbm_destroy(s));

    f = fopen(argv[7], "wb");
    if (!f) return 70;
    fwrite(cells, sizeof(float), cells_n, f);
    fwrite(rho, sizeof(float), (size_t)N * M, f);
    fwrite(u, sizeof(float), (size_t)N * M * 2, f);
    fclose(f);
    printf("abi_host ok: %d x %d, %d steps\n", N, M, steps);
    return 0;
}
