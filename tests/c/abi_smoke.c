/* The C-ABI used from plain C (no Python, no C++): one Swift-Hohenberg gradient through smo_create / smo_forward / smo_adjoint /
 * smo_inner, the device-vector calls, and the error path.  Built by tests/test_capi.py with `gcc -std=c99 -Wall -Werror` (which also
 * proves that include/smo.h is valid C) and run on the GPU by tests/test_sh23_gpu.py, which compares the printed numbers with the Python
 * binding's.   usage: abi_smoke [npts] [n_iters] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "smo.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != SMO_OK) {                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, smo_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(int argc, char** argv) {
    const int npts = argc > 1 ? atoi(argv[1]) : 64, n_iters = argc > 2 ? atoi(argv[2]) : 40;
    const double pi = 3.14159265358979323846;
    smo_config cfg = {0};
    cfg.kind = SMO_SH23; cfg.npts = npts; cfg.x0 = 0.0; cfg.x1 = 12.0 * pi; cfg.dt = 0.1; cfg.n_iters = n_iters; cfg.param = -0.3;
    cfg.cost = 0; cfg.batch = 1; cfg.device = 0; cfg.rank = 0; cfg.world = 1; cfg.ckpt = 1;
    smo_ctx* ctx = NULL;
    CHECK(smo_create(&cfg, &ctx));
    size_t n = 0;
    CHECK(smo_vec_len(ctx, &n));
    if (smo_ncomp(ctx) != 1 || n != (size_t)(2 * npts)) { fprintf(stderr, "unexpected geometry\n"); return 1; }
    double* x = malloc(n * sizeof(double));
    double* g = malloc(n * sizeof(double));
    for (size_t i = 0; i < n; ++i) x[i] = 0.3 * sin(2.0 * pi * 3.0 * (double)i / (double)n) + 0.1 * cos(2.0 * pi * 5.0 * (double)i / (double)n);
    const double* X[1] = {x};
    double* G[1] = {g};
    double J = 0.0, ip = 0.0, ipd = 0.0;
    if (smo_adjoint(ctx, X, SMO_ADJ_DISCRETE, G) != SMO_ERR_STATE) { fprintf(stderr, "adjoint before forward must be refused\n"); return 1; }
    CHECK(smo_forward(ctx, X, &J));
    CHECK(smo_adjoint(ctx, NULL, SMO_ADJ_DISCRETE, G));
    CHECK(smo_inner(ctx, x, g, &ip));
    /* the same inner product on device-resident vectors */
    double *dx = NULL, *dg = NULL;
    CHECK(smo_vec_alloc(0, n, &dx));
    CHECK(smo_vec_alloc(0, n, &dg));
    CHECK(smo_vec_upload(0, dx, x, n));
    CHECK(smo_vec_upload(0, dg, g, n));
    CHECK(smo_inner_dev(ctx, dx, dg, &ipd));
    CHECK(smo_vec_axpby(0, n, 2.0, dx, -1.0, dg, dx));            /* dx <- 2 x - g */
    CHECK(smo_vec_download(0, dx, g, n));                          /* g now holds 2 x - (old g) */
    CHECK(smo_vec_free(0, dx));
    CHECK(smo_vec_free(0, dg));
    double gn = 0.0;
    for (size_t i = 0; i < n; ++i) gn += g[i] * g[i];
    printf("J %.17g\ninner %.17g\ninner_dev %.17g\ncombo_norm2 %.17g\n", J, ip, ipd, gn);
    smo_destroy(ctx);
    free(x); free(g);
    return (ip == ipd) ? 0 : 2;
}
