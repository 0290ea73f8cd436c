/* A plain C99 consumer of include/lfpsqp_hip.h -- what a non-Python host (the reference's `ccall`, a C driver)
 * sees of the boundary: opaque handles, plain pointers and sizes, int status codes.  Solves the benchmark QP
 *     min 1/2 x'Ax - b'x  s.t.  U'x = 0,   A = diag(5 + 4u), U = orthonormalised hash matrix
 * (tangent setup through lfpsqp_factorize, then lfpsqp_projcg) and prints iteration count, residual, |x| and |lambda|
 * as hex floats; tests/test_c_consumer.py compares them with the same calls made through the Python host layer.
 *   usage: consumer n m tol maxit */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "lfpsqp_hip.h"

#define CK(call)                                                                            \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != 0) {                                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? lfpsqp_last_error(ctx) : ""); \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int64_t n = atoll(argv[1]), m = atoll(argv[2]);
    const double tol = atof(argv[3]);
    const int64_t maxit = atoll(argv[4]);
    lfpsqp_ctx* ctx = NULL;
    CK(lfpsqp_ctx_create(0, &ctx));
    char name[128];
    CK(lfpsqp_device_name(ctx, name, (int64_t)sizeof name));

    lfpsqp_mat *J = NULL, *Z = NULL;
    CK(lfpsqp_mat_alloc(ctx, n, m, &J));
    CK(lfpsqp_mat_alloc(ctx, n, m, &Z));
    CK(lfpsqp_mat_hash_fill(ctx, J, 1, 0, n, 1.0, n, m));
    double* Sigma = (double*)malloc(sizeof(double) * (size_t)m);
    double* Vt = (double*)malloc(sizeof(double) * (size_t)(m * m));
    int64_t rank = 0;
    CK(lfpsqp_factorize(ctx, J, NULL, Z, Sigma, Vt, NULL, &rank, 1e-10));

    lfpsqp_vec *a = NULL, *b = NULL, *x = NULL, *lam = NULL;
    lfpsqp_projcg_work w = {NULL, NULL, NULL, NULL};
    CK(lfpsqp_vec_alloc(ctx, n, &a));
    CK(lfpsqp_vec_alloc(ctx, n, &b));
    CK(lfpsqp_vec_alloc(ctx, n, &x));
    CK(lfpsqp_vec_alloc(ctx, m, &lam));
    CK(lfpsqp_vec_alloc(ctx, n, &w.g));
    CK(lfpsqp_vec_alloc(ctx, n, &w.d));
    CK(lfpsqp_vec_alloc(ctx, n, &w.rp));
    CK(lfpsqp_vec_alloc(ctx, m, &w.Utr));
    CK(lfpsqp_vec_hash_fill(ctx, a, 3, 0, 4.0, 5.0));
    CK(lfpsqp_vec_hash_fill(ctx, b, 4, 0, 1.0, 0.0));

    const lfpsqp_diag_op A = {0.0, a};
    const lfpsqp_basis U = {Z, rank, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
    int64_t iters = -1;
    double nr = -1.0, xn = 0.0, ln = 0.0;
    CK(lfpsqp_projcg(ctx, x, lam, &A, &U, b, NULL, tol, maxit, n, LFPSQP_PROJCG_WANT_LAMBDA, &w, &iters, &nr));
    CK(lfpsqp_nrm2(ctx, x, &xn));
    CK(lfpsqp_nrm2(ctx, lam, &ln));
    printf("device=%s\nrank=%" PRId64 "\niters=%" PRId64 "\nnr=%a\nxnorm=%a\nlnorm=%a\nsigma0=%a\n", name, rank, iters, nr, xn, ln, Sigma[0]);

    CK(lfpsqp_vec_free(ctx, a));
    CK(lfpsqp_vec_free(ctx, b));
    CK(lfpsqp_vec_free(ctx, x));
    CK(lfpsqp_vec_free(ctx, lam));
    CK(lfpsqp_vec_free(ctx, w.g));
    CK(lfpsqp_vec_free(ctx, w.d));
    CK(lfpsqp_vec_free(ctx, w.rp));
    CK(lfpsqp_vec_free(ctx, w.Utr));
    CK(lfpsqp_mat_free(ctx, J));
    CK(lfpsqp_mat_free(ctx, Z));
    free(Sigma);
    free(Vt);
    CK(lfpsqp_ctx_destroy(ctx));
    return 0;
}
