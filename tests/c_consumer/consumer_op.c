/* A plain C99 consumer of lfpsqp_projcg_op: the QP of consumer.c with a GENERAL operator
 *     A = tridiag(e, a, e),  a = 5 + 4u (seed 3),  e = 0.8 u (seed 15)
 * supplied as a callback whose body consists of the library's own queued primitives (vmul, ranged copies, axpby) -- what
 * the reference's LinearMap closure (src/optimize.jl:228-230) becomes for a host that keeps its vectors on the device.
 * Prints iteration count, residual, |x|, |lambda| and the number of operator applications.
 *   usage: consumer_op n m tol maxit */
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

typedef struct {
    lfpsqp_ctx* ctx;
    int64_t n;
    lfpsqp_vec *a, *e_up, *e_dn, *sh, *t;
    int calls;
} tri_op;

/* dest = a .* src + e_up .* shift_up(src) + e_dn .* shift_down(src); every call is queued on the context's stream */
static int tri_apply(void* user, const lfpsqp_vec* src, lfpsqp_vec* dest) {
    tri_op* o = (tri_op*)user;
    lfpsqp_ctx* ctx = o->ctx;
    const int64_t n = o->n;
    o->calls++;
    CK(lfpsqp_vmul(ctx, o->a, src, dest));
    CK(lfpsqp_vec_fill(ctx, o->sh, 0.0));
    CK(lfpsqp_vec_copy_range(ctx, o->sh, 0, src, 1, n - 1));
    CK(lfpsqp_vmul(ctx, o->e_up, o->sh, o->t));
    CK(lfpsqp_axpby(ctx, 1.0, o->t, 1.0, dest));
    CK(lfpsqp_vec_fill(ctx, o->sh, 0.0));
    CK(lfpsqp_vec_copy_range(ctx, o->sh, 1, src, 0, n - 1));
    CK(lfpsqp_vmul(ctx, o->e_dn, o->sh, o->t));
    CK(lfpsqp_axpby(ctx, 1.0, o->t, 1.0, dest));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int64_t n = atoll(argv[1]), m = atoll(argv[2]);
    const double tol = atof(argv[3]);
    const int64_t maxit = atoll(argv[4]);
    lfpsqp_ctx* ctx = NULL;
    CK(lfpsqp_ctx_create(0, &ctx));

    lfpsqp_mat *J = NULL, *Z = NULL;
    CK(lfpsqp_mat_alloc(ctx, n, m, &J));
    CK(lfpsqp_mat_alloc(ctx, n, m, &Z));
    CK(lfpsqp_mat_hash_fill(ctx, J, 1, 0, n, 1.0, n, m));
    double* Sigma = (double*)malloc(sizeof(double) * (size_t)m);
    double* Vt = (double*)malloc(sizeof(double) * (size_t)(m * m));
    int64_t rank = 0;
    CK(lfpsqp_factorize(ctx, J, NULL, Z, Sigma, Vt, NULL, &rank, 1e-10));

    tri_op op = {ctx, n, NULL, NULL, NULL, NULL, NULL, 0};
    lfpsqp_vec *b = NULL, *x = NULL, *lam = NULL, *Av = NULL, *e = NULL;
    lfpsqp_projcg_work w = {NULL, NULL, NULL, NULL};
    CK(lfpsqp_vec_alloc(ctx, n, &op.a));
    CK(lfpsqp_vec_alloc(ctx, n, &op.e_up));
    CK(lfpsqp_vec_alloc(ctx, n, &op.e_dn));
    CK(lfpsqp_vec_alloc(ctx, n, &op.sh));
    CK(lfpsqp_vec_alloc(ctx, n, &op.t));
    CK(lfpsqp_vec_alloc(ctx, n, &e));
    CK(lfpsqp_vec_alloc(ctx, n, &b));
    CK(lfpsqp_vec_alloc(ctx, n, &x));
    CK(lfpsqp_vec_alloc(ctx, n, &Av));
    CK(lfpsqp_vec_alloc(ctx, m, &lam));
    CK(lfpsqp_vec_alloc(ctx, n, &w.g));
    CK(lfpsqp_vec_alloc(ctx, n, &w.d));
    CK(lfpsqp_vec_alloc(ctx, n, &w.rp));
    CK(lfpsqp_vec_alloc(ctx, m, &w.Utr));
    CK(lfpsqp_vec_hash_fill(ctx, op.a, 3, 0, 4.0, 5.0));
    CK(lfpsqp_vec_hash_fill(ctx, e, 15, 0, 0.8, 0.0));          /* e[i], i < n-1, couples rows i and i+1 */
    CK(lfpsqp_vec_copy_range(ctx, op.e_up, 0, e, 0, n - 1));      /* e_up[i] = e[i], e_up[n-1] = 0 */
    CK(lfpsqp_vec_copy_range(ctx, op.e_dn, 1, e, 0, n - 1));      /* e_dn[i] = e[i-1], e_dn[0] = 0 */
    CK(lfpsqp_vec_hash_fill(ctx, b, 4, 0, 1.0, 0.0));

    const lfpsqp_basis U = {Z, rank, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
    int64_t iters = -1;
    double nr = -1.0, xn = 0.0, ln = 0.0;
    CK(lfpsqp_projcg_op(ctx, x, lam, tri_apply, &op, Av, &U, b, NULL, tol, maxit, n, LFPSQP_PROJCG_WANT_LAMBDA, &w, &iters, &nr));
    CK(lfpsqp_nrm2(ctx, x, &xn));
    CK(lfpsqp_nrm2(ctx, lam, &ln));
    printf("rank=%" PRId64 "\niters=%" PRId64 "\nnr=%a\nxnorm=%a\nlnorm=%a\ncalls=%d\n", rank, iters, nr, xn, ln, op.calls);
    double* xh = (double*)malloc(sizeof(double) * (size_t)n);
    CK(lfpsqp_vec_download(ctx, x, 0, xh, n));
    printf("x0=%a\nxlast=%a\n", xh[0], xh[n - 1]);
    free(xh);

    lfpsqp_vec* all[] = {op.a, op.e_up, op.e_dn, op.sh, op.t, e, b, x, Av, lam, w.g, w.d, w.rp, w.Utr};
    for (size_t i = 0; i < sizeof all / sizeof all[0]; ++i) CK(lfpsqp_vec_free(ctx, all[i]));
    CK(lfpsqp_mat_free(ctx, J));
    CK(lfpsqp_mat_free(ctx, Z));
    free(Sigma);
    free(Vt);
    CK(lfpsqp_ctx_destroy(ctx));
    return 0;
}
