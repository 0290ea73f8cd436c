/* TEST INFRASTRUCTURE (see oracle/__init__.py): plain C / OpenMP restatement of the
 * reference's projcg! call sequence, used (a) as a second checker next to the numpy
 * oracle and (b) as bench.py's `cpu_baseline` ("port": the reference is Julia-only and
 * cannot run on the GPU box).  It mirrors the reference ONE BLAS CALL AT A TIME -- no
 * fusion -- so it moves the same ~27 n-vector passes + 2 U passes per iteration the
 * reference does:
 *     src/projcg.jl:55-64 (setup), :71-112 (loop), :115-118 (multipliers)
 * with dgemv/ddot/dnrm2 restated as OpenMP loops (the reference threads only its BLAS
 * calls; here the broadcasts are threaded too, which can only make the baseline faster).
 * A is diagonal (the benchmark operator): Ad = adiag .* d.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int port_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void port_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* splitmix64-finaliser hash -> [-1,1)  (SURVEY §8d), identical to oracle/synth.py */
static inline double hash_u(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * 0x1.0p-52 - 1.0;
}

void port_hash_matrix(double* M, int64_t n, int64_t m, int64_t ld, uint64_t seed, int64_t row0, int64_t n_global, double scale) {
#pragma omp parallel for schedule(static) collapse(1)
    for (int64_t j = 0; j < m; ++j)
        for (int64_t i = 0; i < n; ++i) M[j * ld + i] = scale * hash_u(seed, (uint64_t)j * (uint64_t)n_global + (uint64_t)(row0 + i));
}

void port_hash_vector(double* v, int64_t n, uint64_t seed, int64_t offset, double scale, double shift) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) v[i] = scale * hash_u(seed, (uint64_t)(offset + i)) + shift;
}

/* t = M' v : row-blocked so a block of v stays in cache across the m columns (what a
 * tuned dgemv 'T' does); thread-private partials, fixed-order final sum */
void port_gemv_t(int64_t n, int64_t m, const double* M, int64_t ld, const double* v, double* t) {
    const int64_t RB = 2048;
    const int64_t nblk = (n + RB - 1) / RB;
    int nt = port_num_threads();
    double* part = (double*)calloc((size_t)nt * (size_t)(m > 0 ? m : 1), sizeof(double));
#pragma omp parallel
    {
#ifdef _OPENMP
        double* p = part + (size_t)omp_get_thread_num() * (size_t)m;
#else
        double* p = part;
#endif
#pragma omp for schedule(static)
        for (int64_t b = 0; b < nblk; ++b) {
            const int64_t i0 = b * RB, i1 = (i0 + RB < n) ? i0 + RB : n;
            for (int64_t j = 0; j < m; ++j) {
                const double* col = M + j * ld;
                double s = 0.0;
                for (int64_t i = i0; i < i1; ++i) s += col[i] * v[i];
                p[j] += s;
            }
        }
    }
    for (int64_t j = 0; j < m; ++j) {
        double s = 0.0;
        for (int k = 0; k < nt; ++k) s += part[(size_t)k * (size_t)m + j];
        t[j] = s;
    }
    free(part);
}

/* y = alpha M t + beta y */
void port_gemv_n(int64_t n, int64_t m, double alpha, const double* M, int64_t ld, const double* t, double beta, double* y) {
    const int64_t RB = 2048;
    const int64_t nblk = (n + RB - 1) / RB;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t i0 = b * RB, i1 = (i0 + RB < n) ? i0 + RB : n;
        double acc[2048];
        for (int64_t i = i0; i < i1; ++i) acc[i - i0] = 0.0;
        for (int64_t j = 0; j < m; ++j) {
            const double* col = M + j * ld;
            const double tj = t[j];
            for (int64_t i = i0; i < i1; ++i) acc[i - i0] += col[i] * tj;
        }
        if (beta == 0.0)
            for (int64_t i = i0; i < i1; ++i) y[i] = alpha * acc[i - i0];
        else
            for (int64_t i = i0; i < i1; ++i) y[i] = alpha * acc[i - i0] + beta * y[i];
    }
}

/* STREAM-like triad z = a x + y on the host cores: what this box's DRAM gives the CPU baseline (bench.py cpu_baseline.host_triad_GBs) */
void port_triad(int64_t n, double a, const double* x, const double* y, double* z) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) z[i] = a * x[i] + y[i];
}

double port_dot(int64_t n, const double* x, const double* y) {
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int64_t i = 0; i < n; ++i) s += x[i] * y[i];
    return s;
}

/* projcg!(x, lambda, A=diag(adiag), U, b, c; tol, maxit) -> iters (return), *nr_out.
 * work: 6 n-vectors r,g,d,rp,gp,Ad + Utr[m]  (ProjCGWork, src/projcg.jl:1-11) */
int64_t port_projcg(int64_t n, int64_t m, const double* adiag, const double* U, int64_t ld, const double* b, const double* c,
                    double tol, int64_t maxit, double* x, double* lambda, double* work, double* nr_out) {
    double *r = work, *g = work + n, *d = work + 2 * n, *rp = work + 3 * n, *gp = work + 4 * n, *Ad = work + 5 * n;
    double* Utr = work + 6 * n;
    int64_t i;
    port_gemv_n(n, m, 1.0, U, ld, c, 0.0, x);                          /* :55 */
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) r[i] = b[i];                                /* :56 */
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) r[i] = adiag[i] * x[i] - r[i];              /* :57 */
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) g[i] = r[i];                                /* :58 */
    port_gemv_t(n, m, U, ld, r, Utr);                                   /* :59 */
    port_gemv_n(n, m, -1.0, U, ld, Utr, 1.0, g);                        /* :60 */
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) r[i] = g[i];                                /* :61 */
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) d[i] = -1.0 * g[i];                         /* :62 */
    int64_t it = 0;
    double nr = INFINITY;
    const int64_t lim = maxit < n + m ? maxit : n + m;
    while (it < lim) {                                                  /* :71 */
        it += 1;
#pragma omp parallel for schedule(static)
        for (i = 0; i < n; ++i) Ad[i] = adiag[i] * d[i];                /* :74 */
        const double dAd = port_dot(n, d, Ad);                          /* :75 */
        if (dAd <= 0) {                                                 /* :77-82 */
            const double nd = sqrt(port_dot(n, d, d));
            for (i = 0; i < n; ++i) x[i] = d[i] / nd;
            for (i = 0; i < m; ++i) lambda[i] = NAN;
            *nr_out = INFINITY;
            return it;
        }
        const double rg = port_dot(n, r, g);                            /* :84 */
        if (rg <= 0) break;                                             /* :87 */
        const double alpha = rg / dAd;                                  /* :91 */
#pragma omp parallel for schedule(static)
        for (i = 0; i < n; ++i) x[i] += alpha * d[i];                   /* :92 */
#pragma omp parallel for schedule(static)
        for (i = 0; i < n; ++i) rp[i] = r[i] + alpha * Ad[i];           /* :93 */
#pragma omp parallel for schedule(static)
        for (i = 0; i < n; ++i) gp[i] = rp[i];                          /* :95 */
        port_gemv_t(n, m, U, ld, rp, Utr);                              /* :96 */
        port_gemv_n(n, m, -1.0, U, ld, Utr, 1.0, gp);                   /* :97 */
        const double beta = port_dot(n, rp, gp) / rg;                   /* :98 */
#pragma omp parallel for schedule(static)
        for (i = 0; i < n; ++i) d[i] = beta * d[i] - gp[i];             /* :99 */
#pragma omp parallel for schedule(static)
        for (i = 0; i < n; ++i) g[i] = gp[i];                           /* :100 */
#pragma omp parallel for schedule(static)
        for (i = 0; i < n; ++i) r[i] = gp[i];                           /* :101 */
        nr = sqrt(port_dot(n, g, g));                                   /* :103 */
        if (nr < tol) break;                                            /* :107 */
    }
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) r[i] = b[i];                                /* :115 */
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) r[i] = -1.0 * (adiag[i] * x[i]) + r[i];     /* :116 */
    port_gemv_t(n, m, U, ld, r, lambda);                                /* :118 */
    *nr_out = nr;
    return it;
}
