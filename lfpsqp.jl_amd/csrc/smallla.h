// Host-side dense linear algebra on m x m problems (m = number of constraints, 1..~1000):
// the replicated small factor of the tangent setup.  One-sided (Hestenes) Jacobi is used for
// both the symmetric eigenproblem of a Gram matrix and the SVD of the small factor; it is
// simple, accurate to high relative accuracy and needs no external LAPACK.
#pragma once
#include <math.h>
#include <sched.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <algorithm>
#include <numeric>
#include <vector>

namespace lfpsqp {

// host threads for the small dense kernels: the CPUs this process may run on (cgroup / affinity aware), at most 8
inline int small_threads() {
    static int n = 0;
    if (n == 0) {
        cpu_set_t set;
        int c = 1;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) c = CPU_COUNT(&set);
        n = c < 1 ? 1 : (c > 8 ? 8 : c);
    }
    return n;
}

// index of the calling thread inside a `#pragma omp parallel` region of these kernels (0 without OpenMP: the CPU test emulator's build)
inline int small_thread_id() {
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
}

// A: rows x cols, column-major, leading dimension rows (rows >= 1, cols >= 0).
// On exit U (rows x cols, orthonormal columns where S > 0), S (cols, descending, >= 0),
// V (cols x cols, orthogonal), with A = U diag(S) V'.
// One-sided (Hestenes) Jacobi in round-robin order: a sweep is cols-1 rounds of cols/2 DISJOINT column pairs, so the
// pairs of a round rotate in parallel (OpenMP on the host; the result does not depend on the number of threads).
// want_v = false: the rotations are not accumulated (V is returned empty) -- enough when only U and S are wanted.
inline void jacobi_svd(int rows, int cols, const std::vector<double>& Ain, std::vector<double>& U, std::vector<double>& S,
                       std::vector<double>& V, bool want_v = true) {
    std::vector<double> A(Ain);
    V.assign(want_v ? (size_t)cols * cols : 0, 0.0);
    for (int j = 0; want_v && j < cols; ++j) V[(size_t)j * cols + j] = 1.0;
    const double eps = 1e-16;
    const int np = cols + (cols & 1);                    // players of the tournament (one dummy if cols is odd)
    std::vector<int> seat(np);
    for (int k = 0; k < np; ++k) seat[k] = (k < cols) ? k : -1;
    [[maybe_unused]] const int nthreads = (cols >= 192 && rows >= 192) ? small_threads() : 1;   // below that a round is too short to share
    // Squared column norms are carried along (a rotation changes them by -+ t*gamma exactly) and recomputed at the start
    // of every sweep, so a pair costs one dot product instead of three.
    std::vector<double> sq(cols);
    for (int sweep = 0; sweep < 80 && cols > 1; ++sweep) {
        double off = 0.0;
#pragma omp parallel for if (nthreads > 1) num_threads(nthreads) schedule(static)
        for (int j = 0; j < cols; ++j) {
            const double* aj = &A[(size_t)j * rows];
            double s = 0.0;
#pragma omp simd reduction(+ : s)
            for (int i = 0; i < rows; ++i) s += aj[i] * aj[i];
            sq[j] = s;
        }
        for (int round = 0; round < np - 1; ++round) {
#pragma omp parallel for if (nthreads > 1) num_threads(nthreads) schedule(static) reduction(max : off)
            for (int k = 0; k < np / 2; ++k) {
                int p = seat[k], q = seat[np - 1 - k];
                if (p < 0 || q < 0) continue;
                if (p > q) std::swap(p, q);
                double* ap = &A[(size_t)p * rows];
                double* aq = &A[(size_t)q * rows];
                const double alpha = sq[p], beta = sq[q];
                double gamma = 0.0;
#pragma omp simd reduction(+ : gamma)
                for (int i = 0; i < rows; ++i) gamma += ap[i] * aq[i];
                if (gamma == 0.0) continue;
                const double lim = sqrt(alpha * beta);
                if (fabs(gamma) <= eps * lim) continue;
                off = std::max(off, fabs(gamma) / (lim > 0 ? lim : 1.0));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma omp simd
                for (int i = 0; i < rows; ++i) {
                    const double x = ap[i], y = aq[i];
                    ap[i] = c * x - s * y;
                    aq[i] = s * x + c * y;
                }
                sq[p] = alpha - t * gamma;
                sq[q] = beta + t * gamma;
                if (!want_v) continue;
                double* vp = &V[(size_t)p * cols];
                double* vq = &V[(size_t)q * cols];
#pragma omp simd
                for (int i = 0; i < cols; ++i) {
                    const double x = vp[i], y = vq[i];
                    vp[i] = c * x - s * y;
                    vq[i] = s * x + c * y;
                }
            }
            // next round: seat 0 stays, the others move one seat on
            const int last = seat[np - 1];
            for (int k = np - 1; k > 1; --k) seat[k] = seat[k - 1];
            if (np > 1) seat[1] = last;
        }
        if (off <= 1e-15) break;
    }
    std::vector<double> nrm(cols);
    for (int j = 0; j < cols; ++j) {
        double s = 0.0;
        for (int i = 0; i < rows; ++i) s += A[(size_t)j * rows + i] * A[(size_t)j * rows + i];
        nrm[j] = sqrt(s);
    }
    std::vector<int> idx(cols);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return nrm[a] > nrm[b]; });
    U.assign((size_t)rows * cols, 0.0);
    S.assign(cols, 0.0);
    std::vector<double> Vs(want_v ? (size_t)cols * cols : 0);
    for (int jj = 0; jj < cols; ++jj) {
        const int j = idx[jj];
        S[jj] = nrm[j];
        for (int i = 0; i < rows; ++i) U[(size_t)jj * rows + i] = nrm[j] > 0 ? A[(size_t)j * rows + i] / nrm[j] : 0.0;
        for (int i = 0; want_v && i < cols; ++i) Vs[(size_t)jj * cols + i] = V[(size_t)j * cols + i];
    }
    V.swap(Vs);
}

// Lower Cholesky factor of a symmetric positive definite matrix G (m x m, column-major): G = L L', L column-major with
// zeros above the diagonal.  Returns false when a pivot is not safely positive (G numerically singular): the caller
// then takes the rank-revealing route.  Left-looking by columns, so every inner loop runs down a contiguous column.
inline bool cholesky_lower(int m, const std::vector<double>& G, std::vector<double>& L, double rel_floor = 1e-10) {
    L.assign((size_t)m * m, 0.0);
    double dmax = 0.0;
    for (int j = 0; j < m; ++j) dmax = std::max(dmax, G[(size_t)j * m + j]);
    if (!(dmax > 0.0)) return false;
    const double floor_piv = rel_floor * dmax;             // default: cond(G) <~ 1e10, i.e. cond(A) <~ 1e5: far inside what G can resolve
    std::vector<double> col(m);
    for (int j = 0; j < m; ++j) {
        for (int i = j; i < m; ++i) col[i] = G[(size_t)j * m + i];
        for (int k = 0; k < j; ++k) {
            const double ljk = L[(size_t)k * m + j];
            const double* lk = &L[(size_t)k * m];
#pragma omp simd
            for (int i = j; i < m; ++i) col[i] -= lk[i] * ljk;
        }
        if (!(col[j] > floor_piv)) return false;
        const double d = sqrt(col[j]);
        double* lj = &L[(size_t)j * m];
        lj[j] = d;
        for (int i = j + 1; i < m; ++i) lj[i] = col[i] / d;
    }
    return true;
}

// C (ra x cb) = A (ra x ca) * B (ca x cb), all column-major, tight leading dimensions
inline void matmul(int ra, int ca, int cb, const std::vector<double>& A, const std::vector<double>& B, std::vector<double>& C) {
    C.assign((size_t)ra * cb, 0.0);
    for (int j = 0; j < cb; ++j)
        for (int k = 0; k < ca; ++k) {
            const double b = B[(size_t)j * ca + k];
            if (b == 0.0) continue;
            const double* a = &A[(size_t)k * ra];
            double* c = &C[(size_t)j * ra];
            for (int i = 0; i < ra; ++i) c[i] += a[i] * b;
        }
}

}  // namespace lfpsqp
