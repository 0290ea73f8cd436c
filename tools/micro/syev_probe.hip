// How fast is the vendor's symmetric eigensolver on the small replicated factor of the tangent setup (m x m Gram matrix,
// clustered spectrum like the BASELINE configs)?  rocsolver_dsyevd (divide & conquer) and rocsolver_dsyevj (Jacobi), m = 128,
// 256, 512, against the library's host Jacobi (FINDINGS.md 5.3).  hipcc -O3 --offload-arch=gfx950 syev_probe.hip -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    rocblas_handle h; rocblas_create_handle(&h);
    for (int m : {128, 256, 512}) {
        const int n = 8 * m;
        std::mt19937_64 rng(1); std::uniform_real_distribution<double> u(-1, 1);
        std::vector<double> A((size_t)n * m); for (auto& a : A) a = u(rng);
        std::vector<double> G((size_t)m * m, 0.0);
        for (int j = 0; j < m; ++j) for (int k = 0; k <= j; ++k) { double s = 0; for (int i = 0; i < n; ++i) s += A[(size_t)j*n+i]*A[(size_t)k*n+i]; G[(size_t)j*m+k] = G[(size_t)k*m+j] = s; }
        for (int j = 0; j < m; ++j) G[(size_t)j*m+j] += 20.0 * n / 3.0;
        double *dA, *dD, *dE, *dR; rocblas_int *dinfo, *dsw;
        hipMalloc(&dA, sizeof(double) * m * m); hipMalloc(&dD, sizeof(double) * m); hipMalloc(&dE, sizeof(double) * m);
        hipMalloc(&dR, sizeof(double)); hipMalloc(&dinfo, sizeof(rocblas_int)); hipMalloc(&dsw, sizeof(rocblas_int));
        std::vector<double> V((size_t)m * m), D(m);
        for (int algo = 0; algo < 2; ++algo) {
            double best = 1e9; rocblas_int info = -1;
            for (int rep = 0; rep < 4; ++rep) {
                const double t0 = now();
                hipMemcpy(dA, G.data(), sizeof(double) * m * m, hipMemcpyHostToDevice);
                if (algo == 0) rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, m, dA, m, dD, dE, dinfo);
                else rocsolver_dsyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_lower, m, dA, m, 1e-15, dR, 100, dsw, dD, dinfo);
                hipMemcpy(V.data(), dA, sizeof(double) * m * m, hipMemcpyDeviceToHost);
                hipMemcpy(D.data(), dD, sizeof(double) * m, hipMemcpyDeviceToHost);
                hipMemcpy(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost);
                if (rep) best = std::min(best, now() - t0);
            }
            double res = 0, orth = 0;
            for (int j = 0; j < m; ++j) {
                for (int i = 0; i < m; ++i) { double s = 0; for (int k = 0; k < m; ++k) s += G[(size_t)k*m+i] * V[(size_t)j*m+k]; res = std::max(res, fabs(s - D[j] * V[(size_t)j*m+i])); }
                for (int k = 0; k <= j; ++k) { double s = 0; for (int i = 0; i < m; ++i) s += V[(size_t)j*m+i] * V[(size_t)k*m+i]; orth = std::max(orth, fabs(s - (j == k))); }
            }
            printf("m=%d %s: %.2f ms (upload + solve + download)  info %d  resid/eig %.1e  orth %.1e\n", m, algo ? "dsyevj" : "dsyevd", best, (int)info, res / D[m - 1], orth);
        }
        hipFree(dA); hipFree(dD); hipFree(dE); hipFree(dR); hipFree(dinfo); hipFree(dsw);
    }
    return 0;
}
