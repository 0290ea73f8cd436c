// vecprobe: what does it take for a plain elementwise fp64 kernel (copy: 1 read + 1 write; triad: 2 reads + 1 write) to reach the
// streaming rate of the box?  Variants: 16-byte pieces in flight per lane (U = 1, 2, 4, 8: all loads of a lane's U tiles issued before
// the first store), non-temporal loads / stores, grid size.  3.2 GB vectors (far beyond the 256 MB infinity cache).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vecprobe tools/micro/vecprobe.hip && /tmp/vecprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NTL> __device__ __forceinline__ double2 ldv(const double* p) {
    if (NTL) return make_double2(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1));
    return *reinterpret_cast<const double2*>(p);
}
template <bool NTS> __device__ __forceinline__ void stv(double* p, double2 v) {
    if (NTS) { __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1); }
    else *reinterpret_cast<double2*>(p) = v;
}

// tile = 512 rows (256 threads x one 16-byte piece); a block handles tiles blockIdx.x + k * gridDim.x, U of them at a time
template <int OP, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void vk(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ z, int64_t ntiles, double a, double b) {
    for (int64_t t0 = blockIdx.x; t0 < ntiles; t0 += (int64_t)U * gridDim.x) {
        double2 xx[U], yy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = t0 + (int64_t)u * gridDim.x;
            if (t < ntiles) {
                const int64_t i = t * 512 + threadIdx.x * 2;
                xx[u] = ldv<NTL>(x + i);
                if (OP == 1) yy[u] = ldv<NTL>(y + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = t0 + (int64_t)u * gridDim.x;
            if (t < ntiles) {
                const int64_t i = t * 512 + threadIdx.x * 2;
                double2 o = xx[u];
                if (OP == 1) o = make_double2(fma(a, xx[u].x, b * yy[u].x), fma(a, xx[u].y, b * yy[u].y));
                stv<NTS>(z + i, o);
            }
        }
    }
}
// the same with CONSECUTIVE tiles per block (a block owns a contiguous span)
template <int OP, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void vk_span(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ z, int64_t ntiles, double a, double b) {
    const int64_t q = ntiles / gridDim.x, rem = ntiles % gridDim.x;
    const int64_t s0 = (int64_t)blockIdx.x * q + ((int64_t)blockIdx.x < rem ? blockIdx.x : rem);
    const int64_t cnt = q + ((int64_t)blockIdx.x < rem ? 1 : 0);
    for (int64_t k = 0; k < cnt; k += U) {
        double2 xx[U], yy[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u < cnt) {
                const int64_t i = (s0 + k + u) * 512 + threadIdx.x * 2;
                xx[u] = ldv<NTL>(x + i);
                if (OP == 1) yy[u] = ldv<NTL>(y + i);
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u < cnt) {
                const int64_t i = (s0 + k + u) * 512 + threadIdx.x * 2;
                double2 o = xx[u];
                if (OP == 1) o = make_double2(fma(a, xx[u].x, b * yy[u].x), fma(a, xx[u].y, b * yy[u].y));
                stv<NTS>(z + i, o);
            }
    }
}

// block b handles the C consecutive tiles [b*C, (b+1)*C): grid = ceil(ntiles / C) >> resident slots, dispatched in order
template <int OP, int C, bool UNR, int BT>
__global__ __launch_bounds__(BT) void vk_chunk(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ z, int64_t n2, double a, double b) {
    // n2 = number of 16-byte pieces; a tile = BT pieces
    const int64_t p0 = (int64_t)blockIdx.x * C * BT + threadIdx.x;
    if (UNR) {
        double2 xx[C], yy[C];
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int64_t p = p0 + (int64_t)u * BT;
            if (p < n2) { xx[u] = ldv<false>(x + 2 * p); if (OP == 1) yy[u] = ldv<false>(y + 2 * p); }
        }
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int64_t p = p0 + (int64_t)u * BT;
            if (p < n2) {
                double2 o = xx[u];
                if (OP == 1) o = make_double2(fma(a, xx[u].x, b * yy[u].x), fma(a, xx[u].y, b * yy[u].y));
                stv<false>(z + 2 * p, o);
            }
        }
    } else {
#pragma unroll 1
        for (int u = 0; u < C; ++u) {
            const int64_t p = p0 + (int64_t)u * BT;
            if (p < n2) {
                double2 o = ldv<false>(x + 2 * p);
                if (OP == 1) { const double2 yv = ldv<false>(y + 2 * p); o = make_double2(fma(a, o.x, b * yv.x), fma(a, o.y, b * yv.y)); }
                stv<false>(z + 2 * p, o);
            }
        }
    }
}

template <class K>
static double time_ms(K launch, int reps = 5) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int64_t n = 400000000;        // 3.2 GB per vector
    const int64_t ntiles = n / 512;
    double *x, *y, *z;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&z, n * 8));
    CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8)); CK(hipMemset(z, 0, n * 8));
    CK(hipDeviceSynchronize());
    auto rep = [&](const char* name, int streams, double ms) { printf("%-64s %8.4f ms  %7.1f GB/s\n", name, ms, streams * 8.0 * n / ms / 1e6); fflush(stdout); };
    rep("hipMemcpyAsync D2D", 2, time_ms([&] { hipMemcpyAsync(z, x, n * 8, hipMemcpyDeviceToDevice, 0); }));
#define RUN(OP, U, NTL, NTS, GRID)                                                                                              \
    { char nm[128]; snprintf(nm, sizeof nm, "%s U=%d ntl=%d nts=%d grid=%d", OP ? "triad" : "copy ", U, NTL, NTS, GRID);       \
      rep(nm, OP ? 3 : 2, time_ms([&] { hipLaunchKernelGGL((vk<OP, U, NTL, NTS>), dim3(GRID), dim3(256), 0, 0, x, y, z, ntiles, 1.0, 2.0); })); }
#define RUNS(OP, U, NTL, NTS, GRID)                                                                                             \
    { char nm[128]; snprintf(nm, sizeof nm, "%s U=%d ntl=%d nts=%d grid=%d SPAN", OP ? "triad" : "copy ", U, NTL, NTS, GRID);  \
      rep(nm, OP ? 3 : 2, time_ms([&] { hipLaunchKernelGGL((vk_span<OP, U, NTL, NTS>), dim3(GRID), dim3(256), 0, 0, x, y, z, ntiles, 1.0, 2.0); })); }
#define RUNC(OP, C, UNR, BT, N)                                                                                                 \
    { char nm[128]; snprintf(nm, sizeof nm, "%s chunk C=%d unrolled=%d block=%d n=%lld", OP ? "triad" : "copy ", C, UNR, BT, (long long)(N)); \
      const int64_t n2 = (N) / 2; const int64_t g = (n2 + (int64_t)C * BT - 1) / ((int64_t)C * BT);                              \
      const double ms = time_ms([&] { hipLaunchKernelGGL((vk_chunk<OP, C, UNR, BT>), dim3((unsigned)g), dim3(BT), 0, 0, x, y, z, n2, 1.0, 2.0); }, 10); \
      printf("%-64s %8.4f ms  %7.1f GB/s\n", nm, ms, (OP ? 3 : 2) * 8.0 * (N) / ms / 1e6); fflush(stdout); }
    for (int pass = 0; pass < 2; ++pass) {
        printf("---- pass %d\n", pass);
        RUN(0, 1, false, false, 2048) RUN(0, 1, false, false, 781250) RUN(0, 1, false, false, 32768) RUN(0, 1, false, false, 131072)
        RUN(1, 1, false, false, 2048) RUN(1, 1, false, false, 781250) RUN(1, 1, false, false, 32768) RUN(1, 1, false, false, 131072)
        RUNC(0, 1, false, 256, n) RUNC(0, 2, false, 256, n) RUNC(0, 4, false, 256, n) RUNC(0, 8, false, 256, n) RUNC(0, 16, false, 256, n) RUNC(0, 64, false, 256, n)
        RUNC(0, 2, true, 256, n) RUNC(0, 4, true, 256, n) RUNC(0, 8, true, 256, n)
        RUNC(0, 1, false, 512, n) RUNC(0, 1, false, 1024, n) RUNC(0, 2, true, 1024, n) RUNC(0, 1, false, 64, n) RUNC(0, 1, false, 128, n)
        RUNC(1, 1, false, 256, n) RUNC(1, 2, false, 256, n) RUNC(1, 4, false, 256, n) RUNC(1, 8, false, 256, n) RUNC(1, 16, false, 256, n) RUNC(1, 64, false, 256, n)
        RUNC(1, 2, true, 256, n) RUNC(1, 4, true, 256, n) RUNC(1, 8, true, 256, n)
        RUNC(1, 1, false, 512, n) RUNC(1, 1, false, 1024, n) RUNC(1, 1, false, 128, n)
        // the solver's size: n = 1e7 (80 MB per vector: partly served by the 256 MB infinity cache)
        const int64_t ns = 10000000;
        RUNC(0, 1, false, 256, ns) RUNC(0, 4, false, 256, ns) RUNC(0, 8, false, 256, ns) RUNC(0, 4, true, 256, ns)
        RUNC(1, 1, false, 256, ns) RUNC(1, 4, false, 256, ns) RUNC(1, 8, false, 256, ns) RUNC(1, 4, true, 256, ns)
        { const int64_t nt = ns / 512; char nm[64];
          snprintf(nm, sizeof nm, "copy  grid-stride 2048 n=1e7"); const double m0 = time_ms([&] { hipLaunchKernelGGL((vk<0, 1, false, false>), dim3(2048), dim3(256), 0, 0, x, y, z, nt, 1.0, 2.0); }, 10);
          printf("%-64s %8.4f ms  %7.1f GB/s\n", nm, m0, 2 * 8.0 * ns / m0 / 1e6);
          snprintf(nm, sizeof nm, "triad grid-stride 2048 n=1e7"); const double m1 = time_ms([&] { hipLaunchKernelGGL((vk<1, 1, false, false>), dim3(2048), dim3(256), 0, 0, x, y, z, nt, 1.0, 2.0); }, 10);
          printf("%-64s %8.4f ms  %7.1f GB/s\n", nm, m1, 3 * 8.0 * ns / m1 / 1e6); }
    }
    return 0;
}
