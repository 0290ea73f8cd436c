// Does the streaming rate of a 10 GB matrix depend on how/where it was allocated?  hipMalloc vs the virtual-memory API
// (hipMemCreate + hipMemMap), several allocations each (some kept alive to move the next one), same probe kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(const double* __restrict__ M, int64_t ld, int m, double* out) {
    constexpr int CW = 4, RW = 16, kStep = 64, kTiles = 32, UNR = 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & (RW - 1), h = lane / RW;
    const double* base = M + (int64_t)blockIdx.x * 2048 + wave * RW + r + (int64_t)h * ld;
    double acc = 0.0;
    for (int k = 0; k < kTiles; ++k) {
        const double* p = base + (int64_t)k * kStep;
        for (int c = 0; c < m / CW; c += UNR) {
            double v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(p + (int64_t)(c + u) * CW * ld);
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc += v[u];
        }
    }
    if (acc == 123.456) out[0] = acc;
}
static double rate(const double* M, int64_t ld, int64_t n, int m, double* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<<<(unsigned)(n / 2048), 256>>>(M, ld, m, out);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<<<(unsigned)(n / 2048), 256>>>(M, ld, m, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return 8.0 * n * m / (ms / 5 * 1e-3) / 1e9;
}
int main() {
    const int64_t n = 10000384, ld = n; const int m = 128; const size_t bytes = sizeof(double) * ld * m;
    double* out; CK(hipMalloc(&out, 64));
    std::vector<void*> keep;
    for (int i = 0; i < 6; ++i) {
        double* M; CK(hipMalloc(&M, bytes)); CK(hipMemset(M, 0, bytes));
        printf("hipMalloc #%d  %p : %.0f GB/s\n", i, (void*)M, rate(M, ld, n, m, out));
        if (i % 2 == 0) keep.push_back(M); else CK(hipFree(M));
    }
    for (void* p : keep) CK(hipFree(p));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t sz = (bytes + gran - 1) / gran * gran;
    printf("VMM granularity %zu\n", gran);
    for (int i = 0; i < 4; ++i) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, sz, &prop, 0));
        void* va; CK(hipMemAddressReserve(&va, sz, 0, nullptr, 0)); CK(hipMemMap(va, sz, 0, h, 0));
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess(va, sz, &acc, 1));
        CK(hipMemset(va, 0, bytes));
        printf("VMM #%d  %p : %.0f GB/s\n", i, va, rate((double*)va, ld, n, m, out));
        if (i % 2 == 1) { CK(hipMemUnmap(va, sz)); CK(hipMemAddressFree(va, sz)); CK(hipMemRelease(h)); }
    }
    return 0;
}
