// prints what v_permlane32_swap / v_permlane16_swap do to (a = lane, b = 100 + lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned* d; unsigned h[256];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
    for (int q = 0; q < 4; ++q) { printf("%s:", names[q]); for (int i = 0; i < 64; i += 1) printf(" %u", h[q * 64 + i]); printf("\n"); }
    return 0;
}
