// xcd_probe.hip -- which XCD (s_getreg XCC_ID) and CU a workgroup of a plain launch lands on, by blockIdx: is it blockIdx % 8?
// hipcc --offload-arch=gfx950 -O3 tests/native/xcd_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
__global__ void k(unsigned* o, int spin) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));       // HW_REG_XCC_ID [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        // HW_REG_HW_ID
    if (threadIdx.x == 0) { o[2 * blockIdx.x] = xcc; o[2 * blockIdx.x + 1] = hw; }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(10);
}
__global__ void busy(float* p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
int main() {
    unsigned* d; (void)hipMalloc(&d, 8 * 4096); float* f; (void)hipMalloc(&f, 4096);
    hipStream_t s2; (void)hipStreamCreate(&s2);
    for (int trial = 0; trial < 6; trial++) {
        const int grid = trial < 3 ? 256 : 512, nt = 256;
        if (trial % 3 == 2) busy<<<1000, 256, 0, s2>>>(f, 200000);      // a foreign kernel keeps the dispatcher busy
        k<<<grid, nt>>>(d, 50);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(2 * grid);
        (void)hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
        int bad = 0; unsigned cnt[16] = {0};
        for (int b = 0; b < grid; b++) { cnt[h[2 * b] & 15]++; if ((h[2 * b] & 15) != (unsigned)(b % 8)) bad++; }
        printf("trial %d grid %d%s: blocks not on XCD blockIdx %% 8: %d; per XCD:", trial, grid, trial % 3 == 2 ? " (foreign kernel running)" : "", bad);
        for (int x = 0; x < 8; x++) printf(" %u", cnt[x]);
        printf("   first 16:"); for (int b = 0; b < 16; b++) printf(" %u", h[2 * b] & 15);
        printf("\n");
    }
}
