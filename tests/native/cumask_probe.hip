// cumask_probe.hip -- hipExtStreamCreateWithCUMask on MI355X: which XCDs / CUs the workgroups of a launch on a masked stream reach, for
// a few masks (mask bit i = ?).   hipcc --offload-arch=gfx950 -O3 tests/native/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
#include <hip/hip_runtime.h>
__global__ void k(unsigned* o, int spin) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));       // XCC_ID
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        // HW_ID: cu_id [11:8], sh [12], se_id [15:13] on gfx9
    if (threadIdx.x == 0) { o[2 * blockIdx.x] = xcc & 15; o[2 * blockIdx.x + 1] = hw; }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(20);
}
static void run(const char* name, const std::vector<uint32_t>& mask, unsigned* d) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int grid = 1024;
    (void)hipMemsetAsync(d, 0xff, 8 * grid, s);
    k<<<grid, 256, 0, s>>>(d, 200);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * grid);
    (void)hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
    unsigned per_xcd[16] = {0};
    std::set<unsigned> cus;
    for (int b = 0; b < grid; b++) { per_xcd[h[2 * b] & 15]++; cus.insert(((h[2 * b] & 15) << 16) | ((h[2 * b + 1] >> 8) & 0xff)); }
    printf("%-34s workgroups per XCD:", name);
    for (int x = 0; x < 8; x++) printf(" %4u", per_xcd[x]);
    printf("   distinct (xcd, se/sh/cu): %zu\n", cus.size());
    (void)hipStreamDestroy(s);
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 8 * 1024);
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("all 256 bits", all, d);
    std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; run("bits 0..31", m, d);
    m.assign(8, 0); m[7] = 0xffffffffu; run("bits 224..255", m, d);
    m.assign(8, 0); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); run("every 8th bit (0, 8, 16, ...)", m, d);
    m.assign(8, 0); for (int i = 1; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); run("every 8th bit from 1", m, d);
    m.assign(8, 0xffffffffu); for (int i = 0; i < 256; i += 8) m[i / 32] &= ~(1u << (i % 32)); run("all but every 8th bit", m, d);
    m.assign(8, 0); m[0] = 0xff; run("bits 0..7", m, d);
}
