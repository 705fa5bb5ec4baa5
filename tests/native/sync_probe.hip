// sync_probe.hip -- how long after a kernel has finished does the host know?  (round 6: the association loop synchronises once per frame)
//   hipcc --offload-arch=gfx950 -O2 -o sync_probe tests/native/sync_probe.hip && ./sync_probe
// A kernel spins for ~T us and stamps the device wall clock when it ends (into pinned memory); the host (a) hipStreamSynchronize,
// (b) spins on a pinned word that hipStreamWriteValue32 sets behind the kernel, (c) spins on a pinned word the kernel's last thread stores
// with a system-scope release.  Reported: host time from launch to "knows" minus the same for the shortest method, and absolute per method.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin_kernel(unsigned long long ticks, volatile unsigned* flag, unsigned seq) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (flag && threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store((unsigned*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* flag; CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
    unsigned* dflag; CK(hipHostGetDevicePointer((void**)&dflag, flag, 0));
    *flag = 0;
    for (int T : {50, 400}) {
        std::vector<double> a, b, c;
        for (int rep = 0; rep < 60; rep++) {
            const unsigned seq = (unsigned)(rep * 4 + 1);
            double t0 = now_us();
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)T * 100, (volatile unsigned*)nullptr, 0u);
            CK(hipStreamSynchronize(st));
            a.push_back(now_us() - t0);
            t0 = now_us();
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)T * 100, (volatile unsigned*)nullptr, 0u);
            hipError_t e = hipStreamWriteValue32(st, dflag, seq + 1, 0);
            if (e != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); b.push_back(-1); CK(hipStreamSynchronize(st)); }
            else { while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq + 1) { } b.push_back(now_us() - t0); }
            t0 = now_us();
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)T * 100, (volatile unsigned*)dflag, seq + 2);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq + 2) { }
            c.push_back(now_us() - t0);
            CK(hipStreamSynchronize(st));
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("kernel of ~%d us: launch -> host knows, median of 60: hipStreamSynchronize %.1f us | hipStreamWriteValue32 + spin %.1f | kernel's own system-scope store + spin %.1f\n",
               T, med(a), med(b), med(c));
    }
    return 0;
}
