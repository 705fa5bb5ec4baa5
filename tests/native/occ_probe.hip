#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(float* out, int ticks) {
    extern __shared__ float s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) out[blockIdx.x] = s[5];
}
int main() {
    float* out; hipMalloc(&out, 4096 * 4);
    for (int bytes : {65536, 80 * 1024, 81920 - 512, 81920, 81920 + 512}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(512), dim3(256), bytes, 0, out, 200000);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(512), dim3(256), bytes, 0, out, 200000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dynamic LDS %6d B: occupancy API %d blocks/CU; 512 blocks of 200k cycles took %.1f us\n", bytes, nb, ms * 1e3);
    }
}
