// practical fp32 MFMA peak of the device: back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    floatx16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    for (int wg : {256, 512, 1024, 2048}) {
        int iters = 20000;
        hipLaunchKernelGGL(k, dim3(wg), dim3(256), 0, 0, d, 100, 0.5f, 0.25f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(wg), dim3(256), 0, 0, d, iters, 0.5f, 0.25f); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)wg * 4 /*waves*/ * iters * 4 * (32.0 * 32 * 2 * 2);
        printf("wgs=%d: %.2f ms, %.1f TFLOP/s\n", wg, ms, fl / ms / 1e9);
    }
    return 0;
}
