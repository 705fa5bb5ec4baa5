// does the order of dependent v_mfma_f32_32x32x2_f32 matter?  1 / 2 / 4 waves per SIMD, two issue orders
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    floatx16 c0 = {0}, c1 = {0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f, c = a * b, d = a + b;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 4 back-to-back on one accumulator, then the other (what conv_gemm does)
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d, c, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d, c, c1, 0, 0, 0);
        } else {           // alternate accumulators
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d, c, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d, c, c1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float* d, int wg, const char* nm) {
    int iters = 10000;
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(256), 0, 0, d, 100, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(256), 0, 0, d, iters, 0.5f, 0.25f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)wg * 4 * iters * 8 * 4096.0;
    printf("%s wgs=%d (%d waves/SIMD): %.1f TFLOP/s\n", nm, wg, wg / 256, fl / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    for (int wg : {256, 512, 1024}) { run<0>(d, wg, "same-acc x4   "); run<1>(d, wg, "alternating   "); }
    return 0;
}
