// v_mfma_f32_16x16x32_bf16: does a chain of DEPENDENT instructions (the six products of one accumulator block, as the split loop issues them)
// run slower than two chains interleaved?  1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o mfma16_dep mfma16_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    floatx4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    union { bf16x8 v; unsigned u[4]; } a, b;
    for (int i = 0; i < 4; i++) { a.u[i] = 0x3f803f80u + threadIdx.x * 7 + i; b.u[i] = 0x3f003f00u + threadIdx.x * 3 + i; }
    if (seed == 12345.f) a.u[0] = 0;
#define MM(x) x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, x, 0, 0, 0)
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // six dependent, then the next accumulator (the split loop's block order)
            MM(c[0]); MM(c[0]); MM(c[0]); MM(c[0]); MM(c[0]); MM(c[0]);
            MM(c[1]); MM(c[1]); MM(c[1]); MM(c[1]); MM(c[1]); MM(c[1]);
            MM(c[2]); MM(c[2]); MM(c[2]); MM(c[2]); MM(c[2]); MM(c[2]);
            MM(c[3]); MM(c[3]); MM(c[3]); MM(c[3]); MM(c[3]); MM(c[3]);
        } else if (MODE == 1) { // two chains interleaved
            MM(c[0]); MM(c[1]); MM(c[0]); MM(c[1]); MM(c[0]); MM(c[1]); MM(c[0]); MM(c[1]); MM(c[0]); MM(c[1]); MM(c[0]); MM(c[1]);
            MM(c[2]); MM(c[3]); MM(c[2]); MM(c[3]); MM(c[2]); MM(c[3]); MM(c[2]); MM(c[3]); MM(c[2]); MM(c[3]); MM(c[2]); MM(c[3]);
        } else {                // four chains interleaved
            for (int r = 0; r < 6; r++) { MM(c[0]); MM(c[1]); MM(c[2]); MM(c[3]); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}
template <int MODE> void run(int threads, const char* what) {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 100, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 24 * (threads / 64) / 4;      // instructions per SIMD
    printf("%-28s %d waves per SIMD: %.2f ns per instruction and SIMD  (%.1f TFLOP/s for the chip)\n", what, threads / 256, ms * 1e6 / n,
           (double)iters * 24 * (threads / 64) * 256 * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() {
    for (int t : {256, 512}) {
        if (t == 256) { run<0>(256, "six dependent, block order"); run<1>(256, "two chains interleaved"); run<2>(256, "four chains interleaved"); }
        else { run<0>(512, "six dependent, block order"); run<1>(512, "two chains interleaved"); run<2>(512, "four chains interleaved"); }
    }
    return 0;
}
