// Does the 16x16x32 bf16 matrix instruction buy wall time over 32x32x16 in the split-mode ring kernel's inner loop?
// (MI355X_MICROARCH.md, DVFS give-back item 7: ~1.12-1.15x on random data at equal cycles.)  Both variants: 512-thread
// workgroups (two waves per SIMD), one per CU, a 64 x 128 output tile per wave, every product block as six matrix instructions
// (the exact three-way split), all operands re-read from LDS each k-step (random bf16 data), no global traffic in the loop.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip && ./mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, float* out, int iters, unsigned long long* clk) {
    __shared__ uint4 lds[4096];      // 64 KB of operand planes
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        floatx16 acc[2][4];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; it++) {
            const int base = (it * 97) & 2047;
            uint4 b[4][3];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int p = 0; p < 3; p++) b[j][p] = lds[(base + (j * 3 + p) * 64 + lane) & 4095];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                uint4 a[3];
#pragma unroll
                for (int p = 0; p < 3; p++) a[p] = lds[(base + 1024 + (i * 3 + p) * 64 + lane) & 4095];
#pragma unroll
                for (int j = 0; j < 4; j++) {
#define MM(x, y) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[i][j], 0, 0, 0)
                    MM(a[2], b[j][0]); MM(a[1], b[j][1]); MM(a[0], b[j][2]); MM(a[1], b[j][0]); MM(a[0], b[j][1]); MM(a[0], b[j][0]);
#undef MM
                }
            }
        }
        for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    } else {
        floatx4 acc[4][8];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; it += 2) {      // one 32-k step = two 16-k tiles
            const int base = (it * 97) & 2047;
            uint4 a[4][3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int p = 0; p < 3; p++) a[i][p] = lds[(base + 1024 + (i * 3 + p) * 64 + lane) & 4095];
#pragma unroll
            for (int jh = 0; jh < 2; jh++) {
                uint4 b[4][3];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int p = 0; p < 3; p++) b[j][p] = lds[(base + ((jh * 4 + j) * 3 + p) * 64 + lane) & 4095];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
#define MM(x, y) acc[i][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[i][jh * 4 + j], 0, 0, 0)
                        MM(a[i][2], b[j][0]); MM(a[i][1], b[j][1]); MM(a[i][0], b[j][2]); MM(a[i][1], b[j][0]); MM(a[i][0], b[j][1]); MM(a[i][0], b[j][0]);
#undef MM
                    }
            }
        }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    const int n = 4096;
    uint4* h = (uint4*)malloc(n * 16);
    unsigned short* hs = (unsigned short*)h;
    srand(1);
    for (int i = 0; i < n * 8; i++) {      // random bf16 in [-1, 1): sign, exponent 0x3e..0x3f range, random mantissa
        const unsigned m = rand() & 0x7f, e = 0x7c + (rand() & 3), sgn = rand() & 1;
        hs[i] = (unsigned short)((sgn << 15) | (e << 7) | m);
    }
    uint4* d; float* o; unsigned long long* c;
    hipMalloc(&d, n * 16); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 16);
    hipMemcpy(d, h, n * 16, hipMemcpyHostToDevice);
    const int iters = 40000;
    for (int rep = 0; rep < 3; rep++)
        for (int shape : {32, 16}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, d, o, iters, c);
            else hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, d, o, iters, c);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
            const double flops = 256.0 * 8 * iters * 48 * (32.0 * 32 * 16 * 2);      // per 16-k tile: 48 instructions of 32x32x16 (or 96 of 16x16x32)
            printf("shape %2d: %.2f ms  %.0f TFLOP/s issued (%.0f of fp32 work)  in-kernel clock %.2f GHz  cycles per 16-k tile %.0f\n", shape, ms,
                   flops / ms / 1e9, flops / ms / 1e9 / 6, (double)hc[0] / (double)hc[1] * 0.1, (double)hc[0] / iters);
        }
    return 0;
}
