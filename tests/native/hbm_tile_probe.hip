// hbm_tile_probe.hip -- what an epilogue-shaped access pattern gets from HBM: out[m][n] = max(in[m][n] + res[m][n], 0) over [M][N] fp32,
// walked (a) as one linear stream, (b) tile by tile as the 256 x BN tiles of the contraction kernel's epilogue do (a workgroup touches
// BN * 4 bytes of 256 consecutive rows: BN = 256 -> 1 KB pieces 4 KB apart for N = 1024), one 512-thread workgroup per CU-sized tile.
//   hipcc --offload-arch=gfx950 -O3 -o tests/native/hbm_tile_probe tests/native/hbm_tile_probe.hip && ./tests/native/hbm_tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void linear_kernel(const float4* a, const float4* r, float4* o, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 512) {
        const float4 x = a[i], y = r[i];
        o[i] = float4{fmaxf(x.x + y.x, 0.f), fmaxf(x.y + y.y, 0.f), fmaxf(x.z + y.z, 0.f), fmaxf(x.w + y.w, 0.f)};
    }
}
// tile (tm, tn): rows 256 tm .. + 255, columns BN tn .. + BN - 1; a wave takes 32 rows, 16 bytes per lane
template <int BN>
__global__ __launch_bounds__(512) void tile_kernel(const float* a, const float* r, float* o, int M, int N) {
    const int tiles_n = N / BN, tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int LPR = BN / 4 > 64 ? 64 : BN / 4;      // lanes per row
    constexpr int RPI = 64 / LPR;                        // rows per wave instruction
    constexpr int CPL = BN / 4 / LPR;                    // 16-byte pieces per lane and row
    for (int it = 0; it < 32 / RPI; it++) {
        const int m = tm * 256 + wave * 32 + it * RPI + lane / LPR;
        if (m >= M) continue;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const size_t off = (size_t)m * N + tn * BN + ((lane % LPR) + c * LPR) * 4;
            const float4 x = *reinterpret_cast<const float4*>(a + off), y = *reinterpret_cast<const float4*>(r + off);
            *reinterpret_cast<float4*>(o + off) = float4{fmaxf(x.x + y.x, 0.f), fmaxf(x.y + y.y, 0.f), fmaxf(x.z + y.z, 0.f), fmaxf(x.w + y.w, 0.f)};
        }
    }
}

int main() {
    const int M = 107200, N = 1024;
    const size_t n = (size_t)M * N;
    float *a, *r, *o;
    hipMalloc(&a, n * 4); hipMalloc(&r, n * 4); hipMalloc(&o, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(r, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto&& f) {
        for (int i = 0; i < 3; i++) f();
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.1f us  %5.2f TB/s\n", name, ms * 100, 3.0 * n * 4 / (ms / 10 * 1e-3) / 1e12);
    };
    time("linear stream (2 reads + 1 write)", [&] { hipLaunchKernelGGL(linear_kernel, dim3(2048), dim3(512), 0, 0, (const float4*)a, (const float4*)r, (float4*)o, n / 4); });
    const int tm = (M + 255) / 256;
    time("tiles 256 x 128 (512-byte pieces)", [&] { hipLaunchKernelGGL(tile_kernel<128>, dim3(tm * (N / 128)), dim3(512), 0, 0, a, r, o, M, N); });
    time("tiles 256 x 256 (1 KB pieces)", [&] { hipLaunchKernelGGL(tile_kernel<256>, dim3(tm * (N / 256)), dim3(512), 0, 0, a, r, o, M, N); });
    time("tiles 256 x 512 (2 KB pieces)", [&] { hipLaunchKernelGGL(tile_kernel<512>, dim3(tm * (N / 512)), dim3(512), 0, 0, a, r, o, M, N); });
    time("tiles 256 x 1024 (whole rows)", [&] { hipLaunchKernelGGL(tile_kernel<1024>, dim3(tm), dim3(512), 0, 0, a, r, o, M, N); });
    return 0;
}
