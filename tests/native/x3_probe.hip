// Accuracy probe: fp32 GEMM tile on (a) v_mfma_f32_32x32x2_f32 and (b) six v_mfma_f32_32x32x16_bf16 products of an exact
// three-way bf16 split (a = hi + mid + lo by truncation; terms with index sum <= 2), against float64 on the host.
//   hipcc --offload-arch=gfx950 -O3 -o x3_probe x3_probe.hip && ./x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float a, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned ua = __builtin_bit_cast(unsigned, a);
    const float fh = __builtin_bit_cast(float, ua & 0xffff0000u);
    const float r1 = a - fh;
    const unsigned u1 = __builtin_bit_cast(unsigned, r1);
    const float fm = __builtin_bit_cast(float, u1 & 0xffff0000u);
    const float r2 = r1 - fm;
    h = (unsigned short)(ua >> 16); m = (unsigned short)(u1 >> 16); l = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
}

// C[32][32] = A[32][K] * B[32][K]^T, one wavefront
__global__ void probe(const float* A, const float* B, int K, float* C32, float* C3, int terms) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    floatx16 c1, c2;
    for (int i = 0; i < 16; i++) { c1[i] = 0.f; c2[i] = 0.f; }
    for (int k = 0; k < K; k += 2)
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[r * K + k + h], c1, 0, 0, 0);
    for (int k = 0; k < K; k += 16) {
        u16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; j++) {
            unsigned short x, y, z;
            split3(A[r * K + k + 8 * h + j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;
            split3(B[r * K + k + 8 * h + j], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z;
        }
#define MM(x, y) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c2, 0, 0, 0)
        // smallest terms first
        if (terms >= 6) { MM(al, bh); MM(am, bm); MM(ah, bl); }
        if (terms >= 3) { MM(am, bh); MM(ah, bm); }
        MM(ah, bh);
    }
    for (int i = 0; i < 16; i++) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C32[row * 32 + r] = c1[i]; C3[row * 32 + r] = c2[i];
    }
}

int main() {
    for (int K : {64, 256, 1024, 4608}) {
        for (int mode = 0; mode < 2; mode++) {       // 0: uniform [-1,1); 1: positive (ReLU-like activations, no cancellation)
            std::vector<float> A(32 * K), B(32 * K);
            srand(K + mode);
            for (auto& x : A) x = mode ? (float)rand() / RAND_MAX : 2.f * rand() / RAND_MAX - 1.f;
            for (auto& x : B) x = 2.f * rand() / RAND_MAX - 1.f;
            float *dA, *dB, *d1, *d3;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d1, 4096); hipMalloc(&d3, 4096);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            for (int terms : {6, 3, 1}) {
                probe<<<1, 64>>>(dA, dB, K, d1, d3, terms);
                std::vector<float> c1(1024), c3(1024);
                hipMemcpy(c1.data(), d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(c3.data(), d3, 4096, hipMemcpyDeviceToHost);
                double e1 = 0, e3 = 0, d13 = 0, scale = 0;
                for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; k++) { ref += (double)A[i * K + k] * B[j * K + k]; mag += std::fabs((double)A[i * K + k] * B[j * K + k]); }
                    e1 = std::fmax(e1, std::fabs(c1[i * 32 + j] - ref) / mag); e3 = std::fmax(e3, std::fabs(c3[i * 32 + j] - ref) / mag);
                    d13 = std::fmax(d13, std::fabs((double)c1[i * 32 + j] - c3[i * 32 + j]) / mag);
                }
                printf("K=%5d data=%d terms=%d: max |err| / sum|a b|:  fp32 MFMA %.2e   bf16 split %.2e   (between them %.2e)\n", K, mode, terms, e1, e3, d13);
            }
            hipFree(dA); hipFree(dB); hipFree(d1); hipFree(d3);
        }
    }
    return 0;
}
