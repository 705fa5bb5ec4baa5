// sk_wave_probe.hip -- checks the lane exchanges of odam_amd/csrc/sk_wave.h on the device: sk_xor<BIT> against l ^ (1 << BIT), sk_colsum<P>
// against plain sums.   hipcc --offload-arch=gfx950 -O3 -I odam_amd/csrc tests/native/sk_wave_probe.hip -o /tmp/sk_wave_probe && /tmp/sk_wave_probe
#include <cstdio>
#include <vector>
#include "sk_wave.h"
__global__ void k_xor(float* o) {
    const float v = (float)threadIdx.x;
    o[0 * 64 + threadIdx.x] = sk_xor<0>(v); o[1 * 64 + threadIdx.x] = sk_xor<1>(v); o[2 * 64 + threadIdx.x] = sk_xor<2>(v);
    o[3 * 64 + threadIdx.x] = sk_xor<3>(v); o[4 * 64 + threadIdx.x] = sk_xor<4>(v); o[5 * 64 + threadIdx.x] = sk_xor<5>(v);
}
template <int P> __global__ void k_col(const float* in, float* o) {      // in [64][32]
    float t[32];
    for (int c = 0; c < 32; c++) t[c] = c < P ? in[threadIdx.x * 32 + c] : 0.0f;
    o[threadIdx.x] = sk_colsum<P>(t, threadIdx.x);
}
int main() {
    float *d, *in; hipMalloc(&d, 6 * 64 * 4); hipMalloc(&in, 64 * 32 * 4);
    std::vector<float> h(6 * 64), hin(64 * 32);
    k_xor<<<1, 64>>>(d); hipMemcpy(h.data(), d, 6 * 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 6; b++) for (int l = 0; l < 64; l++) if ((int)h[b * 64 + l] != (l ^ (1 << b))) { if (bad < 12) printf("xor bit %d lane %d: got %d\n", b, l, (int)h[b * 64 + l]); bad++; }
    for (int i = 0; i < 64 * 32; i++) hin[i] = (float)((i * 7 + (i >> 5)) % 13);       // small integers: sums are exact
    hipMemcpy(in, hin.data(), 64 * 32 * 4, hipMemcpyHostToDevice);
    for (int P : {8, 16, 32}) {
        if (P == 8) k_col<8><<<1, 64>>>(in, d); else if (P == 16) k_col<16><<<1, 64>>>(in, d); else k_col<32><<<1, 64>>>(in, d);
        hipMemcpy(h.data(), d, 64 * 4, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) {
            const int c = sk_lane(l & 31) & (P - 1);
            float s = 0; for (int r = 0; r < 64; r++) s += hin[r * 32 + c];
            if (h[l] != s) { if (bad < 24) printf("colsum<%d> lane %d (column %d): got %g want %g\n", P, l, c, h[l], s); bad++; }
        }
    }
    printf(bad ? "FAILED: %d mismatches\n" : "sk_wave_probe: all exchanges and column sums correct\n", bad);
    return bad != 0;
}
