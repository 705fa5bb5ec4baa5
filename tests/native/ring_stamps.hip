// Phase timing of the ring contraction kernel (conv_gemm.hip, conv_gemm_big_kernel) on single layers of the detector, and a
// torch-free timing harness for it.  Built with CG_STAMP=1 the kernel accumulates s_memtime deltas per wave:
//   phase 0  s_waitcnt vmcnt + s_barrier          phase 1  fragment ds_reads + LDS-DMA issue (ends when the reads landed)
//   phase 2  first row tile: split + its MFMAs     phase 3  the other row tiles (+ loop bookkeeping)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCG_STAMP=1 -I odam_amd/csrc -o tests/native/ring_stamps tests/native/ring_stamps.hip
//   ./tests/native/ring_stamps [bf16]     (-DCG_STAMP=0: plain timing, no stamps execute)
#include "../../odam_amd/csrc/conv_gemm.hip"
#include "../../odam_amd/csrc/cg_big_f32.hip"
#include "../../odam_amd/csrc/cg_big_bf16.hip"
#include "../../odam_amd/csrc/cg_fused_f32.hip"
#include "../../odam_amd/csrc/cg_fused_bf16.hip"
#include "../../odam_amd/csrc/odam_common.hip"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

struct Shape { const char* name; int B, H, W, Cin, Cout, k, pad, res; };

static unsigned short to_bf16(float v) { unsigned u; memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

int main(int argc, char** argv) {
    const bool bf = argc > 1 && !strcmp(argv[1], "bf16");     // bf16 operands (MODE 0) instead of fp32 through the split
    if (argc > 1 && !strcmp(argv[1], "x16")) odam_cfg::set(odam_cfg::CG_MFMA16, 1);      // the 16x16x32 schedule (MODE 4)
    const Shape shapes[] = {{"l3.c2", 32, 50, 67, 256, 256, 3, 1}, {"l4.c2", 32, 25, 34, 512, 512, 3, 1},
                            {"l3.c1", 32, 50, 67, 1024, 256, 1, 0}, {"l4.c1", 32, 50, 67, 1024, 512, 1, 0},
                            {"l2.c2", 32, 100, 134, 128, 128, 3, 1}, {"l2.c1", 32, 100, 134, 512, 128, 1, 0},
                            {"l1.c2", 32, 200, 267, 64, 64, 3, 1}, {"l3.c3", 32, 50, 67, 256, 1024, 1, 0},
                            {"l3.c3+res", 32, 50, 67, 256, 1024, 1, 0, 1}, {"l4.c3+res", 32, 25, 34, 512, 2048, 1, 0, 1}, {"l2.ds", 32, 100, 134, 256, 512, 1, 0, 0}};
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (const Shape& s : shapes) {
        const int K = s.k * s.k * s.Cin, Kpad = (K + 63) / 64 * 64, M = s.B * s.H * s.W;
        std::vector<float> x((size_t)M * s.Cin), w((size_t)s.Cout * Kpad, 0.f);
        for (auto& v : x) v = nd(rng);
        for (int n = 0; n < s.Cout; n++)
            for (int k = 0; k < K; k++) w[(size_t)n * Kpad + k] = nd(rng) / sqrtf((float)K);
        std::vector<unsigned short> w3((size_t)s.Cout * Kpad * 3);
        odam_cg::split3_filters(w.data(), s.Cout, Kpad, w3.data());
        float *dx, *dw, *dy, *dres = nullptr; void* dw3;
        if (s.res) { hipMalloc(&dres, (size_t)M * s.Cout * 4); hipMemset(dres, 0x3c, (size_t)M * s.Cout * 4); }
        hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dy, (size_t)M * s.Cout * 4); hipMalloc(&dw3, w3.size() * 2);
        if (bf) {
            std::vector<unsigned short> xb(x.size()), wb(w.size());
            for (size_t i = 0; i < x.size(); i++) xb[i] = to_bf16(x[i]);
            for (size_t i = 0; i < w.size(); i++) wb[i] = to_bf16(w[i]);
            hipMemcpy(dx, xb.data(), xb.size() * 2, hipMemcpyHostToDevice);
            hipMemcpy(dw, wb.data(), wb.size() * 2, hipMemcpyHostToDevice);
        } else {
            hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        }
        hipMemcpy(dw3, w3.data(), w3.size() * 2, hipMemcpyHostToDevice);
        odam_cg::ConvGemmArgs a{};
        a.A = dx; a.Wt = dw; a.Wt3 = dw3; a.C = dy; a.res = dres; a.B = s.B; a.H = s.H; a.W = s.W; a.Cin = s.Cin;
        a.log2Cin = 31 - __builtin_clz(s.Cin); a.Ho = s.H; a.Wo = s.W; a.Cout = s.Cout; a.KH = a.KW = s.k; a.stride = 1; a.pad = s.pad;
        a.Kpad = Kpad; a.relu = 1; a.M = M; a.ldc = s.Cout; a.dtype = bf ? 1 : 0; a.k_order = (s.k > 1 && s.Cin % (bf ? 64 : 32) == 0) ? 1 : 0;
        if (bf) a.Wt3 = nullptr;
        for (int i = 0; i < 3; i++) odam_cg::launch_conv_gemm(a, 0);
        hipDeviceSynchronize();
#if CG_STAMP
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(odam_cg::g_stamps), z, sizeof(z));
#endif
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int n = 20;
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; i++) odam_cg::launch_conv_gemm(a, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / n, tf = 2.0 * M * s.Cout * K / us / 1e6;
        printf("%s M=%d N=%d K=%d: %8.1f us %7.1f TF", s.name, M, s.Cout, K, us, tf);
#if CG_STAMP
        hipMemcpyFromSymbol(z, HIP_SYMBOL(odam_cg::g_stamps), sizeof(z));
        const double tiles = (double)z[4];   // wave-k-tiles
        printf("  | per wave and k-tile: barrier %6.0f  vmcnt %6.0f  first half %6.0f  rest %6.0f  (sum %6.0f cycles; %llu waves)",
               z[0] / tiles, z[1] / tiles, z[2] / tiles, z[3] / tiles, (z[0] + z[1] + z[2] + z[3]) / tiles, z[5]);
        if (z[6] || z[7]) printf("  | per wave and tile: entry -> loop %6.0f  loop %7.0f  loop end -> last store issued %6.0f cycles", (double)z[6] / z[5],
                                 (double)(z[0] + z[1] + z[2] + z[3]) / z[5], (double)z[7] / z[5]);
#endif
        printf("\n");
        hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dw3); if (dres) hipFree(dres);
    }
    return 0;
}
