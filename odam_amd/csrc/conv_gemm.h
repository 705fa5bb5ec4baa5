// conv_gemm.h -- fp32 implicit-GEMM convolution / linear layer on the CDNA4 matrix cores.
//
//   C[m, n] = epilogue( sum_k A(m, k) * W[n, k] )
//     m = output pixel (b, oy, ox) of an NHWC tensor, n = output channel,
//     k = (ky * KW + kx) * Cin + ci  -- the input is gathered on the fly (no im2col buffer); see k_order.
//   epilogue: * scale[n] + bias[n] (FrozenBatchNorm folded to scale/bias, or a Linear bias),
//             + residual[m, n], ReLU -- all optional.
//
// A Linear layer is the 1x1 case with H = 1, W = M.  One kernel therefore serves every
// contraction of the detector: ResNet stem/bottleneck convolutions, input_proj, the attention
// projections, the FFN and the prediction heads (reference: src/models/backbone.py:59-94 via
// torchvision ResNet, src/models/detr.py:45,70,73-78, src/models/transformer.py:132-238).
//
// Matrix core use: v_mfma_f32_32x32x2_f32 (exact fp32 multiply-accumulate, 157 TFLOP/s peak); the bf16 mode
// (BASELINE config 4) runs the same tiling on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// Workgroup = 256 threads = 4 wavefronts; tile BM x BN x 32; A and W tiles staged through LDS
// with a 4-float row pad (conflict-free ds_read_b128), register-staged double buffering so the
// global loads of tile t+1 fly under the MFMAs of tile t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace odam_cg {

constexpr int ODAM_CG_F32 = 0;
constexpr int ODAM_CG_BF16 = 1;

struct ConvGemmArgs {
    const void* A;       // NHWC input [B, H, W, Cin] fp32 or bf16; Cin a power of two >= one 16-byte chunk (4 / 8)
    const void* Wt;      // [Cout][Kpad] same type, k-major, zero padded to a multiple of the k-tile (32 / 64)
    const void* Wt3 = nullptr;   // fp32 layers, optional: the same filters split exactly into three bf16 values per weight,
                         //    [Cout][Kpad / 16][3][16] (hi | mid | lo of every 16-k group; split3_filters); lets the split
                         //    contraction mode (set_f32_mode 2) skip splitting them in registers
    const float* scale;  // [Cout] fp32 or nullptr
    const float* bias;   // [Cout] fp32 or nullptr
    const void* res;     // [M, Cout] activation type, or nullptr
    void* C;             // [M, Cout] activation type (fp32 when out_f32)
    int B, H, W, Cin, log2Cin;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad;
    int Kpad;
    int relu;
    int M;
    int ldc;             // row stride of C / res in elements (>= Cout)
    int dtype;           // ODAM_CG_F32 / ODAM_CG_BF16: v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16, fp32 accumulate
    int out_f32;         // bf16 mode only: write C as fp32
    int lda;             // elements between consecutive input pixels (0 = Cin): reads a column block of a wider buffer
    int k_order;         // 0: k = (tap, ci).  1: k = (ci / kt, tap, ci % kt) with kt = the k-tile (32 fp32 / 64 bf16),
                         //    Cin % kt == 0: the taps of one channel chunk are consecutive k-tiles, so a workgroup re-reads
                         //    its input window from L2 while it is still there (tap-major revisits it 8+ k-tiles later)
    // Fused second layer (optional; conv_gemm_big_kernel with pre-split filters, Cout == 64 only): this layer's output
    // tile -- after scale / bias / ReLU -- is not stored but multiplied on chip by a 1x1 layer with F_N = 256 outputs,
    //   F_C[m, n] = act( (relu(C) F_W^T)[m, n] * F_scale[n] + F_bias[n] + F_res[m, n] ),
    // i.e. a ResNet bottleneck's 3x3 + expand + residual in one launch (detr_model.hip, layer1).  F_Wt3 = the expand
    // filters split by split3_filters(w, 256, 64, .); fp32 only; C is ignored.
    const void* F_Wt3 = nullptr;
    const float* F_scale = nullptr;
    const float* F_bias = nullptr;
    const float* F_res = nullptr;   // [M, F_ldc] or null
    float* F_C = nullptr;           // [M, F_ldc]
    int F_ldc = 0, F_relu = 0;
    // ... and a third: the NEXT bottleneck's 1x1 reduce (256 -> G_N channels, scale / bias / ReLU) applied to F_C's tile
    // while it is on chip, G_C[m, 0..G_N-1]; G_Wt3 = split3_filters(w, G_N, 256, .).  Optional; needs the F_* layer.
    const void* G_Wt3 = nullptr;
    const float* G_scale = nullptr;
    const float* G_bias = nullptr;
    float* G_C = nullptr;           // [M, G_N]
    int G_N = 64;                   // 64 (the next block of the same stage) or 128 (layer2's first reduce after layer1's last block)
    // bf16 mode (dtype ODAM_CG_BF16), the same two fusions with plain bf16 filters: F_Wt [F_ldc][Cout] (the 1x1 expand, K = this
    // layer's Cout = 64 / 128 / 256), G_Wt [G_N][F_ldc] (the next reduce).  F_res / F_C / G_C then point to bf16 tensors.
    const void* F_Wt = nullptr;
    const void* G_Wt = nullptr;
    int dil = 1;                    // dilation of the taps (both axes): tap (ky, kx) reads input pixel (oy stride - pad + ky dil, ox stride - pad + kx dil);
                                    // the window loop and the bottleneck-on-the-tile launches take dil = 1 only
    // Stem with the max-pool on the tile (conv_gemm_big_kernel, sixteen-wave 512 x 64 tiles only; launch_conv_gemm checks): pool = 1
    // makes a tile a 2-D patch of (2 pool_ph + 1) x (2 pool_pw + 1) <= 512 conv outputs -- the 3 x 3 / stride 2 / pad 1 windows of
    // pool_ph x pool_pw pooled pixels -- instead of 512 consecutive ones; after scale / bias / ReLU the patch is pooled in LDS and C
    // receives the POOLED tensor [B, Hp, Wp, Cout] (ldc = Cout).  Patches overlap by one conv row / column (recomputed).  The
    // ReLU makes zero padding equivalent to the -inf padding of max_pool2d.  M stays B Ho Wo (the conv's own size).
    // pool_ph x pool_pw is fixed at POOL_PH x POOL_PW below (compile-time divisors in the kernel).
    int pool = 0, pool_ph = 0, pool_pw = 0, Hp = 0, Wp = 0;
    int no_pin = 0;                 // 1: odam_config cg.pin does not apply to this call (its M does not depend on the shard a rank holds: the
                                    //    association network runs on the same rows on every rank, and small tiles are what its sizes want)
    int s1_window = 0;              // set by launch_conv_gemm (bf16 ring kernel, 3x3 stride 1): one LDS window per (channel slice, ky)
                                    //    serves the three horizontal taps (odam_config.h cg.s1)
};

constexpr int POOL_PH = 8, POOL_PW = 14;      // pooled pixels per patch: (2 * 8 + 1) x (2 * 14 + 1) = 493 conv outputs of a 512-row tile

int launch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream);

// experiments / tests: 0 = never use the bf16-native 256-row kernel, 1 = when the problem is large enough (default),
// 2 = whenever the layer is eligible (any size).  = odam_config.h cg.ring.
void set_big_mode(int mode);

// fp32 layers: 0 = v_mfma_f32_32x32x2_f32 on 128x128 tiles, 2 (default) = products on the bf16 matrix instruction through
// an exact three-way bf16 split of both operands (six MFMAs per 16 k; fp32-class accuracy, different last bits).
// = odam_config.h cg.f32.
void set_f32_mode(int mode);
int f32_mode();
// whether launch_conv_gemm would run `a` (with its F_* fields set) on the fused kernel right now
bool fused_second_ok(const ConvGemmArgs& a);    // the mode fp32 layers run in right now (0 when the ring kernel is switched off altogether)

// whether launch_conv_gemm would run `a` (pool fields set) with the max-pool on the tile (odam_config.h: stem.pool)
bool pooled_stem_ok(const ConvGemmArgs& a);

// bf16: whether launch_conv_gemm would run `a` (F_Wt [+ G_Wt] set) as one bottleneck launch (odam_config.h: cg.fuse_bf16)
bool fused_bf16_ok(const ConvGemmArgs& a);

// host: exact three-way bf16 split (truncation) of packed fp32 filters w[Cout][Kpad] (Kpad % 16 == 0) into the Wt3 layout
void split3_filters(const float* w, int Cout, int Kpad, unsigned short* out);

}  // namespace odam_cg
