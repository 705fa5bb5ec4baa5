// conv_gemm.h -- fp32 implicit-GEMM convolution / linear layer on the CDNA4 matrix cores.
//
//   C[m, n] = epilogue( sum_k A(m, k) * W[n, k] )
//     m = output pixel (b, oy, ox) of an NHWC tensor, n = output channel,
//     k = (ky * KW + kx) * Cin + ci  -- the input is gathered on the fly (no im2col buffer).
//   epilogue: * scale[n] + bias[n] (FrozenBatchNorm folded to scale/bias, or a Linear bias),
//             + residual[m, n], ReLU -- all optional.
//
// A Linear layer is the 1x1 case with H = 1, W = M.  One kernel therefore serves every
// contraction of the detector: ResNet stem/bottleneck convolutions, input_proj, the attention
// projections, the FFN and the prediction heads (reference: src/models/backbone.py:59-94 via
// torchvision ResNet, src/models/detr.py:45,70,73-78, src/models/transformer.py:132-238).
//
// Matrix core use: v_mfma_f32_32x32x2_f32 (exact fp32 multiply-accumulate, 157 TFLOP/s peak).
// Workgroup = 256 threads = 4 wavefronts; tile BM x BN x 32; A and W tiles staged through LDS
// with a 4-float row pad (conflict-free ds_read_b128), register-staged double buffering so the
// global loads of tile t+1 fly under the MFMAs of tile t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace odam_cg {

struct ConvGemmArgs {
    const float* A;      // NHWC input [B, H, W, Cin], Cin a power of two >= 4
    const float* Wt;     // [Cout][Kpad], k-major, zero padded to Kpad (multiple of 32)
    const float* scale;  // [Cout] or nullptr
    const float* bias;   // [Cout] or nullptr
    const float* res;    // [M, Cout] or nullptr
    float* C;            // [M, Cout]
    int B, H, W, Cin, log2Cin;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad;
    int Kpad;
    int relu;
    int M;
    int ldc;             // row stride of C / res in floats (>= Cout)
};

int launch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream);

}  // namespace odam_cg
