// cg_launch.h -- entry points of the ring-kernel translation units, called by launch_conv_gemm (conv_gemm.hip).
#pragma once
#include "conv_gemm.h"

namespace odam_cg {

// plain layers, fp32 operands: mode 2 (split in registers), 3 (pre-split filters, 32x32x16), 4 (pre-split, 16x16x32);
// bn = tile columns 64 / 128 / 256; nth = 512, or 1024 for the sixteen-wave 512 x 64 tiles of mode 4
int launch_big_f32(int mode, int bn, int nth, const ConvGemmArgs& a, hipStream_t stream);
// plain layers, bf16 operands: bn 64 / 128 / 256, nth 512 or 1024 (bn 64 and 256)
int launch_big_bf16(int bn, int nth, const ConvGemmArgs& a, hipStream_t stream);
// bottleneck on the tile, fp32 split mode (ConvGemmArgs F_* / G_*)
int launch_big_fused(const ConvGemmArgs& a, hipStream_t stream);
// bottleneck on the tile, bf16 (ConvGemmArgs F_Wt / G_Wt)
int launch_fused_bf16(const ConvGemmArgs& a, hipStream_t stream);

#define g_s1_window odam_cfg::get(odam_cfg::CG_S1)      // bf16 3x3 stride 1: window main loop (0: the generic tap gather)

}  // namespace odam_cg
