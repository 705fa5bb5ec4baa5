// cg_fused_bf16.hip -- bf16 bottlenecks as one launch (BASELINE config 4).  See cg_big.hpp, cg_tail_bf16.hpp.
#include "cg_big.hpp"

namespace odam_cg {

// bf16 bottleneck on the tile: (P, PN) combinations built -- P = 64: PN 0 / 64 / 128; P = 128: PN 0 / 128; P = 256: PN 0
template <int P, int FUSE>
static int launch_fused_bf16_as(const ConvGemmArgs& a, hipStream_t stream) {
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)conv_gemm_big_kernel<0, P, 4, 512, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess;
    }();
    if (!attr_ok) return odam_fail(2, "conv_gemm: cannot raise the dynamic LDS limit");
    ConvGemmArgs b = a;
    b.s1_window = g_s1_window;
    hipLaunchKernelGGL((conv_gemm_big_kernel<0, P, 4, 512, FUSE>), dim3((a.M + 255) / 256), dim3(512), FUSE_LDS_BYTES, stream, b);
    ODAM_HIP(hipGetLastError());
    return 0;
}
int launch_fused_bf16(const ConvGemmArgs& a, hipStream_t stream) {
    const int pn = a.G_Wt ? a.G_N : 0;
    if (a.Cout == 64 && pn == 0) return launch_fused_bf16_as<64, 5>(a, stream);
    if (a.Cout == 64 && pn == 64) return launch_fused_bf16_as<64, 6>(a, stream);
    if (a.Cout == 64 && pn == 128) return launch_fused_bf16_as<64, 7>(a, stream);
    if (a.Cout == 128 && pn == 0) return launch_fused_bf16_as<128, 5>(a, stream);
    if (a.Cout == 128 && pn == 128) return launch_fused_bf16_as<128, 7>(a, stream);
    if (a.Cout == 256 && pn == 0) return launch_fused_bf16_as<256, 5>(a, stream);
    return odam_fail(1, "conv_gemm: fused bf16 bottleneck: channel combination not built");
}

}  // namespace odam_cg
