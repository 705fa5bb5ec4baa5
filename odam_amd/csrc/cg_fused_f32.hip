// cg_fused_f32.hip -- layer1 / layer2 bottlenecks as one launch, fp32 split mode.  See cg_big.hpp, cg_tails_f32.hpp.
#include "cg_big.hpp"

namespace odam_cg {

template <int MODE>
static int launch_big_fused_as(const ConvGemmArgs& a, hipStream_t stream) {
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, 64, 4, 512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, 64, 4, 512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, 128, 4, 512, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, 64, 4, 512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess;
    }();
    if (!attr_ok) return odam_fail(2, "conv_gemm: cannot raise the dynamic LDS limit");
    const int tiles = (a.M + 255) / 256;
    if (a.Cout == 128) hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, 128, 4, 512, 3>), dim3(tiles), dim3(512), FUSE_LDS_BYTES, stream, a);
    else if (a.G_Wt3 && a.G_N == 128) hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, 64, 4, 512, 4>), dim3(tiles), dim3(512), FUSE_LDS_BYTES, stream, a);
    else if (a.G_Wt3) hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, 64, 4, 512, 2>), dim3(tiles), dim3(512), FUSE_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, 64, 4, 512, 1>), dim3(tiles), dim3(512), FUSE_LDS_BYTES, stream, a);
    ODAM_HIP(hipGetLastError());
    return 0;
}
int launch_big_fused(const ConvGemmArgs& a, hipStream_t stream) {
    const int m16 = odam_cfg::get(odam_cfg::CG_MFMA16);      // 2: the 128-column bottleneck launches too, 3: the 64-column ones as well
    const bool x16 = a.Kpad % 32 == 0 && (m16 >= 3 || (m16 == 2 && a.Cout == 128));
    return x16 ? launch_big_fused_as<4>(a, stream) : launch_big_fused_as<3>(a, stream);
}

}  // namespace odam_cg

#if CG_STAMP
extern "C" int odam_cg_tail_stamps(unsigned long long* out8, int reset) {      // diagnostic builds only (-DCG_STAMP=1)
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(odam_cg::g_tail_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 2;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(odam_cg::g_tail_stamps), z, sizeof(z)) != hipSuccess) return 2;
    }
    return 0;
}
#endif
