// cg_big_f32.hip -- plain fp32 layers on the ring kernel (exact three-way bf16 split).  See cg_big.hpp.
#include "cg_big.hpp"

namespace odam_cg {

int launch_big_f32(int mode, int bn, int nth, const ConvGemmArgs& a, hipStream_t stream) {
    if (nth == 1024) {
        if (mode == 4 && bn == 64) return launch_big<4, 64, 4, 1024>(a, stream);
    } else if (bn == 256) {
        if (mode == 4) return launch_big<4, 256, 4>(a, stream);
        if (mode == 3) return launch_big<3, 256, 4>(a, stream);
        if (mode == 2) return launch_big<2, 256, 4>(a, stream);
    } else if (bn == 128) {
        if (mode == 4) return launch_big<4, 128, 4>(a, stream);
        if (mode == 3) return launch_big<3, 128, 4>(a, stream);
        if (mode == 2) return launch_big<2, 128, 4>(a, stream);
    } else if (bn == 64) {
        if (mode == 4) return launch_big<4, 64, 4>(a, stream);
        if (mode == 3) return launch_big<3, 64, 4>(a, stream);
        if (mode == 2) return launch_big<2, 64, 4>(a, stream);
    }
    return odam_fail(1, "conv_gemm: ring kernel (fp32): tile shape not built");
}

}  // namespace odam_cg
