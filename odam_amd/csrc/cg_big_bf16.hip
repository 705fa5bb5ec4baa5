// cg_big_bf16.hip -- plain bf16 layers on the ring kernel (BASELINE config 4).  See cg_big.hpp.
#include "cg_big.hpp"

namespace odam_cg {

int launch_big_bf16(int bn, int nth, const ConvGemmArgs& a, hipStream_t stream) {
    if (nth == 1024) {
        if (bn == 256) return launch_big<0, 256, 4, 1024>(a, stream);
        if (bn == 64) return launch_big<0, 64, 4, 1024>(a, stream);
    } else {
        if (bn == 256) return launch_big<0, 256, 4>(a, stream);
        if (bn == 128) return launch_big<0, 128, 4>(a, stream);
        if (bn == 64) return launch_big<0, 64, 4>(a, stream);
    }
    return odam_fail(1, "conv_gemm: ring kernel (bf16): tile shape not built");
}

}  // namespace odam_cg
