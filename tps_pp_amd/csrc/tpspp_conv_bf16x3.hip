// The "bf16x3" (three-term split) instantiations of the bf16 convolution kernel: see tpspp_conv_bf16.hip / _impl.h.
#include "tpspp_conv_bf16_impl.h"

namespace tpspp {

bool conv_bf16x3_launch(const BParams& P, int KH, int sh, int sw, hipStream_t st)
{
    if (KH == 1 && sh == 1 && sw == 1) return launch_by_shape<1, 1, 1, kKC1, true>(P, st);
    if (KH == 1 && sh == 2 && sw == 2) return launch_by_shape<1, 2, 2, kKC1, true>(P, st);
    if (KH == 3 && sh == 1 && sw == 1) return launch_by_shape<3, 1, 1, kKC3, true>(P, st);
    if (KH == 3 && sh == 2 && sw == 2) return launch_by_shape<3, 2, 2, kKC3, true>(P, st);
    if (KH == 3 && sh == 2 && sw == 1) return launch_by_shape<3, 2, 1, kKC3, true>(P, st);
    return false;
}

}  // namespace tpspp
