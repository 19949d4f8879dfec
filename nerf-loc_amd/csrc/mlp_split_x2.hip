// Split-precision inference in two terms (NERF_PREC_BF16X2): the body of mlp_split_body.h with T = 2.  Its own
// translation unit: each instantiation of the body takes half a minute to compile.
#include "mlp_split_body.h"

namespace nerf {

int launch_mlp_split2(const MlpArgs& a, bool use_viewdirs, int output_ch, unsigned grid, hipStream_t st) {
    const dim3 g(grid), b(MLP_THREADS);
    if (use_viewdirs) hipLaunchKernelGGL((sp::mlp_split_kernel<2, true, 4>), g, b, 0, st, a);
    else if (output_ch == 4) hipLaunchKernelGGL((sp::mlp_split_kernel<2, false, 4>), g, b, 0, st, a);
    else hipLaunchKernelGGL((sp::mlp_split_kernel<2, false, 5>), g, b, 0, st, a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

}  // namespace nerf
