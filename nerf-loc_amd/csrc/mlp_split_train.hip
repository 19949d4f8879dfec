// Split-precision training forward (train precision NERF_PREC_BF16X3): the body of mlp_split_body.h with the tape on.
// Its own translation unit: each instantiation of the body takes half a minute to compile.
#include "mlp_split_body.h"

namespace nerf {
namespace sp {

// the training forward: the same evaluation, and every layer input the backward needs goes to the tape
template <int T, bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_split_train_kernel(MlpArgs a, TapePtrs tp) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    split_body<T, VD, OUT_CH, true>(a, tp, lds);
}

}  // namespace sp

int launch_mlp_split_train(const MlpArgs& a, const TapePtrs& tp, bool use_viewdirs, int output_ch, unsigned grid, hipStream_t st) {
    const dim3 g(grid), b(MLP_THREADS);
#ifdef NERF_EXP_ONLY_VD           // experiment builds only: a third of the compile time
    NERF_REQUIRE(use_viewdirs, "experiment build: view directions only");
    hipLaunchKernelGGL((sp::mlp_split_train_kernel<3, true, 4>), g, b, 0, st, a, tp);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
#endif
    if (use_viewdirs) hipLaunchKernelGGL((sp::mlp_split_train_kernel<3, true, 4>), g, b, 0, st, a, tp);
    else if (output_ch == 4) hipLaunchKernelGGL((sp::mlp_split_train_kernel<3, false, 4>), g, b, 0, st, a, tp);
    else hipLaunchKernelGGL((sp::mlp_split_train_kernel<3, false, 5>), g, b, 0, st, a, tp);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

}  // namespace nerf
