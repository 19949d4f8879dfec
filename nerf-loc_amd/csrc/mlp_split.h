// Host interface of the split-precision network kernels (mlp_split.hip): NERF_PREC_BF16X3 / NERF_PREC_BF16X2.
#pragma once
#include "mlp_device.h"
#include "tape.h"

namespace nerf {

int split_terms(int precision);                                   // 3, 2, or 0 for the other precisions
int split_padded_frags(int terms, bool use_viewdirs);             // stream length in 1 KiB fragments (stage-aligned)
size_t split_packed_bytes(int terms, bool use_viewdirs);          // stream + bias tiles
int split_samples_per_tile();                                     // samples a workgroup takes per pass
int launch_pack_split(const PackArgs& a, int terms, bool use_viewdirs, hipStream_t stream);
int launch_mlp_split(const MlpArgs& a, int terms, bool use_viewdirs, int output_ch, unsigned grid, hipStream_t stream);
int launch_mlp_split2(const MlpArgs& a, bool use_viewdirs, int output_ch, unsigned grid, hipStream_t stream);   // two terms
// the training forward in three terms: as launch_mlp_split, and the float32 tape of tape.h is written (M <= 2^22 samples)
int launch_mlp_split_train(const MlpArgs& a, const TapePtrs& tp, bool use_viewdirs, int output_ch, unsigned grid,
                           hipStream_t stream);

// the input-gradient chain in three terms (dgrad_split.hip): transposed stream, packer, launch.  `bits`: the ReLU bits of
// the forward tape; the gradient tape is written in the float32 layout (M <= 2^22 samples)
int split_bwd_padded_frags(bool use_viewdirs);
int launch_pack_split_backward(const float* const* tensors, bool use_viewdirs, int output_ch, void* packed_out, hipStream_t stream);
int launch_split_dgrad(const void* packed_bwd, bool use_viewdirs, int output_ch, const float* d_raw, long long M,
                       const unsigned* bits, const GTapePtrs& gt, unsigned grid, hipStream_t stream);

}  // namespace nerf
