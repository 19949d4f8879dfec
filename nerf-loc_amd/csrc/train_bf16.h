// bf16 training: tape layout shared by the forward (mlp.hip), the input-gradient chain and the weight-gradient
// kernels (train_bf16.hip).
//
// A tape holds MFMA operands exactly as the layer chain has them in registers.  The unit is a "piece": one
// k-step operand (16 features) of one wave (64 samples) = [column tile ct 2][lane 64] x 16 bytes = 2 KiB, lane
// (h, c) holding features 16 s + 8 (j >> 2) + 4 h + (j & 3), j = 0..7, of sample 256 T + 64 w + 32 ct + c
// (T = workgroup tile, w = wave).  All pieces of one (T, w) are contiguous, in the order given below, so every
// store and every load of a piece is one fully coalesced 1 KiB access per column tile.
#pragma once
#include "mlp_device.h"

namespace nerf {
namespace tape16 {

constexpr int PIECE_BYTES = 2048;
// forward tape: the outputs of pts_linears.0-4 (16 pieces each, scaled as the forward keeps them), the encoded points
// (4 pieces, kernel k-slot order), pts_linears.5-7, feature_linear's output (16), the encoded view (2),
// views_linears.0's output (8).  The order makes the input of every layer one contiguous run of pieces:
// pts_linears.5 reads [h4, points] (20 pieces), views_linears.0 reads [feature, view] (18).
constexpr int PE = 80;
constexpr int h(bool, int i) { return i <= 4 ? 16 * i : 84 + 16 * (i - 5); }
constexpr int FEAT = 132;
constexpr int VW = 148;
constexpr int HV = 150;
// ReLU masks, one piece per layer with a ReLU (pts_linears.0-7, then views_linears.0): [lane 64][tile 8] dwords, bit
// 8 ct + 4 half + qd (+ 16 for the upper bf16) of a lane's dword <-> the unit in packed dword qd of operand quad
// [half][ct] of that tile.  The input-gradient chain reads these instead of the activations.
constexpr int bits(bool vd, int i) { return (vd ? HV + 8 : 132) + i; }
constexpr int total(bool vd) { return vd ? HV + 8 + 9 : 132 + 8; }
// gradient tape: pre-activation gradients of pts_linears.0-7 (16 pieces each), of feature_linear (16), of
// views_linears.0 (8), and the heads' output gradients padded to 32 rows (2 pieces each)
constexpr int dz(int i) { return 16 * i; }
constexpr int DFEAT = 128;
constexpr int DZV = 144;
constexpr int DRGB = 152;
constexpr int DSIG = 154;
constexpr int DOUT = 128;                // without view directions: output_linear's gradient
constexpr int gtotal(bool vd) { return vd ? 156 : 130; }

inline long long pad256(long long M) { return (M + 255) / 256 * 256; }

}  // namespace tape16

// forward with tape (mlp.hip); a.tape set, a.M samples, packed bf16 weights
int launch_mlp_bf16_train(const MlpArgs& a, bool use_viewdirs, int output_ch, hipStream_t st);

}  // namespace nerf
