// Weight-gradient launches (wgrad.hip), used by the network backward in train.hip.
#pragma once
#include "common.h"

namespace nerf {

constexpr int WG_MAX_PARTS = 256;      // workgroups of a weight-gradient launch
constexpr int WG_PART_COLS = 320;      // widest partial tile: 256 hidden + 64 encoded-point columns (pts_linears.5)
// workspace: [WG_MAX_PARTS][256][<= WG_PART_COLS] partial tiles, then [WG_MAX_PARTS][256] bias partials
constexpr size_t WG_BIAS_OFFSET = (size_t)WG_MAX_PARTS * 256 * WG_PART_COLS;

// dW[n, ld*] (+ db[n]) = sum over the M rows of G[m, n] X[m, k]; ws: nerf_train_workspace_bytes() of scratch.
// map: 0 identity, 1 encoded-point k-slot -> reference column, 2 encoded-view k-slot -> reference column.
int launch_wgrad(const float* G, int ldg, int N, const float* X, int ldx, int K, long long M, float* ws, float* dW,
                 int ld, int col_off, int map, float* db, hipStream_t st);
// heads with N <= 5 outputs: G = columns c0 .. c0+N-1 of d_raw [M, ldr]
int launch_wgrad_small(const float* d_raw, int ldr, int c0, int N, const float* X, int K, long long M, float* ws,
                       float* dW, float* db, hipStream_t st);

// Second phase alone, for columns pcol0 .. pcol0 + k_emit - 1 of partial tiles [parts][256][pcol] in ws (bias
// partials at WG_BIAS_OFFSET):
// maps 3 / 4 are the encoded-point / encoded-view columns in the bf16 operand order; *x_scale (device pointer or
// null) is a power of two the inputs were stored multiplied by.
int launch_wgrad_reduce(const float* ws, int parts, int pcol, int pcol0, int n_rows, int k_emit, float* dW, int ld,
                        int col_off, int map, const float* x_scale, float* db, hipStream_t st);

// bf16 training (train_bf16.hip)
size_t tape16_bytes(long long M, bool vd);
size_t gtape16_bytes(long long M, bool vd);
int pack_backward_bf16(const float* const* tensors_host, bool vd, int output_ch, void* packed_out, hipStream_t st);
int backward_bf16(const void* packed_bwd, const void* packed_fwd, bool vd, int output_ch, const float* d_raw, long long M,
                  const void* tape, void* grad_tape, float* ws, float* const* grads_host, hipStream_t st);

}  // namespace nerf
