// Weight-gradient launches (wgrad.hip), used by the network backward in train.hip.
#pragma once
#include "common.h"

namespace nerf {

constexpr int WG_MAX_PARTS = 256;      // workgroups of a weight-gradient launch

// dW[n, ld*] (+ db[n]) = sum over the M rows of G[m, n] X[m, k]; ws: nerf_train_workspace_bytes() of scratch.
// map: 0 identity, 1 encoded-point k-slot -> reference column, 2 encoded-view k-slot -> reference column.
int launch_wgrad(const float* G, int ldg, int N, const float* X, int ldx, int K, long long M, float* ws, float* dW,
                 int ld, int col_off, int map, float* db, hipStream_t st);
// heads with N <= 5 outputs: G = columns c0 .. c0+N-1 of d_raw [M, ldr]
int launch_wgrad_small(const float* d_raw, int ldr, int c0, int N, const float* X, int K, long long M, float* ws,
                       float* dW, float* db, hipStream_t st);

}  // namespace nerf
