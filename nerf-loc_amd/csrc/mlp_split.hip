// Split-precision network kernels, inference side (NERF_PREC_BF16X3; NERF_PREC_BF16X2 is instantiated in mlp_split_x2.hip):
// the instantiations of the body in mlp_split_body.h, the packer of the k-major T-term weight stream, and the host entry points.
#include "mlp_split_body.h"

namespace nerf {
namespace sp {

// ---- packer: the k-major T-term stream --------------------------------------------------------------------------------
// Order: layer, k-step ([encoded points][hidden][encoded view]), 16-row output tile, term.  Then the bias tiles of
// 16 floats (unscaled), as in the bf16 inference stream.
__global__ __launch_bounds__(256) void pack_split_kernel(PackArgs a, int T, int padded_frags, int n_bias_tiles16) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int F = gid >> 6, lane = gid & 63;
    const int r = lane & 15, gg = lane >> 4;
    if (F < padded_frags) {
        int f = F, li = 0;
        for (; li < a.net.n_layers; ++li) {
            const int nf = tiles16(a.net.layer[li]) * ksteps16(a.net.layer[li]) * T;
            if (f < nf) break;
            f -= nf;
        }
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
        if (li < a.net.n_layers) {
            const LayerDesc& L = a.net.layer[li];
            const int nt = tiles16(L);
            const int term = f % T, rest = f / T;
            const int tile = rest % nt, kk = rest / nt;
            int row = -1;
            const float* W = nullptr;
            switch (L.kind) {
                case LK_HIDDEN:
                    if (tile * 16 + r < L.out_dim) { row = tile * 16 + r; W = a.t[L.w_idx]; }
                    break;
                case LK_FEATURE_ALPHA:
                    if (tile < 16) { row = tile * 16 + r; W = a.t[L.w_idx]; }
                    else if (r == 3) { row = 0; W = a.t[L.w2_idx]; }                   // the density row
                    break;
                default:
                    if (r < L.out_dim) { row = r; W = a.t[L.w_idx]; }
            }
            const int npe = L.has_pe ? 2 : 0, nh = L.hid_feats / 32;
            for (int j = 0; j < 8; ++j) {
                int col = -1;
                if (kk < npe) { const int ft = pe16_slot_feature(gg, 8 * kk + j); col = ft < 0 ? -1 : L.pe_col + ft; }
                else if (kk < npe + nh) col = L.hid_col + 32 * (kk - npe) + 16 * (j >> 2) + 4 * gg + (j & 3);
                else { const int ft = view16_slot_feature(gg, j); col = ft < 0 ? -1 : L.view_col + ft; }
                float w = (W && col >= 0) ? W[(size_t)row * L.in_dim + col] : 0.0f;
                __bf16 b = (__bf16)w;
                for (int t = 0; t < term; ++t) {        // exact differences
                    w = __fsub_rn(w, (float)b);
                    b = (__bf16)w;
                }
                v[j] = b;
            }
        }
        *reinterpret_cast<bf16x8*>(a.out + (size_t)F * FRAG_BYTES + lane * 16) = v;
    } else {
        const int bi = gid - padded_frags * 64;
        if (bi >= n_bias_tiles16 * 16) return;
        int t = bi >> 4, li = 0;
        const int rr = bi & 15;
        for (; li < a.net.n_layers; ++li) {
            if (t < tiles16(a.net.layer[li])) break;
            t -= tiles16(a.net.layer[li]);
        }
        const LayerDesc& L = a.net.layer[li];
        float b = 0.0f;
        switch (L.kind) {
            case LK_HIDDEN: if (t * 16 + rr < L.out_dim) b = a.t[L.b_idx][t * 16 + rr]; break;
            case LK_FEATURE_ALPHA:
                if (t < 16) b = a.t[L.b_idx][t * 16 + rr];
                else if (rr == 3) b = a.t[L.b2_idx][0];
                break;
            default: if (rr < L.out_dim) b = a.t[L.b_idx][rr];
        }
        reinterpret_cast<float*>(a.out + (size_t)padded_frags * FRAG_BYTES)[bi] = b;
    }
}

}  // namespace sp

int split_terms(int precision) { return precision == NERF_PREC_BF16X3 ? 3 : (precision == NERF_PREC_BF16X2 ? 2 : 0); }

int split_padded_frags(int terms, bool use_viewdirs) {
    return pad_to_stage(terms * (use_viewdirs ? FRAGS16_VD : FRAGS16_NOVD));
}

size_t split_packed_bytes(int terms, bool use_viewdirs) {
    return (size_t)split_padded_frags(terms, use_viewdirs) * FRAG_BYTES + bias16_block_bytes(use_viewdirs);
}

int launch_pack_split(const PackArgs& a, int terms, bool use_viewdirs, hipStream_t stream) {
    const int padded = split_padded_frags(terms, use_viewdirs);
    const int nb16 = use_viewdirs ? BIAS_TILES16_VD : BIAS_TILES16_NOVD;
    const int threads = padded * 64 + nb16 * 16;
    hipLaunchKernelGGL(sp::pack_split_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, a, terms, padded, nb16);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

int split_samples_per_tile() { return sp::TILE; }

int launch_mlp_split(const MlpArgs& a, int terms, bool use_viewdirs, int output_ch, unsigned grid, hipStream_t st) {
    const dim3 g(grid), b(MLP_THREADS);
#ifdef NERF_EXP_ONLY_VD
    NERF_REQUIRE(use_viewdirs && terms == 3, "experiment build: view directions and three terms only");
    hipLaunchKernelGGL((sp::mlp_split_kernel<3, true, 4>), g, b, 0, st, a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
#endif
    if (terms != 3) return launch_mlp_split2(a, use_viewdirs, output_ch, grid, st);      // mlp_split_x2.hip
    if (use_viewdirs) hipLaunchKernelGGL((sp::mlp_split_kernel<3, true, 4>), g, b, 0, st, a);
    else if (output_ch == 4) hipLaunchKernelGGL((sp::mlp_split_kernel<3, false, 4>), g, b, 0, st, a);
    else hipLaunchKernelGGL((sp::mlp_split_kernel<3, false, 5>), g, b, 0, st, a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

}  // namespace nerf
