// Layout of the weight "fragment stream" shared by the packer and the fused MLP kernels.
//
// The network (run_nerf_helpers.py:67-119) is evaluated transposed: for every layer
//     Out^T[out_feature, sample] = W[out_feature, k] . In^T[k, sample]
// so W is the MFMA A operand and the activations are the B operand.  The 32x32 result tile
// of one MFMA chain has the sample on the lane and the output features in the accumulator
// registers, which is exactly the B-operand layout of the next layer: activations never leave
// the register file.  Only the weights move: they are pre-arranged on the host side into 1 KiB
// fragments (64 lanes x 16 B, in the order the kernel consumes them) and streamed
// L2 -> LDS -> registers.
//
// Stream order: layer, then 32-row output tile, then k-step.  The k-steps of a tile are
// [positional-encoding part][hidden part][view-encoding part].
//
// k-slot conventions (h = lane >> 5, r = lane & 31):
//   bf16, v_mfma_f32_32x32x16_bf16: a k-step has 16 slots (h, j), j = 0..7.
//     hidden k-step s : slot (h, j) <-> input feature 16 s + 8 (j >> 2) + 4 h + (j & 3)
//                       (registers 8(s&1) .. 8(s&1)+7 of accumulator tile s >> 1, see the C layout
//                       row = (reg & 3) + 8 (reg >> 2) + 4 h)
//   f32, v_mfma_f32_32x32x2_f32: a k-step has 2 slots (h); accumulator register e of input
//     tile T is used directly as the B operand:
//     hidden k-step 16 T + e : slot h <-> input feature 32 T + (e & 3) + 8 (e >> 2) + 4 h
//   encoded inputs: each lane half computes 32 (points) / 16 (view) values, "slot q":
//     points q < 30 : frequency 5 h + q / 6, coordinate (q % 6) / 2, sin if q even else cos
//            q = 30 : x (h = 0) or z (h = 1);  q = 31 : y (h = 0) or zero padding (h = 1)
//     view   q < 12 : frequency 2 h + q / 6, coordinate (q % 6) / 2, sin / cos as above
//            q = 12..14 : the direction itself (h = 0 only);  rest zero padding
//     bf16: k-step p holds slots 8 p .. 8 p + 7;  f32: k-step e holds slot e.
#pragma once

namespace nerf {

constexpr int FRAG_BYTES = 1024;
constexpr int STAGE_FRAGS = 16;
constexpr int STAGE_BYTES = STAGE_FRAGS * FRAG_BYTES;
constexpr int MAX_LAYERS = 11;

enum LayerKind { LK_HIDDEN = 0, LK_FEATURE_ALPHA = 1, LK_RGB = 2, LK_OUTPUT = 3 };

struct LayerDesc {
    int kind;
    int n_tiles;     // 32-row output tiles
    int out_dim;     // rows of the (main) weight tensor
    int has_pe;      // k part 1: encoded points (63 features, 64 slots)
    int hid_feats;   // k part 2: hidden features (0, 128 or 256)
    int has_view;    // k part 3: encoded view direction (27 features, 32 slots)
    int w_idx, b_idx;        // index into the state_dict-ordered tensor list
    int w2_idx, b2_idx;      // second tensor (alpha_linear for LK_FEATURE_ALPHA), else -1
    int in_dim;      // columns of the weight tensor
    int pe_col, hid_col, view_col;   // first column of each k part in the weight tensor
    // Backward (input-gradient) layers use the TRANSPOSE of a weight tensor: output row r of the layer is
    // column row_off + r of the tensor, k index c is its row c (valid for c < k_limit, zero beyond).
    int trans, row_off, k_limit;
    // optional extra k part of `extra_frags` float32 fragments taken from the transposed tensor w2_idx
    // (row stride in_dim2, rows c < k_limit2): the density head joining the feature head's input gradient
    int extra_frags, in_dim2, k_limit2;
    // bf16 kernel only: outputs of a layer are kept multiplied by a power of two chosen at pack time so that they
    // provably stay below 1; v_cvt_pk_bf16_f32 with the clamp bit (saturate to [0, 1]) is then exactly ReLU.
    int src_layer;    // layer whose (scaled) output is this layer's hidden k part, -1: none
    int scaled_out;   // 1: the main rows of this layer are stored scaled (hidden layers, feature, view layer)
};

struct NetDesc {
    int n_layers;
    int n_bias_tiles;
    LayerDesc layer[MAX_LAYERS];
};

__host__ __device__ constexpr int ksteps_bf16(const LayerDesc& L) {
    return (L.has_pe ? 4 : 0) + L.hid_feats / 16 + (L.has_view ? 2 : 0) + L.extra_frags;
}
__host__ __device__ constexpr int frags_f32(const LayerDesc& L) {
    return (L.has_pe ? 8 : 0) + L.hid_feats / 8 + (L.has_view ? 4 : 0) + L.extra_frags;
}
__host__ __device__ constexpr int frags_per_tile(const LayerDesc& L, int prec_bf16) {
    return prec_bf16 ? ksteps_bf16(L) : frags_f32(L);
}

// reference feature index of encoded-point slot q of lane half h, or -1 (run_nerf_helpers.py:24-45:
// [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...], each block of three being (x, y, z))
__host__ __device__ constexpr int pe_slot_feature(int h, int q) {
    if (q < 30) return 3 + 6 * (5 * h + q / 6) + 3 * (q & 1) + (q % 6) / 2;
    if (q == 30) return h == 0 ? 0 : 2;
    return h == 0 ? 1 : -1;
}
__host__ __device__ constexpr int view_slot_feature(int h, int q) {
    if (q < 12) return 3 + 6 * (2 * h + q / 6) + 3 * (q & 1) + (q % 6) / 2;
    if (q < 15) return h == 0 ? q - 12 : -1;
    return -1;
}

// ---- v_mfma_f32_16x16x32_bf16 conventions (the bf16 INFERENCE kernel, mlp16.h; g = lane >> 4, c = lane & 15) ----------
// A fragment = 16 output rows x 32 k-slots: lane (c, g) holds row c, slots (g, j), j = 0..7.  B operand = 32 k-slots x 16
// samples: lane (c, g) holds sample c, slots (g, j).  Result: lane (c, g) holds sample c, rows 4 g + reg, reg = 0..3.
//   hidden k-step s : slot (g, j) <-> input feature 32 s + 16 (j >> 2) + 4 g + (j & 3)
//                     = accumulator register j & 3 of the 16-row output tile 2 s + (j >> 2) of the previous layer
//   encoded points  : 2 k-steps; lane group g computes 16 values, "slot q" = 8 (k-step) + j = 4 m + t with
//                     t = 0 sin(f0), 1 cos(f0), 2 sin(f0 + 1), 3 cos(f0 + 1) (octaves f0, f0 + 1 of one coordinate):
//                     m < 3: coordinate m, f0 = 2 g;  m = 3: g < 3: coordinate g, f0 = 8;  g = 3: x, y, z, zero padding
//   encoded view    : 1 k-step; g < 3: coordinate g, slot j = 2 f + (1 if cos), f = 0..3;  g = 3: dx, dy, dz, zeros
__host__ __device__ constexpr int pe16_slot_feature(int g, int q) {
    const int m = q >> 2, t = q & 3;
    if (m == 3 && g == 3) return t < 3 ? t : -1;
    return 3 + 6 * ((m < 3 ? 2 * g : 8) + (t >> 1)) + 3 * (t & 1) + (m < 3 ? m : g);
}
__host__ __device__ constexpr int view16_slot_feature(int g, int j) {
    if (g == 3) return j < 3 ? j : -1;
    return 3 + 6 * (j >> 1) + 3 * (j & 1) + g;
}
// 16-row output tiles and 32-slot k-steps of a (forward) layer in this layout
__host__ __device__ constexpr int tiles16(const LayerDesc& L) {
    return L.kind == LK_HIDDEN ? L.out_dim / 16 : (L.kind == LK_FEATURE_ALPHA ? 17 : 1);
}
__host__ __device__ constexpr int ksteps16(const LayerDesc& L) {
    return (L.has_pe ? 2 : 0) + L.hid_feats / 32 + (L.has_view ? 1 : 0);
}

inline NetDesc make_net_desc(bool use_viewdirs, int output_ch) {
    NetDesc n{};
    int li = 0, bt = 0;
    for (int i = 0; i < 8; ++i) {
        LayerDesc L{};
        L.kind = LK_HIDDEN; L.n_tiles = 8; L.out_dim = 256;
        L.has_pe = (i == 0 || i == 5); L.hid_feats = (i == 0) ? 0 : 256; L.has_view = 0;
        L.w_idx = 2 * i; L.b_idx = 2 * i + 1; L.w2_idx = L.b2_idx = -1;
        L.in_dim = (i == 0) ? 63 : (i == 5 ? 319 : 256);
        L.pe_col = 0; L.hid_col = (i == 5) ? 63 : 0; L.view_col = 0;
        L.src_layer = i - 1; L.scaled_out = 1;
        n.layer[li++] = L; bt += L.n_tiles;
    }
    if (use_viewdirs) {
        LayerDesc fa{};
        fa.kind = LK_FEATURE_ALPHA; fa.n_tiles = 9; fa.out_dim = 256; fa.hid_feats = 256;
        fa.w_idx = 18; fa.b_idx = 19; fa.w2_idx = 20; fa.b2_idx = 21; fa.in_dim = 256;
        fa.src_layer = 7; fa.scaled_out = 1;      // the feature rows; the density row is a network output (scale 1)
        n.layer[li++] = fa; bt += fa.n_tiles;
        LayerDesc vw{};
        vw.kind = LK_HIDDEN; vw.n_tiles = 4; vw.out_dim = 128; vw.hid_feats = 256; vw.has_view = 1;
        vw.w_idx = 16; vw.b_idx = 17; vw.w2_idx = vw.b2_idx = -1; vw.in_dim = 283; vw.view_col = 256;
        vw.src_layer = 8; vw.scaled_out = 1;
        n.layer[li++] = vw; bt += vw.n_tiles;
        LayerDesc rgb{};
        rgb.kind = LK_RGB; rgb.n_tiles = 1; rgb.out_dim = 3; rgb.hid_feats = 128;
        rgb.w_idx = 22; rgb.b_idx = 23; rgb.w2_idx = rgb.b2_idx = -1; rgb.in_dim = 128;
        rgb.src_layer = 9; rgb.scaled_out = 0;
        n.layer[li++] = rgb; bt += rgb.n_tiles;
    } else {
        LayerDesc o{};
        o.kind = LK_OUTPUT; o.n_tiles = 1; o.out_dim = output_ch; o.hid_feats = 256;
        o.w_idx = 18; o.b_idx = 19; o.w2_idx = o.b2_idx = -1; o.in_dim = 256;
        o.src_layer = 7; o.scaled_out = 0;
        n.layer[li++] = o; bt += o.n_tiles;
    }
    n.n_layers = li;
    n.n_bias_tiles = bt;
    return n;
}

// Input-gradient chain.  Layers in execution order, with view directions:
//   rgb_linear^T (3 -> 128), views_linears.0^T restricted to the feature columns (128 -> 256),
//   feature_linear^T plus alpha_linear^T (256 + 1 -> 256), then pts_linears 7..1 transposed
//   (layer 5: only the 256 hidden columns, the encoded-input columns have no leaves behind them);
// without: output_linear^T (output_ch -> 256), then pts_linears 7..1.  The smallest k part is one fragment: 8 input
// features in float32, 16 in bf16.
inline NetDesc make_backward_desc(bool use_viewdirs, int output_ch, bool bf16 = false) {
    NetDesc n{};
    int li = 0;
    auto add = [&](int tiles, int out_dim, int kfeats, int w_idx, int in_dim, int row_off, int k_limit) -> LayerDesc& {
        LayerDesc L{};
        L.kind = LK_HIDDEN; L.n_tiles = tiles; L.out_dim = out_dim; L.hid_feats = kfeats;
        L.w_idx = w_idx; L.b_idx = -1; L.w2_idx = L.b2_idx = -1; L.in_dim = in_dim;
        L.trans = 1; L.row_off = row_off; L.k_limit = k_limit; L.src_layer = -1;
        n.layer[li] = L;
        return n.layer[li++];
    };
    if (use_viewdirs) {
        add(4, 128, bf16 ? 16 : 8, 22, 128, 0, 3);     // d(view hidden) = rgb_linear.weight^T d(rgb)
        add(8, 256, 128, 16, 283, 0, 128);             // d(feature) = views_linears.0.weight[:, :256]^T d(view hidden)
        LayerDesc& fa = add(8, 256, 256, 18, 256, 0, 256);   // d(h8) = feature^T d(feature) + alpha^T d(sigma)
        fa.extra_frags = 1; fa.w2_idx = 20; fa.in_dim2 = 256; fa.k_limit2 = 1;
    } else {
        add(8, 256, bf16 ? 16 : 8, 18, 256, 0, output_ch);   // d(h8) = output_linear.weight^T d(raw)
    }
    for (int i = 7; i >= 1; --i) add(8, 256, 256, 2 * i, i == 5 ? 319 : 256, i == 5 ? 63 : 0, 256);
    n.n_layers = li;
    n.n_bias_tiles = 0;
    return n;
}
constexpr int FRAGS_BWD_F32_VD = 4 * 1 + 8 * 16 + 8 * 33 + 7 * 8 * 32;               // 2188
constexpr int FRAGS_BWD_F32_NOVD = 8 * 1 + 7 * 8 * 32;                                // 1800
constexpr int FRAGS_BWD_BF16_VD = 4 * 1 + 8 * 8 + 8 * 17 + 7 * 8 * 16;               // 1100
constexpr int FRAGS_BWD_BF16_NOVD = 8 * 1 + 7 * 8 * 16;                               // 904
constexpr int pad_to_stage(int f) { return (f + STAGE_FRAGS - 1) / STAGE_FRAGS * STAGE_FRAGS; }

inline int total_frags(const NetDesc& n, int prec_bf16) {
    int t = 0;
    for (int i = 0; i < n.n_layers; ++i) t += n.layer[i].n_tiles * frags_per_tile(n.layer[i], prec_bf16);
    return t;
}

// compile-time totals the kernels rely on (checked against make_net_desc at pack time)
constexpr int FRAGS_BF16_VD = 8 * 4 + 4 * 8 * 16 + 8 * 20 + 2 * 8 * 16 + 9 * 16 + 4 * 18 + 8;   // 1184
constexpr int FRAGS_BF16_NOVD = 8 * 4 + 4 * 8 * 16 + 8 * 20 + 2 * 8 * 16 + 16;                  // 976
constexpr int FRAGS_F32_VD = 8 * 8 + 4 * 8 * 32 + 8 * 40 + 2 * 8 * 32 + 9 * 32 + 4 * 36 + 16;    // 2368
constexpr int FRAGS_F32_NOVD = 8 * 8 + 4 * 8 * 32 + 8 * 40 + 2 * 8 * 32 + 32;                    // 1952
constexpr int BIAS_TILES_VD = 64 + 9 + 4 + 1;    // 78
constexpr int BIAS_TILES_NOVD = 64 + 1;          // 65
constexpr int BIAS_TILE_BYTES = 32 * 4;
constexpr int SCALE_BYTES = 64;                  // MAX_LAYERS output scales behind the bias block
constexpr int ROWSUM_BYTES = MAX_LAYERS * 256 * 2 * 4;   // behind the scales: pack-time scratch (per-row |W| sums)
constexpr float INPUT_BOUND = 16777216.0f;       // |network input| assumed <= 2^24 when the scales are chosen

// The 16x16x32 stream of the inference kernel: 1172 / 968 fragments, zero-padded to the same totals as above
constexpr int FRAGS16_VD = 16 * 2 + 4 * 16 * 8 + 16 * 10 + 2 * 16 * 8 + 17 * 8 + 8 * 9 + 4;     // 1172
constexpr int FRAGS16_NOVD = 16 * 2 + 4 * 16 * 8 + 16 * 10 + 2 * 16 * 8 + 8;                     // 968
constexpr int BIAS_TILES16_VD = 8 * 16 + 17 + 8 + 1;     // 154 tiles of 16 floats
constexpr int BIAS_TILES16_NOVD = 8 * 16 + 1;            // 129
constexpr int BIAS16_TILE_BYTES = 16 * 4;
constexpr int bias16_block_bytes(bool vd) { return ((vd ? BIAS_TILES16_VD : BIAS_TILES16_NOVD) * BIAS16_TILE_BYTES + 127) / 128 * 128; }
static_assert(pad_to_stage(FRAGS16_VD) == FRAGS_BF16_VD && pad_to_stage(FRAGS16_NOVD) == FRAGS_BF16_NOVD, "same stream length");

static_assert(FRAGS_BF16_VD % STAGE_FRAGS == 0 && FRAGS_BF16_NOVD % STAGE_FRAGS == 0, "stage-aligned");
static_assert(FRAGS_F32_VD % STAGE_FRAGS == 0 && FRAGS_F32_NOVD % STAGE_FRAGS == 0, "stage-aligned");

}  // namespace nerf
