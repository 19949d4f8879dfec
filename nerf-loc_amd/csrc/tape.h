// Layout of the float32 "tape": what a training forward pass keeps of one launch for the backward kernels
// (train.hip).  Written by mlp_f32_train_kernel (train.hip) and by the split-precision training forward
// (mlp_split.hip), read by the input-gradient chain (ReLU bits) and the weight-gradient kernels (layer inputs).
//
// Per sample, sample-major rows: enc(x) [M,64] and enc(d) [M,32] in the k-slot order of the float32 kernel
// (mlp_layout.h: pe_slot_feature / view_slot_feature, position 32 h + q resp. 16 h + q), the post-ReLU outputs of
// pts_linears.0-7 h[0..7] [M,256], the feature vector [M,256] (no ReLU) and the post-ReLU view hidden layer [M,128].
// ReLU bits: per sample and ReLU layer (pts_linears.0-7, then views_linears.0) two lane halves x 4 dwords; dword j of
// half h holds the 32-row output tiles 2 j (bits 0-15) and 2 j + 1 (bits 16-31), bit e of a tile <-> feature
// 32 tile + (e & 3) + 8 (e >> 2) + 4 h (accumulator register e of the 32x32 result layout).
#pragma once
#include <stddef.h>

namespace nerf {

constexpr int TAPE_PE = 64, TAPE_VW = 32;
constexpr int TAPE_BITS_LAYER = 8;
constexpr int tape_bits(bool vd) { return (vd ? 9 : 8) * TAPE_BITS_LAYER; }
constexpr int tape_floats(bool vd) {   // per sample
    return (vd ? TAPE_PE + TAPE_VW + 8 * 256 + 256 + 128 : TAPE_PE + 8 * 256) + tape_bits(vd);
}

struct TapePtrs {
    float* pe; float* vw; float* h[8]; float* feat; float* hv; unsigned* bits;
};

// Rows are allocated for whole 128-sample workgroup tiles so that the kernels store without bounds checks; rows
// beyond M hold values of clamped (repeated) inputs and are never read by the weight-gradient kernels.
inline long long pad128(long long M) { return (M + 127) / 128 * 128; }

inline TapePtrs tape_ptrs(float* base, long long M, bool vd) {
    M = pad128(M);
    TapePtrs t{};
    size_t o = 0;
    t.pe = base + o; o += (size_t)M * TAPE_PE;
    if (vd) { t.vw = base + o; o += (size_t)M * TAPE_VW; }
    for (int i = 0; i < 8; ++i) { t.h[i] = base + o; o += (size_t)M * 256; }
    if (vd) {
        t.feat = base + o; o += (size_t)M * 256;
        t.hv = base + o; o += (size_t)M * 128;
    }
    t.bits = reinterpret_cast<unsigned*>(base + o);
    return t;
}

// The gradient tape (scratch of the backward): the gradient at every pre-activation, same row layout.
constexpr int gtape_floats(bool vd) { return vd ? 8 * 256 + 256 + 128 : 8 * 256; }

struct GTapePtrs {
    float* dz[8]; float* dfeat; float* dzv;
};

inline GTapePtrs gtape_ptrs(float* base, long long M, bool vd) {
    M = pad128(M);
    GTapePtrs t{};
    size_t o = 0;
    for (int i = 0; i < 8; ++i) { t.dz[i] = base + o; o += (size_t)M * 256; }
    if (vd) {
        t.dfeat = base + o; o += (size_t)M * 256;
        t.dzv = base + o;
    }
    return t;
}

}  // namespace nerf
