// Device-side pieces shared by the split-precision kernels (mlp_split.hip: inference and training forward;
// dgrad_split.hip: input-gradient chain): the fragment queue over the LDS ring, the T-term split of float32 pairs into
// packed bf16 dwords, the operand window of one k-step.  Operand conventions of the 16x16x32 stream: mlp_layout.h.
#pragma once
#include "mlp_bf16.h"
#include "mlp16.h"

namespace nerf {
namespace sp {

constexpr int NCT = 2;                       // 16-column tiles per wave
constexpr int WSAMP = NCT * 16;              // samples per wave and pass
constexpr int TILE = 4 * WSAMP;              // samples per workgroup and pass
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int T>
struct SStream {
    Pipe p;
    unsigned slot0;                          // ring slot of stage 0 of the current pass
    u32x4 q[2 * T];                          // fragment k sits in q[k % (2 T)]

    __device__ __forceinline__ const unsigned char* addr(int k) const {
        const unsigned slot = (slot0 + (unsigned)(k >> 4)) & (NSTAGE - 1);
        return p.lds + slot * STAGE_BYTES + (k & (STAGE_FRAGS - 1)) * FRAG_BYTES + p.lane16;
    }
    __device__ __forceinline__ void fetch(int k) { q[k % (2 * T)] = *reinterpret_cast<const u32x4*>(addr(k)); }
    __device__ __forceinline__ void tick(int k) {
        if ((k & (STAGE_FRAGS - 1)) == STAGE_FRAGS / 2) p.advance();
    }
    __device__ __forceinline__ void end_pass(int n_frags) { slot0 = (slot0 + (unsigned)(n_frags >> 4)) & (NSTAGE - 1); }
};

// ---- the T-term split of a pair of float32 values into packed bf16 dwords ------------------------------------------
__device__ __forceinline__ float lo_as_float(unsigned packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float hi_as_float(unsigned packed) { return __builtin_bit_cast(float, packed & 0xffff0000u); }

// Single instructions, written out: fmaxf() costs a second v_max_f32 (NaN canonicalisation) and hipcc pairs the two
// subtractions into one v_pk_add_f32, which is slower than two plain ones in the shadow of an MFMA.
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float sub1(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int T>
__device__ __forceinline__ void split_pair(float a, float b, bool relu, unsigned (&out)[T]) {
    if (relu) { a = relu1(a); b = relu1(b); }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const unsigned p = pack_pair(a, b, false);          // v_cvt_pk_bf16_f32, round to nearest even
        out[t] = p;
        if (t + 1 < T) {                                    // exact: the rounded value shares the leading bits
            a = sub1(a, lo_as_float(p));
            b = sub1(b, hi_as_float(p));
        }
    }
}

// B operands of one k-step: [column tile][term], a quad of 4 packed dwords each
template <int T>
struct Ops {
    u32x4 v[NCT][T];
};

}  // namespace sp
}  // namespace nerf
