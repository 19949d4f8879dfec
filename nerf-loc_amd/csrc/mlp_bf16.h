// bf16 layer loop shared by the inference kernel (mlp.hip) and the bf16 training kernels (train_bf16.hip):
// 2 column tiles (64 samples) per wave, software-pipelined by hand:
//   * weight fragments are read from LDS PF k-steps ahead into a 4-deep register queue;
//   * the epilogue of an output tile (relu + v_cvt_pk_bf16_f32 into the next layer's operand
//     registers) is cut into 16 slices that are issued under the MFMAs of the NEXT tile;
//   * the bias of the next tile is fetched half a tile ahead.
// The running position `pos` in the fragment stream is a compile-time constant everywhere
// (everything below is fully unrolled), only ring slots and addresses are run-time values.
//
// The training kernels hook into the epilogue through an `Epi` policy: the packed operand quad of a finished tile
// half is what they store ("tape"), and the input-gradient chain masks it with the forward activation first.
#pragma once
#include "mlp_device.h"

namespace nerf {

#ifdef NERF_EXP_STAMP
// diagnostic build only: per-block cycle sums {input+encoding, layers 0-4, layer 5-7, heads, store, passes}
static __device__ unsigned long long g_stamps[256 * 8];
#define NERF_STAMP(var)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);
#else
#define NERF_STAMP(var)
#endif

constexpr int NCT = 2;
#ifndef NERF_PF
#define NERF_PF 3
#endif
constexpr int PF = NERF_PF;                  // LDS read-ahead in fragments
constexpr int QD = PF + 1 <= 4 ? 4 : 8;      // queue depth (power of two > PF)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Stream {                              // position-independent view of the LDS ring
    Pipe p;
    unsigned slot0;                          // ring slot of stage 0 of the current pass
    u32x4 q[QD];                             // fragment queue: q[k & (QD-1)] holds fragment k

    __device__ __forceinline__ const unsigned char* addr(int k) const {
        const unsigned slot = (slot0 + (unsigned)(k >> 4)) & (NSTAGE - 1);
        return p.lds + slot * STAGE_BYTES + (k & (STAGE_FRAGS - 1)) * FRAG_BYTES + p.lane16;
    }
    __device__ __forceinline__ void fetch(int k) { q[k & (QD - 1)] = *reinterpret_cast<const u32x4*>(addr(k)); }
    // before fragment `pos` is consumed: at mid-stage make the next stage visible
    __device__ __forceinline__ void tick(int pos) {
        if ((pos & (STAGE_FRAGS - 1)) == STAGE_FRAGS / 2) p.advance();
    }
    __device__ __forceinline__ void end_pass(int n_frags) { slot0 = (slot0 + (unsigned)(n_frags >> 4)) & (NSTAGE - 1); }
};

__device__ __forceinline__ unsigned pack_pair(float a, float b, bool relu) {
    if (relu) {
        // Activations are stored scaled into [0, 1) (mlp_layout.h, scale_kernel): the clamp bit of the conversion
        // (saturate to [0, 1], negatives and -0 to +0) is then ReLU, at no extra instruction.
        unsigned r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    const f32x2 f = {a, b};
    const bf16x2 v = __builtin_convertvector(f, bf16x2);   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, v);
}

// Tape view of one output tile for the lane: its two operand pieces (k-steps 2t, 2t+1 of the next layer) are four
// 16-byte quads at [half][ct] * 1 KiB from these pointers.
struct EpiTile {
    const unsigned char* act;    // the lane's word of ReLU bits of the tile: written by the training forward, read by
                                 // the input-gradient chain
    unsigned char* out;          // where the packed tile is stored
    bool has_act, has_out;       // compile-time constants at every call site
    int idx;                     // running number of the tile among the masked tiles of the pass (compile-time)
};

struct EpiNone {                 // inference: nothing leaves the registers
    static constexpr bool ZERO_BIAS = false;
    __device__ __forceinline__ void prefetch(const EpiTile&, int) {}
    __device__ __forceinline__ unsigned post(unsigned v, const EpiTile&, int, int, int, int) const { return v; }
    __device__ __forceinline__ void store(const u32x4&, const EpiTile&, int, int) const {}
};

// slice sl (0..15) of the epilogue of a finished tile: two accumulator registers -> one packed
// operand dword of the next layer.  dst = &h[2t]: rows 2t <- registers 0-7, 2t+1 <- registers 8-15.
template <class Epi>
__device__ __forceinline__ void epi_slice(const f32x16 (&accp)[NCT], u32x4 (*dst)[NCT], bool relu, int sl, Epi& epi,
                                          const EpiTile& prev, int prev_par) {
    const int g = sl >> 2, qd = sl & 3, ct = g >> 1, half = g & 1;
    const unsigned v = pack_pair(accp[ct][8 * half + 2 * qd], accp[ct][8 * half + 2 * qd + 1], relu);
    dst[half][ct][qd] = epi.post(v, prev, prev_par, half, ct, qd);
    if (qd == 3) epi.store(dst[half][ct], prev, half, ct);
}

// One 32-row output tile: k-steps [pe KPE][hidden KH][view KVW] accumulate into `acc`, while the
// epilogue of the previous tile (accumulators `accp`, destination pdst) is issued, if has_prev.
// The two accumulator sets alternate from tile to tile, so nothing is ever copied.
template <int KPE, int KH, int KVW, bool early, class Epi>
__device__ __forceinline__ void tile_bf16(Stream& st, int pos, int h, int bias_next_idx, f32x16& bias,
                                          const u32x4 (*pe)[NCT], const u32x4 (*hin)[NCT], const u32x4 (*vw)[NCT],
                                          f32x16 (&acc)[NCT], const f32x16 (&accp)[NCT], u32x4 (*pdst)[NCT],
                                          bool prelu, bool has_prev, Epi& epi, const EpiTile& cur, const EpiTile& prev,
                                          int par) {
    constexpr int KS = KPE + KH + KVW;
    // The 16 epilogue slices of the previous tile are spread over k-steps 1..EPI_N (the previous
    // tile's last MFMA result is not readable for a few cycles, hence not k-step 0).  When this
    // tile reads what the epilogue writes (first tile of a layer: rows 14, 15 of its input) it must
    // be done early, otherwise it may take the whole tile: at most one slice per MFMA gap.
    constexpr int EPI_N = early ? (KS >= 16 ? 12 : KS / 2) : KS - 1;
    static_assert(!early || KS >= 2, "the first tile of a layer needs a k-step to hide the previous epilogue under");
    f32x16 bias_next;
    if (!Epi::ZERO_BIAS) bias_next = bias;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // The order below is the issue order: hipcc's scheduler would otherwise sink every LDS read
    // next to its MFMA (one s_waitcnt lgkmcnt(0) per k-step) and bunch the epilogue in front of the
    // tile, so each group is fenced with sched_barrier(0).  Per k-step: 1 ds_read_b128 (PF ahead),
    // MFMA, <= 8 VALU of the previous tile's epilogue, MFMA, <= 8 VALU.
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = pos + s;
        st.tick(k);
        __builtin_amdgcn_sched_barrier(0);
        st.fetch(k + PF);
        if (!Epi::ZERO_BIAS && s == KS / 2) bias_next = bias_tile(st.p.lds, bias_next_idx, h);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4* b = s < KPE ? pe[s] : (s < KPE + KH ? hin[s - KPE] : vw[s - KPE - KH]);
        const bf16x8 w = __builtin_bit_cast(bf16x8, st.q[k & (QD - 1)]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, __builtin_bit_cast(bf16x8, b[ct]),
                                                             s == 0 ? (Epi::ZERO_BIAS ? zero16 : bias) : acc[ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0 && ct == NCT - 1) epi.prefetch(cur, par);
            if (has_prev && EPI_N >= 1 && s >= 1 && s <= EPI_N) {
                constexpr int EN = EPI_N >= 1 ? EPI_N : 1;
                const int lo = (16 * (s - 1) + EN - 1) / EN, hi = (16 * s + EN - 1) / EN;
                const int cnt = (hi < 16 ? hi : 16) - lo, first = cnt / 2;   // ct0 gap: first, ct1 gap: the rest
                const int b0 = ct == 0 ? lo : lo + first, b1 = ct == 0 ? lo + first : lo + cnt;
#pragma unroll
                for (int sl = b0; sl < b1; ++sl) epi_slice(accp, pdst, prelu, sl, epi, prev, par ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (has_prev && EPI_N < 1) {          // a one-k-step tile has no MFMA shadow to hide the slices under
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) epi_slice(accp, pdst, prelu, sl, epi, prev, par ^ 1);
    }
    if (!Epi::ZERO_BIAS) bias = bias_next;
}

// NT tiles of one layer.  (prev_dst, prev_relu, has_prev) describe the tile that ran just before
// this layer; tile number bt of the pass accumulates into accs[bt & 1]; on return the last tile of this
// layer is pending there (destination hout + 2(NT-1)).  `carry` is the tape view of the pending tile: on entry the
// one that ran before this layer, on return this layer's last; (act0, out0) are this layer's tile 0.
template <int NT, int KPE, int KH, int KVW, class Epi>
__device__ __forceinline__ void layer_bf16(Stream& st, int& pos, int& bt, int n_bias_tiles, int h, f32x16& bias,
                                           const u32x4 (*pe)[NCT], const u32x4 (*hin)[NCT], const u32x4 (*vw)[NCT],
                                           f32x16 (&accs)[2][NCT], u32x4 (*prev_dst)[NCT], bool prev_relu, bool has_prev,
                                           u32x4 (*hout)[NCT], bool relu, Epi& epi, EpiTile& carry,
                                           const unsigned char* act0, bool has_act, unsigned char* out0, bool has_out,
                                           int idx0 = 0) {
    constexpr int ACT_STRIDE = 4;        // the ReLU bit words of a layer: [lane 64][tile 8] dwords
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const EpiTile cur{act0 + t * ACT_STRIDE, out0 + t * 4096, has_act, has_out, idx0 + t};
        const int nb = Epi::ZERO_BIAS ? 0 : (bt + 1) % n_bias_tiles;
        if (t == 0)
            tile_bf16<KPE, KH, KVW, (KPE + KH + KVW >= 2)>(st, pos, h, nb, bias, pe, hin, vw, accs[bt & 1],
                                                            accs[(bt & 1) ^ 1], prev_dst, prev_relu, has_prev, epi, cur, carry, bt & 1);
        else
            tile_bf16<KPE, KH, KVW, false>(st, pos, h, nb, bias, pe, hin, vw, accs[bt & 1],
                                           accs[(bt & 1) ^ 1], hout + 2 * (t - 1), relu, true, epi, cur, carry, bt & 1);
        carry = cur;
        pos += KPE + KH + KVW;
        bt += 1;
    }
}

}  // namespace nerf
