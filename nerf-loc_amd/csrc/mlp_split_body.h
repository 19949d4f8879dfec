// float32-accurate network evaluation on the bf16 matrix cores (NERF_PREC_BF16X3, NERF_PREC_BF16X2): the kernel body,
// instantiated by mlp_split.hip (inference, three and two terms) and mlp_split_train.hip (training forward).
//
// NeRF.forward (run_nerf_helpers.py:96-119) computes in float32.  The exact float32 MFMA runs at 1/16 of the bf16
// rate, so this kernel writes every matrix-product operand as a sum of T bfloat16 terms,
//     x = x0 + x1 (+ x2),   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (differences are exact),
// and evaluates  W x  as the products  W_i x_j  with i + j < T  on v_mfma_f32_16x16x32_bf16 with float32
// accumulation: six products for T = 3 (what is dropped is below 2^-24 of the leading product: float32 accuracy at 6/16
// of the exact instruction's cost), three for T = 2 (2^-16).
//
// Data flow (operand conventions of the 16x16x32 stream: mlp_layout.h).  A wave owns 32 samples (two 16-column tiles)
// and carries them through all layers in registers, as in the other network kernels; but the activations of a layer
// are kept as FLOAT32 accumulators (4 bytes per value instead of 2 T), and a layer is evaluated K-MAJOR: for each 32-wide
// k-step, for each 16-row output tile, T weight fragments and their products.  All output tiles of a layer accumulate
// side by side (2 x 16 tiles x 2 column tiles x 4 registers = 256 accumulator registers for the layer being read and the
// one being written), and the bf16 terms of an input k-step exist only while that k-step runs: the split of k-step
// s + 1 (ReLU, 3 conversions, 4 subtractions per pair of values) is issued in the MFMA shadow of k-step s.
// Weights stream L2 -> LDS ring -> registers exactly as in the bf16 kernel (Pipe), T fragments per (k-step, tile).
//
// The same body is the float32-accurate TRAINING forward (mlp_split_train_kernel): the moment a layer's accumulators
// are turned into the next layer's operands is also the moment their post-ReLU float32 values exist in registers, so
// the conversion slices store them (and the encodings, and one ReLU bit per value) in the float32 tape layout of
// tape.h, and the float32 backward kernels of train.hip run on that tape unchanged.
#pragma once
#include "mlp_split_device.h"
#include "mlp_split.h"
#include "tape.h"

namespace nerf {
namespace sp {

// ---- tape of the training forward ------------------------------------------------------------------------------------
// Where the conversion slices of one layer's output go: rows of `ld` floats (256, or 128 for the view hidden layer).
struct TapeDst {
    float* base;
    int ld;
};
// Per pass and lane: rowb[ct] = 1024 m (byte offset of the sample's row in a 256-float tape; 4 194 304 samples at
// most), g16 = 16 g; nib[ct] = one bit per value (> 0) of the layer being converted, nibble t of the 64-bit word <->
// accumulator tile t.
struct TapeCtx {
    unsigned rowb[NCT], g16;
    unsigned nib[NCT][2];
};

__device__ __forceinline__ unsigned positive_bit(float x) {      // 1 if x > 0 (on the bits as a signed integer)
    unsigned b;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(b) : "v"(x));
    return b;
}

// nibble i of the low 16 bits -> nibble 2 i
__device__ __forceinline__ unsigned spread_nibbles(unsigned x) {
    x = (x | (x << 8)) & 0x00ff00ffu;
    return (x | (x << 4)) & 0x0f0f0f0fu;
}

// The ReLU bits of a converted layer in the layout of tape.h.  Lane group g holds features 16 t + 4 g + r of tile t;
// the float32 layout wants, for lane half h = g & 1, 32-row tile t32 = t >> 1, bit 4 g' + r with g' = 2 (t & 1) + (g >> 1):
// groups g and g ^ 2 (32 lanes apart) share every dword, g < 2 supplies its even nibbles, g >= 2 the odd ones.
// After the exchange group g < 2 writes dwords 0, 1 and group g >= 2 dwords 2, 3 of the half's four.
__device__ __forceinline__ void flush_bits(TapeCtx& tc, unsigned* bits, int bits_per_sample, int layer) {
    const int g = (int)(tc.g16 >> 4), lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const unsigned lo = tc.nib[ct][0], hi = tc.nib[ct][1];
        const int partner = (lane ^ 32) << 2;
        const unsigned plo = (unsigned)__builtin_amdgcn_ds_bpermute(partner, (int)lo);
        const unsigned phi = (unsigned)__builtin_amdgcn_ds_bpermute(partner, (int)hi);
        const bool low = g < 2;
        const unsigned even = low ? lo : phi, odd = low ? plo : hi;
        uint2 w;
        w.x = spread_nibbles(even & 0xffffu) | (spread_nibbles(odd & 0xffffu) << 4);
        w.y = spread_nibbles(even >> 16) | (spread_nibbles(odd >> 16) << 4);
        unsigned* row = bits + (size_t)(tc.rowb[ct] >> 10) * bits_per_sample + layer * TAPE_BITS_LAYER + 4 * (g & 1) + (low ? 0 : 2);
        *reinterpret_cast<uint2*>(row) = w;
        tc.nib[ct][0] = tc.nib[ct][1] = 0u;
    }
}

// pair-conversion number sl (0 .. 4 NCT - 1) of the k-step made of accumulator tiles src[0], src[1] (tiles tile0,
// tile0 + 1 of their layer).  TAPE: the second slice of a tile stores its four values and notes their signs.
template <int T, bool TAPE>
__device__ __forceinline__ void convert_slice(const f32x4v (*src)[NCT], bool relu, int sl, Ops<T>& dst, TapeCtx& tc,
                                              const TapeDst& td, int tile0) {
    const int ct = sl >> 2, d = sl & 3;
    const f32x4v& acc = src[d >> 1][ct];
    unsigned w[T];
    split_pair<T>(acc[2 * (d & 1)], acc[2 * (d & 1) + 1], relu, w);
#pragma unroll
    for (int t = 0; t < T; ++t) dst.v[ct][t][d] = w[t];
    if (TAPE && (d & 1)) {
        const int tile = tile0 + (d >> 1);
        float4 v;
        v.x = relu ? relu1(acc[0]) : acc[0]; v.y = relu ? relu1(acc[1]) : acc[1];
        v.z = relu ? relu1(acc[2]) : acc[2]; v.w = relu ? relu1(acc[3]) : acc[3];
        const unsigned off = (td.ld == 256 ? tc.rowb[ct] : tc.rowb[ct] >> 1) + tc.g16;
#ifndef NERF_SPX_NOTAPESTORE      // timing experiment only (the tape is garbage): what the activation stores cost
        *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(td.base) + off + 64 * tile) = v;
#else
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(off));
#endif
        if (relu) {
            const unsigned n = positive_bit(v.x) | (positive_bit(v.y) << 1) | (positive_bit(v.z) << 2) | (positive_bit(v.w) << 3);
            tc.nib[ct][tile >> 3] |= n << (4 * (tile & 7));
        }
    }
}

// Tape positions of the lane's encoded inputs (mlp_layout.h: the float32 kernel's slot (h, q) sits at 32 h + q of the
// 64-float row, resp. 16 h + q of the 32-float row).  The lane's 16 point slots q = 4 m + t (pe16_slot_feature) fall
// into four runs whose first position depends on the lane group only; within a run the position is a constant away:
//   m < 3, t < 2: octave 2 g      -> base[0] + 2 m + t          m < 3, t >= 2: octave 2 g + 1 -> base[1] + 2 m + (t - 2)
//   m = 3, t < 2: base[2] + t                                   m = 3, t >= 2: base[3] + (t - 2)
//   (g < 3: octaves 8, 9 of coordinate g;  g = 3: x, y at 30, 31 and z, zero padding at 62, 63)
struct TapeSlots {
    int pe[4];
};
__device__ __forceinline__ int octave_pos(int octave) {          // slot of (octave, coordinate 0, sin)
    const int hh = octave >= 5 ? 1 : 0;
    return 32 * hh + 6 * (octave - 5 * hh);
}
__device__ __forceinline__ TapeSlots tape_slots(int g) {
    TapeSlots s;
    s.pe[0] = octave_pos(2 * g);
    s.pe[1] = octave_pos(2 * g + 1);
    s.pe[2] = g == 3 ? 30 : octave_pos(8) + 2 * g;
    s.pe[3] = g == 3 ? 62 : octave_pos(9) + 2 * g;
    return s;
}
// byte offsets into the tapes are 32-bit (at most 2^22 samples per launch)
__device__ __forceinline__ void store_pe_tape(float* tape, unsigned m, const TapeSlots& sl, const float (&pv)[16]) {
    unsigned char* base = reinterpret_cast<unsigned char*>(tape);
    const unsigned row = m * (unsigned)(TAPE_PE * sizeof(float));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned off = row + 4u * (unsigned)sl.pe[r];
#pragma unroll
        for (int i = 0; i < (r < 2 ? 6 : 2); ++i) {
            // run r < 2: values (m = i >> 1, t = 2 r + (i & 1)) at 2 m + (i & 1) = i;  run r >= 2: (m = 3, t = 2 (r - 2) + i) at i
            const int q = r < 2 ? 4 * (i >> 1) + 2 * r + (i & 1) : 12 + 2 * (r - 2) + i;
            *reinterpret_cast<float*>(base + off + 4 * i) = pv[q];
        }
    }
}
__device__ __forceinline__ void store_view_tape(float* tape, unsigned m, int g, const float (&vv)[8]) {
    unsigned char* base = reinterpret_cast<unsigned char*>(tape);
    const unsigned row = m * (unsigned)(TAPE_VW * sizeof(float));
#pragma unroll
    for (int j = 0; j < 8; ++j) {    // view16_slot_feature: g < 3: frequency j >> 1 of coordinate g; g = 3: d, then padding
        const int f = j >> 1;
        const int pos3 = j < 3 ? 12 + j : (j == 3 ? 15 : 24 + j);
        const int pos = g == 3 ? pos3 : 16 * (f >> 1) + 6 * (f & 1) + 2 * g + (j & 1);
        *reinterpret_cast<float*>(base + row + 4u * (unsigned)pos) = vv[j];
    }
}

// ---- encodings with the float32 library sine / cosine (what the reference's torch.sin / torch.cos round to) ------
__device__ __forceinline__ void encode_point_exact(float px, float py, float pz, int g, float (&v)[16]) {
    const float p[3] = {px, py, pz};
    const float f0 = g == 0 ? 1.0f : (g == 1 ? 4.0f : (g == 2 ? 16.0f : 64.0f));          // octaves 2 g, 2 g + 1
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        sincosf(p[m] * f0, &v[4 * m], &v[4 * m + 1]);                                     // power-of-two scaling is exact
        sincosf(p[m] * (2.0f * f0), &v[4 * m + 2], &v[4 * m + 3]);
    }
    const float pc = g == 0 ? px : (g == 1 ? py : pz);                                    // octaves 8, 9 of coordinate g
    sincosf(pc * 256.0f, &v[12], &v[13]);
    sincosf(pc * 512.0f, &v[14], &v[15]);
    if (g == 3) { v[12] = px; v[13] = py; v[14] = pz; v[15] = 0.0f; }
}

__device__ __forceinline__ void encode_view_exact(float dx, float dy, float dz, int g, float (&v)[8]) {
    const float dc = g == 0 ? dx : (g == 1 ? dy : dz);
#pragma unroll
    for (int f = 0; f < 4; ++f) sincosf(dc * (float)(1 << f), &v[2 * f], &v[2 * f + 1]);
    if (g == 3) {
        v[0] = dx; v[1] = dy; v[2] = dz;
#pragma unroll
        for (int j = 3; j < 8; ++j) v[j] = 0.0f;
    }
}

// ---- one k-step of a layer: NT output tiles ------------------------------------------------------------------------
// k: index of the k-step's first fragment in the pass (compile-time after unrolling).  first: the layer's first k-step
// (accumulators start from the bias; tile t's bias was fetched one position earlier).  bias_after: bias tile to fetch
// at the last position (the first tile of the next layer), or -1.  conv: while this k-step runs, split the NEXT
// k-step's operands (accumulator tiles src[0], src[1]) into nxt; cross: those tiles are outputs of THIS k-step (it is its
// layer's last), complete after positions 0 and 1.
template <int T, int NT, bool TAPE>
__device__ __forceinline__ void kstep(SStream<T>& st, int& k, int g, bool first, int bias_tile0, int bias_after, f32x4v& bias,
                                      const Ops<T>& cur, f32x4v (*out)[NCT], bool conv, const f32x4v (*src)[NCT],
                                      bool relu, bool cross, Ops<T>& nxt, TapeCtx& tc, const TapeDst& td, int tile0) {
    constexpr int NSL = 4 * NCT;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < T; ++j) st.tick(k + j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < T; ++j) st.fetch(k + T + j);
        f32x4v bias_next = bias;
        if (first && t + 1 < NT) bias_next = m16::bias16(st.p.lds, bias_tile0 + t + 1, g);
        if (t + 1 == NT && bias_after >= 0) bias_next = m16::bias16(st.p.lds, bias_after, g);
        // products W_i x_j, i + j < T, smallest first
        bool started = !first;
#pragma unroll
        for (int sum = T - 1; sum >= 0; --sum) {
#pragma unroll
            for (int i = sum; i >= 0; --i) {
                const bf16x8 w = __builtin_bit_cast(bf16x8, st.q[(k + i) % (2 * T)]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    out[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, cur.v[ct][sum - i]),
                                                                         started ? out[t][ct] : bias, 0, 0, 0);
                started = true;
            }
        }
        if (conv) {
            const int t0 = cross ? (NT > 2 ? 2 : NT - 1) : 0, avail = NT - t0;
            if (t >= t0) {
                const int lo = NSL * (t - t0) / avail, hi = NSL * (t - t0 + 1) / avail;
#pragma unroll
                for (int sl = lo; sl < hi; ++sl) convert_slice<T, TAPE>(src, relu, sl, nxt, tc, td, tile0);
            }
        }
        // issue order: an MFMA, then in its shadow one LDS read of the next position (T fragments and possibly a bias
        // tile: measured 1.9 % faster than reading them in front of the position) and up to two vector instructions of
        // the split
#pragma unroll
        for (int i = 0; i < NCT * T * (T + 1) / 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < T + 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        bias = bias_next;
        k += T;
    }
}

// running state of a pass: fragment position, bias tile, which operand window is current
template <int T>
struct Pass {
    int k = 0, bt = 0, wp = 0;
};

// A layer: [KPE k-steps of encoded points][KH k-steps of hidden input `in` (ReLU'd if relu_in)][KVW of encoded view] ->
// NT output tiles `out`.  What follows decides the last k-step's conversion: next_src = the accumulator tiles the next
// layer's first k-step is made of (this layer's `out`), or nullptr when that k-step is an encoding (or nothing follows).
// TAPE: td_in / td_out are the tape rows of `in` and `out`; bits_in is the ReLU-bit layer of `in` (-1: none), written
// once all of `in` has been converted, i.e. in front of the last k-step.
template <int T, int NT, int KPE, int KH, int KVW, bool TAPE>
__device__ __forceinline__ void layer(SStream<T>& st, Pass<T>& ps, int g, f32x4v& bias, const Ops<T>* pe, const Ops<T>* vw,
                                      const f32x4v (*in)[NCT], bool relu_in, f32x4v (*out)[NCT], Ops<T> (&win)[2],
                                      bool next_from_out, bool relu_out, int n_bias_tiles, TapeCtx& tc, const TapeDst& td_in,
                                      const TapeDst& td_out, int bits_in, unsigned* bits, int bits_per_sample) {
    constexpr int KS = KPE + KH + KVW;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bool hidden = s >= KPE && s < KPE + KH;
        const Ops<T>& cur = s < KPE ? pe[s] : (hidden ? win[ps.wp] : vw[s - KPE - KH]);
        // the k-step after this one
        const bool last = s + 1 == KS;
        const bool next_hidden = !last && (s + 1 >= KPE && s + 1 < KPE + KH);
        const bool conv = last ? next_from_out : next_hidden;
        const f32x4v (*src)[NCT] = last ? out : in + 2 * (s + 1 - KPE);
        const int nwp = hidden ? ps.wp ^ 1 : ps.wp;       // a hidden k-step occupies win[wp]; the next one gets the other
        const int bias_after = last ? (ps.bt + NT) % n_bias_tiles : -1;
        if (TAPE && last && KH > 0 && bits_in >= 0) flush_bits(tc, bits, bits_per_sample, bits_in);
        kstep<T, NT, TAPE>(st, ps.k, g, s == 0, ps.bt, bias_after, bias, cur, out, conv, src, last ? relu_out : relu_in, last,
                           win[nwp], tc, last ? td_out : td_in, last ? 0 : 2 * (s + 1 - KPE));
        ps.wp = nwp;
    }
    ps.bt += NT;
}

template <int T, bool VD, int OUT_CH, bool TAPE>
__device__ __forceinline__ void split_body(MlpArgs& a, const TapePtrs& tp, unsigned char* lds) {
    const TileWalk tw = tile_walk(a, (a.M + TILE - 1) / TILE);
    SStream<T> st;
    pipe_start(st.p, a, lds);
    st.slot0 = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    constexpr int NB = VD ? BIAS_TILES16_VD : BIAS_TILES16_NOVD;
    constexpr int NFRAGS = T * (VD ? FRAGS16_VD : FRAGS16_NOVD);
    constexpr int NPAD = pad_to_stage(NFRAGS);

    // the first T fragments of a pass are read ahead of it, behind the previous pass's last layer (TAPE: the encodings
    // need the registers, they are read in front of the first layer instead)
    if (!TAPE) {
#pragma unroll
        for (int j = 0; j < T; ++j) st.fetch(j);
    }
    f32x4v bias = m16::bias16(st.p.lds, 0, g);

    const bool feat_mode = a.mode == NERF_IN_FEATURES;
    const unsigned s0 = (unsigned)tw.sample0, s_end = (unsigned)tw.sample_end;
    RawIn nxt_in[NCT];
    if (!TAPE && !feat_mode && tw.first < tw.end) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = s0 + (unsigned)tw.first * TILE + wave * WSAMP + ct * 16 + c;
            fetch_raw<VD>(a, m < s_end ? m : s_end - 1, nxt_in[ct]);
        }
    }

    for (long long t = tw.first; t < tw.end; t += tw.stride) {
        const unsigned pass0 = s0 + (unsigned)t * TILE + wave * WSAMP;
        RawIn cur_in[NCT];
        if (!TAPE) {
            // the raw inputs of the next pass travel while this one computes
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) cur_in[ct] = nxt_in[ct];
#ifndef NERF_SPX_NOIN      // timing experiment only (results are garbage): what fetching the next pass's inputs costs
            if (!feat_mode && t + tw.stride < tw.end) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const unsigned m = s0 + (unsigned)(t + tw.stride) * TILE + wave * WSAMP + ct * 16 + c;
                    fetch_raw<VD>(a, m < s_end ? m : s_end - 1, nxt_in[ct]);
                }
            }
#endif
        } else if (!feat_mode) {
            // TAPE: fetched when needed.  Twenty registers held across the layer loop for the next pass's inputs do not
            // fit beside the tape addressing; the exposed latency is about 1 % of a pass
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const unsigned m = pass0 + ct * 16 + c;
                fetch_raw<VD>(a, m < s_end ? m : s_end - 1, cur_in[ct]);
            }
        }
        Ops<T> pe[2], vw[1];
        TapeCtx tc;
        // TAPE: everything the tape addresses derive from the lane group is recomputed per pass (a few instructions)
        // from a copy of g the compiler cannot see through; hoisted out of the pass loop these lane constants would
        // occupy some forty registers across the layer loop and end up in scratch memory
        int gt = g;
        if (TAPE) asm volatile("" : "+v"(gt));
        TapeSlots slots;
        if (TAPE) slots = tape_slots(gt);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = pass0 + ct * 16 + c;
            float pv[16], vv[8];
            if (feat_mode) {
                int gf = g;      // TAPE: keep the 24 column indices (functions of g) from being hoisted out of the pass
                if (TAPE) asm volatile("" : "+v"(gf));      // loop, where they would be held in scratch memory
                m16::load_features16<VD>(a, m < s_end ? m : s_end - 1, gf, pv, vv);
            } else {
                const RawIn& r = cur_in[ct];
#ifdef NERF_SPX_NOENC      // timing experiment only (results are garbage): what the sine / cosine evaluation costs
#pragma unroll
                for (int q = 0; q < 16; ++q) pv[q] = r.o[q % 3] + r.z;
#pragma unroll
                for (int q = 0; q < 8; ++q) vv[q] = r.v[q % 3];
                if (false)
#endif
                // pts = rays_o + rays_d * z (run_nerf.py:387): product and sum rounded separately
                encode_point_exact(fadd(r.o[0], fmul(r.d[0], r.z)), fadd(r.o[1], fmul(r.d[1], r.z)),
                                   fadd(r.o[2], fmul(r.d[2], r.z)), g, pv);
                if (VD) encode_view_exact(r.v[0], r.v[1], r.v[2], g, vv);
            }
            if (TAPE) {      // rows exist for whole 128-sample tiles: every lane stores
                store_pe_tape(tp.pe, m, slots, pv);
                if (VD) store_view_tape(tp.vw, m, gt, vv);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned w[T];
                    split_pair<T>(pv[ks * 8 + 2 * j], pv[ks * 8 + 2 * j + 1], false, w);
#pragma unroll
                    for (int tt = 0; tt < T; ++tt) pe[ks].v[ct][tt][j] = w[tt];
                }
            if (VD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned w[T];
                    split_pair<T>(vv[2 * j], vv[2 * j + 1], false, w);
#pragma unroll
                    for (int tt = 0; tt < T; ++tt) vw[0].v[ct][tt][j] = w[tt];
                }
            }
        }

        f32x4v FA[17][NCT], FB[16][NCT];
        Ops<T> win[2];
        Pass<T> ps;
        if (TAPE) {
            __builtin_amdgcn_sched_barrier(0);
            tc.g16 = 16u * gt;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                tc.rowb[ct] = (pass0 + ct * 16 + c) << 10;
                tc.nib[ct][0] = tc.nib[ct][1] = 0u;
            }
#pragma unroll
            for (int j = 0; j < T; ++j) st.fetch(j);
        }
        constexpr int NBITS = tape_bits(VD);
        const TapeDst none{nullptr, 256};
#define NERF_TD(p) (TAPE ? TapeDst{p, 256} : none)
#define NERF_SL(NT, KPE, KH, KVW, in, relu_in, out, next_from_out, relu_out, td_in, td_out, bits_in) \
    layer<T, NT, KPE, KH, KVW, TAPE>(st, ps, g, bias, pe, vw, in, relu_in, out, win, next_from_out, relu_out, NB, tc, td_in, \
                                     td_out, bits_in, tp.bits, NBITS)
        NERF_SL(16, 2, 0, 0, FA, false, FA, true, true, none, NERF_TD(tp.h[0]), -1);             // L0: pe -> FA
        NERF_SL(16, 0, 8, 0, FA, true, FB, true, true, NERF_TD(tp.h[0]), NERF_TD(tp.h[1]), 0);   // L1
        NERF_SL(16, 0, 8, 0, FB, true, FA, true, true, NERF_TD(tp.h[1]), NERF_TD(tp.h[2]), 1);   // L2
        NERF_SL(16, 0, 8, 0, FA, true, FB, true, true, NERF_TD(tp.h[2]), NERF_TD(tp.h[3]), 2);   // L3
        NERF_SL(16, 0, 8, 0, FB, true, FA, false, true, NERF_TD(tp.h[3]), none, 3);              // L4 -> FA; L5 starts with the encoded points
        NERF_SL(16, 2, 8, 0, FA, true, FB, true, true, NERF_TD(tp.h[4]), NERF_TD(tp.h[5]), 4);   // L5: [pe, h4]
        NERF_SL(16, 0, 8, 0, FB, true, FA, true, true, NERF_TD(tp.h[5]), NERF_TD(tp.h[6]), 5);   // L6
        NERF_SL(16, 0, 8, 0, FA, true, FB, true, true, NERF_TD(tp.h[6]), NERF_TD(tp.h[7]), 6);   // L7 -> FB
        float o[NCT][5];
        if (VD) {
            // feature (tiles 0..15, no ReLU) and density (tile 16, row 3)
            NERF_SL(17, 0, 8, 0, FB, true, FA, true, false, NERF_TD(tp.h[7]), NERF_TD(tp.feat), 7);
            // views: [feature, view] -> FB[0..7]
            NERF_SL(8, 0, 8, 1, FA, false, FB, true, true, NERF_TD(tp.feat), (TAPE ? TapeDst{tp.hv, 128} : none), -1);
            // rgb: rows 0..2 of FA[0]
            NERF_SL(1, 0, 4, 0, FB, true, FA, false, false, (TAPE ? TapeDst{tp.hv, 128} : none), none, 8);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                o[ct][0] = FA[0][ct][0]; o[ct][1] = FA[0][ct][1]; o[ct][2] = FA[0][ct][2]; o[ct][3] = FA[16][ct][3];
                o[ct][4] = 0.0f;
            }
        } else {
            // output_linear: rows 0..3 (4: register 0 of lane group 1)
            NERF_SL(1, 0, 8, 0, FB, true, FA, false, false, NERF_TD(tp.h[7]), none, 7);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                o[ct][0] = FA[0][ct][0]; o[ct][1] = FA[0][ct][1]; o[ct][2] = FA[0][ct][2]; o[ct][3] = FA[0][ct][3];
                o[ct][4] = FA[0][ct][0];
            }
        }
#undef NERF_SL
#undef NERF_TD
        // zero padding at the end of the stream: nothing to compute, the ring keeps turning
#pragma unroll
        for (int k = NFRAGS; k < NPAD; ++k) st.tick(k);
        __builtin_amdgcn_sched_barrier(0);
        st.end_pass(NPAD);
        if (!TAPE) {
#pragma unroll
            for (int j = 0; j < T; ++j) st.fetch(j);  // the first fragments of the next pass (queue slots 0 .. T - 1)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = pass0 + ct * 16 + c;
            if (m < s_end) {
                if (OUT_CH == 4) {
                    if (g == 0) *reinterpret_cast<float4*>(a.out + (size_t)m * 4) = make_float4(o[ct][0], o[ct][1], o[ct][2], o[ct][3]);
                } else {
                    float* q = a.out + (size_t)m * OUT_CH;
                    if (g == 0) { q[0] = o[ct][0]; q[1] = o[ct][1]; q[2] = o[ct][2]; q[3] = o[ct][3]; }
                    else if (g == 1) { q[4] = o[ct][4]; }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

template <int T, bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_split_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    split_body<T, VD, OUT_CH, false>(a, TapePtrs{}, lds);
}

}  // namespace sp
}  // namespace nerf
