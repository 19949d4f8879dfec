// Fused NeRF MLP for gfx950: positional encoding + 8x256 skip MLP + heads in ONE kernel.
//
// Replaces Embedder.embed (run_nerf_helpers.py:44-45), run_network / batchify
// (run_nerf.py:27-51) and NeRF.forward (run_nerf_helpers.py:96-119).
//
// Design (see mlp_layout.h for the operand conventions):
//   * a wavefront owns 64 (bf16) or 32 (f32) samples and carries them through all 11 matrix
//     products; activations live in VGPR/AGPRs the whole time: the accumulator tile of one
//     layer IS the B operand of the next (bf16: after relu + v_cvt_pk_bf16_f32);
//   * weights are the A operand.  They are streamed as 1 KiB fragments through an 8 x 16 KiB
//     LDS ring by global_load_lds (LDS-DMA, no VGPR staging), prefetched PREFETCH stages ahead
//     behind a counted s_waitcnt vmcnt(N) and one raw s_barrier per stage; the 4 waves of the
//     workgroup share every fragment (ds_read_b128, lane-linear, conflict-free);
//   * persistent grid (one 256-thread workgroup per CU, 512 registers per lane): the ring
//     keeps streaming across sample tiles, the stream wraps around at the end of the network;
//   * positions (o + d z), sin/cos features and the per-ray view encoding are computed in
//     registers from 44 B/ray + 4 B/sample; nothing but raw[.,4] is written back.
// Bound: MFMA.  1 186 816 algorithmic FLOP per sample (1 212 416 issued with padding).
#include "mlp_device.h"

namespace nerf {


// ------------------------------------------------------------------------------------------
// bf16 kernel: 2 column tiles (64 samples) per wave, software-pipelined by hand:
//   * weight fragments are read from LDS PF k-steps ahead into a 4-deep register queue;
//   * the epilogue of an output tile (relu + v_cvt_pk_bf16_f32 into the next layer's operand
//     registers) is cut into 16 slices that are issued under the MFMAs of the NEXT tile;
//   * the bias of the next tile is fetched half a tile ahead.
// The running position `pos` in the fragment stream is a compile-time constant everywhere
// (everything below is fully unrolled), only ring slots and addresses are run-time values.
// ------------------------------------------------------------------------------------------
constexpr int NCT = 2;
#ifndef NERF_PF
#define NERF_PF 3
#endif
constexpr int PF = NERF_PF;                  // LDS read-ahead in fragments
constexpr int QD = PF + 1 <= 4 ? 4 : 8;      // queue depth (power of two > PF)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
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

// slice sl (0..15) of the epilogue of a finished tile: two accumulator registers -> one packed
// operand dword of the next layer.  dst = &h[2t]: rows 2t <- registers 0-7, 2t+1 <- registers 8-15.
__device__ __forceinline__ void epi_slice(const f32x16 (&accp)[NCT], u32x4 (*dst)[NCT], bool relu, int sl) {
    const int g = sl >> 2, qd = sl & 3, ct = g >> 1, half = g & 1;
    dst[half][ct][qd] = pack_pair(accp[ct][8 * half + 2 * qd], accp[ct][8 * half + 2 * qd + 1], relu);
}

// One 32-row output tile: k-steps [pe KPE][hidden KH][view KVW] accumulate into `acc`, while the
// epilogue of the previous tile (accumulators `accp`, destination pdst) is issued, if has_prev.
// The two accumulator sets alternate from tile to tile, so nothing is ever copied.
template <int KPE, int KH, int KVW, bool early>
__device__ __forceinline__ void tile_bf16(Stream& st, int pos, int h, int bias_next_idx, f32x16& bias,
                                          const u32x4 (*pe)[NCT], const u32x4 (*hin)[NCT], const u32x4 (*vw)[NCT],
                                          f32x16 (&acc)[NCT], const f32x16 (&accp)[NCT], u32x4 (*pdst)[NCT],
                                          bool prelu, bool has_prev) {
    constexpr int KS = KPE + KH + KVW;
    // The 16 epilogue slices of the previous tile are spread over k-steps 1..EPI_N (the previous
    // tile's last MFMA result is not readable for a few cycles, hence not k-step 0).  When this
    // tile reads what the epilogue writes (first tile of a layer: rows 14, 15 of its input) it must
    // be done early, otherwise it may take the whole tile: at most one slice per MFMA gap.
    constexpr int EPI_N = early ? (KS >= 16 ? 12 : KS / 2) : KS - 1;
    f32x16 bias_next = bias;
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
        if (s == KS / 2) bias_next = bias_tile(st.p.lds, bias_next_idx, h);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4* b = s < KPE ? pe[s] : (s < KPE + KH ? hin[s - KPE] : vw[s - KPE - KH]);
        const bf16x8 w = __builtin_bit_cast(bf16x8, st.q[k & (QD - 1)]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, __builtin_bit_cast(bf16x8, b[ct]),
                                                             s == 0 ? bias : acc[ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_prev && s >= 1 && s <= EPI_N) {
                const int lo = (16 * (s - 1) + EPI_N - 1) / EPI_N, hi = (16 * s + EPI_N - 1) / EPI_N;
                const int cnt = (hi < 16 ? hi : 16) - lo, first = cnt / 2;   // ct0 gap: first, ct1 gap: the rest
                const int b0 = ct == 0 ? lo : lo + first, b1 = ct == 0 ? lo + first : lo + cnt;
#pragma unroll
                for (int sl = b0; sl < b1; ++sl) epi_slice(accp, pdst, prelu, sl);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    bias = bias_next;
}

// NT tiles of one layer.  (prev_dst, prev_relu, has_prev) describe the tile that ran just before
// this layer; tile number bt of the pass accumulates into accs[bt & 1]; on return the last tile of this
// layer is pending there (destination hout + 2(NT-1)).
template <int NT, int KPE, int KH, int KVW>
__device__ __forceinline__ void layer_bf16(Stream& st, int& pos, int& bt, int n_bias_tiles, int h, f32x16& bias,
                                           const u32x4 (*pe)[NCT], const u32x4 (*hin)[NCT], const u32x4 (*vw)[NCT],
                                           f32x16 (&accs)[2][NCT], u32x4 (*prev_dst)[NCT], bool prev_relu, bool has_prev,
                                           u32x4 (*hout)[NCT], bool relu) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t == 0)
            tile_bf16<KPE, KH, KVW, true>(st, pos, h, (bt + 1) % n_bias_tiles, bias, pe, hin, vw, accs[bt & 1],
                                          accs[(bt & 1) ^ 1], prev_dst, prev_relu, has_prev);
        else
            tile_bf16<KPE, KH, KVW, false>(st, pos, h, (bt + 1) % n_bias_tiles, bias, pe, hin, vw, accs[bt & 1],
                                           accs[(bt & 1) ^ 1], hout + 2 * (t - 1), relu, true);
        pos += KPE + KH + KVW;
        bt += 1;
    }
}

#ifdef NERF_EXP_STAMP
// diagnostic build only: per-block cycle sums {input+encoding, layers 0-4, layer 5-7, heads, store, passes}
__device__ unsigned long long g_stamps[256 * 8];
#define NERF_STAMP(var)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);
#else
#define NERF_STAMP(var)
#endif

template <bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_bf16_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    Stream st;
    pipe_start(st.p, a, lds);
    st.slot0 = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const long long n_tiles = (a.M + 255) / 256;
    constexpr int NB = VD ? BIAS_TILES_VD : BIAS_TILES_NOVD;
    constexpr int NFRAGS = VD ? FRAGS_BF16_VD : FRAGS_BF16_NOVD;

#pragma unroll
    for (int k = 0; k < PF; ++k) st.fetch(k);
    f32x16 bias = bias_tile(st.p.lds, 0, h);

    // raw inputs are fetched one whole pass ahead: with a single wave per SIMD nothing else could
    // hide the HBM latency of a load issued where its value is needed
    const bool feat_mode = a.mode == NERF_IN_FEATURES;
    const unsigned Mu = (unsigned)a.M;
    RawIn nxt[NCT];
    if (!feat_mode && (long long)blockIdx.x < n_tiles) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = (unsigned)blockIdx.x * 256u + wave * 64 + ct * 32 + c;
            fetch_raw<VD>(a, m < Mu ? m : Mu - 1, nxt[ct]);
        }
    }

#ifdef NERF_EXP_STAMP
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, acc_t[6] = {0, 0, 0, 0, 0, 0};
#endif
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        NERF_STAMP(t0)
        u32x4 pe[4][NCT], vw[2][NCT];
        unsigned m_of[NCT];
        RawIn cur[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) cur[ct] = nxt[ct];
        if (!feat_mode && tile + gridDim.x < n_tiles) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const unsigned m = (unsigned)(tile + gridDim.x) * 256u + wave * 64 + ct * 32 + c;
                fetch_raw<VD>(a, m < Mu ? m : Mu - 1, nxt[ct]);
            }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = (unsigned)tile * 256u + wave * 64 + ct * 32 + c;
            m_of[ct] = m;
            float pv[32], vv[16];
            if (feat_mode) load_features<VD>(a, m < Mu ? m : Mu - 1, h, pv, vv);
            else encode_raw<VD, true>(cur[ct], h, pv, vv);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) pe[ks][ct][j] = pack_pair(pv[ks * 8 + 2 * j], pv[ks * 8 + 2 * j + 1], false);
            if (VD) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j) vw[ks][ct][j] = pack_pair(vv[ks * 8 + 2 * j], vv[ks * 8 + 2 * j + 1], false);
            }
        }

        NERF_STAMP(t1)
        u32x4 A[16][NCT], B[16][NCT];
        f32x16 accp[2][NCT];     // accumulators of the current and of the previous output tile
        int pos = 0, bt = 0;
        layer_bf16<8, 4, 0, 0>(st, pos, bt, NB, h, bias, pe, A, vw, accp, A, false, false, A, true);         // L0: pe -> A
        layer_bf16<8, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, A, vw, accp, A + 14, true, true, B, true);     // L1
        layer_bf16<8, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, B, vw, accp, B + 14, true, true, A, true);     // L2
        layer_bf16<8, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, A, vw, accp, A + 14, true, true, B, true);     // L3
        layer_bf16<8, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, B, vw, accp, B + 14, true, true, A, true);     // L4
        NERF_STAMP(t2)
        layer_bf16<8, 4, 16, 0>(st, pos, bt, NB, h, bias, pe, A, vw, accp, A + 14, true, true, B, true);     // L5: [pe, h]
        layer_bf16<8, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, B, vw, accp, B + 14, true, true, A, true);     // L6
        layer_bf16<8, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, A, vw, accp, A + 14, true, true, B, true);     // L7 -> B

        NERF_STAMP(t3)
        float o0[NCT], o1[NCT], o2[NCT], o3[NCT], o4[NCT];
        if (VD) {
            layer_bf16<8, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, B, vw, accp, B + 14, true, true, A, false);   // feature -> A (no relu)
            layer_bf16<1, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, B, vw, accp, A + 14, false, true, A, false);  // alpha: row 3
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) o3[ct] = accp[(bt - 1) & 1][ct][3];
            layer_bf16<4, 0, 16, 2>(st, pos, bt, NB, h, bias, pe, A, vw, accp, A, false, false, B, true);       // views -> B[0..7]
            layer_bf16<1, 0, 8, 0>(st, pos, bt, NB, h, bias, pe, B, vw, accp, B + 6, true, true, B, false);     // rgb: rows 0..2
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x16& r = accp[(bt - 1) & 1][ct];
                o0[ct] = r[0]; o1[ct] = r[1]; o2[ct] = r[2]; o4[ct] = 0.f;
            }
        } else {
            layer_bf16<1, 0, 16, 0>(st, pos, bt, NB, h, bias, pe, B, vw, accp, B + 14, true, true, B, false);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x16& r = accp[(bt - 1) & 1][ct];
                o0[ct] = r[0]; o1[ct] = r[1]; o2[ct] = r[2]; o3[ct] = r[3];
                o4[ct] = r[0];   // row 4 sits in register 0 of the upper lane half
            }
        }
        st.end_pass(NFRAGS);
        NERF_STAMP(t4)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = m_of[ct];
            if (m < Mu) {
                if (OUT_CH == 4) {
                    if (h == 0) *reinterpret_cast<float4*>(a.out + (size_t)m * 4) = make_float4(o0[ct], o1[ct], o2[ct], o3[ct]);
                } else {
                    float* o = a.out + (size_t)m * OUT_CH;
                    if (h == 0) { o[0] = o0[ct]; o[1] = o1[ct]; o[2] = o2[ct]; o[3] = o3[ct]; }
                    else { o[4] = o4[ct]; }
                }
            }
        }
#ifdef NERF_EXP_STAMP
        NERF_STAMP(t5)
        acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += t5 - t4; acc_t[5] += 1;
#endif
    }
#ifdef NERF_EXP_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 256)
        for (int i = 0; i < 6; ++i) g_stamps[blockIdx.x * 8 + i] = acc_t[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}


template <bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_f32_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    Pipe p;
    pipe_start(p, a, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const long long n_tiles = (a.M + 127) / 128;

    const bool feat_mode = a.mode == NERF_IN_FEATURES;
    const unsigned Mu = (unsigned)a.M;
    RawIn nxt;
    if (!feat_mode && (long long)blockIdx.x < n_tiles) {
        const unsigned m0 = (unsigned)blockIdx.x * 128u + wave * 32 + c;
        fetch_raw<VD>(a, m0 < Mu ? m0 : Mu - 1, nxt);
    }

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned m = (unsigned)tile * 128u + wave * 32 + c;
        const RawIn cur = nxt;
        if (!feat_mode && tile + gridDim.x < n_tiles) {
            const unsigned mn = (unsigned)(tile + gridDim.x) * 128u + wave * 32 + c;
            fetch_raw<VD>(a, mn < Mu ? mn : Mu - 1, nxt);
        }
        float pe[32], vw[16];
        if (feat_mode) load_features<VD>(a, m < Mu ? m : Mu - 1, h, pe, vw);
        else encode_raw<VD, false>(cur, h, pe, vw);

        float A[8][16], B[8][16];
        int pos = 0, bt = 0;
        layer_f32<8, true, 0, false, true>(p, pos, bt, h, pe, A, vw, A);   // L0
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, A, vw, B);  // L1
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, B, vw, A);  // L2
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, A, vw, B);  // L3
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, B, vw, A);  // L4
        layer_f32<8, true, 8, false, true>(p, pos, bt, h, pe, A, vw, B);   // L5
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, B, vw, A);  // L6
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, A, vw, B);  // L7 -> B

        float o0, o1, o2, o3, o4 = 0.f;
        if (VD) {
            layer_f32<8, false, 8, false, false>(p, pos, bt, h, pe, B, vw, A);   // feature -> A
            {
                f32x16 acc = bias_tile(p.lds, bt, h);
                mma_run_f32<32>(p, pos, acc, &B[0][0]);
                pos += 32; bt += 1;
                o3 = acc[3];
            }
            layer_f32<4, false, 8, true, true>(p, pos, bt, h, pe, A, vw, B);     // views -> B[0..3]
            {
                f32x16 acc = bias_tile(p.lds, bt, h);
                mma_run_f32<16>(p, pos, acc, &B[0][0]);
                pos += 16; bt += 1;
                o0 = acc[0]; o1 = acc[1]; o2 = acc[2];
            }
        } else {
            f32x16 acc = bias_tile(p.lds, bt, h);
            mma_run_f32<32>(p, pos, acc, &B[0][0]);
            pos += 32; bt += 1;
            o0 = acc[0]; o1 = acc[1]; o2 = acc[2]; o3 = acc[3]; o4 = acc[0];
        }
        if (m < Mu) {
            if (OUT_CH == 4) {
                if (h == 0) *reinterpret_cast<float4*>(a.out + (size_t)m * 4) = make_float4(o0, o1, o2, o3);
            } else {
                float* o = a.out + (size_t)m * OUT_CH;
                if (h == 0) { o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; }
                else { o[4] = o4; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// weight packer
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ const float* pack_row(const PackArgs& a, const LayerDesc& L, int tile, int r, bool bias, int& row) {
    row = -1;
    switch (L.kind) {
        case LK_HIDDEN:
            row = tile * 32 + r;
            if (row >= L.out_dim) { row = -1; return nullptr; }
            return a.t[bias ? L.b_idx : L.w_idx];
        case LK_FEATURE_ALPHA:
            if (tile < 8) { row = tile * 32 + r; return a.t[bias ? L.b_idx : L.w_idx]; }
            if (r == 3) { row = 0; return a.t[bias ? L.b2_idx : L.w2_idx]; }
            return nullptr;
        default:   // LK_RGB, LK_OUTPUT: rows 0..out_dim-1 of a single tile
            if (r < L.out_dim) { row = r; return a.t[bias ? L.b_idx : L.w_idx]; }
            return nullptr;
    }
}

// Output scales of the bf16 network, one block.  B_l bounds |pre-activation| of layer l for any input with
// |feature| <= INPUT_BOUND: B_l = max_row (|b| + sum_j |W_j| * bound of input j).  The stored activations are
// s_l * relu(.) with s_l = 2^-ceil(log2(2 B_l)) < 1 / (2 B_l): the factor 2 covers the bf16 rounding of weights
// and activations.  Powers of two commute with every rounding step, so the network's results do not change.
__global__ __launch_bounds__(256) void scale_kernel(PackArgs a, float* scales) {
    __shared__ float red[256];
    __shared__ float bound[MAX_LAYERS];
    for (int li = 0; li < a.net.n_layers; ++li) {
        const LayerDesc& L = a.net.layer[li];
        const float bin = L.src_layer >= 0 ? bound[L.src_layer] : 0.0f;
        float m = 0.0f;
        for (int row = threadIdx.x; row < L.out_dim; row += 256) {
            const float* W = a.t[L.w_idx] + (size_t)row * L.in_dim;
            float sum = fabsf(a.t[L.b_idx][row]);
            for (int col = 0; col < L.in_dim; ++col) {
                const bool hid = col >= L.hid_col && col < L.hid_col + L.hid_feats;
                sum += fabsf(W[col]) * (hid ? bin : INPUT_BOUND);
            }
            m = fmaxf(m, sum);
        }
        red[threadIdx.x] = m;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + off]);
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float B = fmaxf(red[0], 1e-30f);
            bound[li] = B;
            scales[li] = (a.bf16 && L.scaled_out) ? exp2f(-ceilf(log2f(2.0f * B))) : 1.0f;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int F = gid >> 6, lane = gid & 63;
    const int r = lane & 31, h = lane >> 5;
    if (F >= a.total_frags && F < a.padded_frags) {
        *reinterpret_cast<float4*>(a.out + (size_t)F * FRAG_BYTES + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    if (F < a.total_frags) {
        int f = F, li = 0;
        for (; li < a.net.n_layers; ++li) {
            const int nf = a.net.layer[li].n_tiles * frags_per_tile(a.net.layer[li], a.bf16);
            if (f < nf) break;
            f -= nf;
        }
        const LayerDesc& L = a.net.layer[li];
        const int fpt = frags_per_tile(L, a.bf16);
        const int tile = f / fpt, g = f - tile * fpt;
        int row;
        const float* W = pack_row(a, L, tile, r, false, row);
        const int n_el = a.bf16 ? 8 : 4;
        // bf16: weights carry (scale of the tile's output) / (scale of the k part's input)
        const float* scales = reinterpret_cast<const float*>(a.out + (size_t)a.padded_frags * FRAG_BYTES +
                                                             (size_t)a.net.n_bias_tiles * BIAS_TILE_BYTES);
        float s_out = 1.0f, s_hid = 1.0f;
        if (a.bf16) {
            const bool head_row = L.kind == LK_FEATURE_ALPHA && tile == 8;
            s_out = (L.scaled_out && !head_row) ? scales[li] : 1.0f;
            s_hid = L.src_layer >= 0 ? scales[L.src_layer] : 1.0f;
        }
        float vals[8];
        for (int j = 0; j < n_el; ++j) {
            int col = -1;
            bool extra = false;
            if (a.bf16) {
                const int npe = L.has_pe ? 4 : 0, nh = L.hid_feats / 16;
                if (g < npe) { const int ft = pe_slot_feature(h, 8 * g + j); col = ft < 0 ? -1 : L.pe_col + ft; }
                else if (g < npe + nh) { const int hs = g - npe; col = L.hid_col + 16 * hs + 8 * (j >> 2) + 4 * h + (j & 3); }
                else { const int ft = view_slot_feature(h, 8 * (g - npe - nh) + j); col = ft < 0 ? -1 : L.view_col + ft; }
            } else {
                const int npe = L.has_pe ? 8 : 0, nh = L.hid_feats / 8;
                if (g < npe) { const int ft = pe_slot_feature(h, 4 * g + j); col = ft < 0 ? -1 : L.pe_col + ft; }
                else if (g < npe + nh) {
                    const int gh = g - npe, T = gh >> 2, e = (gh & 3) * 4 + j;
                    col = L.hid_col + 32 * T + (e & 3) + 8 * (e >> 2) + 4 * h;
                } else if (g < npe + nh + (L.has_view ? 4 : 0)) {
                    const int ft = view_slot_feature(h, 4 * (g - npe - nh) + j); col = ft < 0 ? -1 : L.view_col + ft;
                } else {
                    const int e = (g - npe - nh - (L.has_view ? 4 : 0)) * 4 + j;
                    col = (e & 3) + 8 * (e >> 2) + 4 * h;
                    extra = true;
                }
            }
            if (L.trans) {
                // transposed read: layer row -> tensor column, k index -> tensor row
                if (extra) {
                    const float* W2 = a.t[L.w2_idx];
                    vals[j] = (row >= 0 && col < L.k_limit2) ? W2[(size_t)col * L.in_dim2 + row] : 0.0f;
                } else {
                    vals[j] = (W && col >= 0 && col < L.k_limit) ? W[(size_t)col * L.in_dim + L.row_off + row] : 0.0f;
                }
                continue;
            }
            vals[j] = (W && col >= 0) ? W[(size_t)row * L.in_dim + col] : 0.0f;
            if (a.bf16) {
                const int npe = L.has_pe ? 4 : 0, nh = L.hid_feats / 16;
                vals[j] *= (g >= npe && g < npe + nh) ? s_out / s_hid : s_out;
            }
        }
        unsigned char* dst = a.out + (size_t)F * FRAG_BYTES + lane * 16;
        if (a.bf16) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)vals[j];
            *reinterpret_cast<bf16x8*>(dst) = v;
        } else {
            *reinterpret_cast<float4*>(dst) = make_float4(vals[0], vals[1], vals[2], vals[3]);
        }
    } else {
        // bias block: n_bias_tiles x 32 floats in natural row order
        const int bi = gid - a.padded_frags * 64;
        if (bi >= a.net.n_bias_tiles * 32) return;
        int t = bi >> 5, li = 0;
        const int rr = bi & 31;
        for (; li < a.net.n_layers; ++li) {
            if (t < a.net.layer[li].n_tiles) break;
            t -= a.net.layer[li].n_tiles;
        }
        int row;
        const float* Bp = pack_row(a, a.net.layer[li], t, rr, true, row);
        float v = Bp ? Bp[row] : 0.0f;
        if (a.bf16) {
            const LayerDesc& L = a.net.layer[li];
            const float* scales = reinterpret_cast<const float*>(a.out + (size_t)a.padded_frags * FRAG_BYTES +
                                                                 (size_t)a.net.n_bias_tiles * BIAS_TILE_BYTES);
            const bool head_row = L.kind == LK_FEATURE_ALPHA && t == 8;
            if (L.scaled_out && !head_row) v *= scales[li];
        }
        reinterpret_cast<float*>(a.out + (size_t)a.padded_frags * FRAG_BYTES)[bi] = v;
    }
}

// ------------------------------------------------------------------------------------------
// MFMA lane-map self test (exact small integers, asymmetric operands)
// ------------------------------------------------------------------------------------------
__global__ void selftest_kernel(int* bad) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    auto Aval = [](int i, int k) { return (float)((i * 3 + k * 5) % 7 - 3); };
    auto Bval = [](int k, int j) { return (float)((k * 2 + j * 7) % 5 - 2); };
    int nbad = 0;
    {   // bf16 32x32x16: A[row r][k = 8h + j], B[k = 8h + j][col r]
        bf16x8 fa, fb;
        for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)Aval(r, 8 * h + j); fb[j] = (__bf16)Bval(8 * h + j, r); }
        f32x16 acc;
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h, col = r;
            float ref = 0.f;
            for (int k = 0; k < 16; ++k) ref += Aval(row, k) * Bval(k, col);
            if (acc[e] != ref) ++nbad;
        }
    }
    {   // f32 32x32x2: A[i = r][k = h], B[k = h][j = r]
        f32x16 acc;
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Aval(r, h), Bval(h, r), acc, 0, 0, 0);
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h, col = r;
            const float ref = Aval(row, 0) * Bval(0, col) + Aval(row, 1) * Bval(1, col);
            if (acc[e] != ref) ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int launch_pack(const PackArgs& a, hipStream_t stream) {
    if (a.bf16) {   // output scales first: the pack kernel reads them from behind the bias block
        float* scales = reinterpret_cast<float*>(a.out + (size_t)a.padded_frags * FRAG_BYTES +
                                                 (size_t)a.net.n_bias_tiles * BIAS_TILE_BYTES);
        hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(256), 0, stream, a, scales);
        NERF_LAUNCH_CHECK();
    }
    const int threads = a.padded_frags * 64 + a.net.n_bias_tiles * 32;
    hipLaunchKernelGGL(pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

}  // namespace nerf

using namespace nerf;

#ifdef NERF_EXP_STAMP
extern "C" int nerf_debug_read_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int nerf_selftest_mfma(void) {
    int* d = nullptr;
    NERF_HIP_CHECK(hipMalloc(&d, sizeof(int)));
    NERF_HIP_CHECK(hipMemset(d, 0, sizeof(int)));
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, d);
    NERF_LAUNCH_CHECK();
    int hbad = -1;
    NERF_HIP_CHECK(hipMemcpy(&hbad, d, sizeof(int), hipMemcpyDeviceToHost));
    NERF_HIP_CHECK(hipFree(d));
    if (hbad != 0) {
        set_error("MFMA lane-map self test: %d mismatching accumulator elements", hbad);
        return NERF_E_HIP;
    }
    return NERF_OK;
}

static int frag_total(int precision, int use_viewdirs) {
    if (precision == NERF_PREC_BF16) return use_viewdirs ? FRAGS_BF16_VD : FRAGS_BF16_NOVD;
    return use_viewdirs ? FRAGS_F32_VD : FRAGS_F32_NOVD;
}

extern "C" size_t nerf_mlp_packed_bytes(int precision, int use_viewdirs) {
    if (precision != NERF_PREC_BF16 && precision != NERF_PREC_F32) return 0;
    return (size_t)frag_total(precision, use_viewdirs) * FRAG_BYTES +
           (size_t)(use_viewdirs ? BIAS_TILES_VD : BIAS_TILES_NOVD) * BIAS_TILE_BYTES + SCALE_BYTES;
}

extern "C" int nerf_mlp_pack(const float* const* tensors_host, int use_viewdirs, int output_ch, int precision,
                             void* packed_out, void* stream) {
    NERF_REQUIRE(tensors_host && packed_out, "nerf_mlp_pack: null pointer");
    NERF_REQUIRE(precision == NERF_PREC_BF16 || precision == NERF_PREC_F32, "nerf_mlp_pack: unknown precision %d", precision);
    if (use_viewdirs) NERF_REQUIRE(output_ch == 4, "nerf_mlp_pack: with view directions the output is [rgb, sigma] (4 channels)");
    else NERF_REQUIRE(output_ch == 4 || output_ch == 5, "nerf_mlp_pack: output_ch must be 4 or 5, got %d", output_ch);
    PackArgs a;
    a.net = make_net_desc(use_viewdirs != 0, output_ch);
    a.bf16 = precision == NERF_PREC_BF16;
    a.total_frags = total_frags(a.net, a.bf16);
    NERF_REQUIRE(a.total_frags == frag_total(precision, use_viewdirs) &&
                     a.net.n_bias_tiles == (use_viewdirs ? BIAS_TILES_VD : BIAS_TILES_NOVD),
                 "nerf_mlp_pack: internal layout mismatch");
    const int n_t = use_viewdirs ? 24 : 20;
    for (int i = 0; i < 24; ++i) a.t[i] = i < n_t ? tensors_host[i] : nullptr;
    for (int i = 0; i < n_t; ++i)
        if (use_viewdirs || (i != 16 && i != 17)) NERF_REQUIRE(a.t[i], "nerf_mlp_pack: tensor %d is null", i);
    a.out = static_cast<unsigned char*>(packed_out);
    a.padded_frags = a.total_frags;
    return launch_pack(a, as_stream(stream));
}

extern "C" int nerf_mlp_forward(const void* packed, int precision, int use_viewdirs, int output_ch, int input_mode,
                                const float* in0, const float* in1, int R, int S, int ray_cols, float* raw_out,
                                void* stream) {
    NERF_REQUIRE(packed && (R == 0 || (in0 && raw_out)), "nerf_mlp_forward: null pointer");
    NERF_REQUIRE(precision == NERF_PREC_BF16 || precision == NERF_PREC_F32, "nerf_mlp_forward: unknown precision %d", precision);
    NERF_REQUIRE(R >= 0 && S >= 1, "nerf_mlp_forward: bad shape R=%d S=%d", R, S);
    if (use_viewdirs) NERF_REQUIRE(output_ch == 4, "nerf_mlp_forward: output_ch must be 4 with view directions");
    else NERF_REQUIRE(output_ch == 4 || output_ch == 5, "nerf_mlp_forward: output_ch must be 4 or 5");
    switch (input_mode) {
        case NERF_IN_FEATURES: NERF_REQUIRE(S == 1, "nerf_mlp_forward: feature input is flat (S = 1)"); break;
        case NERF_IN_POINTS: NERF_REQUIRE(R == 0 || !use_viewdirs || in1, "nerf_mlp_forward: view directions missing"); break;
        case NERF_IN_RAYS:
            NERF_REQUIRE(R == 0 || in1, "nerf_mlp_forward: z_vals missing");
            NERF_REQUIRE(ray_cols >= (use_viewdirs ? 11 : 8), "nerf_mlp_forward: rays need %d columns, got %d",
                         use_viewdirs ? 11 : 8, ray_cols);
            break;
        default: NERF_REQUIRE(false, "nerf_mlp_forward: unknown input mode %d", input_mode);
    }
    if (output_ch == 4) NERF_REQUIRE(((uintptr_t)raw_out & 15) == 0, "nerf_mlp_forward: raw_out must be 16-byte aligned");
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0, "nerf_mlp_forward: packed weights must be 16-byte aligned");
    const long long M = (long long)R * S;
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(M < (1ll << 31) - 512, "nerf_mlp_forward: at most 2^31 - 512 samples per launch, got %lld", M);
    MlpArgs a;
    a.packed = static_cast<const unsigned char*>(packed);
    a.n_stages = frag_total(precision, use_viewdirs) / STAGE_FRAGS;
    a.n_bias_tiles = use_viewdirs ? BIAS_TILES_VD : BIAS_TILES_NOVD;
    a.mode = input_mode; a.in0 = in0; a.in1 = in1; a.R = R; a.S = S; a.ray_cols = ray_cols; a.M = M; a.out = raw_out;
    const int per_tile = precision == NERF_PREC_BF16 ? 256 : 128;
    const long long n_tiles = (M + per_tile - 1) / per_tile;
    const int grid = (int)(n_tiles < cu_count() ? n_tiles : cu_count());
    hipStream_t st = as_stream(stream);
    if (precision == NERF_PREC_BF16) {
        if (use_viewdirs) hipLaunchKernelGGL((mlp_bf16_kernel<true, 4>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
        else if (output_ch == 4) hipLaunchKernelGGL((mlp_bf16_kernel<false, 4>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
        else hipLaunchKernelGGL((mlp_bf16_kernel<false, 5>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
    } else {
        if (use_viewdirs) hipLaunchKernelGGL((mlp_f32_kernel<true, 4>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
        else if (output_ch == 4) hipLaunchKernelGGL((mlp_f32_kernel<false, 4>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
        else hipLaunchKernelGGL((mlp_f32_kernel<false, 5>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
    }
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}
