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
#include <type_traits>
#include "mlp_bf16.h"
#include "mlp16.h"
#include "train_bf16.h"
#include "mlp_split.h"

namespace nerf {


// TAPE: the training forward.  Every layer input a weight gradient needs is also stored, in the operand layout it
// has in the registers (train_bf16.h).
struct EpiTape {
    static constexpr bool ZERO_BIAS = false;
    unsigned bits, ones;         // ReLU bits of the tile being converted; 0x00010001
    __device__ __forceinline__ void prefetch(const EpiTile&, int) {}
    // the packed pair is non-negative after the clamp: min(bits, 1) per half is "unit active"
    __device__ __forceinline__ unsigned post(unsigned v, const EpiTile& t, int, int half, int ct, int qd) {
        if (t.has_act) {
            unsigned m;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(v), "v"(ones));
            const int k = 8 * ct + 4 * half + qd;
            bits = k == 0 ? m : (bits | (m << k));
        }
        return v;
    }
    __device__ __forceinline__ void store(const u32x4& v, const EpiTile& t, int half, int ct) const {
        // streamed: 5 KB per sample must not push the weight stream out of L2 (0.69 -> 0.50 ms per forward launch)
        if (t.has_out) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(t.out + (half * 2 + ct) * 1024));
        if (t.has_act && half == 1 && ct == NCT - 1)      // the 8 words of a lane fill one 32-byte sector: let L2 merge them
            *reinterpret_cast<unsigned*>(const_cast<unsigned char*>(t.act)) = bits;
    }
};

__device__ __forceinline__ void epi_init(EpiNone&) {}
__device__ __forceinline__ void epi_init(EpiTape& e) { e.bits = 0u; e.ones = 0x00010001u; }

template <bool VD, int OUT_CH, bool TAPE>
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
        // tape of this wave's 64 samples (training forward): piece p at tw + p * 2 KiB, lane quad at + ct KiB
        unsigned char* tw = nullptr;
        if (TAPE) {
            const unsigned long long blk = (unsigned long long)(tile * 4 + wave) * (tape16::total(VD) * tape16::PIECE_BYTES);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)blk), hi = __builtin_amdgcn_readfirstlane((unsigned)(blk >> 32));
            tw = a.tape + (((unsigned long long)hi << 32) | lo) + lane * 16;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    *reinterpret_cast<u32x4*>(tw + (tape16::PE + ks) * tape16::PIECE_BYTES + ct * 1024) = pe[ks][ct];
                if (VD) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        *reinterpret_cast<u32x4*>(tw + (tape16::VW + ks) * tape16::PIECE_BYTES + ct * 1024) = vw[ks][ct];
                }
            }
        }
        typename std::conditional<TAPE, EpiTape, EpiNone>::type epi;
        epi_init(epi);
        EpiTile carry{nullptr, nullptr, false, false, 0};
        auto out = [&](int piece) { return tw + piece * tape16::PIECE_BYTES; };
        // ReLU bit words: 32 bytes per lane (tw carries 16 lane)
        auto bitp = [&](int piece) { return tw + piece * tape16::PIECE_BYTES + lane * 16; };
#define NERF_L(NT, KPE, KH, KVW, hin, pdst, prelu, hasp, hout, relu, piece, store, bpiece)                             \
    layer_bf16<NT, KPE, KH, KVW>(st, pos, bt, NB, h, bias, pe, hin, vw, accp, pdst, prelu, hasp, hout, relu, epi, carry, \
                                 bitp(bpiece), TAPE && (store) && (relu), out(piece), TAPE && (store))
        u32x4 A[16][NCT], B[16][NCT];
        f32x16 accp[2][NCT];     // accumulators of the current and of the previous output tile
        int pos = 0, bt = 0;
        NERF_L(8, 4, 0, 0, A, A, false, false, A, true, tape16::h(VD, 0), true, tape16::bits(VD, 0));        // L0: pe -> A
        NERF_L(8, 0, 16, 0, A, A + 14, true, true, B, true, tape16::h(VD, 1), true, tape16::bits(VD, 1));    // L1
        NERF_L(8, 0, 16, 0, B, B + 14, true, true, A, true, tape16::h(VD, 2), true, tape16::bits(VD, 2));    // L2
        NERF_L(8, 0, 16, 0, A, A + 14, true, true, B, true, tape16::h(VD, 3), true, tape16::bits(VD, 3));    // L3
        NERF_L(8, 0, 16, 0, B, B + 14, true, true, A, true, tape16::h(VD, 4), true, tape16::bits(VD, 4));    // L4
        NERF_STAMP(t2)
        NERF_L(8, 4, 16, 0, A, A + 14, true, true, B, true, tape16::h(VD, 5), true, tape16::bits(VD, 5));    // L5: [pe, h]
        NERF_L(8, 0, 16, 0, B, B + 14, true, true, A, true, tape16::h(VD, 6), true, tape16::bits(VD, 6));    // L6
        NERF_L(8, 0, 16, 0, A, A + 14, true, true, B, true, tape16::h(VD, 7), true, tape16::bits(VD, 7));    // L7 -> B

        NERF_STAMP(t3)
        float o0[NCT], o1[NCT], o2[NCT], o3[NCT], o4[NCT];
        if (VD) {
            NERF_L(8, 0, 16, 0, B, B + 14, true, true, A, false, tape16::FEAT, true, 0);   // feature -> A (no relu)
            NERF_L(1, 0, 16, 0, B, A + 14, false, true, A, false, 0, false, 0);            // alpha: row 3
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) o3[ct] = accp[(bt - 1) & 1][ct][3];
            NERF_L(4, 0, 16, 2, A, A, false, false, B, true, tape16::HV, true, tape16::bits(VD, 8));         // views -> B[0..7]
            NERF_L(1, 0, 8, 0, B, B + 6, true, true, B, false, 0, false, 0);               // rgb: rows 0..2
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x16& r = accp[(bt - 1) & 1][ct];
                o0[ct] = r[0]; o1[ct] = r[1]; o2[ct] = r[2]; o4[ct] = 0.f;
            }
        } else {
            NERF_L(1, 0, 16, 0, B, B + 14, true, true, B, false, 0, false, 0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x16& r = accp[(bt - 1) & 1][ct];
                o0[ct] = r[0]; o1[ct] = r[1]; o2[ct] = r[2]; o3[ct] = r[3];
                o4[ct] = r[0];   // row 4 sits in register 0 of the upper lane half
            }
        }
#undef NERF_L
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
    const TileWalk tw = tile_walk(a, (a.M + 127) / 128);
    Pipe p;
    pipe_start(p, a, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;

    const bool feat_mode = a.mode == NERF_IN_FEATURES;
    // the samples of the network this workgroup serves: tile t covers s0 + 128 t ..., the last tile may be partial
    const unsigned s0 = (unsigned)tw.sample0, Mu = (unsigned)tw.sample_end;
    RawIn nxt;
    if (!feat_mode && tw.first < tw.end) {
        const unsigned m0 = s0 + (unsigned)tw.first * 128u + wave * 32 + c;
        fetch_raw<VD>(a, m0 < Mu ? m0 : Mu - 1, nxt);
    }

    for (long long t = tw.first; t < tw.end; t += tw.stride) {
        const unsigned m = s0 + (unsigned)t * 128u + wave * 32 + c;
        const RawIn cur = nxt;
        if (!feat_mode && t + tw.stride < tw.end) {
            const unsigned mn = s0 + (unsigned)(t + tw.stride) * 128u + wave * 32 + c;
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

// Output scales of the bf16 network.  B_l bounds |pre-activation| of layer l for any input with
// |feature| <= INPUT_BOUND: B_l = max_row (|b| + sum_j |W_j| * bound of input j).  The stored activations are
// s_l * relu(.) with s_l = 2^-ceil(log2(2 B_l)) < 1 / (2 B_l): the factor 2 covers the bf16 rounding of weights
// and activations.  Powers of two commute with every rounding step, so the network's results do not change.
// Two launches: per-row sums of |W| (every row of every layer in parallel, a wave per row), then the short chain
// over the layers in one block.
__global__ __launch_bounds__(256) void rowsum_kernel(PackArgs a, float* rs /*[layer][256][2]*/) {
    const int li = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const LayerDesc& L = a.net.layer[li];
    for (int k = 0; k < 8; ++k) {
        const int row = blockIdx.y * 32 + wave * 8 + k;
        if (row >= L.out_dim) break;                          // wave-uniform
        const float* W = a.t[L.w_idx] + (size_t)row * L.in_dim;
        float hid = 0.0f, other = 0.0f;
        for (int col = lane; col < L.in_dim; col += 64) {
            const float w = fabsf(W[col]);
            if (col >= L.hid_col && col < L.hid_col + L.hid_feats) hid += w; else other += w;
        }
        for (int off = 32; off > 0; off >>= 1) {
            hid += __shfl_xor(hid, off, WAVE);
            other += __shfl_xor(other, off, WAVE);
        }
        if (lane == 0) {
            rs[((size_t)li * 256 + row) * 2 + 0] = fabsf(a.t[L.b_idx][row]) + other * INPUT_BOUND;
            rs[((size_t)li * 256 + row) * 2 + 1] = hid;
        }
    }
}

__global__ __launch_bounds__(256) void scale_kernel(PackArgs a, const float* rs, float* scales) {
    __shared__ float red[256];
    __shared__ float bound[MAX_LAYERS];
    for (int li = 0; li < a.net.n_layers; ++li) {
        const LayerDesc& L = a.net.layer[li];
        const float bin = L.src_layer >= 0 ? bound[L.src_layer] : 0.0f;
        float m = 0.0f;
        if ((int)threadIdx.x < L.out_dim) {
            const float* r = rs + ((size_t)li * 256 + threadIdx.x) * 2;
            m = r[0] + r[1] * bin;
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
        if (a.bf16 && a.with_scales) {
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
                else if (g < npe + nh + (L.has_view ? 2 : 0)) {
                    const int ft = view_slot_feature(h, 8 * (g - npe - nh) + j); col = ft < 0 ? -1 : L.view_col + ft;
                } else {
                    col = 8 * (j >> 2) + 4 * h + (j & 3);
                    extra = true;
                }
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
            if (a.bf16 && a.with_scales) {
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
        if (a.bf16 && a.with_scales) {
            const LayerDesc& L = a.net.layer[li];
            const float* scales = reinterpret_cast<const float*>(a.out + (size_t)a.padded_frags * FRAG_BYTES +
                                                                 (size_t)a.net.n_bias_tiles * BIAS_TILE_BYTES);
            const bool head_row = L.kind == LK_FEATURE_ALPHA && t == 8;
            if (L.scaled_out && !head_row) v *= scales[li];
        }
        reinterpret_cast<float*>(a.out + (size_t)a.padded_frags * FRAG_BYTES)[bi] = v;
    }
}

// The same network as the 16x16x32 stream of the bf16 inference kernel (mlp16.h): fragments of 16 output rows x 32
// k-slots, bias tiles of 16 floats, the output scales computed above folded in the same way.
__device__ __forceinline__ const float* pack_row16(const PackArgs& a, const LayerDesc& L, int tile, int r, bool bias, int& row) {
    row = -1;
    switch (L.kind) {
        case LK_HIDDEN:
            row = tile * 16 + r;
            if (row >= L.out_dim) { row = -1; return nullptr; }
            return a.t[bias ? L.b_idx : L.w_idx];
        case LK_FEATURE_ALPHA:
            if (tile < 16) { row = tile * 16 + r; return a.t[bias ? L.b_idx : L.w_idx]; }
            if (r == 3) { row = 0; return a.t[bias ? L.b2_idx : L.w2_idx]; }      // the density row
            return nullptr;
        default:   // LK_RGB, LK_OUTPUT: rows 0..out_dim-1 of a single tile
            if (r < L.out_dim) { row = r; return a.t[bias ? L.b_idx : L.w_idx]; }
            return nullptr;
    }
}

__global__ __launch_bounds__(256) void pack16_kernel(PackArgs a, unsigned char* out16, int padded_frags, int n_bias_tiles16,
                                                     const float* scales) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int F = gid >> 6, lane = gid & 63;
    const int r = lane & 15, gg = lane >> 4;
    if (F < padded_frags) {
        int f = F, li = 0;
        for (; li < a.net.n_layers; ++li) {
            const int nf = tiles16(a.net.layer[li]) * ksteps16(a.net.layer[li]);
            if (f < nf) break;
            f -= nf;
        }
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
        if (li < a.net.n_layers) {
            const LayerDesc& L = a.net.layer[li];
            const int ks = ksteps16(L);
            const int tile = f / ks, kk = f - tile * ks;
            int row;
            const float* W = pack_row16(a, L, tile, r, false, row);
            const bool head_row = L.kind == LK_FEATURE_ALPHA && tile == 16;
            const float s_out = (L.scaled_out && !head_row) ? scales[li] : 1.0f;
            const float s_hid = L.src_layer >= 0 ? scales[L.src_layer] : 1.0f;
            const int npe = L.has_pe ? 2 : 0, nh = L.hid_feats / 32;
            for (int j = 0; j < 8; ++j) {
                int col = -1;
                const bool hidden = kk >= npe && kk < npe + nh;
                if (kk < npe) { const int ft = pe16_slot_feature(gg, 8 * kk + j); col = ft < 0 ? -1 : L.pe_col + ft; }
                else if (hidden) col = L.hid_col + 32 * (kk - npe) + 16 * (j >> 2) + 4 * gg + (j & 3);
                else { const int ft = view16_slot_feature(gg, j); col = ft < 0 ? -1 : L.view_col + ft; }
                float w = (W && col >= 0) ? W[(size_t)row * L.in_dim + col] : 0.0f;
                w *= hidden ? s_out / s_hid : s_out;
                v[j] = (__bf16)w;
            }
        }
        *reinterpret_cast<bf16x8*>(out16 + (size_t)F * FRAG_BYTES + lane * 16) = v;
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
        int row;
        const float* Bp = pack_row16(a, L, t, rr, true, row);
        float b = Bp ? Bp[row] : 0.0f;
        const bool head_row = L.kind == LK_FEATURE_ALPHA && t == 16;
        if (L.scaled_out && !head_row) b *= scales[li];
        reinterpret_cast<float*>(out16 + (size_t)padded_frags * FRAG_BYTES)[bi] = b;
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
    {   // bf16 16x16x32: A[row c][k = 8g + j], B[k = 8g + j][col c]; result rows 4g + e of column c
        const int c = lane & 15, g = lane >> 4;
        bf16x8 fa, fb;
        for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)Aval(c, 8 * g + j); fb[j] = (__bf16)Bval(8 * g + j, c); }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
        for (int e = 0; e < 4; ++e) {
            float ref = 0.f;
            for (int k = 0; k < 32; ++k) ref += Aval(4 * g + e, k) * Bval(k, c);
            if (acc[e] != ref) ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int launch_pack(const PackArgs& a, hipStream_t stream) {
    if (a.bf16 && a.with_scales) {   // output scales first: the pack kernel reads them from behind the bias block
        float* scales = reinterpret_cast<float*>(a.out + (size_t)a.padded_frags * FRAG_BYTES +
                                                 (size_t)a.net.n_bias_tiles * BIAS_TILE_BYTES);
        float* rs = scales + SCALE_BYTES / 4;
        hipLaunchKernelGGL(rowsum_kernel, dim3(a.net.n_layers, 8), dim3(256), 0, stream, a, rs);
        NERF_LAUNCH_CHECK();
        hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(256), 0, stream, a, rs, scales);
        NERF_LAUNCH_CHECK();
    }
    const int threads = a.padded_frags * 64 + a.net.n_bias_tiles * 32;
    hipLaunchKernelGGL(pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

int launch_mlp_bf16_train(const MlpArgs& a, bool use_viewdirs, int output_ch, hipStream_t st) {
    const long long n_tiles = (a.M + 255) / 256;
    const dim3 grid((unsigned)(n_tiles < cu_count() ? n_tiles : cu_count())), block(MLP_THREADS);
    if (use_viewdirs) hipLaunchKernelGGL((mlp_bf16_kernel<true, 4, true>), grid, block, 0, st, a);
    else if (output_ch == 4) hipLaunchKernelGGL((mlp_bf16_kernel<false, 4, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((mlp_bf16_kernel<false, 5, true>), grid, block, 0, st, a);
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

// mismatch counter of the self test: a device global, so the test allocates nothing.  Host code reaches it only
// through the address hipGetSymbolAddress returns for the current device.
__device__ int g_selftest_bad;

extern "C" int nerf_selftest_mfma(void) {
    int* d = nullptr;
    NERF_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void**>(&d), HIP_SYMBOL(g_selftest_bad)));
    NERF_HIP_CHECK(hipMemset(d, 0, sizeof(int)));
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, d);
    NERF_LAUNCH_CHECK();
    int hbad = -1;
    NERF_HIP_CHECK(hipMemcpy(&hbad, d, sizeof(int), hipMemcpyDeviceToHost));
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

// bf16 buffer: [32x32x16 stream | its bias tiles | scales | row-sum scratch] as the training kernels read it, then
// [16x16x32 stream | its bias tiles] for the inference kernel
static size_t stream16_offset(int use_viewdirs) {
    return (size_t)frag_total(NERF_PREC_BF16, use_viewdirs) * FRAG_BYTES +
           (size_t)(use_viewdirs ? BIAS_TILES_VD : BIAS_TILES_NOVD) * BIAS_TILE_BYTES + SCALE_BYTES + ROWSUM_BYTES;
}

extern "C" size_t nerf_mlp_packed_bytes(int precision, int use_viewdirs) {
    if (split_terms(precision)) return split_packed_bytes(split_terms(precision), use_viewdirs != 0);
    if (precision != NERF_PREC_BF16 && precision != NERF_PREC_F32) return 0;
    if (precision == NERF_PREC_BF16)
        return stream16_offset(use_viewdirs) + (size_t)frag_total(NERF_PREC_BF16, use_viewdirs) * FRAG_BYTES +
               bias16_block_bytes(use_viewdirs != 0);
    return (size_t)frag_total(precision, use_viewdirs) * FRAG_BYTES +
           (size_t)(use_viewdirs ? BIAS_TILES_VD : BIAS_TILES_NOVD) * BIAS_TILE_BYTES + SCALE_BYTES;
}

extern "C" int nerf_mlp_pack(const float* const* tensors_host, int use_viewdirs, int output_ch, int precision,
                             void* packed_out, void* stream) {
    NERF_REQUIRE(tensors_host && packed_out, "nerf_mlp_pack: null pointer");
    NERF_REQUIRE(precision == NERF_PREC_BF16 || precision == NERF_PREC_F32 || split_terms(precision),
                 "nerf_mlp_pack: unknown precision %d", precision);
    if (use_viewdirs) NERF_REQUIRE(output_ch == 4, "nerf_mlp_pack: with view directions the output is [rgb, sigma] (4 channels)");
    else NERF_REQUIRE(output_ch == 4 || output_ch == 5, "nerf_mlp_pack: output_ch must be 4 or 5, got %d", output_ch);
    PackArgs a;
    a.net = make_net_desc(use_viewdirs != 0, output_ch);
    if (const int terms = split_terms(precision)) {
        const int n_t = use_viewdirs ? 24 : 20;
        for (int i = 0; i < 24; ++i) a.t[i] = i < n_t ? tensors_host[i] : nullptr;
        for (int i = 0; i < n_t; ++i)
            if (use_viewdirs || (i != 16 && i != 17)) NERF_REQUIRE(a.t[i], "nerf_mlp_pack: tensor %d is null", i);
        a.bf16 = 1; a.with_scales = 0; a.total_frags = a.padded_frags = split_padded_frags(terms, use_viewdirs != 0);
        a.out = static_cast<unsigned char*>(packed_out);
        return launch_pack_split(a, terms, use_viewdirs != 0, as_stream(stream));
    }
    a.bf16 = precision == NERF_PREC_BF16;
    a.with_scales = a.bf16;
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
    if (int rc = launch_pack(a, as_stream(stream))) return rc;
    if (a.bf16) {
        const int nb16 = use_viewdirs ? BIAS_TILES16_VD : BIAS_TILES16_NOVD;
        const float* scales = reinterpret_cast<const float*>(a.out + (size_t)a.padded_frags * FRAG_BYTES +
                                                             (size_t)a.net.n_bias_tiles * BIAS_TILE_BYTES);
        const int threads = a.total_frags * 64 + nb16 * 16;
        hipLaunchKernelGGL(pack16_kernel, dim3((threads + 255) / 256), dim3(256), 0, as_stream(stream), a,
                           a.out + stream16_offset(use_viewdirs), a.total_frags, nb16, scales);
        NERF_LAUNCH_CHECK();
    }
    return NERF_OK;
}

static int mlp_forward_impl(const char* who, const void* packed, const void* const* table, int n_objects, int precision,
                            int use_viewdirs, int output_ch, int input_mode, const float* in0, const float* in1, int R,
                            int S, int ray_cols, float* raw_out, void* stream) {
    NERF_REQUIRE((packed || table) && (R == 0 || (in0 && raw_out)), "%s: null pointer", who);
    const int terms = split_terms(precision);
    NERF_REQUIRE(precision == NERF_PREC_BF16 || precision == NERF_PREC_F32 || terms, "%s: unknown precision %d", who, precision);
    NERF_REQUIRE(R >= 0 && S >= 1, "%s: bad shape R=%d S=%d", who, R, S);
    if (use_viewdirs) NERF_REQUIRE(output_ch == 4, "%s: output_ch must be 4 with view directions", who);
    else NERF_REQUIRE(output_ch == 4 || output_ch == 5, "%s: output_ch must be 4 or 5", who);
    switch (input_mode) {
        case NERF_IN_FEATURES: NERF_REQUIRE(S == 1, "%s: feature input is flat (S = 1)", who); break;
        case NERF_IN_POINTS: NERF_REQUIRE(R == 0 || !use_viewdirs || in1, "%s: view directions missing", who); break;
        case NERF_IN_RAYS:
            NERF_REQUIRE(R == 0 || in1, "%s: z_vals missing", who);
            NERF_REQUIRE(ray_cols >= (use_viewdirs ? 11 : 8), "%s: rays need %d columns, got %d", who,
                         use_viewdirs ? 11 : 8, ray_cols);
            break;
        default: NERF_REQUIRE(false, "%s: unknown input mode %d", who, input_mode);
    }
    if (output_ch == 4) NERF_REQUIRE(((uintptr_t)raw_out & 15) == 0, "%s: raw_out must be 16-byte aligned", who);
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0, "%s: packed weights must be 16-byte aligned", who);
    const long long M = (long long)R * S;
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(M < (1ll << 31) - 512, "%s: at most 2^31 - 512 samples per launch, got %lld", who, M);
    const int per_tile = terms ? split_samples_per_tile() : (precision == NERF_PREC_BF16 ? m16::TILE16 : 128);
    MlpArgs a;
    a.packed = static_cast<const unsigned char*>(packed);
    a.n_stages = (terms ? split_padded_frags(terms, use_viewdirs != 0) : frag_total(precision, use_viewdirs)) / STAGE_FRAGS;
    a.n_bias_tiles = use_viewdirs ? BIAS_TILES_VD : BIAS_TILES_NOVD;
    a.mode = input_mode; a.in0 = in0; a.in1 = in1; a.R = R; a.S = S; a.ray_cols = ray_cols; a.M = M; a.out = raw_out;
    set_sample_divisor(a);
    a.tape = nullptr;
    a.table = nullptr; a.n_obj = 1; a.tiles_per_obj = 0; a.table_off = 0; a.samples_per_obj = M;
    const long long n_tiles = (M + per_tile - 1) / per_tile;
    long long grid = n_tiles < cu_count() ? n_tiles : cu_count();
    if (table) {
        NERF_REQUIRE(n_objects >= 1 && R % n_objects == 0, "%s: %d rays do not split evenly over %d networks", who, R, n_objects);
        const long long per_obj = M / n_objects;
        // both kernels bound every network's last tile by its sample count
        NERF_REQUIRE(n_objects <= cu_count(), "%s: at most %d networks per launch, got %d", who, cu_count(), n_objects);
        a.table = reinterpret_cast<const unsigned char* const*>(table);
        a.n_obj = n_objects;
        a.tiles_per_obj = (per_obj + per_tile - 1) / per_tile;
        a.samples_per_obj = per_obj;
        // a network is served by floor(CUs / networks) workgroups or one more (tile_walk), never by more than it has tiles
        grid = a.tiles_per_obj * n_objects <= cu_count() ? a.tiles_per_obj * n_objects : cu_count();
    }
    hipStream_t st = as_stream(stream);
    if (terms) {
        a.n_bias_tiles = ((use_viewdirs ? BIAS_TILES16_VD : BIAS_TILES16_NOVD) * 16 + 31) / 32;    // in units of 32 floats
        return launch_mlp_split(a, terms, use_viewdirs != 0, output_ch, (unsigned)grid, st);
    }
    if (precision == NERF_PREC_BF16) {
        if (table) a.table_off = stream16_offset(use_viewdirs); else a.packed += stream16_offset(use_viewdirs);
        a.n_bias_tiles = ((use_viewdirs ? BIAS_TILES16_VD : BIAS_TILES16_NOVD) * 16 + 31) / 32;    // in units of 32 floats
        if (use_viewdirs) hipLaunchKernelGGL((m16::mlp16_bf16_kernel<true, 4>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
        else if (output_ch == 4) hipLaunchKernelGGL((m16::mlp16_bf16_kernel<false, 4>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
        else hipLaunchKernelGGL((m16::mlp16_bf16_kernel<false, 5>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
    } else {
        if (use_viewdirs) hipLaunchKernelGGL((mlp_f32_kernel<true, 4>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
        else if (output_ch == 4) hipLaunchKernelGGL((mlp_f32_kernel<false, 4>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
        else hipLaunchKernelGGL((mlp_f32_kernel<false, 5>), dim3(grid), dim3(MLP_THREADS), 0, st, a);
    }
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

extern "C" int nerf_mlp_forward(const void* packed, int precision, int use_viewdirs, int output_ch, int input_mode,
                                const float* in0, const float* in1, int R, int S, int ray_cols, float* raw_out,
                                void* stream) {
    NERF_REQUIRE(packed, "nerf_mlp_forward: null pointer");
    return mlp_forward_impl("nerf_mlp_forward", packed, nullptr, 1, precision, use_viewdirs, output_ch, input_mode, in0, in1,
                            R, S, ray_cols, raw_out, stream);
}

extern "C" int nerf_mlp_forward_objects(const void* const* packed_table_dev, int n_objects, int precision, int use_viewdirs,
                                        int output_ch, int input_mode, const float* in0, const float* in1, int R, int S,
                                        int ray_cols, float* raw_out, void* stream) {
    NERF_REQUIRE(packed_table_dev, "nerf_mlp_forward_objects: null pointer");
    return mlp_forward_impl("nerf_mlp_forward_objects", nullptr, packed_table_dev, n_objects, precision, use_viewdirs, output_ch,
                            input_mode, in0, in1, R, S, ray_cols, raw_out, stream);
}
