// float32-accurate network evaluation on the bf16 matrix cores (NERF_PREC_BF16X3, NERF_PREC_BF16X2).
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
#include "mlp_bf16.h"
#include "mlp16.h"
#include "mlp_split.h"

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

// pair-conversion number sl (0 .. 4 NCT - 1) of the k-step made of accumulator tiles src[0], src[1]
template <int T>
__device__ __forceinline__ void convert_slice(const f32x4v (*src)[NCT], bool relu, int sl, Ops<T>& dst) {
    const int ct = sl >> 2, d = sl & 3;
    const f32x4v& acc = src[d >> 1][ct];
    unsigned w[T];
    split_pair<T>(acc[2 * (d & 1)], acc[2 * (d & 1) + 1], relu, w);
#pragma unroll
    for (int t = 0; t < T; ++t) dst.v[ct][t][d] = w[t];
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
template <int T, int NT>
__device__ __forceinline__ void kstep(SStream<T>& st, int& k, int g, bool first, int bias_tile0, int bias_after, f32x4v& bias,
                                      const Ops<T>& cur, f32x4v (*out)[NCT], bool conv, const f32x4v (*src)[NCT],
                                      bool relu, bool cross, Ops<T>& nxt) {
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
                for (int sl = lo; sl < hi; ++sl) convert_slice<T>(src, relu, sl, nxt);
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
template <int T, int NT, int KPE, int KH, int KVW>
__device__ __forceinline__ void layer(SStream<T>& st, Pass<T>& ps, int g, f32x4v& bias, const Ops<T>* pe, const Ops<T>* vw,
                                      const f32x4v (*in)[NCT], bool relu_in, f32x4v (*out)[NCT], Ops<T> (&win)[2],
                                      bool next_from_out, bool relu_out, int n_bias_tiles) {
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
        kstep<T, NT>(st, ps.k, g, s == 0, ps.bt, bias_after, bias, cur, out, conv, src, last ? relu_out : relu_in, last,
                     win[nwp]);
        ps.wp = nwp;
    }
    ps.bt += NT;
}

template <int T, bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_split_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    const TileWalk tw = tile_walk(a, (a.M + TILE - 1) / TILE);
    SStream<T> st;
    pipe_start(st.p, a, lds);
    st.slot0 = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    constexpr int NB = VD ? BIAS_TILES16_VD : BIAS_TILES16_NOVD;
    constexpr int NFRAGS = T * (VD ? FRAGS16_VD : FRAGS16_NOVD);
    constexpr int NPAD = pad_to_stage(NFRAGS);

#pragma unroll
    for (int j = 0; j < T; ++j) st.fetch(j);
    f32x4v bias = m16::bias16(st.p.lds, 0, g);

    const bool feat_mode = a.mode == NERF_IN_FEATURES;
    const unsigned s0 = (unsigned)tw.sample0, s_end = (unsigned)tw.sample_end;
    RawIn nxt_in[NCT];
    if (!feat_mode && tw.first < tw.end) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = s0 + (unsigned)tw.first * TILE + wave * WSAMP + ct * 16 + c;
            fetch_raw<VD>(a, m < s_end ? m : s_end - 1, nxt_in[ct]);
        }
    }

    for (long long t = tw.first; t < tw.end; t += tw.stride) {
        const unsigned pass0 = s0 + (unsigned)t * TILE + wave * WSAMP;
        RawIn cur_in[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) cur_in[ct] = nxt_in[ct];
#ifndef NERF_SPX_NOIN      // timing experiment only (results are garbage): what fetching the next pass's inputs costs
        if (!feat_mode && t + tw.stride < tw.end) {
#else
        if (false) {
#endif
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const unsigned m = s0 + (unsigned)(t + tw.stride) * TILE + wave * WSAMP + ct * 16 + c;
                fetch_raw<VD>(a, m < s_end ? m : s_end - 1, nxt_in[ct]);
            }
        }
        Ops<T> pe[2], vw[1];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = pass0 + ct * 16 + c;
            float pv[16], vv[8];
            if (feat_mode) {
                m16::load_features16<VD>(a, m < s_end ? m : s_end - 1, g, pv, vv);
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
#define NERF_SL(NT, KPE, KH, KVW, in, relu_in, out, next_from_out, relu_out) \
    layer<T, NT, KPE, KH, KVW>(st, ps, g, bias, pe, vw, in, relu_in, out, win, next_from_out, relu_out, NB)
        NERF_SL(16, 2, 0, 0, FA, false, FA, true, true);       // L0: pe -> FA
        NERF_SL(16, 0, 8, 0, FA, true, FB, true, true);        // L1
        NERF_SL(16, 0, 8, 0, FB, true, FA, true, true);        // L2
        NERF_SL(16, 0, 8, 0, FA, true, FB, true, true);        // L3
        NERF_SL(16, 0, 8, 0, FB, true, FA, false, true);       // L4 -> FA; L5 starts with the encoded points
        NERF_SL(16, 2, 8, 0, FA, true, FB, true, true);        // L5: [pe, h4]
        NERF_SL(16, 0, 8, 0, FB, true, FA, true, true);        // L6
        NERF_SL(16, 0, 8, 0, FA, true, FB, true, true);        // L7 -> FB
        float o[NCT][5];
        if (VD) {
            NERF_SL(17, 0, 8, 0, FB, true, FA, true, false);   // feature (tiles 0..15, no ReLU) and density (tile 16, row 3)
            NERF_SL(8, 0, 8, 1, FA, false, FB, true, true);    // views: [feature, view] -> FB[0..7]
            NERF_SL(1, 0, 4, 0, FB, true, FA, false, false);   // rgb: rows 0..2 of FA[0]
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                o[ct][0] = FA[0][ct][0]; o[ct][1] = FA[0][ct][1]; o[ct][2] = FA[0][ct][2]; o[ct][3] = FA[16][ct][3];
                o[ct][4] = 0.0f;
            }
        } else {
            NERF_SL(1, 0, 8, 0, FB, true, FA, false, false);   // output_linear: rows 0..3 (4: register 0 of lane group 1)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                o[ct][0] = FA[0][ct][0]; o[ct][1] = FA[0][ct][1]; o[ct][2] = FA[0][ct][2]; o[ct][3] = FA[0][ct][3];
                o[ct][4] = FA[0][ct][0];
            }
        }
#undef NERF_SL
        // zero padding at the end of the stream: nothing to compute, the ring keeps turning
#pragma unroll
        for (int k = NFRAGS; k < NPAD; ++k) st.tick(k);
        __builtin_amdgcn_sched_barrier(0);
        st.end_pass(NPAD);
#pragma unroll
        for (int j = 0; j < T; ++j) st.fetch(j);      // the first fragments of the next pass (queue slots 0 .. T - 1)
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
    if (terms == 3) {
        if (use_viewdirs) hipLaunchKernelGGL((sp::mlp_split_kernel<3, true, 4>), g, b, 0, st, a);
        else if (output_ch == 4) hipLaunchKernelGGL((sp::mlp_split_kernel<3, false, 4>), g, b, 0, st, a);
        else hipLaunchKernelGGL((sp::mlp_split_kernel<3, false, 5>), g, b, 0, st, a);
    } else {
        if (use_viewdirs) hipLaunchKernelGGL((sp::mlp_split_kernel<2, true, 4>), g, b, 0, st, a);
        else if (output_ch == 4) hipLaunchKernelGGL((sp::mlp_split_kernel<2, false, 4>), g, b, 0, st, a);
        else hipLaunchKernelGGL((sp::mlp_split_kernel<2, false, 5>), g, b, 0, st, a);
    }
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

}  // namespace nerf
