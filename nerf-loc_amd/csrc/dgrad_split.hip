// Input-gradient chain of the training step in three bf16 terms (train precision NERF_PREC_BF16X3).
//
// The reference gets these gradients from autograd over NeRF.forward (run_nerf_helpers.py:96-119, run_nerf.py:770-796).
// mlp_dgrad_f32_kernel (train.hip) runs the register-resident layer chain on the TRANSPOSED weights with the exact
// float32 MFMA; this kernel runs the same chain the way mlp_split.hip runs the forward: every operand a sum of three
// bf16 terms, six products per multiply-add on v_mfma_f32_16x16x32_bf16, float32 accumulation, layers evaluated
// k-major with both the layer being read and the one being written resident as float32 accumulators.  With view
// directions, for the lane's sample:
//     d(view hidden) = rgb_linear^T d(rgb)                        masked by the ReLU bits of views_linears.0   -> dzv
//     d(feature)     = views_linears.0[:, :256]^T d(view hidden)                                               -> dfeat
//     d(h8)          = feature_linear^T d(feature) + alpha_linear^T d(sigma)      masked by pts_linears.7's    -> dz[7]
//     d(h_i)         = pts_linears.i[:, hidden columns]^T dz[i], i = 7..1         masked by pts_linears.(i-1)'s -> dz[i-1]
// without: d(h8) = output_linear^T d(raw), then the same seven layers.  The moment a gradient tile is split into the
// next layer's operands is the moment its masked float32 values exist in registers: the conversion slices store them
// to the gradient tape (tape.h), which the float32 weight-gradient kernels read unchanged.  The ReLU bits are the
// forward's (tape.h layout), fetched one layer before they are needed.
#include "mlp_split_device.h"
#include "mlp_split.h"
#include "tape.h"

namespace nerf {
namespace sp {

struct SplitDgradArgs {
    const unsigned char* packed;   // transposed k-major three-term stream (pack_split_bwd_kernel)
    int n_stages;
    long long M;
    const float* d_raw;            // [M, out_ch]
    const unsigned* bits;          // the forward tape's ReLU bits
    GTapePtrs gt;
};

struct GradDst {
    float* base;
    int ld;                        // 256, or 128 (view hidden layer)
};

// per pass and lane: rowb[ct] = 1024 m (byte offset of the sample's row in a 256-float tape), g16 = 16 g,
// gsh = 4 (g >> 1): where the lane's nibbles start in a byte of the ReLU words
struct RowCtx {
    unsigned rowb[NCT], g16, gsh;
};

// The four ReLU dwords of the lane's half (h = g & 1) of one layer and sample, shifted right by gsh: bit 8 (t & 3) + r of
// dword t >> 2 belongs to register r of 16-row tile t (tape.h: bit e of 32-row tile t32 <-> feature
// 32 t32 + (e & 3) + 8 (e >> 2) + 4 h, and the lane holds features 16 t + 4 g + r).
struct MaskWords {
    unsigned w[NCT][4];
};

__device__ __forceinline__ void fetch_mask(const SplitDgradArgs& a, const RowCtx& rc, int bits_per_sample, int layer, MaskWords& mw) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const unsigned m = rc.rowb[ct] >> 10;
        const uint4 q = *reinterpret_cast<const uint4*>(a.bits + (size_t)m * bits_per_sample + layer * TAPE_BITS_LAYER +
                                                         ((rc.g16 >> 4) & 1u) * 4u);
        mw.w[ct][0] = q.x >> rc.gsh; mw.w[ct][1] = q.y >> rc.gsh; mw.w[ct][2] = q.z >> rc.gsh; mw.w[ct][3] = q.w >> rc.gsh;
    }
}

__device__ __forceinline__ float keep_if(float x, unsigned word, int pos) {      // x if bit pos of word is set, else +0
    const int m = __builtin_amdgcn_sbfe((int)word, pos, 1);                        // 0 or -1
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m);
}

// pair-conversion number sl of the k-step made of gradient tiles src[0], src[1] (tiles tile0, tile0 + 1 of their
// layer): mask, split into the next layer's operand terms, and (second slice of a tile) store the tile's four values
template <int T>
__device__ __forceinline__ void dconvert_slice(const f32x4v (*src)[NCT], const MaskWords& mask, bool masked, int sl,
                                               Ops<T>& dst, const GradDst& gd, const RowCtx& rc, int tile0) {
    const int ct = sl >> 2, d = sl & 3, tile = tile0 + (d >> 1);
    const f32x4v& acc = src[d >> 1][ct];
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = masked ? keep_if(acc[r], mask.w[ct][tile >> 2], 8 * (tile & 3) + r) : acc[r];
    unsigned w[T];
    split_pair<T>(v[2 * (d & 1)], v[2 * (d & 1) + 1], false, w);
#pragma unroll
    for (int t = 0; t < T; ++t) dst.v[ct][t][d] = w[t];
    if (d & 1) {
        const unsigned off = (gd.ld == 256 ? rc.rowb[ct] : rc.rowb[ct] >> 1) + rc.g16;
        f32x4v* dstp = reinterpret_cast<f32x4v*>(reinterpret_cast<unsigned char*>(gd.base) + off + 64 * tile);
        const f32x4v val = {v[0], v[1], v[2], v[3]};
        *dstp = val;      // plain: a non-temporal store is 4 % faster here and costs the weight-gradient kernels more
    }
}

// One k-step of a layer, NT output tiles (mlp_split.hip's kstep without biases: gradients accumulate from zero).
template <int T, int NT>
__device__ __forceinline__ void dkstep(SStream<T>& st, int& k, bool first, const Ops<T>& cur, f32x4v (*out)[NCT], bool conv,
                                       const f32x4v (*src)[NCT], const MaskWords& mask, bool masked, bool cross, Ops<T>& nxt,
                                       const GradDst& gd, const RowCtx& rc, int tile0) {
    constexpr int NSL = 4 * NCT;
    const f32x4v zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < T; ++j) st.tick(k + j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < T; ++j) st.fetch(k + T + j);
        bool started = !first;
#pragma unroll
        for (int sum = T - 1; sum >= 0; --sum) {
#pragma unroll
            for (int i = sum; i >= 0; --i) {
                const bf16x8 w = __builtin_bit_cast(bf16x8, st.q[(k + i) % (2 * T)]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    out[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, cur.v[ct][sum - i]),
                                                                         started ? out[t][ct] : zero, 0, 0, 0);
                started = true;
            }
        }
        if (conv) {
            const int t0 = cross ? (NT > 2 ? 2 : NT - 1) : 0, avail = NT - t0;
            if (t >= t0) {
                const int lo = NSL * (t - t0) / avail, hi = NSL * (t - t0 + 1) / avail;
#pragma unroll
                for (int sl = lo; sl < hi; ++sl) dconvert_slice<T>(src, mask, masked, sl, nxt, gd, rc, tile0);
            }
        }
#pragma unroll
        for (int i = 0; i < NCT * T * (T + 1) / 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < T) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        k += T;
    }
}

// A layer: [KD k-steps of d(raw) operands][KH hidden k-steps of `in`][KX k-steps of d(raw) operands again] -> NT tiles
// `out`.  Conversions of `in` (masked by mask_in, stored to gd_in) run one k-step ahead; the last k-step converts tiles
// 0, 1 of `out` for the next layer (masked by mask_out, stored to gd_out) if next_from_out.
template <int T, int NT, int KD, int KH, int KX>
__device__ __forceinline__ void dlayer(SStream<T>& st, int& k, int& wp, const Ops<T>& draw, const f32x4v (*in)[NCT],
                                       const MaskWords& mask_in, bool masked_in, const GradDst& gd_in, f32x4v (*out)[NCT],
                                       Ops<T> (&win)[2], bool next_from_out, const MaskWords& mask_out, bool masked_out,
                                       const GradDst& gd_out, const RowCtx& rc) {
    constexpr int KS = KD + KH + KX;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bool hidden = s >= KD && s < KD + KH;
        const Ops<T>& cur = hidden ? win[wp] : draw;
        const bool last = s + 1 == KS;
        const bool next_hidden = !last && (s + 1 >= KD && s + 1 < KD + KH);
        const int nwp = hidden ? wp ^ 1 : wp;
        // two calls rather than selected arguments: the mask words stay in registers only if no pointer to them is chosen
        // at run time
        if (last) dkstep<T, NT>(st, k, s == 0, cur, out, next_from_out, out, mask_out, masked_out, true, win[nwp], gd_out, rc, 0);
        else dkstep<T, NT>(st, k, s == 0, cur, out, next_hidden, in + 2 * (s + 1 - KD), mask_in, masked_in, false, win[nwp], gd_in, rc,
                           2 * (s + 1 - KD));
        wp = nwp;
    }
}

constexpr int BWD_FRAGS16_VD = 8 * 1 + 16 * 4 + 16 * 9 + 7 * 16 * 8;      // 1112 per term
constexpr int BWD_FRAGS16_NOVD = 16 * 1 + 7 * 16 * 8;                     // 912

template <bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_split_dgrad_kernel(SplitDgradArgs a) {
    constexpr int T = 3;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    SStream<T> st;
    {
        MlpArgs ma{};
        ma.packed = a.packed; ma.n_stages = a.n_stages; ma.n_bias_tiles = 0;
        pipe_start(st.p, ma, lds);
    }
    st.slot0 = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15;
    constexpr int NFRAGS = T * (VD ? BWD_FRAGS16_VD : BWD_FRAGS16_NOVD);
    constexpr int NPAD = pad_to_stage(NFRAGS);
    constexpr int NBITS = tape_bits(VD);
    const long long n_tiles = (a.M + TILE - 1) / TILE;
    const unsigned Mu = (unsigned)a.M;

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned pass0 = (unsigned)tile * TILE + wave * WSAMP;
        // the lane group, through a copy the compiler cannot see through: what derives from it is recomputed per pass
        // instead of being carried across the layer loop
        int g = lane >> 4;
        asm volatile("" : "+v"(g));
        RowCtx rc;
        rc.g16 = 16u * g;
        rc.gsh = 4u * (g >> 1);
        // d(raw) as the B operand of a one-fragment k part: k-slot (g, j) = 8 g + j; channels 0..3 sit in slots 0..3 of
        // lane group 0, channel 4 (output_ch 5) in slot 0 of lane group 1
        Ops<T> draw;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = pass0 + ct * 16 + c;
            rc.rowb[ct] = m << 10;                                   // gradient-tape rows exist for whole 128-sample tiles
            const float* drp = a.d_raw + (size_t)(m < Mu ? m : Mu - 1) * OUT_CH;
            float dr[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (OUT_CH == 4) {
                const float4 q = *reinterpret_cast<const float4*>(drp);
                dr[0] = q.x; dr[1] = q.y; dr[2] = q.z; dr[3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 5; ++i) dr[i] = drp[i];
            }
            const float s0 = g == 0 ? dr[0] : (g == 1 ? dr[4] : 0.0f);
            const float s1 = g == 0 ? dr[1] : 0.0f, s2 = g == 0 ? dr[2] : 0.0f, s3 = g == 0 ? dr[3] : 0.0f;
            unsigned w01[T], w23[T];
            split_pair<T>(s0, s1, false, w01);
            split_pair<T>(s2, s3, false, w23);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                draw.v[ct][t][0] = w01[t]; draw.v[ct][t][1] = w23[t]; draw.v[ct][t][2] = 0u; draw.v[ct][t][3] = 0u;
            }
        }
        MaskWords mA, mB;
        f32x4v FA[16][NCT], FB[16][NCT];
        Ops<T> win[2];
        int k = 0, wp = 0;
        const GradDst none{nullptr, 256};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < T; ++j) st.fetch(j);
#define NERF_GD(p) GradDst{p, 256}
        if (VD) {
            fetch_mask(a, rc, NBITS, 8, mA);
            // rgb_linear^T: d(rgb) -> d(view hidden), 8 tiles
            dlayer<T, 8, 1, 0, 0>(st, k, wp, draw, FB, mA, false, none, FB, win, true, mA, true, GradDst{a.gt.dzv, 128}, rc);
            // views_linears.0[:, :256]^T: 128 -> 256, no mask on the feature gradient
            dlayer<T, 16, 0, 4, 0>(st, k, wp, draw, FB, mA, true, GradDst{a.gt.dzv, 128}, FA, win, true, mA, false, NERF_GD(a.gt.dfeat), rc);
            fetch_mask(a, rc, NBITS, 7, mB);
            // feature_linear^T and alpha_linear^T (the extra k-step: d(sigma) in slot 3)
            dlayer<T, 16, 0, 8, 1>(st, k, wp, draw, FA, mA, false, NERF_GD(a.gt.dfeat), FB, win, true, mB, true, NERF_GD(a.gt.dz[7]), rc);
        } else {
            fetch_mask(a, rc, NBITS, 7, mB);
            dlayer<T, 16, 1, 0, 0>(st, k, wp, draw, FB, mB, false, none, FB, win, true, mB, true, NERF_GD(a.gt.dz[7]), rc);
        }
        // pts_linears.7 .. 1 transposed: dz[i] (in, mask of layer i) -> dz[i - 1] (out, mask of layer i - 1)
        fetch_mask(a, rc, NBITS, 6, mA);
        dlayer<T, 16, 0, 8, 0>(st, k, wp, draw, FB, mB, true, NERF_GD(a.gt.dz[7]), FA, win, true, mA, true, NERF_GD(a.gt.dz[6]), rc);
        fetch_mask(a, rc, NBITS, 5, mB);
        dlayer<T, 16, 0, 8, 0>(st, k, wp, draw, FA, mA, true, NERF_GD(a.gt.dz[6]), FB, win, true, mB, true, NERF_GD(a.gt.dz[5]), rc);
        fetch_mask(a, rc, NBITS, 4, mA);
        dlayer<T, 16, 0, 8, 0>(st, k, wp, draw, FB, mB, true, NERF_GD(a.gt.dz[5]), FA, win, true, mA, true, NERF_GD(a.gt.dz[4]), rc);
        fetch_mask(a, rc, NBITS, 3, mB);
        dlayer<T, 16, 0, 8, 0>(st, k, wp, draw, FA, mA, true, NERF_GD(a.gt.dz[4]), FB, win, true, mB, true, NERF_GD(a.gt.dz[3]), rc);
        fetch_mask(a, rc, NBITS, 2, mA);
        dlayer<T, 16, 0, 8, 0>(st, k, wp, draw, FB, mB, true, NERF_GD(a.gt.dz[3]), FA, win, true, mA, true, NERF_GD(a.gt.dz[2]), rc);
        fetch_mask(a, rc, NBITS, 1, mB);
        dlayer<T, 16, 0, 8, 0>(st, k, wp, draw, FA, mA, true, NERF_GD(a.gt.dz[2]), FB, win, true, mB, true, NERF_GD(a.gt.dz[1]), rc);
        fetch_mask(a, rc, NBITS, 0, mA);
        dlayer<T, 16, 0, 8, 0>(st, k, wp, draw, FB, mB, true, NERF_GD(a.gt.dz[1]), FA, win, false, mA, false, none, rc);
#undef NERF_GD
        // zero padding at the end of the stream: nothing to compute, the ring keeps turning
#pragma unroll
        for (int kk = NFRAGS; kk < NPAD; ++kk) st.tick(kk);
        __builtin_amdgcn_sched_barrier(0);
        st.end_pass(NPAD);
        // dz[0]: nothing follows that would convert it; masked and stored from the accumulators
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4v& acc = FA[t][ct];
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = keep_if(acc[r], mA.w[ct][t >> 2], 8 * (t & 3) + r);
                const f32x4v val = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4v*>(reinterpret_cast<unsigned char*>(a.gt.dz[0]) + rc.rowb[ct] + rc.g16 + 64 * t) = val;
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

// ---- packer: the transposed k-major three-term stream ----------------------------------------------------------------
// Order: layer, k-step, 16-row output tile, term.  t: the 24 (20) state_dict-ordered tensors.
struct BwdPackArgs {
    const float* t[24];
    unsigned char* out;
    int vd, out_ch, T, padded_frags;
};

__global__ __launch_bounds__(256) void pack_split_bwd_kernel(BwdPackArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int F = gid >> 6, lane = gid & 63;
    if (F >= a.padded_frags) return;
    const int r = lane & 15, gg = lane >> 4;
    // layer table: tiles, k-steps, and what a k-step reads
    int f = F / a.T;
    const int term = F % a.T;
    int tiles = 0, ksteps = 0, li = 0;
    const int n_layers = a.vd ? 10 : 8;
    for (; li < n_layers; ++li) {
        if (a.vd) { tiles = li == 0 ? 8 : 16; ksteps = li == 0 ? 1 : (li == 1 ? 4 : (li == 2 ? 9 : 8)); }
        else { tiles = 16; ksteps = li == 0 ? 1 : 8; }
        if (f < tiles * ksteps) break;
        f -= tiles * ksteps;
    }
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
    if (li < n_layers) {
        const int tile = f % tiles, kk = f / tiles;
        const int row = tile * 16 + r;                    // output feature of the layer = COLUMN of the weight tensor
        for (int j = 0; j < 8; ++j) {
            float w = 0.0f;
            const int hid = 32 * kk + 16 * (j >> 2) + 4 * gg + (j & 3);      // hidden k index of slot (gg, j) in k-step kk
            const int ch = gg == 0 && j < 4 ? j : (gg == 1 && j == 0 ? 4 : -1);   // d(raw) channel of a one-fragment part
            if (a.vd) {
                if (li == 0) { if (ch >= 0 && ch < 3) w = a.t[22][ch * 128 + row]; }                      // rgb_linear [3,128]
                else if (li == 1) w = a.t[16][(size_t)hid * 283 + row];                                    // views [128,283]
                else if (li == 2) {
                    if (kk < 8) w = a.t[18][(size_t)hid * 256 + row];                                      // feature [256,256]
                    else if (ch == 3) w = a.t[20][row];                                                    // alpha [1,256]
                } else {
                    const int i = 10 - li;                                                                 // pts_linears.7 .. 1
                    w = a.t[2 * i][(size_t)hid * (i == 5 ? 319 : 256) + (i == 5 ? 63 : 0) + row];
                }
            } else {
                if (li == 0) { if (ch >= 0 && ch < a.out_ch) w = a.t[18][ch * 256 + row]; }                // output_linear
                else {
                    const int i = 8 - li;
                    w = a.t[2 * i][(size_t)hid * (i == 5 ? 319 : 256) + (i == 5 ? 63 : 0) + row];
                }
            }
            __bf16 b = (__bf16)w;
            for (int t = 0; t < term; ++t) {        // exact differences
                w = __fsub_rn(w, (float)b);
                b = (__bf16)w;
            }
            v[j] = b;
        }
    }
    *reinterpret_cast<bf16x8*>(a.out + (size_t)F * FRAG_BYTES + lane * 16) = v;
}

}  // namespace sp

int split_bwd_padded_frags(bool use_viewdirs) {
    return pad_to_stage(3 * (use_viewdirs ? sp::BWD_FRAGS16_VD : sp::BWD_FRAGS16_NOVD));
}

int launch_pack_split_backward(const float* const* tensors, bool use_viewdirs, int output_ch, void* packed_out, hipStream_t stream) {
    sp::BwdPackArgs a;
    const int n_t = use_viewdirs ? 24 : 20;
    for (int i = 0; i < 24; ++i) a.t[i] = i < n_t ? tensors[i] : nullptr;
    a.out = static_cast<unsigned char*>(packed_out);
    a.vd = use_viewdirs; a.out_ch = output_ch; a.T = 3; a.padded_frags = split_bwd_padded_frags(use_viewdirs);
    const int threads = a.padded_frags * 64;
    hipLaunchKernelGGL(sp::pack_split_bwd_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

int launch_split_dgrad(const void* packed_bwd, bool use_viewdirs, int output_ch, const float* d_raw, long long M,
                       const unsigned* bits, const GTapePtrs& gt, unsigned grid, hipStream_t st) {
    sp::SplitDgradArgs a;
    a.packed = static_cast<const unsigned char*>(packed_bwd);
    a.n_stages = split_bwd_padded_frags(use_viewdirs) / STAGE_FRAGS;
    a.M = M; a.d_raw = d_raw; a.bits = bits; a.gt = gt;
    const dim3 g(grid), b(MLP_THREADS);
    if (use_viewdirs) hipLaunchKernelGGL((sp::mlp_split_dgrad_kernel<true, 4>), g, b, 0, st, a);
    else if (output_ch == 4) hipLaunchKernelGGL((sp::mlp_split_dgrad_kernel<false, 4>), g, b, 0, st, a);
    else hipLaunchKernelGGL((sp::mlp_split_dgrad_kernel<false, 5>), g, b, 0, st, a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

}  // namespace nerf
