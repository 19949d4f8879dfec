// Training kernels of the view-dependent NeRF network (float32, exact MFMA): the forward pass that
// keeps its activations, the input-gradient chain, and the weight-gradient reductions.
//
// The reference trains through PyTorch autograd (run_nerf.py:770-796: loss.backward() over
// NeRF.forward, run_nerf_helpers.py:96-119); nothing below has a line-by-line counterpart there.
// What autograd would compute for the 24 parameter tensors is produced by three kinds of launches:
//
//   forward (mlp_f32_train_kernel)   the inference kernel's layer loop with a per-tile hook that stores
//        every layer input a weight gradient needs ("tape", sample-major rows):
//        enc(x) [M,64], enc(d) [M,32] (kernel k-slot order), h1..h8 [M,256], feature [M,256], view hidden [M,128]
//   input gradients (mlp_dgrad_f32_kernel)   the same register-resident chain run on the TRANSPOSED
//        weights (packed by the same packer, mlp_layout.h make_backward_desc): the gradient tile of one
//        layer is the MFMA B operand of the next; ReLU masks come from the tape; every pre-activation
//        gradient is stored ("gradient tape": dz1..dz8 [M,256], d feature [M,256], d view hidden [M,128])
//   weight gradients (wgrad_kernel)   dW[out, in] = sum_m G[m, out] X[m, in]: samples are the MFMA k
//        dimension, a workgroup owns a contiguous sample range and the full 256 x K tile in registers,
//        partial tiles go to a workspace and a second kernel adds them in a fixed order (deterministic).
//
// Bounds: forward and input gradients MFMA (1.19 and 1.11 MFLOP/sample); weight gradients MFMA with
// 2 KB/sample/layer of tape traffic (HBM-balanced at float32).
#include <math.h>
#include <mutex>
#include "mlp_device.h"
#include "mlp_split.h"
#include "tape.h"
#include "train_bf16.h"
#include "wgrad.h"

namespace nerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Tape rows are plain stores.  Non-temporal ones were measured: the storing kernel gains 4 %, the weight-gradient kernels
// that read the rows next lose more (4096-ray iteration 29.6 -> 31.1 ms).
#define NERF_TAPE_STORE(val, ptr) (*(ptr) = (val))

// per-tile hooks -----------------------------------------------------------------------------
// Accumulator register 4 g + k of tile t holds feature 32 t + 8 g + 4 h + k of the lane's sample: four
// 16-byte pieces per tile and lane in a sample-major row.
struct SaveTile {
    float* row;      // dst + m * ld + 4 h (the tapes are padded to whole 128-sample tiles: every lane stores)
    unsigned* bits;  // 4 dwords of the layer being computed (ReLU layers), or nullptr
    __device__ __forceinline__ void operator()(int t, float (&v)[16]) const {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 q = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            NERF_TAPE_STORE(q, reinterpret_cast<f32x4*>(row + 32 * t + 8 * g));
        }
        if (bits) {
            unsigned w = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {   // > 0 on the float's bits as a signed integer: +0 and -0 give 0
                unsigned b;
                asm("v_med3_i32 %0, %1, 0, 1" : "=v"(b) : "v"(v[e]));
                w |= b << e;
            }
            bits[t >> 1] = (t & 1) ? (bits[t >> 1] | (w << 16)) : w;
        }
    }
};

// gradient tile: zero where the forward activation was not positive (ReLU bits of the forward tape), then store
struct ActTile {};
template <bool MASK>
struct MaskSaveTile {
    const unsigned* bits;   // 4 dwords of the layer (kernel registers), unused without MASK
    float* row;             // destination row (+ 4 h)
    __device__ __forceinline__ ActTile fetch(int) const { return ActTile{}; }
    __device__ __forceinline__ void apply(int t, float (&v)[16], const ActTile&) const {
        const unsigned w = MASK ? bits[t >> 1] >> (16 * (t & 1)) : 0u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MASK) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[4 * g + k] = ((w >> (4 * g + k)) & 1u) ? v[4 * g + k] : 0.0f;
            }
            const f32x4 q = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            NERF_TAPE_STORE(q, reinterpret_cast<f32x4*>(row + 32 * t + 8 * g));
        }
    }
};

// ------------------------------------------------------------------------------------------
// forward with tape
// ------------------------------------------------------------------------------------------
template <bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_f32_train_kernel(MlpArgs a, TapePtrs tp) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    Pipe p;
    pipe_start(p, a, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const long long n_tiles = (a.M + 127) / 128;
    const bool feat_mode = a.mode == NERF_IN_FEATURES;
    const unsigned Mu = (unsigned)a.M;

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned m = (unsigned)tile * 128u + wave * 32 + c;
        const bool valid = m < Mu;
        const unsigned mc = valid ? m : Mu - 1;
        float pe[32], vw[16];
        if (feat_mode) {
            load_features<VD>(a, mc, h, pe, vw);
        } else {
            RawIn cur;
            fetch_raw<VD>(a, mc, cur);
            encode_raw<VD, false>(cur, h, pe, vw);
        }
        {
            float4* q = reinterpret_cast<float4*>(tp.pe + (size_t)m * TAPE_PE + 32 * h);
#pragma unroll
            for (int g = 0; g < 8; ++g) q[g] = make_float4(pe[4 * g], pe[4 * g + 1], pe[4 * g + 2], pe[4 * g + 3]);
            if (VD) {
                float4* qv = reinterpret_cast<float4*>(tp.vw + (size_t)m * TAPE_VW + 16 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g) qv[g] = make_float4(vw[4 * g], vw[4 * g + 1], vw[4 * g + 2], vw[4 * g + 3]);
            }
        }
        unsigned lb[4];                     // ReLU bits of the layer in progress
        auto save256 = [&](float* base, bool relu = true) { return SaveTile{base + (size_t)m * 256 + 4 * h, relu ? lb : nullptr}; };
        uint4* brow = reinterpret_cast<uint4*>(tp.bits + (size_t)m * tape_bits(VD)) + h;     // + 2 per layer
        auto flush_bits = [&](int layer) { brow[2 * layer] = make_uint4(lb[0], lb[1], lb[2], lb[3]); };

        float A[8][16], B[8][16];
        int pos = 0, bt = 0;
        layer_f32<8, true, 0, false, true>(p, pos, bt, h, pe, A, vw, A, save256(tp.h[0]));
        flush_bits(0);
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, A, vw, B, save256(tp.h[1]));
        flush_bits(1);
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, B, vw, A, save256(tp.h[2]));
        flush_bits(2);
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, A, vw, B, save256(tp.h[3]));
        flush_bits(3);
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, B, vw, A, save256(tp.h[4]));
        flush_bits(4);
        layer_f32<8, true, 8, false, true>(p, pos, bt, h, pe, A, vw, B, save256(tp.h[5]));
        flush_bits(5);
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, B, vw, A, save256(tp.h[6]));
        flush_bits(6);
        layer_f32<8, false, 8, false, true>(p, pos, bt, h, pe, A, vw, B, save256(tp.h[7]));
        flush_bits(7);
        if (VD) {
            layer_f32<8, false, 8, false, false>(p, pos, bt, h, pe, B, vw, A, save256(tp.feat, false));
            float o3;
            {
                f32x16 acc = bias_tile(p.lds, bt, h);
                mma_run_f32<32>(p, pos, acc, &B[0][0]);
                pos += 32; bt += 1;
                o3 = acc[3];
            }
            lb[2] = lb[3] = 0u;
            layer_f32<4, false, 8, true, true>(p, pos, bt, h, pe, A, vw, B, SaveTile{tp.hv + (size_t)m * 128 + 4 * h, lb});
            flush_bits(8);
            f32x16 acc = bias_tile(p.lds, bt, h);
            mma_run_f32<16>(p, pos, acc, &B[0][0]);
            if (valid && h == 0) *reinterpret_cast<float4*>(a.out + (size_t)m * 4) = make_float4(acc[0], acc[1], acc[2], o3);
        } else {
            f32x16 acc = bias_tile(p.lds, bt, h);
            mma_run_f32<32>(p, pos, acc, &B[0][0]);
            if (valid) {
                float* o = a.out + (size_t)m * OUT_CH;
                if (OUT_CH == 4) {
                    if (h == 0) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                } else if (h == 0) {
                    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
                } else {
                    o[4] = acc[0];       // row 4 of the tile lives in register 0 of the upper lane half
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// input-gradient chain
// ------------------------------------------------------------------------------------------
struct DgradArgs {
    const unsigned char* packed;   // transposed-weight fragment stream (make_backward_desc)
    int n_stages;
    long long M;
    const float* d_raw;            // [M,4]: gradient of the loss w.r.t. raw (rgb pre-sigmoid, sigma pre-relu)
    TapePtrs tp;
    GTapePtrs gt;
};

template <int NT, int NF, bool EXTRA, class Hook>
__device__ __forceinline__ void layer_bwd_f32(Pipe& p, int& pos, const float* hin, const float (&ex)[4],
                                              float (&hout)[8][16], Hook hook) {
    ActTile nxt = hook.fetch(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const ActTile cur = nxt;
        if (t + 1 < NT) nxt = hook.fetch(t + 1);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        mma_run_f32<NF>(p, pos, acc, hin);
        pos += NF;
        if (EXTRA) { mma_run_f32<1>(p, pos, acc, &ex[0]); pos += 1; }
#pragma unroll
        for (int e = 0; e < 16; ++e) hout[t][e] = acc[e];
        hook.apply(t, hout[t], cur);
    }
}

template <bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_dgrad_f32_kernel(DgradArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    Pipe p;
    {
        MlpArgs ma{};
        ma.packed = a.packed; ma.n_stages = a.n_stages; ma.n_bias_tiles = 0;
        pipe_start(p, ma, lds);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const long long n_tiles = (a.M + 127) / 128;
    const unsigned Mu = (unsigned)a.M;

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned m = (unsigned)tile * 128u + wave * 32 + c;
        const bool valid = m < Mu;
        const unsigned mc = valid ? m : Mu - 1;
        // k slot (j, h) of a one-fragment part is input index j + 4 h (mlp_layout.h, hidden convention, tile 0)
        const float* drp = a.d_raw + (size_t)mc * OUT_CH;
        float dr[5];
#pragma unroll
        for (int k = 0; k < OUT_CH; ++k) dr[k] = drp[k];
        const float none[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        // ReLU bits of all layers of the pass: no load is left inside the layer loop, where the wait for it would
        // also wait for the gradient-tape stores issued before it
        unsigned lb[VD ? 9 : 8][4];
        {
            const uint4* brow = reinterpret_cast<const uint4*>(a.tp.bits + (size_t)m * tape_bits(VD)) + h;
#pragma unroll
            for (int l = 0; l < (VD ? 9 : 8); ++l) {
                const uint4 q = brow[2 * l];
                lb[l][0] = q.x; lb[l][1] = q.y; lb[l][2] = q.z; lb[l][3] = q.w;
            }
        }
        auto rows = [&](int layer, float* dst, int ld) {
            return MaskSaveTile<true>{lb[layer], dst + (size_t)m * ld + 4 * h};
        };
        float A[8][16], B[8][16];
        int pos = 0;
        if (VD) {
            const float drgb[4] = {h ? 0.0f : dr[0], h ? 0.0f : dr[1], h ? 0.0f : dr[2], 0.0f};
            const float dsig[4] = {h ? 0.0f : dr[3], 0.0f, 0.0f, 0.0f};
            layer_bwd_f32<4, 1, false>(p, pos, &drgb[0], none, B, rows(8, a.gt.dzv, 128));
            layer_bwd_f32<8, 16, false>(p, pos, &B[0][0], none, A,
                                        MaskSaveTile<false>{nullptr, a.gt.dfeat + (size_t)m * 256 + 4 * h});
            layer_bwd_f32<8, 32, true>(p, pos, &A[0][0], dsig, B, rows(7, a.gt.dz[7], 256));
        } else {
            const float dout[4] = {h ? (OUT_CH > 4 ? dr[OUT_CH > 4 ? 4 : 0] : 0.0f) : dr[0], h ? 0.0f : dr[1], h ? 0.0f : dr[2],
                                   h ? 0.0f : dr[3]};
            layer_bwd_f32<8, 1, false>(p, pos, &dout[0], none, B, rows(7, a.gt.dz[7], 256));
        }
        layer_bwd_f32<8, 32, false>(p, pos, &B[0][0], none, A, rows(6, a.gt.dz[6], 256));   // pts_linears.7^T
        layer_bwd_f32<8, 32, false>(p, pos, &A[0][0], none, B, rows(5, a.gt.dz[5], 256));   // .6
        layer_bwd_f32<8, 32, false>(p, pos, &B[0][0], none, A, rows(4, a.gt.dz[4], 256));   // .5 (hidden columns)
        layer_bwd_f32<8, 32, false>(p, pos, &A[0][0], none, B, rows(3, a.gt.dz[3], 256));   // .4
        layer_bwd_f32<8, 32, false>(p, pos, &B[0][0], none, A, rows(2, a.gt.dz[2], 256));   // .3
        layer_bwd_f32<8, 32, false>(p, pos, &A[0][0], none, B, rows(1, a.gt.dz[1], 256));   // .2
        layer_bwd_f32<8, 32, false>(p, pos, &B[0][0], none, A, rows(0, a.gt.dz[0], 256));   // .1
        // zero fragments that pad the stream to a whole stage
#pragma unroll
        for (int g = 0; g < pad_to_stage(VD ? FRAGS_BWD_F32_VD : FRAGS_BWD_F32_NOVD) - (VD ? FRAGS_BWD_F32_VD : FRAGS_BWD_F32_NOVD); ++g)
            p.tick(pos + g);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// Adam (run_nerf.py:225: torch.optim.Adam(lr, betas=(0.9, 0.999))), all parameter tensors in one launch
// ------------------------------------------------------------------------------------------
constexpr int ADAM_MAX_TENSORS = 64;
struct AdamArgs {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    long long end[ADAM_MAX_TENSORS];      // exclusive prefix ends of the element counts
    int n_tensors;
    float lr, beta1, beta2, eps, weight_decay;
    float bias1, bias2_sqrt;              // 1 - beta1^t, sqrt(1 - beta2^t)
};

// torch.optim.Adam's single-tensor formula, operation by operation:
//   g += wd * p;  m = m + (1 - b1) (g - m)  [lerp];  v = b2 v + (1 - b2) g g;
//   p -= (lr / bias1) * m / (sqrt(v) / sqrt(bias2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const long long total = a.end[a.n_tensors - 1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int t = 0;
        while (i >= a.end[t]) ++t;
        const long long j = i - (t ? a.end[t - 1] : 0);
        float g = a.g[t][j];
        const float p = a.p[t][j];
        if (a.weight_decay != 0.0f) g = fadd(g, fmul(a.weight_decay, p));
        const float m0 = a.m[t][j], v0 = a.v[t][j];
        const float m = fadd(m0, fmul(fsub(1.0f, a.beta1), fsub(g, m0)));
        const float v = fadd(fmul(a.beta2, v0), fmul(fmul(fsub(1.0f, a.beta2), g), g));
        const float denom = fadd(fdiv(sqrtf(v), a.bias2_sqrt), a.eps);
        a.m[t][j] = m;
        a.v[t][j] = v;
        a.p[t][j] = fsub(p, fmul(fdiv(a.lr, a.bias1), fdiv(m, denom)));
    }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_adam_step(int n_tensors, float* const* params_host, const float* const* grads_host,
                              float* const* exp_avg_host, float* const* exp_avg_sq_host, const long long* numel_host,
                              long long step, float lr, float beta1, float beta2, float eps, float weight_decay,
                              void* stream) {
    NERF_REQUIRE(n_tensors >= 1 && n_tensors <= ADAM_MAX_TENSORS, "nerf_adam_step: 1..%d tensors per call, got %d",
                 ADAM_MAX_TENSORS, n_tensors);
    NERF_REQUIRE(params_host && grads_host && exp_avg_host && exp_avg_sq_host && numel_host, "nerf_adam_step: null pointer");
    NERF_REQUIRE(step >= 1, "nerf_adam_step: step counts from 1, got %lld", step);
    AdamArgs a;
    long long end = 0;
    for (int i = 0; i < n_tensors; ++i) {
        NERF_REQUIRE(numel_host[i] >= 0, "nerf_adam_step: negative size");
        NERF_REQUIRE(numel_host[i] == 0 || (params_host[i] && grads_host[i] && exp_avg_host[i] && exp_avg_sq_host[i]),
                     "nerf_adam_step: tensor %d has a null buffer", i);
        a.p[i] = params_host[i]; a.g[i] = grads_host[i]; a.m[i] = exp_avg_host[i]; a.v[i] = exp_avg_sq_host[i];
        end += numel_host[i];
        a.end[i] = end;
    }
    if (end == 0) return NERF_OK;
    a.n_tensors = n_tensors;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bias1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    long long blocks = (end + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

static int check_arch(const char* who, int use_viewdirs, int output_ch) {
    if (use_viewdirs) NERF_REQUIRE(output_ch == 4, "%s: output_ch must be 4 with view directions", who);
    else NERF_REQUIRE(output_ch == 4 || output_ch == 5, "%s: output_ch must be 4 or 5, got %d", who, output_ch);
    return NERF_OK;
}
static int bwd_frags(int use_viewdirs) { return use_viewdirs ? FRAGS_BWD_F32_VD : FRAGS_BWD_F32_NOVD; }

// NERF_PREC_BF16X3 trains on the float32 tapes: forward pass (mlp_split.hip) and input-gradient chain (dgrad_split.hip)
// evaluate the float32 network in three bf16 terms, the weight gradients are the float32 kernels'.
static bool known_precision(int p) { return p == NERF_PREC_F32 || p == NERF_PREC_BF16 || p == NERF_PREC_BF16X3; }

extern "C" size_t nerf_train_tape_bytes(long long M, int use_viewdirs, int precision) {
    if (M < 0 || !known_precision(precision)) return 0;
    if (precision == NERF_PREC_BF16) return tape16_bytes(M, use_viewdirs != 0);
    return (size_t)pad128(M) * tape_floats(use_viewdirs != 0) * sizeof(float);
}
extern "C" size_t nerf_train_grad_tape_bytes(long long M, int use_viewdirs, int precision) {
    if (M < 0 || !known_precision(precision)) return 0;
    if (precision == NERF_PREC_BF16) return gtape16_bytes(M, use_viewdirs != 0);
    return (size_t)pad128(M) * gtape_floats(use_viewdirs != 0) * sizeof(float);
}
extern "C" size_t nerf_train_workspace_bytes(void) {
    return (WG_BIAS_OFFSET + (size_t)WG_MAX_PARTS * 256) * sizeof(float);
}
extern "C" size_t nerf_mlp_backward_packed_bytes(int use_viewdirs, int precision) {
    if (!known_precision(precision)) return 0;
    if (precision == NERF_PREC_BF16)
        return (size_t)pad_to_stage(use_viewdirs ? FRAGS_BWD_BF16_VD : FRAGS_BWD_BF16_NOVD) * FRAG_BYTES;
    if (precision == NERF_PREC_BF16X3) return (size_t)split_bwd_padded_frags(use_viewdirs != 0) * FRAG_BYTES;
    return (size_t)pad_to_stage(bwd_frags(use_viewdirs)) * FRAG_BYTES;
}

extern "C" int nerf_mlp_pack_backward(const float* const* tensors_host, int use_viewdirs, int output_ch, int precision,
                                      void* packed_out, void* stream) {
    NERF_REQUIRE(tensors_host && packed_out, "nerf_mlp_pack_backward: null pointer");
    NERF_REQUIRE(known_precision(precision), "nerf_mlp_pack_backward: unknown precision %d", precision);
    if (int rc = check_arch("nerf_mlp_pack_backward", use_viewdirs, output_ch)) return rc;
    NERF_REQUIRE(((uintptr_t)packed_out & 15) == 0, "nerf_mlp_pack_backward: packed_out must be 16-byte aligned");
    if (precision == NERF_PREC_BF16)
        return pack_backward_bf16(tensors_host, use_viewdirs != 0, output_ch, packed_out, as_stream(stream));
    if (precision == NERF_PREC_BF16X3) {
        const int n_tensors = use_viewdirs ? 24 : 20;
        for (int i = 0; i < n_tensors; i += 2)
            if (use_viewdirs || i != 16) NERF_REQUIRE(tensors_host[i], "nerf_mlp_pack_backward: weight tensor %d is null", i);
        return launch_pack_split_backward(tensors_host, use_viewdirs != 0, output_ch, packed_out, as_stream(stream));
    }
    PackArgs a;
    a.net = make_backward_desc(use_viewdirs != 0, output_ch);
    a.bf16 = 0;
    a.with_scales = 0;
    a.total_frags = total_frags(a.net, 0);
    a.padded_frags = pad_to_stage(bwd_frags(use_viewdirs));
    NERF_REQUIRE(a.total_frags == bwd_frags(use_viewdirs), "nerf_mlp_pack_backward: internal layout mismatch");
    const int n_t = use_viewdirs ? 24 : 20;
    for (int i = 0; i < 24; ++i) a.t[i] = i < n_t ? tensors_host[i] : nullptr;
    for (int i = 0; i < a.net.n_layers; ++i) {
        NERF_REQUIRE(a.t[a.net.layer[i].w_idx], "nerf_mlp_pack_backward: weight tensor %d is null", a.net.layer[i].w_idx);
        if (a.net.layer[i].extra_frags) NERF_REQUIRE(a.t[a.net.layer[i].w2_idx], "nerf_mlp_pack_backward: weight tensor %d is null", a.net.layer[i].w2_idx);
    }
    a.out = static_cast<unsigned char*>(packed_out);
    return launch_pack(a, as_stream(stream));
}

extern "C" int nerf_mlp_forward_train(const void* packed, int precision, int use_viewdirs, int output_ch, int input_mode,
                                      const float* in0, const float* in1, int R, int S, int ray_cols, float* raw_out,
                                      void* tape_v, void* stream) {
    float* tape = static_cast<float*>(tape_v);
    NERF_REQUIRE(packed && (R == 0 || (in0 && raw_out && tape)), "nerf_mlp_forward_train: null pointer");
    NERF_REQUIRE(known_precision(precision), "nerf_mlp_forward_train: unknown precision %d", precision);
    NERF_REQUIRE(R >= 0 && S >= 1, "nerf_mlp_forward_train: bad shape R=%d S=%d", R, S);
    if (int rc = check_arch("nerf_mlp_forward_train", use_viewdirs, output_ch)) return rc;
    switch (input_mode) {
        case NERF_IN_FEATURES: NERF_REQUIRE(S == 1, "nerf_mlp_forward_train: feature input is flat (S = 1)"); break;
        case NERF_IN_POINTS: NERF_REQUIRE(R == 0 || !use_viewdirs || in1, "nerf_mlp_forward_train: view directions missing"); break;
        case NERF_IN_RAYS:
            NERF_REQUIRE(R == 0 || in1, "nerf_mlp_forward_train: z_vals missing");
            NERF_REQUIRE(ray_cols >= (use_viewdirs ? 11 : 8), "nerf_mlp_forward_train: rays need %d columns, got %d",
                         use_viewdirs ? 11 : 8, ray_cols);
            break;
        default: NERF_REQUIRE(false, "nerf_mlp_forward_train: unknown input mode %d", input_mode);
    }
    NERF_REQUIRE(((uintptr_t)tape & 15) == 0 && ((uintptr_t)packed & 15) == 0 && (output_ch != 4 || ((uintptr_t)raw_out & 15) == 0),
                 "nerf_mlp_forward_train: buffers must be 16-byte aligned");
    const long long M = (long long)R * S;
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(M < (1ll << 31) - 512, "nerf_mlp_forward_train: at most 2^31 - 512 samples per launch, got %lld", M);
    MlpArgs a;
    a.packed = static_cast<const unsigned char*>(packed);
    a.n_stages = (use_viewdirs ? FRAGS_F32_VD : FRAGS_F32_NOVD) / STAGE_FRAGS;
    a.n_bias_tiles = use_viewdirs ? BIAS_TILES_VD : BIAS_TILES_NOVD;
    a.mode = input_mode; a.in0 = in0; a.in1 = in1; a.R = R; a.S = S; a.ray_cols = ray_cols; a.M = M; a.out = raw_out;
    set_sample_divisor(a);
    a.tape = nullptr;
    a.table = nullptr; a.n_obj = 1; a.tiles_per_obj = 0; a.table_off = 0; a.samples_per_obj = M;
    hipStream_t st = as_stream(stream);
    if (precision == NERF_PREC_BF16) {
        a.n_stages = (use_viewdirs ? FRAGS_BF16_VD : FRAGS_BF16_NOVD) / STAGE_FRAGS;
        a.tape = static_cast<unsigned char*>(tape_v);
        return launch_mlp_bf16_train(a, use_viewdirs != 0, output_ch, st);
    }
    const long long n_tiles = (M + 127) / 128;
    const dim3 grid((unsigned)(n_tiles < cu_count() ? n_tiles : cu_count())), block(MLP_THREADS);
    const TapePtrs tp = tape_ptrs(tape, M, use_viewdirs != 0);
    if (precision == NERF_PREC_BF16X3) {
        static_assert(sizeof(float) * 256 == 1024, "the split kernel addresses tape rows with 32-bit byte offsets");
        NERF_REQUIRE(M <= (1ll << 22), "nerf_mlp_forward_train: at most 4 194 304 samples per launch in bf16x3, got %lld", M);
        NERF_REQUIRE(split_samples_per_tile() == 128, "nerf_mlp_forward_train: internal tile size mismatch");
        a.n_stages = split_padded_frags(3, use_viewdirs != 0) / STAGE_FRAGS;
        a.n_bias_tiles = ((use_viewdirs ? BIAS_TILES16_VD : BIAS_TILES16_NOVD) * 16 + 31) / 32;    // in units of 32 floats
        return launch_mlp_split_train(a, tp, use_viewdirs != 0, output_ch, grid.x, st);
    }
    if (use_viewdirs) hipLaunchKernelGGL((mlp_f32_train_kernel<true, 4>), grid, block, 0, st, a, tp);
    else if (output_ch == 4) hipLaunchKernelGGL((mlp_f32_train_kernel<false, 4>), grid, block, 0, st, a, tp);
    else hipLaunchKernelGGL((mlp_f32_train_kernel<false, 5>), grid, block, 0, st, a, tp);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

extern "C" int nerf_mlp_backward(const void* packed_bwd, const void* packed_fwd, int precision, int use_viewdirs,
                                 int output_ch, const float* d_raw, long long M, const void* tape_v, void* grad_tape_v,
                                 float* workspace, size_t workspace_bytes, float* const* grads_host, void* stream) {
    const float* tape = static_cast<const float*>(tape_v);
    float* grad_tape = static_cast<float*>(grad_tape_v);
    NERF_REQUIRE(packed_bwd && d_raw && tape && grad_tape && workspace && grads_host, "nerf_mlp_backward: null pointer");
    NERF_REQUIRE(known_precision(precision), "nerf_mlp_backward: unknown precision %d", precision);
    NERF_REQUIRE(precision != NERF_PREC_BF16 || packed_fwd, "nerf_mlp_backward: the bf16 backward needs the forward stream");
    if (int rc = check_arch("nerf_mlp_backward", use_viewdirs, output_ch)) return rc;
    NERF_REQUIRE(workspace_bytes >= nerf_train_workspace_bytes(), "nerf_mlp_backward: workspace of %zu bytes, %zu needed",
                 workspace_bytes, nerf_train_workspace_bytes());
    NERF_REQUIRE(((uintptr_t)d_raw & 15) == 0 && ((uintptr_t)tape & 15) == 0 && ((uintptr_t)grad_tape & 15) == 0,
                 "nerf_mlp_backward: buffers must be 16-byte aligned");
    NERF_REQUIRE(M >= 1 && M < (1ll << 31) - 512, "nerf_mlp_backward: bad sample count %lld", M);
    const int n_t = use_viewdirs ? 24 : 20;
    for (int i = 0; i < n_t; ++i)
        if (use_viewdirs || (i != 16 && i != 17)) NERF_REQUIRE(grads_host[i], "nerf_mlp_backward: gradient tensor %d is null", i);
    hipStream_t st = as_stream(stream);
    if (precision == NERF_PREC_BF16)
        return backward_bf16(packed_bwd, packed_fwd, use_viewdirs != 0, output_ch, d_raw, M, tape_v, grad_tape_v, workspace,
                             grads_host, st);
    const TapePtrs tp = tape_ptrs(const_cast<float*>(tape), M, use_viewdirs != 0);
    const GTapePtrs gt = gtape_ptrs(grad_tape, M, use_viewdirs != 0);
    const long long n_tiles = (M + 127) / 128;
    const dim3 grid((unsigned)(n_tiles < cu_count() ? n_tiles : cu_count())), block(MLP_THREADS);
    if (precision == NERF_PREC_BF16X3) {
        NERF_REQUIRE(M <= (1ll << 22), "nerf_mlp_backward: at most 4 194 304 samples per call in bf16x3, got %lld", M);
        if (int rc = launch_split_dgrad(packed_bwd, use_viewdirs != 0, output_ch, d_raw, M, tp.bits, gt, grid.x, st)) return rc;
    } else {
    DgradArgs d;
    d.packed = static_cast<const unsigned char*>(packed_bwd);
    d.n_stages = pad_to_stage(bwd_frags(use_viewdirs)) / STAGE_FRAGS;
    d.M = M; d.d_raw = d_raw; d.tp = tp; d.gt = gt;
    if (use_viewdirs) hipLaunchKernelGGL((mlp_dgrad_f32_kernel<true, 4>), grid, block, 0, st, d);
    else if (output_ch == 4) hipLaunchKernelGGL((mlp_dgrad_f32_kernel<false, 4>), grid, block, 0, st, d);
    else hipLaunchKernelGGL((mlp_dgrad_f32_kernel<false, 5>), grid, block, 0, st, d);
    NERF_LAUNCH_CHECK();
    }

    // state_dict order: pts_linears.i -> 2i, 2i+1; views_linears.0 16, 17; then with view directions feature 18, 19,
    // alpha 20, 21, rgb 22, 23; without them output_linear 18, 19 (views_linears.0 is unused and gets no gradient)
    float* const* g = grads_host;
    int rc;
#define NERF_TRY(x) do { rc = (x); if (rc != NERF_OK) return rc; } while (0)
    NERF_TRY(launch_wgrad(gt.dz[0], 256, 256, tp.pe, TAPE_PE, 64, M, workspace, g[0], 63, 0, 1, g[1], st));
    for (int i = 1; i < 8; ++i) {
        const int ld = i == 5 ? 319 : 256, off = i == 5 ? 63 : 0;
        NERF_TRY(launch_wgrad(gt.dz[i], 256, 256, tp.h[i - 1], 256, 256, M, workspace, g[2 * i], ld, off, 0, g[2 * i + 1], st));
        if (i == 5) NERF_TRY(launch_wgrad(gt.dz[5], 256, 256, tp.pe, TAPE_PE, 64, M, workspace, g[10], 319, 0, 1, nullptr, st));
    }
    if (use_viewdirs) {
        NERF_TRY(launch_wgrad(gt.dfeat, 256, 256, tp.h[7], 256, 256, M, workspace, g[18], 256, 0, 0, g[19], st));
        NERF_TRY(launch_wgrad(gt.dzv, 128, 128, tp.feat, 256, 256, M, workspace, g[16], 283, 0, 0, g[17], st));
        NERF_TRY(launch_wgrad(gt.dzv, 128, 128, tp.vw, TAPE_VW, 32, M, workspace, g[16], 283, 256, 2, nullptr, st));
        NERF_TRY(launch_wgrad_small(d_raw, 4, 3, 1, tp.h[7], 256, M, workspace, g[20], g[21], st));
        NERF_TRY(launch_wgrad_small(d_raw, 4, 0, 3, tp.hv, 128, M, workspace, g[22], g[23], st));
    } else {
        NERF_TRY(launch_wgrad_small(d_raw, output_ch, 0, output_ch, tp.h[7], 256, M, workspace, g[18], g[19], st));
    }
#undef NERF_TRY
    return NERF_OK;
}

// ------------------------------------------------------------------------------------------
// nerf_mlp_backward without caller-provided scratch: the gradient tape and the partial-sum workspace live in a
// static device array of this library, one copy per GPU, held until the process ends.
// ------------------------------------------------------------------------------------------
// largest gradient tape at the sample limit: float32, with view directions, rows padded to 128 samples
constexpr size_t STATIC_GTAPE_BYTES =
    (size_t)((NERF_STATIC_MAX_SAMPLES + 127) / 128 * 128) * gtape_floats(true) * sizeof(float);
static_assert(STATIC_GTAPE_BYTES % 256 == 0, "the workspace behind the tape stays 256-byte aligned");
constexpr size_t STATIC_WS_BYTES = (WG_BIAS_OFFSET + (size_t)WG_MAX_PARTS * 256) * sizeof(float);
constexpr size_t STATIC_SCRATCH_BYTES = STATIC_GTAPE_BYTES + STATIC_WS_BYTES;

static __device__ __align__(256) unsigned char g_static_scratch[STATIC_SCRATCH_BYTES];

// Device address of g_static_scratch on the CURRENT device.  In host code the array's name is only a handle: the
// memory is reached through what hipGetSymbolAddress returns, looked up once per device and kept.
static int static_scratch_base(unsigned char** out) {
    constexpr int MAX_DEV = 64;
    static unsigned char* base[MAX_DEV] = {nullptr};
    static std::mutex mu;
    int dev = -1;
    NERF_HIP_CHECK(hipGetDevice(&dev));
    NERF_REQUIRE(dev >= 0 && dev < MAX_DEV, "nerf_mlp_backward_static: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!base[dev]) {
        void* p = nullptr;
        NERF_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch)));
        NERF_REQUIRE(p && ((uintptr_t)p & 255) == 0, "nerf_mlp_backward_static: bad address of the static scratch array");
        base[dev] = static_cast<unsigned char*>(p);
    }
    *out = base[dev];
    return NERF_OK;
}

static_assert(STATIC_SCRATCH_BYTES == NERF_STATIC_SCRATCH_BYTES, "the header states the size of the static array");

extern "C" int nerf_mlp_backward_static(const void* packed_bwd, const void* packed_fwd, int precision, int use_viewdirs,
                                        int output_ch, const float* d_raw, long long M, const void* tape,
                                        float* const* grads_host, void* stream) {
    NERF_REQUIRE(known_precision(precision), "nerf_mlp_backward_static: unknown precision %d", precision);
    NERF_REQUIRE(M >= 1, "nerf_mlp_backward_static: bad sample count %lld", M);
    NERF_REQUIRE(M <= NERF_STATIC_MAX_SAMPLES,
                 "nerf_mlp_backward_static: %lld samples exceed the static scratch (at most %d per call); use "
                 "nerf_mlp_backward with caller-provided buffers", M, NERF_STATIC_MAX_SAMPLES);
    NERF_REQUIRE(nerf_train_grad_tape_bytes(M, use_viewdirs, precision) <= STATIC_GTAPE_BYTES,
                 "nerf_mlp_backward_static: internal size mismatch");
    unsigned char* base = nullptr;
    if (int rc = static_scratch_base(&base)) return rc;
    // nothing in the backward reads scratch it has not written in the same call (the caller-provided buffers of
    // nerf_mlp_backward are uninitialised as well), so the previous call's contents need no clearing
    return nerf_mlp_backward(packed_bwd, packed_fwd, precision, use_viewdirs, output_ch, d_raw, M, tape, base,
                             reinterpret_cast<float*>(base + STATIC_GTAPE_BYTES), STATIC_WS_BYTES, grads_host, stream);
}
