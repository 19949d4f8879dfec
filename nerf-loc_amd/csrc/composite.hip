// Alpha compositing (raw2outputs, run_nerf.py:268-311; nerf_utils.py:344-395).
//
// One wavefront per ray.  The 64 lanes take 64 consecutive samples, the exclusive
// transmittance product is a wave-level scan (6 shuffle steps), longer rays carry the
// running product from chunk to chunk.  All loads are coalesced: one float4 of `raw` and one
// float of `z` per lane.  HBM-bound: 24*S + 36 algorithmic bytes per ray.
#include "common.h"

namespace nerf {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// sigmoid and 1 - exp(-x) through the hardware exp2 / rcp units (about 1 ulp each): the maps are compared at
// 1e-4 against the reference, nothing here is part of a bit-exact contract.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_exp_neg(float x) {   // exp(-x)
    return __builtin_amdgcn_exp2f(-1.4426950408889634f * x);
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
constexpr int DPP_ROW_SHR = 0x110, DPP_ROW_SHL = 0x100, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138;
__device__ __forceinline__ float row_sum_to_last(float s) {      // lane 15 of each row ends up with the row's sum
    s += dpp_f<DPP_ROW_SHR + 1>(0.f, s);
    s += dpp_f<DPP_ROW_SHR + 2>(0.f, s);
    s += dpp_f<DPP_ROW_SHR + 4>(0.f, s);
    s += dpp_f<DPP_ROW_SHR + 8>(0.f, s);
    return s;
}
// sum over the wavefront, valid in lane 63: row sums, then the totals of rows 0 and 2 into rows 1 and 3, then the total
// of the lower half into the upper one
__device__ __forceinline__ float wave_sum_to_last(float s) {
    s = row_sum_to_last(s);
    s += dpp_f<DPP_ROW_BCAST15, 0xa>(0.f, s);
    s += dpp_f<DPP_ROW_BCAST31, 0xc>(0.f, s);
    return s;
}
// inclusive product scan over the wavefront (no LDS traffic: six DPP multiplications)
__device__ __forceinline__ float wave_prod_scan(float p) {
    p *= dpp_f<DPP_ROW_SHR + 1>(1.f, p);
    p *= dpp_f<DPP_ROW_SHR + 2>(1.f, p);
    p *= dpp_f<DPP_ROW_SHR + 4>(1.f, p);
    p *= dpp_f<DPP_ROW_SHR + 8>(1.f, p);
    p *= dpp_f<DPP_ROW_BCAST15, 0xa>(1.f, p);
    p *= dpp_f<DPP_ROW_BCAST31, 0xc>(1.f, p);
    return p;
}

// NCH chunks of 64 samples, all loaded before the first scan so that a wave keeps NCH x 24 B per lane in flight.
// A wavefront takes RPW consecutive rays (unrolled, so the loads of all of them are in flight together).
template <int RAW_CH, int NCH, int RPW>
__global__ __launch_bounds__(256) void raw2outputs_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ d_ptr,
    int d_stride, const float* __restrict__ noise, int R, int S, int white_bkgd,
    float* __restrict__ rgb_out, float* __restrict__ disp_out, float* __restrict__ acc_out,
    float* __restrict__ depth_out, float* __restrict__ weights_out) {
    const int lane = threadIdx.x & 63;
    const int ray0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW;
#pragma unroll
    for (int rq = 0; rq < RPW; ++rq) {
    const int ray = ray0 + rq;
    if (ray >= R) break;   // whole wave leaves together

    const float* zr = z_vals + (size_t)ray * S;
    const float* rr = raw + (size_t)ray * S * RAW_CH;
    const float* nr = noise ? noise + (size_t)ray * S : nullptr;
    const float* dp = d_ptr + (size_t)ray * d_stride;

    const float dx = dp[0], dy = dp[1], dz = dp[2];
    const float dnorm = sqrtf(fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz)));
    float carry = 1.0f;                    // product of (1-alpha+1e-10) over all earlier chunks
    float s_r = 0.f, s_g = 0.f, s_b = 0.f, s_depth = 0.f, s_acc = 0.f;
    for (int base = 0; base < S; base += 64 * NCH) {    // one trip unless S > 64 NCH
    float r0[NCH], r1[NCH], r2[NCH], r3[NCH], zc[NCH], zn[NCH], nz[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = base + 64 * k + lane;
        const int ic = i < S ? i : S - 1;
        if (RAW_CH == 4) {
            const float4 q = reinterpret_cast<const float4*>(rr)[ic];
            r0[k] = q.x; r1[k] = q.y; r2[k] = q.z; r3[k] = q.w;
        } else {
            const float* q = rr + (size_t)ic * RAW_CH;
            r0[k] = q[0]; r1[k] = q[1]; r2[k] = q[2]; r3[k] = q[3];
        }
        zc[k] = zr[ic];
        zn[k] = zr[ic + 1 < S ? ic + 1 : S - 1];
        nz[k] = nr ? nr[ic] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = base + 64 * k + lane;
        const bool valid = i < S;
        float alpha = 0.f, z = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (valid) {
            z = zc[k];
            // dists = [z[1:]-z[:-1], 1e10] * |d|
            const float dist = fmul((i + 1 < S) ? fsub(zn[k], z) : 1e10f, dnorm);
            const float sig = fmaxf(fadd(r3[k], nz[k]), 0.0f);
            alpha = fsub(1.0f, fast_exp_neg(fmul(sig, dist)));
            c0 = fast_sigmoid(r0[k]);
            c1 = fast_sigmoid(r1[k]);
            c2 = fast_sigmoid(r2[k]);
        }
        // inclusive product scan of (1 - alpha + 1e-10) over the wave, made exclusive by a shift
        const float one_m = fadd(fsub(1.0f, alpha), 1e-10f);
        const float incl = wave_prod_scan(one_m);
        const float excl = dpp_f<DPP_WAVE_SHR1>(1.0f, incl);      // lane 0 keeps 1
        const float T = carry * excl;
        const float w = alpha * T;
        carry *= __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), WAVE - 1));
        if (valid && weights_out) weights_out[(size_t)ray * S + i] = w;
        s_r += w * c0; s_g += w * c1; s_b += w * c2;
        s_depth += w * z;
        s_acc += w;
    }
    }
    s_r = wave_sum_to_last(s_r); s_g = wave_sum_to_last(s_g); s_b = wave_sum_to_last(s_b);
    s_depth = wave_sum_to_last(s_depth); s_acc = wave_sum_to_last(s_acc);
    if (lane == WAVE - 1) {
        if (rgb_out) {
            const float bg = white_bkgd ? fsub(1.0f, s_acc) : 0.0f;
            rgb_out[(size_t)ray * 3 + 0] = s_r + bg;
            rgb_out[(size_t)ray * 3 + 1] = s_g + bg;
            rgb_out[(size_t)ray * 3 + 2] = s_b + bg;
        }
        if (disp_out) {
            // 1 / max(1e-10, depth/acc); torch.max propagates the NaN of 0/0, fmaxf would not
            const float q = fdiv(s_depth, s_acc);
            const float m = (q != q) ? q : fmaxf(1e-10f, q);
            disp_out[ray] = fdiv(1.0f, m);
        }
        if (acc_out) acc_out[ray] = s_acc;
        if (depth_out) depth_out[ray] = s_depth;
    }
    }
}

// Rays of at most 64 samples (the coarse pass): a ray is taken by ONE 16-lane row of the wavefront, four consecutive
// samples per lane, so a wave works on four rays at once and every scan / sum is four DPP row shifts: no cross-row
// traffic, no LDS.  Loads are 16-byte (raw: one float4 per sample, z: one float4 per lane) when the row length allows.
template <int RAW_CH, bool VEC>
__global__ __launch_bounds__(256) void raw2outputs_row_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ d_ptr,
    int d_stride, const float* __restrict__ noise, int R, int S, int white_bkgd,
    float* __restrict__ rgb_out, float* __restrict__ disp_out, float* __restrict__ acc_out,
    float* __restrict__ depth_out, float* __restrict__ weights_out) {
    const int j = threadIdx.x & 15;                                   // lane in the row: samples 4 j .. 4 j + 3
    const long long ray_u = (long long)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const bool ray_ok = ray_u < R;
    const size_t ray = ray_ok ? (size_t)ray_u : (size_t)(R - 1);      // idle rows redo the last ray and store nothing
    const float* zr = z_vals + ray * S;
    const float* rr = raw + ray * S * RAW_CH;
    const float* nr = noise ? noise + ray * S : nullptr;
    const float* dp = d_ptr + ray * d_stride;
    const float dx = dp[0], dy = dp[1], dz = dp[2];
    const float dnorm = sqrtf(fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz)));

    float r0[4], r1[4], r2[4], r3[4], z[5], nz[4];
    const int i0 = 4 * j;
    if (VEC) {          // S % 4 == 0, RAW_CH == 4, 16-byte aligned bases: whole quads are valid or not
        const int q = i0 < S ? i0 : S - 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = reinterpret_cast<const float4*>(rr)[q + k];
            r0[k] = v.x; r1[k] = v.y; r2[k] = v.z; r3[k] = v.w;
        }
        const float4 zq = *reinterpret_cast<const float4*>(zr + q);
        z[0] = zq.x; z[1] = zq.y; z[2] = zq.z; z[3] = zq.w;
        if (nr) { const float4 nq = *reinterpret_cast<const float4*>(nr + q); nz[0] = nq.x; nz[1] = nq.y; nz[2] = nq.z; nz[3] = nq.w; }
        else { nz[0] = nz[1] = nz[2] = nz[3] = 0.f; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ic = i0 + k < S ? i0 + k : S - 1;
            const float* v = rr + (size_t)ic * RAW_CH;
            r0[k] = v[0]; r1[k] = v[1]; r2[k] = v[2]; r3[k] = v[3];
            z[k] = zr[ic];
            nz[k] = nr ? nr[ic] : 0.f;
        }
    }
    // the depth after this lane's four: the next lane's first (row shift left by one; the last lane keeps its own)
    z[4] = dpp_f<DPP_ROW_SHL + 1>(z[3], z[0]);

    float alpha[4], om[4], c0[4], c1[4], c2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = i0 + k;
        const bool valid = i < S;
        alpha[k] = 0.f; c0[k] = c1[k] = c2[k] = 0.f;
        if (valid) {
            const float dist = fmul((i + 1 < S) ? fsub(z[k + 1], z[k]) : 1e10f, dnorm);
            const float sig = fmaxf(fadd(r3[k], nz[k]), 0.0f);
            alpha[k] = fsub(1.0f, fast_exp_neg(fmul(sig, dist)));
            c0[k] = fast_sigmoid(r0[k]);
            c1[k] = fast_sigmoid(r1[k]);
            c2[k] = fast_sigmoid(r2[k]);
        }
        om[k] = fadd(fsub(1.0f, alpha[k]), 1e-10f);
    }
    // exclusive transmittance: product over the lane's earlier samples times the product over all earlier lanes
    const float p1 = om[0], p2 = p1 * om[1], p3 = p2 * om[2], p4 = p3 * om[3];
    float incl = p4;
    incl *= dpp_f<DPP_ROW_SHR + 1>(1.f, incl);
    incl *= dpp_f<DPP_ROW_SHR + 2>(1.f, incl);
    incl *= dpp_f<DPP_ROW_SHR + 4>(1.f, incl);
    incl *= dpp_f<DPP_ROW_SHR + 8>(1.f, incl);
    const float before = dpp_f<DPP_ROW_SHR + 1>(1.f, incl);
    const float T[4] = {before, before * p1, before * p2, before * p3};
    float w[4], s_r = 0.f, s_g = 0.f, s_b = 0.f, s_depth = 0.f, s_acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w[k] = alpha[k] * T[k];
        s_r += w[k] * c0[k]; s_g += w[k] * c1[k]; s_b += w[k] * c2[k];
        s_depth += w[k] * z[k];
        s_acc += w[k];
    }
    if (weights_out && ray_ok) {
        float* wo = weights_out + ray * S;
        if (VEC) { if (i0 < S) *reinterpret_cast<float4*>(wo + i0) = make_float4(w[0], w[1], w[2], w[3]); }
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i0 + k < S) wo[i0 + k] = w[k];
        }
    }
    s_r = row_sum_to_last(s_r); s_g = row_sum_to_last(s_g); s_b = row_sum_to_last(s_b);
    s_depth = row_sum_to_last(s_depth); s_acc = row_sum_to_last(s_acc);
    if (j == 15 && ray_ok) {
        if (rgb_out) {
            const float bg = white_bkgd ? fsub(1.0f, s_acc) : 0.0f;
            rgb_out[ray * 3 + 0] = s_r + bg;
            rgb_out[ray * 3 + 1] = s_g + bg;
            rgb_out[ray * 3 + 2] = s_b + bg;
        }
        if (disp_out) {
            const float q = fdiv(s_depth, s_acc);
            const float m = (q != q) ? q : fmaxf(1e-10f, q);
            disp_out[ray] = fdiv(1.0f, m);
        }
        if (acc_out) acc_out[ray] = s_acc;
        if (depth_out) depth_out[ray] = s_depth;
    }
}

// Backward of raw2outputs.  The reference obtains it from autograd over run_nerf.py:268-311; here the
// forward quantities are recomputed per ray and the chain rule is applied in closed form:
//   c = sigmoid(raw_rgb), e = exp(-sigma dist), alpha = 1 - e, T_i = prod_{j<i} (1 - alpha_j + 1e-10), w = alpha T
//   q_i = dL/dw_i = g_rgb . c_i - [white] sum(g_rgb) + g_acc + g_depth z_i
//   dL/dalpha_i = q_i T_i - (sum_{k>i} q_k w_k) / (1 - alpha_i + 1e-10)        (suffix sum: reverse wave scan)
//   dL/draw_sigma_i = [raw_sigma_i + noise_i > 0] dL/dalpha_i dist_i e_i,   dL/draw_rgb_i = w_i g_rgb c_i (1 - c_i)
// g_disp folds into g_depth / g_acc through disp = acc / depth.  One wavefront per ray, NCH chunks of 64
// samples kept in registers.
template <int RAW_CH, int NCH>
__global__ __launch_bounds__(256) void raw2outputs_bwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ d_ptr, int d_stride,
    const float* __restrict__ noise, int R, int S, int white_bkgd, const float* __restrict__ g_rgb,
    const float* __restrict__ g_disp, const float* __restrict__ g_acc, const float* __restrict__ g_depth,
    const float* __restrict__ g_weights, float* __restrict__ d_raw) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float* dp = d_ptr + (size_t)ray * d_stride;
    const float dnorm = sqrtf(fadd(fadd(fmul(dp[0], dp[0]), fmul(dp[1], dp[1])), fmul(dp[2], dp[2])));
    const float* zr = z_vals + (size_t)ray * S;
    const float* rr = raw + (size_t)ray * S * RAW_CH;
    const float* nr = noise ? noise + (size_t)ray * S : nullptr;

    float alpha[NCH], T[NCH], e_[NCH], dist[NCH], z[NCH], c0[NCH], c1[NCH], c2[NCH], one_m[NCH];
    bool gate[NCH];
    float carry = 1.0f, s_acc = 0.f, s_depth = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = 64 * k + lane;
        const bool valid = i < S;
        alpha[k] = 0.f; e_[k] = 1.f; dist[k] = 0.f; z[k] = 0.f; c0[k] = c1[k] = c2[k] = 0.f; gate[k] = false;
        if (valid) {
            const float* q = rr + (size_t)i * RAW_CH;
            z[k] = zr[i];
            dist[k] = fmul((i + 1 < S) ? fsub(zr[i + 1], z[k]) : 1e10f, dnorm);
            float sig = q[3];
            if (nr) sig = fadd(sig, nr[i]);
            gate[k] = sig > 0.0f;
            sig = fmaxf(sig, 0.0f);
            e_[k] = expf(-fmul(sig, dist[k]));
            alpha[k] = fsub(1.0f, e_[k]);
            c0[k] = fdiv(1.0f, fadd(1.0f, expf(-q[0])));
            c1[k] = fdiv(1.0f, fadd(1.0f, expf(-q[1])));
            c2[k] = fdiv(1.0f, fadd(1.0f, expf(-q[2])));
        }
        one_m[k] = fadd(fsub(1.0f, alpha[k]), 1e-10f);
        float incl = one_m[k];
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const float up = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl *= up;
        }
        float excl = __shfl_up(incl, 1, WAVE);
        if (lane == 0) excl = 1.0f;
        T[k] = carry * excl;
        carry *= __shfl(incl, WAVE - 1, WAVE);
        const float w = alpha[k] * T[k];
        s_acc += w;
        s_depth += w * z[k];
    }
    const float g0 = g_rgb ? g_rgb[(size_t)ray * 3 + 0] : 0.f, g1 = g_rgb ? g_rgb[(size_t)ray * 3 + 1] : 0.f,
                g2 = g_rgb ? g_rgb[(size_t)ray * 3 + 2] : 0.f;
    float ga = g_acc ? g_acc[ray] : 0.f, gd = g_depth ? g_depth[ray] : 0.f;
    if (g_disp) {
        const float gq = g_disp[ray];
        if (gq != 0.0f) {
            s_acc = wave_sum(s_acc);
            s_depth = wave_sum(s_depth);
            const float q = s_depth / s_acc;
            if (!(q <= 1e-10f)) {            // above the clamp, or NaN (0/0), which propagates like in autograd
                const float disp = 1.0f / q;
                gd += gq * (-disp * disp / s_acc);
                ga += gq * (disp * disp * s_depth / (s_acc * s_acc));
            }
        }
    }
    const float bgterm = white_bkgd ? (g0 + g1 + g2) : 0.0f;
    float later = 0.0f;                       // sum of q_k w_k over all later chunks
    float* outr = d_raw + (size_t)ray * S * RAW_CH;
#pragma unroll
    for (int k = NCH - 1; k >= 0; --k) {
        const int i = 64 * k + lane;
        const bool valid = i < S;
        const float w = alpha[k] * T[k];
        float q = g0 * c0[k] + g1 * c1[k] + g2 * c2[k] - bgterm + ga + gd * z[k];
        if (g_weights && valid) q += g_weights[(size_t)ray * S + i];      // a loss on the weights themselves
        const float term = valid ? q * w : 0.0f;
        float incl = term;                    // inclusive suffix sum within the chunk
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const float dn = __shfl_down(incl, off, WAVE);
            if (lane + off < WAVE) incl += dn;
        }
        float excl = __shfl_down(incl, 1, WAVE);
        if (lane == WAVE - 1) excl = 0.0f;
        const float suffix = excl + later;
        later += __shfl(incl, 0, WAVE);
        if (valid) {
            const float dalpha = q * T[k] - suffix / one_m[k];
            const float dsig = gate[k] ? dalpha * dist[k] * e_[k] : 0.0f;
            const float r0 = w * g0 * c0[k] * (1.0f - c0[k]);
            const float r1 = w * g1 * c1[k] * (1.0f - c1[k]);
            const float r2 = w * g2 * c2[k] * (1.0f - c2[k]);
            if (RAW_CH == 4) {
                reinterpret_cast<float4*>(outr)[i] = make_float4(r0, r1, r2, dsig);
            } else {
                float* o = outr + (size_t)i * RAW_CH;
                o[0] = r0; o[1] = r1; o[2] = r2; o[3] = dsig; o[4] = 0.0f;
            }
        }
    }
}

// One sample per ray: the reference builds its interval list from the S - 1 differences of the depths and appends the
// 1e10 tail with the shape of an (empty) slice of them (run_nerf.py:283-284), so for S = 1 the list is empty, every
// per-sample tensor broadcasts to [R, 0] and the sums run over nothing: acc = depth = 0, disp = 1 / max(1e-10, 0 / 0) =
// NaN, rgb = the background, and `weights` has no columns.
__global__ void raw2outputs_single_sample_kernel(int R, int white, float* rgb, float* disp, float* acc, float* depth) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (rgb) { const float c = white ? 1.0f : 0.0f; rgb[3 * (size_t)r] = c; rgb[3 * (size_t)r + 1] = c; rgb[3 * (size_t)r + 2] = c; }
    if (disp) disp[r] = __builtin_nanf("");
    if (acc) acc[r] = 0.0f;
    if (depth) depth[r] = 0.0f;
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_raw2outputs(const float* raw, int raw_ch, const float* z_vals, const float* d_ptr,
                                int d_stride, const float* noise, int R, int S, int white_bkgd,
                                float* rgb, float* disp, float* acc, float* depth, float* weights,
                                void* stream) {
    NERF_REQUIRE(R == 0 || (raw && z_vals && d_ptr), "nerf_raw2outputs: null input");
    NERF_REQUIRE(raw_ch == 4 || raw_ch == 5, "nerf_raw2outputs: raw_ch must be 4 or 5, got %d", raw_ch);
    NERF_REQUIRE(R >= 0 && S >= 1 && d_stride >= 3, "nerf_raw2outputs: bad shape R=%d S=%d d_stride=%d", R, S, d_stride);
    if (raw_ch == 4) NERF_REQUIRE(((uintptr_t)raw & 15) == 0, "nerf_raw2outputs: raw must be 16-byte aligned");
    if (R == 0) return NERF_OK;
    if (S == 1) {
        hipLaunchKernelGGL(raw2outputs_single_sample_kernel, dim3((R + 255) / 256), dim3(256), 0, as_stream(stream), R,
                           white_bkgd, rgb, disp, acc, depth);
        NERF_LAUNCH_CHECK();
        return NERF_OK;
    }
    const int nch = (S + 63) / 64;
    if (nch == 1) {      // a 16-lane row per ray
        const bool vec = raw_ch == 4 && S % 4 == 0 && ((uintptr_t)z_vals & 15) == 0 && (!noise || ((uintptr_t)noise & 15) == 0) &&
                         (!weights || ((uintptr_t)weights & 15) == 0);
        dim3 rgrid((R + 15) / 16), rblock(256);
#define NERF_R2O_ROW(CH, V)                                                                                              \
    hipLaunchKernelGGL((raw2outputs_row_kernel<CH, V>), rgrid, rblock, 0, as_stream(stream), raw, z_vals, d_ptr, d_stride, \
                       noise, R, S, white_bkgd, rgb, disp, acc, depth, weights)
        if (vec) NERF_R2O_ROW(4, true);
        else if (raw_ch == 4) NERF_R2O_ROW(4, false);
        else NERF_R2O_ROW(5, false);
#undef NERF_R2O_ROW
        NERF_LAUNCH_CHECK();
        return NERF_OK;
    }
    const int rpw = nch == 2 ? 2 : 1;
    const int rays_per_block = 4 * rpw;
    dim3 grid((R + rays_per_block - 1) / rays_per_block), block(4 * WAVE);
#define NERF_R2O_FWD(CH, n, q)                                                                                       \
    hipLaunchKernelGGL((raw2outputs_kernel<CH, n, q>), grid, block, 0, as_stream(stream), raw, z_vals, d_ptr, d_stride, \
                       noise, R, S, white_bkgd, rgb, disp, acc, depth, weights)
#define NERF_R2O_SWITCH(CH)                                                                  \
    switch (nch) {                                                                           \
        case 1: NERF_R2O_FWD(CH, 1, 4); break;                                               \
        case 2: NERF_R2O_FWD(CH, 2, 2); break;                                               \
        case 3: NERF_R2O_FWD(CH, 3, 1); break;                                               \
        default: NERF_R2O_FWD(CH, 4, 1); break;                                              \
    }
    if (raw_ch == 4) { NERF_R2O_SWITCH(4) } else { NERF_R2O_SWITCH(5) }
#undef NERF_R2O_SWITCH
#undef NERF_R2O_FWD
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

template <int RAW_CH>
static void launch_r2o_bwd(int nch, dim3 grid, dim3 block, hipStream_t st, const float* raw, const float* z,
                           const float* d_ptr, int d_stride, const float* noise, int R, int S, int white,
                           const float* g_rgb, const float* g_disp, const float* g_acc, const float* g_depth,
                           const float* g_weights, float* d_raw) {
#define NERF_R2O_CASE(n)                                                                                           \
    case n: hipLaunchKernelGGL((raw2outputs_bwd_kernel<RAW_CH, n>), grid, block, 0, st, raw, z, d_ptr, d_stride,   \
                               noise, R, S, white, g_rgb, g_disp, g_acc, g_depth, g_weights, d_raw); break;
    switch (nch) { NERF_R2O_CASE(1) NERF_R2O_CASE(2) NERF_R2O_CASE(3) default: NERF_R2O_CASE(4) }
#undef NERF_R2O_CASE
}

extern "C" int nerf_raw2outputs_backward(const float* raw, int raw_ch, const float* z_vals, const float* d_ptr,
                                         int d_stride, const float* noise, int R, int S, int white_bkgd,
                                         const float* g_rgb, const float* g_disp, const float* g_acc,
                                         const float* g_depth, const float* g_weights, float* d_raw, void* stream) {
    NERF_REQUIRE(R == 0 || (raw && z_vals && d_ptr && d_raw), "nerf_raw2outputs_backward: null pointer");
    NERF_REQUIRE(raw_ch == 4 || raw_ch == 5, "nerf_raw2outputs_backward: raw_ch must be 4 or 5, got %d", raw_ch);
    NERF_REQUIRE(R >= 0 && S >= 1 && d_stride >= 3, "nerf_raw2outputs_backward: bad shape R=%d S=%d d_stride=%d", R, S, d_stride);
    if (S > 256) {
        set_error("nerf_raw2outputs_backward: at most 256 samples per ray, got %d", S);
        return NERF_E_UNSUPPORTED;
    }
    if (raw_ch == 4) NERF_REQUIRE(((uintptr_t)d_raw & 15) == 0, "nerf_raw2outputs_backward: d_raw must be 16-byte aligned");
    if (R == 0) return NERF_OK;
    if (S == 1) {      // nothing is composited from a single sample (see raw2outputs_single_sample_kernel): no gradient
        NERF_HIP_CHECK(hipMemsetAsync(d_raw, 0, (size_t)R * raw_ch * sizeof(float), as_stream(stream)));
        return NERF_OK;
    }
    dim3 grid((R + 3) / 4), block(4 * WAVE);
    const int nch = (S + 63) / 64;
    if (raw_ch == 4) launch_r2o_bwd<4>(nch, grid, block, as_stream(stream), raw, z_vals, d_ptr, d_stride, noise, R, S,
                                       white_bkgd, g_rgb, g_disp, g_acc, g_depth, g_weights, d_raw);
    else launch_r2o_bwd<5>(nch, grid, block, as_stream(stream), raw, z_vals, d_ptr, d_stride, noise, R, S, white_bkgd,
                           g_rgb, g_disp, g_acc, g_depth, g_weights, d_raw);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}
