// Ray generation / packing and stratified depth kernels.
//   nerf_make_rays : get_rays + view-direction normalisation + ndc_rays + [N,8|11] packing
//                    (run_nerf_helpers.py:153-192, run_nerf.py:95-123)
//   nerf_z_vals    : z_vals = near*(1-t) + far*t | lindisp, stratified jitter (run_nerf.py:363-385)
// Both are HBM-bound streaming kernels: 32-44 B written per ray, 4 B per sample.
#include "common.h"

namespace nerf {

struct MakeRaysArgs {
    int n, H, W;
    const float* o_in;
    const float* d_in;
    const float* c2w;
    const float* c2w_view;
    float fx, fy, cx, cy;
    float near, far;
    const float* near_arr;
    const float* far_arr;
    int use_viewdirs, ndc;
    float ndc_c0, ndc_c1, ndc_near, ndc_2near;
    int W_img;
    float* out;
};

__device__ __forceinline__ void pixel_dir(const float* __restrict__ c2w, float i, float j, float fx,
                                          float fy, float cx, float cy, float d[3]) {
    // dirs = [(i-cx)/fx, -(j-cy)/fy, -1];  rays_d[k] = ((dirs0*R[k][0] + dirs1*R[k][1]) + dirs2*R[k][2])
    const float dx = fdiv(fsub(i, cx), fx);
    const float dy = fdiv(-fsub(j, cy), fy);
    const float dz = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p0 = fmul(dx, c2w[k * 4 + 0]);
        const float p1 = fmul(dy, c2w[k * 4 + 1]);
        const float p2 = fmul(dz, c2w[k * 4 + 2]);
        d[k] = fadd(fadd(p0, p1), p2);
    }
}

__global__ __launch_bounds__(256) void make_rays_kernel(MakeRaysArgs a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.n) return;
    float o[3], d[3], v[3];
    if (a.o_in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.o_in[n * 3 + k];
            d[k] = a.d_in[n * 3 + k];
            v[k] = d[k];
        }
    } else {
        const int h = n / a.W, w = n - h * a.W;
        pixel_dir(a.c2w, (float)w, (float)h, a.fx, a.fy, a.cx, a.cy, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = a.c2w[k * 4 + 3];
        if (a.c2w_view) {
            pixel_dir(a.c2w_view, (float)w, (float)h, a.fx, a.fy, a.cx, a.cy, v);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = d[k];
        }
    }
    if (a.use_viewdirs) {
        // viewdirs / torch.norm(viewdirs): taken from the world-space direction, before NDC
        const float nrm = sqrtf(fadd(fadd(fmul(v[0], v[0]), fmul(v[1], v[1])), fmul(v[2], v[2])));
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = fdiv(v[k], nrm);
    }
    if (a.ndc) {
        const float t = fdiv(-fadd(a.ndc_near, o[2]), d[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = fadd(o[k], fmul(t, d[k]));
        const float o0 = fdiv(fmul(a.ndc_c0, o[0]), o[2]);
        const float o1 = fdiv(fmul(a.ndc_c1, o[1]), o[2]);
        const float o2 = fadd(1.0f, fdiv(a.ndc_2near, o[2]));
        const float d0 = fmul(a.ndc_c0, fsub(fdiv(d[0], d[2]), fdiv(o[0], o[2])));
        const float d1 = fmul(a.ndc_c1, fsub(fdiv(d[1], d[2]), fdiv(o[1], o[2])));
        const float d2 = fdiv(-a.ndc_2near, o[2]);
        o[0] = o0; o[1] = o1; o[2] = o2;
        d[0] = d0; d[1] = d1; d[2] = d2;
    }
    const int cols = a.use_viewdirs ? 11 : 8;
    float* r = a.out + (size_t)n * cols;
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = a.near_arr ? a.near_arr[n] : a.near;
    r[7] = a.far_arr ? a.far_arr[n] : a.far;
    if (a.use_viewdirs) { r[8] = v[0]; r[9] = v[1]; r[10] = v[2]; }
}

// The pixel grids of several views in one launch (blockIdx.y = view): views[v] = c2w (3x4, row-major, 12 floats) followed
// by fx, fy, cx, cy.  Same arithmetic as make_rays_kernel without the NDC warp; view v fills rows [v H W, (v + 1) H W).
__global__ __launch_bounds__(256) void make_rays_views_kernel(int H, int W, const float* __restrict__ views, float near,
                                                              float far, int use_viewdirs, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= H * W) return;
    const float* vw = views + (size_t)blockIdx.y * 16;
    const int h = n / W, w = n - h * W;
    float d[3], v[3];
    pixel_dir(vw, (float)w, (float)h, vw[12], vw[13], vw[14], vw[15], d);
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = d[k];
    if (use_viewdirs) {
        const float nrm = sqrtf(fadd(fadd(fmul(v[0], v[0]), fmul(v[1], v[1])), fmul(v[2], v[2])));
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = fdiv(v[k], nrm);
    }
    const int cols = use_viewdirs ? 11 : 8;
    float* r = out + ((size_t)blockIdx.y * H * W + n) * cols;
    r[0] = vw[3]; r[1] = vw[7]; r[2] = vw[11];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = near;
    r[7] = far;
    if (use_viewdirs) { r[8] = v[0]; r[9] = v[1]; r[10] = v[2]; }
}

__global__ __launch_bounds__(256) void z_vals_kernel(const float* __restrict__ rays, int ray_cols, int R,
                                                     int S, const float* __restrict__ t_vals,
                                                     const float* __restrict__ t_rand, int lindisp,
                                                     float* __restrict__ z_out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * S) return;
    const int r = (int)(idx / S), s = (int)(idx - (size_t)r * S);
    const float near = rays[(size_t)r * ray_cols + 6], far = rays[(size_t)r * ray_cols + 7];
    auto z_at = [&](int k) -> float {
        const float t = t_vals[k];
        const float omt = fsub(1.0f, t);
        if (!lindisp) return fadd(fmul(near, omt), fmul(far, t));
        return fdiv(1.0f, fadd(fmul(fdiv(1.0f, near), omt), fmul(fdiv(1.0f, far), t)));
    };
    float z = z_at(s);
    if (t_rand) {
        // mids = .5*(z[1:]+z[:-1]); upper = [mids, z_last]; lower = [z_first, mids]
        const float up = (s + 1 < S) ? fmul(0.5f, fadd(z_at(s + 1), z)) : z;
        const float lo = (s > 0) ? fmul(0.5f, fadd(z, z_at(s - 1))) : z;
        z = fadd(lo, fmul(fsub(up, lo), t_rand[idx]));
    }
    z_out[idx] = z;
}

// Embedder.embed (run_nerf_helpers.py:24-45) as a standalone op: one thread per output element.
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ x, long long M, int L,
                                                    int include_input, float* __restrict__ out) {
    const int C = (include_input ? 3 : 0) + 6 * L;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * C) return;
    const long long m = idx / C;
    int col = (int)(idx - m * C);
    float v;
    if (include_input && col < 3) {
        v = x[m * 3 + col];
    } else {
        if (include_input) col -= 3;
        const int f = col / 6, rem = col - f * 6, c3 = rem % 3;
        const float arg = fmul(x[m * 3 + c3], (float)(1 << f));
        v = rem < 3 ? sinf(arg) : cosf(arg);
    }
    out[idx] = v;
}

// NaN / Inf census of up to CENSUS_MAX tensors in one launch (the check the reference makes key by key with
// torch.isnan(...).any() / torch.isinf(...).any(), run_nerf.py:420-422, nerf_utils.py:336-340): streaming read,
// one atomic per workgroup and tensor that has any.
constexpr int CENSUS_MAX = 16;
struct CensusArgs {
    const float* p[CENSUS_MAX];
    long long end[CENSUS_MAX];     // running end (elements) of tensor t in the concatenated index space
    int n;
    int* counts;
};
__global__ __launch_bounds__(256) void census_kernel(CensusArgs a) {
    __shared__ int cnt[CENSUS_MAX];
    if (threadIdx.x < CENSUS_MAX) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long long)gridDim.x * blockDim.x;
    auto bad = [](float v) { return (int)((__float_as_uint(v) & 0x7f800000u) == 0x7f800000u); };
    for (int t = 0; t < a.n; ++t) {                          // uniform loop: every workgroup sweeps every tensor
        const float* p = a.p[t];
        const long long n = a.end[t] - (t ? a.end[t - 1] : 0);
        // 16-byte loads from the first aligned element on; the few elements in front and behind one by one
        const long long head = ((16 - ((uintptr_t)p & 15)) & 15) / 4 < n ? ((16 - ((uintptr_t)p & 15)) & 15) / 4 : n;
        const long long n4 = (n - head) / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p + head);
        int local = 0;
        for (long long i = tid; i < n4; i += 4 * nthr) {     // four independent loads in flight per lane
            const long long i1 = i + nthr, i2 = i + 2 * nthr, i3 = i + 3 * nthr;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 q0 = p4[i], q1 = i1 < n4 ? p4[i1] : z, q2 = i2 < n4 ? p4[i2] : z, q3 = i3 < n4 ? p4[i3] : z;
            local += bad(q0.x) + bad(q0.y) + bad(q0.z) + bad(q0.w) + bad(q1.x) + bad(q1.y) + bad(q1.z) + bad(q1.w) +
                     bad(q2.x) + bad(q2.y) + bad(q2.z) + bad(q2.w) + bad(q3.x) + bad(q3.y) + bad(q3.z) + bad(q3.w);
        }
        if (tid < head) local += bad(p[tid]);
        const long long tail0 = head + 4 * n4;
        if (tid < n - tail0) local += bad(p[tail0 + tid]);
        if (local) atomicAdd(&cnt[t], local);
    }
    __syncthreads();
    if ((int)threadIdx.x < a.n && cnt[threadIdx.x]) atomicAdd(&a.counts[threadIdx.x], cnt[threadIdx.x]);
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_count_nonfinite(int n_tensors, const float* const* ptrs_host, const long long* numel_host,
                                    int* counts_dev, void* stream) {
    NERF_REQUIRE(n_tensors >= 1 && n_tensors <= CENSUS_MAX, "nerf_count_nonfinite: 1..%d tensors per call, got %d", CENSUS_MAX, n_tensors);
    NERF_REQUIRE(ptrs_host && numel_host && counts_dev, "nerf_count_nonfinite: null pointer");
    CensusArgs a;
    long long end = 0;
    for (int t = 0; t < n_tensors; ++t) {
        NERF_REQUIRE(numel_host[t] >= 0 && (numel_host[t] == 0 || ptrs_host[t]), "nerf_count_nonfinite: tensor %d is null", t);
        a.p[t] = ptrs_host[t];
        end += numel_host[t];
        a.end[t] = end;
    }
    a.n = n_tensors;
    a.counts = counts_dev;
    NERF_HIP_CHECK(hipMemsetAsync(counts_dev, 0, sizeof(int) * n_tensors, as_stream(stream)));
    if (end == 0) return NERF_OK;
    long long blocks = (end + 256 * 16 - 1) / (256 * 16);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(census_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

extern "C" int nerf_embed(const float* x, long long M, int multires, int include_input, float* out, void* stream) {
    NERF_REQUIRE(M == 0 || (x && out), "nerf_embed: null pointer");
    NERF_REQUIRE(M >= 0 && multires >= 0 && multires <= 24, "nerf_embed: bad sizes M=%lld L=%d", M, multires);
    const long long total = M * ((include_input ? 3 : 0) + 6 * multires);
    if (total == 0) return NERF_OK;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, M,
                       multires, include_input, out);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

extern "C" int nerf_make_rays(int n_rays, int H, int W, const float* o_in, const float* d_in,
                              const float* c2w, const float* c2w_view, float fx, float fy, float cx,
                              float cy, float near, float far, const float* near_arr,
                              const float* far_arr, int use_viewdirs, int ndc, float ndc_focal,
                              float ndc_near, float* rays_out, void* stream) {
    NERF_REQUIRE(n_rays >= 0 && rays_out, "nerf_make_rays: bad n_rays or null output");
    NERF_REQUIRE((o_in == nullptr) == (d_in == nullptr), "nerf_make_rays: o_in and d_in go together");
    if (!o_in) {
        NERF_REQUIRE(c2w && H > 0 && W > 0 && (long long)H * W == n_rays,
                     "nerf_make_rays: pixel-grid mode needs c2w and H*W == n_rays (H=%d W=%d n=%d)", H, W, n_rays);
        NERF_REQUIRE(fx != 0.f && fy != 0.f, "nerf_make_rays: zero focal length");
    }
    if (ndc) NERF_REQUIRE(H > 0 && W > 0 && ndc_focal != 0.f, "nerf_make_rays: ndc needs H, W, focal");
    if (n_rays == 0) return NERF_OK;
    MakeRaysArgs a;
    a.n = n_rays; a.H = H; a.W = W; a.o_in = o_in; a.d_in = d_in; a.c2w = c2w; a.c2w_view = c2w_view;
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.near = near; a.far = far;
    a.near_arr = near_arr; a.far_arr = far_arr; a.use_viewdirs = use_viewdirs; a.ndc = ndc;
    // python-float scalars of ndc_rays are formed in double, then enter float32 tensor ops
    a.ndc_c0 = ndc ? (float)(-1.0 / ((double)W / (2.0 * (double)ndc_focal))) : 0.f;
    a.ndc_c1 = ndc ? (float)(-1.0 / ((double)H / (2.0 * (double)ndc_focal))) : 0.f;
    a.ndc_near = ndc_near;
    a.ndc_2near = (float)(2.0 * (double)ndc_near);
    a.W_img = W; a.out = rays_out;
    const int block = 256;
    hipLaunchKernelGGL(make_rays_kernel, dim3((n_rays + block - 1) / block), dim3(block), 0, as_stream(stream), a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

extern "C" int nerf_make_rays_views(int n_views, int H, int W, const float* views, float near, float far, int use_viewdirs,
                                    float* rays_out, void* stream) {
    NERF_REQUIRE(n_views >= 0 && H >= 1 && W >= 1 && (long long)H * W < (1ll << 31), "nerf_make_rays_views: bad shape %d x %d x %d",
                 n_views, H, W);
    NERF_REQUIRE(n_views <= 65535, "nerf_make_rays_views: at most 65535 views per call, got %d", n_views);
    NERF_REQUIRE(n_views == 0 || (views && rays_out), "nerf_make_rays_views: null pointer");
    if (n_views == 0) return NERF_OK;
    const int block = 256;
    hipLaunchKernelGGL(make_rays_views_kernel, dim3((H * W + block - 1) / block, n_views), dim3(block), 0, as_stream(stream),
                       H, W, views, near, far, use_viewdirs, rays_out);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

extern "C" int nerf_z_vals(const float* rays, int ray_cols, int R, int S, const float* t_vals,
                           const float* t_rand, int lindisp, float* z_out, void* stream) {
    NERF_REQUIRE(t_vals && (R == 0 || (rays && z_out)), "nerf_z_vals: null pointer");
    NERF_REQUIRE(R >= 0 && S >= 1 && ray_cols >= 8, "nerf_z_vals: bad shape R=%d S=%d cols=%d", R, S, ray_cols);
    if (R == 0) return NERF_OK;
    const size_t total = (size_t)R * S;
    const int block = 256;
    hipLaunchKernelGGL(z_vals_kernel, dim3((unsigned)((total + block - 1) / block)), dim3(block), 0,
                       as_stream(stream), rays, ray_cols, R, S, t_vals, t_rand, lindisp, z_out);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}
