// Hierarchical (inverse-CDF) sampling: sample_pdf (run_nerf_helpers.py:196-239) with the
// mid-point, sort(cat) and z_std code around it in render_rays (run_nerf.py:397-401, 418).
//
// A wavefront takes 8 rays at a time, rows staged in LDS:
//   weights -> +1e-5 -> sum -> pdf -> cdf        (serial along the ray: one ray per lane)
//   u -> searchsorted(right) by binary search over the LDS cdf, one sample per lane per step
//   samples -> LDS -> merge with the coarse depths by rank -> coalesced store
//
// Bit-exactness: indices are the reference's contract, so the float32 arithmetic reproduces
// ATen's CPU kernels operation by operation:
//   torch.sum(w, -1)   : vectorised row reduction, 8 lanes x 4 interleaved accumulators with
//                        a 4-level cascade (row_sum_aten below), scalar tail, lanes in order;
//   torch.cumsum       : sequential, float64 accumulator, each prefix rounded to float32;
//   divisions          : IEEE (correctly rounded), no fused multiply-add anywhere.
#include "common.h"

namespace nerf {

// ATen row_sum over the n elements x[0], x[stride], ... (SumKernel.cpp: row_sum +
// multi_row_sum with 4 interleaved accumulators and 4 cascade levels).
__device__ float row_sum_aten(const float* x, int n, int stride) {
    const int size = n >> 2;
    const int ceil_log2 = size <= 1 ? 0 : 32 - __clz(size - 1);
    const int level_power = max(4, ceil_log2 / 4);
    const int level_step = 1 << level_power;
    const int level_mask = level_step - 1;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    float a2[4] = {0.f, 0.f, 0.f, 0.f}, a3[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    while (i + level_step <= size) {
        for (int j = 0; j < level_step; ++j, ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a0[k] = fadd(a0[k], x[(i * 4 + k) * stride]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1[k] = fadd(a1[k], a0[k]); a0[k] = 0.f; }
        if ((i & (level_mask << level_power)) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a2[k] = fadd(a2[k], a1[k]); a1[k] = 0.f; }
            if ((i & (level_mask << (2 * level_power))) == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { a3[k] = fadd(a3[k], a2[k]); a2[k] = 0.f; }
            }
        }
    }
    for (; i < size; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a0[k] = fadd(a0[k], x[(i * 4 + k) * stride]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a0[k] = fadd(fadd(fadd(a0[k], a1[k]), a2[k]), a3[k]);
    float p0 = a0[0];
    for (i = size * 4; i < n; ++i) p0 = fadd(p0, x[i * stride]);
    p0 = fadd(p0, a0[1]);
    p0 = fadd(p0, a0[2]);
    p0 = fadd(p0, a0[3]);
    return p0;
}

struct SamplePdfArgs {
    const float* bins; int bins_stride; int bins_are_z; int nb;
    const float* weights; int w_stride;
    const float* u; int n_u; int u_per_ray; int R;
    const float* z_vals; int n_z;
    float* z_samples; int32_t* inds; float* z_sorted; float* z_std;
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// number of entries of the sorted array a[0..n) that are <= v (LE) or < v (!LE)
template <bool LE>
__device__ __forceinline__ int count_sorted(const float* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float c = a[mid];
        const bool go = LE ? (c <= v) : (c < v);
        if (go) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int PDF_WAVES = 1;      // wavefronts per workgroup
constexpr int PDF_GROUP = 8;      // rays a wavefront takes through the serial phases together

// ATen's sum of one row w[0..n) (unit stride): the vectorised path for n >= 8 is eight strided
// row sums combined after the scalar tail, lanes in order (SumKernel.cpp).
__device__ float row_sum_full(const float* w, int n) {
    if (n < 8) return row_sum_aten(w, n, 1);
    const int nvec = n >> 3;
    float acc = 0.f;
    for (int k = nvec * 8; k < n; ++k) acc = fadd(acc, w[k]);
#pragma nounroll
    for (int l = 0; l < 8; ++l) acc = fadd(acc, row_sum_aten(w + l, nvec, 8));   // one copy of the code, low register use
    return acc;
}

// A wavefront owns PDF_GROUP consecutive rays.  The parts of the algorithm that are serial along a ray
// (the ATen-ordered sum and the float64 running sum of the cdf) run with ONE RAY PER LANE on rows staged in
// LDS with an odd row stride (conflict-free); the parts that are parallel along a ray (searchsorted,
// interpolation, merge) then run ray by ray with one sample per lane.
__global__ __launch_bounds__(PDF_WAVES * 64) void sample_pdf_kernel(SamplePdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nb = a.nb, nw = a.nb - 1, n_u = a.n_u, n_z = a.z_sorted ? a.n_z : 0;
    const int sw = nw | 1, sc = nb | 1;                       // odd row strides
    const int per_wave = PDF_GROUP * (sw + sc) + nb + 2 * n_u + n_z;
    float* w_g = smem + (size_t)wv * per_wave;                // [PDF_GROUP][sw] weights + 1e-5
    float* cdf_g = w_g + PDF_GROUP * sw;                      // [PDF_GROUP][sc]
    float* bins_s = cdf_g + PDF_GROUP * sc;                   // the ray in flight: bins, samples, coarse depths
    float* zs_s = bins_s + nb;
    float* zc_s = zs_s + n_u;
    int* ind_s = reinterpret_cast<int*>(zc_s + n_z);          // searchsorted results of the ray in flight
    // coarse depths and bins come from the same array: the rank of a sample among the coarse depths follows from
    // its bin (checked, with the binary search as fallback)
    const bool z_is_bins = a.bins_are_z && a.z_sorted && a.z_vals == a.bins && a.n_z == nb + 1;
    const long long r0 = ((long long)blockIdx.x * PDF_WAVES + wv) * PDF_GROUP;
    const long long last = a.R - 1;

    // ---- stage the weight rows (coalesced), + 1e-5
    for (int g = 0; g < PDF_GROUP; ++g) {
        const long long ray = r0 + g < a.R ? r0 + g : last;
        const float* wr = a.weights + (size_t)ray * a.w_stride;
        for (int i = lane; i < nw; i += WAVE) w_g[g * sw + i] = fadd(wr[i], 1e-5f);
    }
    __syncthreads();
    // ---- one ray per lane: total in ATen order
    float total = 1.0f;
    if (lane < PDF_GROUP) total = row_sum_full(w_g + lane * sw, nw);
    // pdf = w / total, one element per lane (IEEE division is long: keep it parallel along the ray)
    for (int g = 0; g < PDF_GROUP; ++g) {
        const float tg = __shfl(total, g, WAVE);
        for (int i = lane; i < nw; i += WAVE) w_g[g * sw + i] = fdiv(w_g[g * sw + i], tg);
    }
    __syncthreads();
    // ---- one ray per lane: float64 running sum, each prefix rounded to float32
    if (lane < PDF_GROUP) {
        const float* w = w_g + lane * sw;
        float* cdf = cdf_g + lane * sc;
        double c = 0.0;
        cdf[0] = 0.f;
        for (int i = 0; i < nw; ++i) {
            c = __dadd_rn(c, (double)w[i]);
            cdf[i + 1] = (float)c;
        }
    }
    __syncthreads();

    // ---- ray by ray: searchsorted, interpolation, z_std, merge
    for (int g = 0; g < PDF_GROUP; ++g) {
        const bool live = r0 + g < a.R;
        const long long ray = live ? r0 + g : last;
        const float* cdf_s = cdf_g + g * sc;
        const float* br = a.bins + (size_t)ray * a.bins_stride;
        for (int i = lane; i < nb; i += WAVE)
            bins_s[i] = a.bins_are_z ? fmul(0.5f, fadd(br[i + 1], br[i])) : br[i];
        if (n_z) {
            const float* zr = a.z_vals + (size_t)ray * a.n_z;
            for (int i = lane; i < n_z; i += WAVE) zc_s[i] = zr[i];
        }
        __syncthreads();

        const float* ur = a.u + (a.u_per_ray ? (size_t)ray * n_u : 0);
        float lsum = 0.f;
        for (int j = lane; j < n_u; j += WAVE) {
            const float u = ur[j];
            const int ind = count_sorted<true>(cdf_s, nb, u);
            const int below = max(ind - 1, 0), above = min(ind, nb - 1);
            const float cb = cdf_s[below], ca = cdf_s[above];
            float denom = fsub(ca, cb);
            if (denom < 1e-5f) denom = 1.0f;
            const float t = fdiv(fsub(u, cb), denom);
            const float bb = bins_s[below], ba = bins_s[above];
            const float s = fadd(bb, fmul(t, fsub(ba, bb)));
            zs_s[j] = s;
            ind_s[j] = ind;
            lsum += s;
            if (live) {
                if (a.z_samples) a.z_samples[(size_t)ray * n_u + j] = s;
                if (a.inds) a.inds[(size_t)ray * n_u + j] = ind;
            }
        }
        __syncthreads();

        if (a.z_std) {
            // population standard deviation of the new samples (torch.std(unbiased=False))
            const float mean = wave_sum_f(lsum) / (float)n_u;
            float lv = 0.f;
            for (int j = lane; j < n_u; j += WAVE) { const float d = zs_s[j] - mean; lv += d * d; }
            const float var = wave_sum_f(lv) / (float)n_u;
            if (live && lane == 0) a.z_std[ray] = sqrtf(var);
        }

        if (a.z_sorted) {
            // sort(cat([z_vals, z_samples])) by rank.  z_vals are non-decreasing by construction
            // (run_nerf.py:363-385).  Ties: coarse depths first, then samples by index.
            float* out = a.z_sorted + (size_t)ray * (n_z + n_u);
            bool sorted_l = true;
            for (int j = lane; j + 1 < n_u; j += WAVE) sorted_l = sorted_l && (zs_s[j] <= zs_s[j + 1]);
            const bool sorted = __all(sorted_l);
            if (sorted) {
                for (int i = lane; i < n_z; i += WAVE) {
                    const float v = zc_s[i];
                    const int r = i + count_sorted<false>(zs_s, n_u, v);
                    if (live) out[r] = v;
                }
                for (int j = lane; j < n_u; j += WAVE) {
                    const float v = zs_s[j];
                    int rc = -1;
                    if (z_is_bins) {
                        // v lies between the mid-points around coarse depth number ind: ind or ind + 1 depths are <= v
                        const int ind = ind_s[j];
                        const int c = ind + ((ind < n_z && zc_s[ind] <= v) ? 1 : 0);
                        const bool ok = (c == 0 || zc_s[c - 1] <= v) && (c == n_z || zc_s[c] > v);
                        if (ok) rc = c;
                    }
                    if (rc < 0) rc = count_sorted<true>(zc_s, n_z, v);
                    if (live) out[j + rc] = v;
                }
            } else {
                for (int i = lane; i < n_z; i += WAVE) {
                    const float v = zc_s[i];
                    int r = i;
                    for (int k = 0; k < n_u; ++k) r += (zs_s[k] < v) ? 1 : 0;
                    if (live) out[r] = v;
                }
                for (int j = lane; j < n_u; j += WAVE) {
                    const float v = zs_s[j];
                    int r = count_sorted<true>(zc_s, n_z, v);
                    for (int k = 0; k < n_u; ++k) {
                        const float o = zs_s[k];
                        r += (o < v || (o == v && k < j)) ? 1 : 0;
                    }
                    if (live) out[r] = v;
                }
            }
        }
        __syncthreads();      // the per-ray buffers are reused by the next ray
    }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_sample_pdf(const float* bins, int bins_stride, int bins_are_z, int nb,
                               const float* weights, int w_stride, const float* u, int n_u,
                               int u_per_ray, int R, const float* z_vals, int n_z, float* z_samples,
                               int32_t* inds, float* z_sorted, float* z_std, void* stream) {
    NERF_REQUIRE(u && (R == 0 || (bins && weights)), "nerf_sample_pdf: null input");
    NERF_REQUIRE(nb >= 2 && n_u >= 1 && R >= 0, "nerf_sample_pdf: bad sizes nb=%d n_u=%d R=%d", nb, n_u, R);
    NERF_REQUIRE(bins_stride >= (bins_are_z ? nb + 1 : nb) && w_stride >= nb - 1,
                 "nerf_sample_pdf: row strides shorter than the rows");
    if (z_sorted) NERF_REQUIRE(z_vals && n_z >= 1, "nerf_sample_pdf: z_sorted needs z_vals");
    if (R == 0) return NERF_OK;
    SamplePdfArgs a{bins, bins_stride, bins_are_z, nb, weights, w_stride, u, n_u, u_per_ray, R,
                    z_vals, n_z, z_samples, inds, z_sorted, z_std};
    const size_t per_wave = (size_t)PDF_GROUP * (((nb - 1) | 1) + (nb | 1)) + nb + 2 * (size_t)n_u + (z_sorted ? n_z : 0);
    const size_t shmem = per_wave * PDF_WAVES * sizeof(float);
    NERF_REQUIRE(shmem <= 64 * 1024, "nerf_sample_pdf: rows too long for the LDS staging (%zu bytes)", shmem);
    const int rays_per_block = PDF_WAVES * PDF_GROUP;
    dim3 grid((R + rays_per_block - 1) / rays_per_block), block(PDF_WAVES * WAVE);
    hipLaunchKernelGGL(sample_pdf_kernel, grid, block, shmem, as_stream(stream), a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}
