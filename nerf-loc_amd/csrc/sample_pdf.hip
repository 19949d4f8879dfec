// Hierarchical (inverse-CDF) sampling: sample_pdf (run_nerf_helpers.py:196-239) with the
// mid-point, sort(cat) and z_std code around it in render_rays (run_nerf.py:397-401, 418).
//
// A wavefront takes 64 rays at a time, rows staged in LDS:
//   weights -> +1e-5 -> sum -> pdf -> cdf        (serial along the ray: one ray per lane, all 64 lanes busy)
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
#ifndef NERF_PDF_GROUP
#define NERF_PDF_GROUP 16
#endif
// Rays a wavefront takes through the serial phases together, one per lane.  More rays per group use more lanes there,
// but their LDS rows (252 B per ray) cost occupancy, and the ray-by-ray phases are latency-bound: measured 0.40 ms
// (16), 0.53 ms (32), 0.86 ms (64) for 640 000 rays.
constexpr int PDF_GROUP = NERF_PDF_GROUP;

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

// The serial phases for the wave's n_live rays: rows of [0, cdf 1..nw] in LDS (row stride sc floats).
__device__ __forceinline__ void build_cdf_rows(const SamplePdfArgs& a, float* cdf_g, int sc, long long r0, int n_live, int lane) {
    const int nw = a.nb - 1;
    // ---- stage the weight rows (coalesced, one row per trip), + 1e-5, at positions 1..nw of the row
    for (int g = 0; g < n_live; ++g) {
        const float* wr = a.weights + (size_t)(r0 + g) * a.w_stride;
        for (int i = lane; i < nw; i += WAVE) cdf_g[g * sc + 1 + i] = fadd(wr[i], 1e-5f);
    }
    __syncthreads();
    // ---- one ray per lane: total in ATen order, then pdf = w / total (IEEE) and its float64 running sum, each prefix
    //      rounded to float32, written over the weights
    if (lane < n_live) {
        float* row = cdf_g + lane * sc;
        const float total = row_sum_full(row + 1, nw);
        double c = 0.0;
        row[0] = 0.f;
        for (int i = 1; i <= nw; ++i) {
            c = __dadd_rn(c, (double)fdiv(row[i], total));
            row[i] = (float)c;
        }
    }
    __syncthreads();

}

// ------------------------------------------------------------------------------------------------------------------
// The shape render_rays uses (run_nerf.py:393-401: 64 coarse depths -> 63 mid-point bins, 128 variates shared by all
// rays, merged with the coarse depths): everything is a compile-time size, the searches are branch-free with
// immediate LDS offsets, the variates stay in registers, sums go through DPP.  Same arithmetic, operation by
// operation, as the general kernel below.
constexpr int FAST_NB = 63, FAST_NU = 128, FAST_NZ = 64, FAST_SC = 63;

template <int CTRL>
__device__ __forceinline__ float pdf_dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {          // total in every lane
    v += pdf_dpp<0x111>(0.f, v);      // row_shr 1, 2, 4, 8: lane 15 of each row holds the row's sum
    v += pdf_dpp<0x112>(0.f, v);
    v += pdf_dpp<0x114>(0.f, v);
    v += pdf_dpp<0x118>(0.f, v);
    auto at = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    return (at(15) + at(31)) + (at(47) + at(63));
}
// number of entries of the sorted array a[0 .. 2^LOG - 1) that are <= v (LE) or < v
template <int LOG, bool LE>
__device__ __forceinline__ int count_fixed(const float* a, float v) {
    int pos = 0;
#pragma unroll
    for (int step = 1 << (LOG - 1); step >= 1; step >>= 1) {
        const float c = a[pos + step - 1];
        pos += (LE ? (c <= v) : (c < v)) ? step : 0;
    }
    return pos;
}

__global__ __launch_bounds__(64) void sample_pdf_fast_kernel(SamplePdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    float* cdf_g = smem;                                      // [PDF_GROUP][63]
    float* bins_s = cdf_g + PDF_GROUP * FAST_SC;              // 64 (63 used)
    float* zs_s = bins_s + 64;                                // 128
    float* zc_s = zs_s + FAST_NU;                             // 64
    const long long r0 = (long long)blockIdx.x * PDF_GROUP;
    const int n_live = (int)(a.R - r0 < PDF_GROUP ? a.R - r0 : PDF_GROUP);
    build_cdf_rows(a, cdf_g, FAST_SC, r0, n_live, lane);
    const float u0 = a.u[lane], u1 = a.u[lane + 64];

    for (int g = 0; g < n_live; ++g) {
        const long long ray = r0 + g;
        const float* cdf_s = cdf_g + g * FAST_SC;
        const float zc = a.bins[(size_t)ray * a.bins_stride + lane];              // the coarse depths ARE the bin source
        zc_s[lane] = zc;
        __syncthreads();
        const float zn = zc_s[lane < 63 ? lane + 1 : 63];
        bins_s[lane] = fmul(0.5f, fadd(zn, zc));                                  // lane 63: unused
        __syncthreads();

        float sv[2];
        int iv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float u = q ? u1 : u0;
            const int ind = count_fixed<6, true>(cdf_s, u);                       // 63 entries
            const int below = max(ind - 1, 0), above = min(ind, FAST_NB - 1);
            const float cb = cdf_s[below], ca = cdf_s[above];
            float denom = fsub(ca, cb);
            if (denom < 1e-5f) denom = 1.0f;
            const float t = fdiv(fsub(u, cb), denom);
            const float bb = bins_s[below], ba = bins_s[above];
            sv[q] = fadd(bb, fmul(t, fsub(ba, bb)));
            iv[q] = ind;
            zs_s[lane + 64 * q] = sv[q];
            const size_t o = (size_t)ray * FAST_NU + lane + 64 * q;
            if (a.z_samples) a.z_samples[o] = sv[q];
            if (a.inds) a.inds[o] = ind;
        }
        __syncthreads();

        if (a.z_std) {
            const float mean = wave_sum_dpp(sv[0] + sv[1]) / (float)FAST_NU;
            const float d0 = sv[0] - mean, d1 = sv[1] - mean;
            const float var = wave_sum_dpp(d0 * d0 + d1 * d1) / (float)FAST_NU;
            if (lane == 0) a.z_std[ray] = sqrtf(var);
        }

        // sort(cat([z_vals, z_samples])) by rank; ties: coarse depths first, then samples by index
        float* out = a.z_sorted + (size_t)ray * (FAST_NZ + FAST_NU);
        const float nx0 = zs_s[lane + 1], nx1 = zs_s[lane < 63 ? lane + 65 : 127];
        const bool sorted = __all(sv[0] <= nx0 && sv[1] <= nx1);
        if (sorted) {
            // 128 sorted samples: all of them below zc, or the count among the first 127
            out[lane + ((zs_s[FAST_NU - 1] < zc) ? FAST_NU : count_fixed<7, false>(zs_s, zc))] = zc;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float v = sv[q];
                const int ind = iv[q];
                // v lies between the mid-points around coarse depth number ind: ind or ind + 1 depths are <= v
                int c = ind + ((ind < FAST_NZ && zc_s[min(ind, FAST_NZ - 1)] <= v) ? 1 : 0);
                const bool ok = (c == 0 || zc_s[max(c - 1, 0)] <= v) && (c == FAST_NZ || zc_s[min(c, FAST_NZ - 1)] > v);
                if (!ok) c = count_sorted<true>(zc_s, FAST_NZ, v);
                out[lane + 64 * q + c] = v;
            }
        } else {
            int r = lane;
            for (int k = 0; k < FAST_NU; ++k) r += (zs_s[k] < zc) ? 1 : 0;
            out[r] = zc;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float v = sv[q];
                const int j = lane + 64 * q;
                int rr = count_sorted<true>(zc_s, FAST_NZ, v);
                for (int k = 0; k < FAST_NU; ++k) {
                    const float o = zs_s[k];
                    rr += (o < v || (o == v && k < j)) ? 1 : 0;
                }
                out[rr] = v;
            }
        }
        __syncthreads();      // the per-ray buffers are reused by the next ray
    }
}

// A wavefront owns PDF_GROUP = 64 consecutive rays.  The parts of the algorithm that are serial along a ray (the
// ATen-ordered sum, the division by it and the float64 running sum of the cdf) run with ONE RAY PER LANE on rows staged
// in LDS with an odd row stride (conflict-free); the cdf overwrites the weights in place.  The parts that are parallel
// along a ray (searchsorted, interpolation, merge) then run ray by ray with one sample per lane.
__global__ __launch_bounds__(PDF_WAVES * 64) void sample_pdf_kernel(SamplePdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nb = a.nb, n_u = a.n_u, n_z = a.z_sorted ? a.n_z : 0;
    const int sc = nb | 1;                                    // odd row stride
    const int per_wave = PDF_GROUP * sc + nb + 2 * n_u + n_z;
    float* cdf_g = smem + (size_t)wv * per_wave;              // [PDF_GROUP][sc]: row = [0, cdf 1..nw]; first holds w + 1e-5
    float* bins_s = cdf_g + PDF_GROUP * sc;                   // the ray in flight: bins, samples, coarse depths
    float* zs_s = bins_s + nb;
    float* zc_s = zs_s + n_u;
    int* ind_s = reinterpret_cast<int*>(zc_s + n_z);          // searchsorted results of the ray in flight
    // coarse depths and bins come from the same array: the rank of a sample among the coarse depths follows from
    // its bin (checked, with the binary search as fallback)
    const bool z_is_bins = a.bins_are_z && a.z_sorted && a.z_vals == a.bins && a.n_z == nb + 1;
    const long long r0 = ((long long)blockIdx.x * PDF_WAVES + wv) * PDF_GROUP;
    const int n_live = (int)(a.R - r0 < PDF_GROUP ? a.R - r0 : PDF_GROUP);

    build_cdf_rows(a, cdf_g, sc, r0, n_live, lane);

    // ---- ray by ray: searchsorted, interpolation, z_std, merge
    for (int g = 0; g < n_live; ++g) {
        const bool live = true;
        const long long ray = r0 + g;
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
    if (nb == FAST_NB && n_u == FAST_NU && !u_per_ray && bins_are_z && z_sorted && z_vals == bins && n_z == FAST_NZ) {
        const size_t fast_shmem = ((size_t)PDF_GROUP * FAST_SC + 64 + FAST_NU + FAST_NZ) * sizeof(float);
        hipLaunchKernelGGL(sample_pdf_fast_kernel, dim3((R + PDF_GROUP - 1) / PDF_GROUP), dim3(WAVE), fast_shmem,
                           as_stream(stream), a);
        NERF_LAUNCH_CHECK();
        return NERF_OK;
    }
    const size_t per_wave = (size_t)PDF_GROUP * (nb | 1) + nb + 2 * (size_t)n_u + (z_sorted ? n_z : 0);
    const size_t shmem = per_wave * PDF_WAVES * sizeof(float);
    NERF_REQUIRE(shmem <= 64 * 1024, "nerf_sample_pdf: rows too long for the LDS staging (%zu bytes)", shmem);
    const int rays_per_block = PDF_WAVES * PDF_GROUP;
    dim3 grid((R + rays_per_block - 1) / rays_per_block), block(PDF_WAVES * WAVE);
    hipLaunchKernelGGL(sample_pdf_kernel, grid, block, shmem, as_stream(stream), a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}
