// Hierarchical (inverse-CDF) sampling: sample_pdf (run_nerf_helpers.py:196-239) with the
// mid-point, sort(cat) and z_std code around it in render_rays (run_nerf.py:397-401, 418).
//
// Three kernels, one arithmetic:
//   sample_pdf_kernel       any sizes, per-ray or shared variates.  A wavefront takes a group of rays; the parts that
//                           are serial along a ray (sum, pdf, cdf) run one ray per lane on rows staged in LDS, then ray
//                           by ray: searchsorted(right) by binary search over the LDS cdf, one sample per lane and
//                           step; samples -> LDS -> merge with the coarse depths by rank -> coalesced store
//   sample_pdf_fast_kernel  the shape render_rays uses (64 depths -> 63 mid-point bins, 128 shared variates, merge):
//                           the same with compile-time sizes and unrolled fixed-depth searches
//   sample_pdf_row_kernel   that shape with torch.linspace variates: one 16-lane row per ray, counting instead of
//                           searching (described at the kernel)
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
    int group;             // rays a wavefront of the bisection kernels takes together, 1 .. PDF_GROUP (few rays: more waves)
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

__device__ __forceinline__ bool variates_on_grid(const float* u, int lane);

constexpr int FAST_LDS = PDF_GROUP * FAST_SC + 64 + FAST_NU + FAST_NZ;      // floats

// One group of a.group rays, rays r0 ...; smem: FAST_LDS floats.  Ends with a workgroup barrier.
__device__ __forceinline__ void sample_pdf_fast_group(const SamplePdfArgs& a, float* smem, long long r0, int lane) {
    float* cdf_g = smem;                                      // [PDF_GROUP][63]
    float* bins_s = cdf_g + PDF_GROUP * FAST_SC;              // 64 (63 used)
    float* zs_s = bins_s + 64;                                // 128
    float* zc_s = zs_s + FAST_NU;                             // 64
    const int n_live = (int)(a.R - r0 < a.group ? a.R - r0 : a.group);
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

__global__ __launch_bounds__(64) void sample_pdf_fast_kernel(SamplePdfArgs a, int n_groups) {
    __shared__ __attribute__((aligned(16))) float smem[FAST_LDS];
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) sample_pdf_fast_group(a, smem, (long long)grp * a.group, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------------------------
// The same shape when the shared variates ascend along the regular grid k / 127 (what render_rays passes when it does
// not perturb: torch.linspace(0, 1, 128), run_nerf_helpers.py:210-212; also numpy.linspace of the pytest hook): ONE 16-LANE ROW PER RAY, four rays per
// wavefront.  A lane owns 4 consecutive weights / cdf entries / coarse depths and 8 consecutive samples, so all
// per-ray bookkeeping (sums, scans, checks) is a handful of DPP row operations shared by four rays, and nothing is
// searched:
//   searchsorted(cdf, u_k, right) = #{m : cdf_m <= u_k} = #{m : cnt_m <= k},   cnt_m = #{k : u_k < cdf_m}
// because u ascends; cnt_m comes from the grid (candidate trunc(127 cdf_m), settled by comparing cdf_m with the three
// variates around it: the float comparisons a search would make), a 129-bin histogram of the counts in LDS and a
// prefix sum give the index of every sample.  Merge: a sample of bin i lies between the mid-points around coarse
// depth i, so i or i + 1 depths are <= it (checked against its neighbours); depth m is preceded by the cnt_{m-1}
// samples of lower bins and by the samples of its own bin that lie below it (second histogram).
// The float64 running sum of the cdf is done as a parallel scan: every pdf value is a float32 of at least 2^-28 and
// the sum stays below 2, so all partial sums are multiples of 2^-52 below 2 - exactly representable, in any order.
// Rows for which that cannot be shown (checked), or whose samples / counts do not ascend, take slow exact paths.
#ifndef NERF_ROW_WAVES
#define NERF_ROW_WAVES 5
#endif
// LDS of one row (floats): the weights + 1e-5 as staged for the row sum; the nine partial sums; cdf and bin mid-points
// INTERLEAVED (entry i: cdf_i at 2 i, mid-point i at 2 i + 1) with sentinel entries -1 and 63 that are copies of
// entries 0 and 62, so that the reference's clamped lookups [max(i - 1, 0)], [min(i, 62)] of both arrays are the four
// consecutive floats at entry i - 1: one LDS instruction per sample; the coarse depths; the two marker arrays.
constexpr int ROW_FS = 0, ROW_PARTS = 64, ROW_CB = 84, ROW_ZC = 216, ROW_MARK = 284, ROW_H2 = 284 + 132, ROW_LDS = 284 + 132 + 68;
constexpr int MARK_DUMMY = 129, H2_DUMMY = 64;      // where a lane that has nothing to mark writes

template <int CTRL>
__device__ __forceinline__ int row_dpp_i(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float row_dpp_f(float old, float src) {
    return __builtin_bit_cast(float, row_dpp_i<CTRL>(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src)));
}
template <int CTRL>
__device__ __forceinline__ double row_dpp_d(double old, double src) {
    const long long o = __builtin_bit_cast(long long, old), v = __builtin_bit_cast(long long, src);
    const unsigned lo = (unsigned)row_dpp_i<CTRL>((int)(unsigned)o, (int)(unsigned)v);
    const unsigned hi = (unsigned)row_dpp_i<CTRL>((int)(unsigned)(o >> 32), (int)(unsigned)(v >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
constexpr int ROW_SHR = 0x110, ROW_SHL = 0x100;

// Do the 128 shared variates ascend along the regular grid k / 127 (within 1e-3 of a grid step)?  torch.linspace and
// numpy.linspace tables of any host do; the candidate count trunc(127 cdf) is then within one of the true count, which
// three comparisons against the actual variates settle.  (lane: variates 2 lane, 2 lane + 1)
__device__ __forceinline__ bool variates_on_grid(const float* u, int lane) {
    const float ua = u[2 * lane], ub = u[2 * lane + 1], uc = u[min(2 * lane + 2, FAST_NU - 1)];
    return __all(ua <= ub && ub <= uc && fabsf(ua * 127.0f - (float)(2 * lane)) <= 1e-3f &&
                 fabsf(ub * 127.0f - (float)(2 * lane + 1)) <= 1e-3f);
}

static_assert(4 * ROW_LDS + FAST_NU >= FAST_LDS, "the bisection path fits the row kernel's LDS");
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NERF_ROW_WAVES))) void sample_pdf_row_kernel(SamplePdfArgs a, int n_quads,
                                                                                                              int n_groups) {
    __shared__ __attribute__((aligned(16))) float smem[4 * ROW_LDS + FAST_NU];
    const int lane = threadIdx.x, j = lane & 15, row = lane >> 4, row0 = lane & 48;
    if (!variates_on_grid(a.u, lane)) {
        // any other shared variates: the bisection path, in this same launch (decided on the device: the host never
        // sees the variates), groups of a.group rays dealt to the workgroups
        for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) sample_pdf_fast_group(a, smem, (long long)grp * a.group, lane);
        return;
    }
    float* u_s = smem + 4 * ROW_LDS;
    u_s[lane] = a.u[lane];
    u_s[lane + 64] = a.u[lane + 64];
    float* base = smem + row * ROW_LDS;
    float* fs_s = base + ROW_FS;
    float* parts_s = base + ROW_PARTS;
    float* cb_s = base + ROW_CB;
    float* zc_s = base + ROW_ZC;
    int* mark = reinterpret_cast<int*>(base + ROW_MARK);        // slot c: 1 + the last cdf entry with count c
    int* h2 = reinterpret_cast<int*>(base + ROW_H2);            // slot i: 1 + the last sample of bin i below depth i
    float* out_s = base + ROW_MARK;                             // the merged row takes the markers' place once they are read
    float* zs_s = base + ROW_MARK;                              // so do the samples in the slow merge
    if (j == 0) cb_s[-2] = 0.f;                                 // sentinel entry -1: cdf[0] is 0 (its mid-point follows per ray)
    __syncthreads();
    float u[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) u[t] = u_s[8 * j + t];

    // the global reads of a quad are issued one quad ahead
    float zq[4], wq[4];
    auto request = [&](int quad) {
        const long long ru = (long long)quad * 4 + row;
        const size_t r = (size_t)(ru < a.R ? ru : a.R - 1);
        const float* zr = a.bins + r * a.bins_stride;
        const float* wr = a.weights + r * a.w_stride;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = 4 * j + t;
            zq[t] = zr[m];
            wq[t] = wr[min(max(m - 1, 0), FAST_NB - 2)];
        }
    };
    request(blockIdx.x < n_quads ? blockIdx.x : 0);
    for (int quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
        const long long ray_u = (long long)quad * 4 + row;
        const bool ray_ok = ray_u < a.R;
        const size_t ray = ray_ok ? (size_t)ray_u : (size_t)(a.R - 1);      // idle rows redo the last ray and store nothing
        // ---- coarse depths 4 j .. 4 j + 3, weights with the same numbers (weight m - 1 sits at cdf entry m)
        float z[5], f[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = 4 * j + t;
            z[t] = zq[t];
            f[t] = (m >= 1 && m <= FAST_NB - 1) ? fadd(wq[t], 1e-5f) : 0.f;
        }
        request(quad + (int)gridDim.x < n_quads ? quad + (int)gridDim.x : quad);
        __syncthreads();                                        // the previous quad's readers are done
        *reinterpret_cast<float4*>(fs_s + 4 * j) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(zc_s + 4 * j) = make_float4(z[0], z[1], z[2], z[3]);
        {
            float4* mz = reinterpret_cast<float4*>(base + ROW_MARK);             // 200 floats of markers: 50 quads
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            mz[j] = zero; mz[16 + j] = zero; mz[32 + j] = zero;
            if (j < 2) mz[48 + j] = zero;
        }
        z[4] = row_dpp_f<ROW_SHL + 1>(z[3], z[0]);              // the next lane's first depth (lane 15: its own last)
        __syncthreads();

        // ---- total of the 62 weights in ATen's order (row_sum_full for n = 62): eight strided partial sums and the
        //      tail 56..61, each a chain of adds, combined tail first, then the partial sums in order
        float part;
        {
            const int l = j < 8 ? j : 0;
            // weight i sits at fs_s[i + 1]
            const float x0 = fs_s[1 + (j == 8 ? 56 : l)], x1 = fs_s[1 + (j == 8 ? 57 : 32 + l)], x2 = fs_s[1 + (j == 8 ? 58 : 40 + l)],
                        x3 = fs_s[1 + (j == 8 ? 59 : 48 + l)], x4 = fs_s[1 + (j == 8 ? 60 : 8 + l)], x5 = fs_s[1 + (j == 8 ? 61 : 16 + l)],
                        x6 = fs_s[1 + 24 + l];
            // ATen starts every accumulator at 0: 0 + x0 is x0 except that it turns -0 into +0
            float sacc = fadd(0.f, x0);
            sacc = fadd(sacc, x1); sacc = fadd(sacc, x2); sacc = fadd(sacc, x3); sacc = fadd(sacc, x4); sacc = fadd(sacc, x5);
            part = j == 8 ? sacc : fadd(sacc, x6);
        }
        parts_s[j] = part;
        __syncthreads();
        float total;
        {
            const float4 p0 = *reinterpret_cast<const float4*>(parts_s), p1 = *reinterpret_cast<const float4*>(parts_s + 4);
            total = parts_s[8];
            total = fadd(total, p0.x); total = fadd(total, p0.y); total = fadd(total, p0.z); total = fadd(total, p0.w);
            total = fadd(total, p1.x); total = fadd(total, p1.y); total = fadd(total, p1.z); total = fadd(total, p1.w);
        }

        // ---- pdf, cdf
        float p[4];
        bool exact = true;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = 4 * j + t;
            const bool valid = m >= 1 && m <= FAST_NB - 1;
            p[t] = valid ? fdiv(f[t], total) : 0.f;
            exact = exact && (!valid || (p[t] >= 3.7252903e-09f && p[t] <= 1.0f));      // 2^-28
        }
        const double d0 = (double)p[0], d1 = __dadd_rn(d0, (double)p[1]), d2 = __dadd_rn(d1, (double)p[2]),
                     d3 = __dadd_rn(d2, (double)p[3]);
        double incl = d3;
        incl = __dadd_rn(incl, row_dpp_d<ROW_SHR + 1>(0.0, incl));
        incl = __dadd_rn(incl, row_dpp_d<ROW_SHR + 2>(0.0, incl));
        incl = __dadd_rn(incl, row_dpp_d<ROW_SHR + 4>(0.0, incl));
        incl = __dadd_rn(incl, row_dpp_d<ROW_SHR + 8>(0.0, incl));
        const double before = row_dpp_d<ROW_SHR + 1>(0.0, incl);
        float c[4] = {(float)__dadd_rn(before, d0), (float)__dadd_rn(before, d1), (float)__dadd_rn(before, d2),
                      (float)__dadd_rn(before, d3)};
        const unsigned long long ex_mask = __ballot(exact && incl < 1.5);
        const bool row_exact = ((ex_mask >> row0) & 0xffffull) == 0xffffull;
        {
            // entry 63 (the lane-15 slot past the end) is the sentinel: a copy of entry 62
            const float b0 = fmul(0.5f, fadd(z[1], z[0])), b1 = fmul(0.5f, fadd(z[2], z[1])), b2 = fmul(0.5f, fadd(z[3], z[2])),
                        b3 = fmul(0.5f, fadd(z[4], z[3]));
            *reinterpret_cast<float4*>(cb_s + 8 * j) = make_float4(c[0], b0, c[1], b1);
            *reinterpret_cast<float4*>(cb_s + 8 * j + 4) = make_float4(c[2], b2, j == 15 ? c[2] : c[3], j == 15 ? b2 : b3);
            if (j == 0) cb_s[-1] = b0;
        }
        __syncthreads();
        if (ex_mask != ~0ull) {
            if (!row_exact && j == 0) {
                // the sequential float64 sum, each prefix rounded (one lane; the rest of the wave waits)
                double cs = 0.0;
                cb_s[0] = 0.f;
#pragma nounroll
                for (int i = 1; i <= FAST_NB - 1; ++i) {
                    cs = __dadd_rn(cs, (double)fdiv(fs_s[i], total));
                    cb_s[2 * i] = (float)cs;
                }
                cb_s[2 * FAST_NB] = cb_s[2 * FAST_NB - 2];
            }
            __syncthreads();
            if (!row_exact) {
#pragma unroll
                for (int t = 0; t < 4; ++t) c[t] = cb_s[2 * (4 * j + t)];
            }
        }

        // ---- how many variates lie below each cdf entry
        int cnt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float cj = c[t];
            int k0 = (int)fmul(cj, 127.0f);                                    // NaN -> 0
            k0 = min(max(k0, 0), FAST_NU);
            const int km = k0 - 1, kp = k0 + 1;
            const float um = u_s[max(km, 0)], uc = u_s[min(k0, FAST_NU - 1)], up = u_s[min(kp, FAST_NU - 1)];
            cnt[t] = max(km, 0) + ((km >= 0 && um < cj) ? 1 : 0) + ((k0 < FAST_NU && uc < cj) ? 1 : 0) +
                     ((kp < FAST_NU && up < cj) ? 1 : 0);                      // 0 .. 128
        }
        // #{m : cnt_m <= k} for ascending counts = 1 + the last m with cnt_m <= k: the last entry of every run of equal
        // counts leaves its number + 1 at slot cnt (one writer per slot, no atomics), a running maximum fills the gaps
        const int cnt_next = row_dpp_i<ROW_SHL + 1>(cnt[3], cnt[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = 4 * j + t;
            const int nxt = t < 3 ? cnt[t + 1] : cnt_next;
            const bool writes = m == FAST_NB - 1 || (m < FAST_NB - 1 && nxt != cnt[t]);
            mark[writes ? cnt[t] : MARK_DUMMY] = m + 1;
        }
        __syncthreads();
        int iv[8];
        {
            const int4 ha = *reinterpret_cast<const int4*>(mark + 8 * j), hb = *reinterpret_cast<const int4*>(mark + 8 * j + 4);
            iv[0] = ha.x; iv[1] = max(iv[0], ha.y); iv[2] = max(iv[1], ha.z); iv[3] = max(iv[2], ha.w);
            iv[4] = max(iv[3], hb.x); iv[5] = max(iv[4], hb.y); iv[6] = max(iv[5], hb.z); iv[7] = max(iv[6], hb.w);
            int sc = iv[7];
            sc = max(sc, row_dpp_i<ROW_SHR + 1>(0, sc));
            sc = max(sc, row_dpp_i<ROW_SHR + 2>(0, sc));
            sc = max(sc, row_dpp_i<ROW_SHR + 4>(0, sc));
            sc = max(sc, row_dpp_i<ROW_SHR + 8>(0, sc));
            const int lower = row_dpp_i<ROW_SHR + 1>(0, sc);
#pragma unroll
            for (int t = 0; t < 8; ++t) iv[t] = max(iv[t], lower);
        }

        // ---- inverse cdf (the pair at ind - 1 is (cdf[below], cdf[above]) thanks to the sentinels)
        // A sample lies between the mid-points it was interpolated from (roundings are monotone), and those lie
        // between coarse depths i - 1, i and i, i + 1: whether depth i is <= the sample is all that is left to look up.
        float sv[8];
        bool good = true;
        unsigned on_edge = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float2* q = reinterpret_cast<const float2*>(cb_s + 2 * iv[t] - 2);
            const float2 lo = q[0], hi = q[1];                                // (cdf, mid-point) below and above
            float denom = fsub(hi.x, lo.x);
            if (denom < 1e-5f) denom = 1.0f;
            const float tt = fdiv(fsub(u[t], lo.x), denom);
            sv[t] = fadd(lo.y, fmul(tt, fsub(hi.y, lo.y)));
            good = good && lo.y <= sv[t] && sv[t] <= hi.y;
            if (sv[t] == hi.y && iv[t] != FAST_NB) on_edge |= 1u << t;
        }
        if (__any(on_edge != 0)) {
            // a sample equal to its upper mid-point (the last variate when the cdf ends above 1, typically): the
            // mid-point can round up to the depth above it, so that depth is looked at
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (on_edge & (1u << t)) good = good && zc_s[min(iv[t] + 1, FAST_NZ - 1)] > sv[t];
        }
        if (ray_ok) {
            const size_t o = ray * FAST_NU + 8 * j;
            if (a.z_samples) {
                *reinterpret_cast<float4*>(a.z_samples + o) = make_float4(sv[0], sv[1], sv[2], sv[3]);
                *reinterpret_cast<float4*>(a.z_samples + o + 4) = make_float4(sv[4], sv[5], sv[6], sv[7]);
            }
            if (a.inds) {
                *reinterpret_cast<int4*>(a.inds + o) = make_int4(iv[0], iv[1], iv[2], iv[3]);
                *reinterpret_cast<int4*>(a.inds + o + 4) = make_int4(iv[4], iv[5], iv[6], iv[7]);
            }
        }
        if (a.z_std) {
            float ls = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
            ls += row_dpp_f<ROW_SHR + 1>(0.f, ls); ls += row_dpp_f<ROW_SHR + 2>(0.f, ls);
            ls += row_dpp_f<ROW_SHR + 4>(0.f, ls); ls += row_dpp_f<ROW_SHR + 8>(0.f, ls);
            const float mean = __shfl(ls, row0 + 15, WAVE) / (float)FAST_NU;
            float lv = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) { const float d = sv[t] - mean; lv += d * d; }
            lv += row_dpp_f<ROW_SHR + 1>(0.f, lv); lv += row_dpp_f<ROW_SHR + 2>(0.f, lv);
            lv += row_dpp_f<ROW_SHR + 4>(0.f, lv); lv += row_dpp_f<ROW_SHR + 8>(0.f, lv);
            if (j == 15 && ray_ok) a.z_std[ray] = sqrtf(lv / (float)FAST_NU);
        }

        // ---- merge with the coarse depths
        const float s_next = row_dpp_f<ROW_SHL + 1>(sv[7], sv[0]);           // lane 15: its own last
        good = good && sv[7] <= s_next && cnt[0] <= cnt[1] && cnt[1] <= cnt[2] && (j == 15 || (cnt[2] <= cnt[3] && cnt[3] <= cnt_next));
#pragma unroll
        for (int t = 0; t < 7; ++t) good = good && sv[t] <= sv[t + 1];
        int pos[8];
        bool le[9];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            le[t] = zc_s[iv[t]] <= sv[t];
            pos[t] = 8 * j + t + iv[t] + (le[t] ? 1 : 0);
        }
        // the samples of a bin that lie below its coarse depth are the first of the bin (samples ascend): the last of
        // them leaves its number + 1 at the bin's slot (one writer per slot)
        const int i_next = row_dpp_i<ROW_SHL + 1>(-1, iv[0]);                // lane 15: no bin follows
        le[8] = row_dpp_i<ROW_SHL + 1>(1, le[0] ? 1 : 0) != 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int inx = t < 7 ? iv[t + 1] : i_next;
            const bool writes = !le[t] && (inx != iv[t] || le[t + 1]);
            h2[writes ? iv[t] : H2_DUMMY] = 8 * j + t + 1;
        }
        __syncthreads();
        const unsigned long long good_mask = __ballot(good);
        const bool row_good = ((good_mask >> row0) & 0xffffull) == 0xffffull;
        float* out = a.z_sorted + ray * (FAST_NZ + FAST_NU);
        const int cnt_prev = row_dpp_i<ROW_SHR + 1>(0, cnt[3]);               // lane 0: no entry before its first
        int rank[4];
        {
            const int4 hq = *reinterpret_cast<const int4*>(h2 + 4 * j);
            rank[0] = 4 * j + max(cnt_prev, hq.x);
            rank[1] = 4 * j + 1 + max(cnt[0], hq.y);
            rank[2] = 4 * j + 2 + max(cnt[1], hq.z);
            rank[3] = 4 * j + 3 + max(cnt[2], hq.w);
#pragma unroll
            for (int t = 0; t < 4; ++t) rank[t] = min(rank[t], FAST_NZ + FAST_NU - 1);
        }
        __syncthreads();                                        // both marker arrays are read: their space takes the merged row
        if (row_good) {
            // scattered 4-byte stores to memory cost more than the rest of the kernel together: scatter in LDS, then
            // three 16-byte stores per lane that are contiguous across the row
#pragma unroll
            for (int t = 0; t < 4; ++t) out_s[rank[t]] = z[t];
#pragma unroll
            for (int t = 0; t < 8; ++t) out_s[pos[t]] = sv[t];
        }
        __syncthreads();
        if (row_good && ray_ok) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = (16 * i + j) * 4;
                *reinterpret_cast<float4*>(out + q) = *reinterpret_cast<const float4*>(out_s + q);
            }
        }
        if (good_mask != ~0ull) {
            // ranks by counting (ties: coarse depths first, then samples by index); everything by index out of LDS,
            // loops left rolled: this path is for inputs that break the ordering assumptions, it only has to be right
            // and must not cost the fast path registers
            __syncthreads();
            if (!row_good) {
#pragma unroll
                for (int t = 0; t < 8; ++t) zs_s[8 * j + t] = sv[t];
            }
            __syncthreads();
            if (!row_good && ray_ok) {
#pragma nounroll
                for (int t = 0; t < 4; ++t) {
                    const float v = zc_s[4 * j + t];
                    int r = 4 * j + t;
#pragma nounroll
                    for (int k = 0; k < FAST_NU; ++k) r += (zs_s[k] < v) ? 1 : 0;
                    out[r] = v;
                }
#pragma nounroll
                for (int t = 0; t < 8; ++t) {
                    const int own = 8 * j + t;
                    const float v = zs_s[own];
                    int r = 0;
#pragma nounroll
                    for (int i = 0; i < FAST_NZ; ++i) r += (zc_s[i] <= v) ? 1 : 0;
#pragma nounroll
                    for (int k = 0; k < FAST_NU; ++k) {
                        const float o = zs_s[k];
                        r += (o < v || (o == v && k < own)) ? 1 : 0;
                    }
                    out[r] = v;
                }
            }
        }
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
    const long long r0 = ((long long)blockIdx.x * PDF_WAVES + wv) * a.group;
    const int n_live = (int)(a.R - r0 < a.group ? a.R - r0 : a.group);

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
    // The serial phases run one ray per lane for a group of rays and the rest ray by ray: with few rays (a training
    // batch) smaller groups give more wavefronts and a shorter critical path (512 rays: 0.257 ms with groups of 16).
    int group = R / 2048;
    group = group < 1 ? 1 : (group > PDF_GROUP ? PDF_GROUP : group);
    SamplePdfArgs a{group, bins, bins_stride, bins_are_z, nb, weights, w_stride, u, n_u, u_per_ray, R,
                    z_vals, n_z, z_samples, inds, z_sorted, z_std};
    const int n_groups = (R + group - 1) / group;
    if (nb == FAST_NB && n_u == FAST_NU && !u_per_ray && bins_are_z && z_sorted && z_vals == bins && n_z == FAST_NZ) {
        // ONE launch: the row kernel works when the variates are the regular ascending grid and runs the bisection path
        // itself otherwise (decided on the device); outputs that are not 16-byte aligned go to the bisection kernel
        if ((((uintptr_t)z_samples | (uintptr_t)inds | (uintptr_t)z_sorted) & 15) == 0) {
            const int n_quads = (R + 3) / 4;
            hipLaunchKernelGGL(sample_pdf_row_kernel, dim3(n_quads < 16384 ? n_quads : 16384), dim3(WAVE), 0, as_stream(stream), a,
                               n_quads, n_groups);
        } else {
            hipLaunchKernelGGL(sample_pdf_fast_kernel, dim3(n_groups < 65536 ? n_groups : 65536), dim3(WAVE), 0, as_stream(stream),
                               a, n_groups);
        }
        NERF_LAUNCH_CHECK();
        return NERF_OK;
    }
    const size_t per_wave = (size_t)PDF_GROUP * (nb | 1) + nb + 2 * (size_t)n_u + (z_sorted ? n_z : 0);
    const size_t shmem = per_wave * PDF_WAVES * sizeof(float);
    NERF_REQUIRE(shmem <= 64 * 1024, "nerf_sample_pdf: rows too long for the LDS staging (%zu bytes)", shmem);
    const int rays_per_block = PDF_WAVES * group;
    dim3 grid((R + rays_per_block - 1) / rays_per_block), block(PDF_WAVES * WAVE);
    hipLaunchKernelGGL(sample_pdf_kernel, grid, block, shmem, as_stream(stream), a);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}
