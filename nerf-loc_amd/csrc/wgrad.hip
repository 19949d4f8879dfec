// Weight gradients of the NeRF network: dW[out, in] = sum_m G[m, out] X[m, in] with the sample index as the MFMA k
// dimension (what autograd computes for the Linear layers of run_nerf_helpers.py:67-119 in loss.backward(),
// run_nerf.py:785).  Kept in its own translation unit: the 256 accumulator registers of a wave live in AGPRs here,
// whereas the layer-chain kernels are compiled with accumulators in arch VGPRs.
#include <stdlib.h>
#include "wgrad.h"
#include "mlp_layout.h"
#include "mlp_device.h"

namespace nerf {

// ------------------------------------------------------------------------------------------
// weight gradients
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* G; int ldg; int N;      // gradient rows   [M, ldg], N <= 256 output features used
    const float* X; int ldx; int K;      // layer inputs    [M, ldx], K <= 32 CT input features used
    long long M;
    long long chunk;                     // samples per workgroup (even)
    float* part;                         // [grid][256][32 CT]
    float* bpart;                        // [grid][256] column sums of G, or nullptr
};

// v_mfma_f32_32x32x2_f32 with i = output feature, j = input feature, k = sample.  Which feature an MFMA row or column
// stands for is free as long as the partial tile is stored accordingly, so a lane takes CONSECUTIVE features and reads
// them with one wide load: lane (r, h) of wave w feeds G[m + h][64 w + 2 r + i] (i: the wave's two row tiles, one
// 8-byte load) and X[m + h][128 g + 4 r + e] (column tile 4 g + e, one 16-byte load per group g of four tiles; with
// fewer than four tiles 2 r + e or r).  Three loads per k-step instead of ten, and the partial tile leaves in 16-byte
// stores.  N is a multiple of 64, K = 32 CT, ldg even, ldx a multiple of 4 (checked by the host).
template <int V> struct WgVec { typedef float T __attribute__((ext_vector_type(V))); };
template <> struct WgVec<1> { typedef float T; };

template <int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void wgrad_kernel(WgradArgs a) {
    constexpr int V = CT < 4 ? CT : 4;             // consecutive input columns of a lane per group
    constexpr int NG = CT / V;                     // groups of 32 V columns
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // N = 128 (views_linears.0) needs two of the four waves for its rows: the other two take the second half of the
    // workgroup's samples and leave their partial tile in rows 128..255 of the slot, which nobody else writes.
    const bool halves = a.N == 128;
    const int half = halves ? wave >> 1 : 0;
    const int row0 = halves ? 64 * (wave & 1) : 64 * wave;
    const int prow0 = row0 + 128 * half;           // where the wave's rows go in the partial tile
    float* part = a.part + (size_t)blockIdx.x * 256 * (32 * CT);
    if (row0 >= a.N) return;                       // whole wave: these rows are never read back
    long long m0 = (long long)blockIdx.x * a.chunk;
    long long m1 = m0 + a.chunk < a.M ? m0 + a.chunk : a.M;
    const long long last_row = m1 - 1;             // a row of the workgroup's range: where clamped requests go
    if (halves) {
        const long long mid = m0 + (((m1 - m0) / 2 + 1) & ~1LL);
        if (half == 0) m1 = mid < m1 ? mid : m1; else m0 = mid < m1 ? mid : m1;
    }

    f32x16 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][ct][e] = 0.0f;
    float bs0 = 0.0f, bs1 = 0.0f;

    // Only the sample index needs a bound.  Loads are unconditional from a clamped row and zeroed by a select: no
    // branches in the loop.  The row pointers advance by addition (two rows per k-step); the requests of a k-step are
    // issued after the first MFMA pair of an earlier one, so that address arithmetic and load issue run in the shadow
    // of the 64-cycle MFMAs and not between the k-steps.
#ifndef NERF_WG_DEPTH
#define NERF_WG_DEPTH 8
#endif
    constexpr int DEPTH = NERF_WG_DEPTH;           // k-steps of operands in flight
    static_assert(DEPTH >= 2, "the slot being refilled must not be the one being consumed");
    static_assert(DEPTH * (1 + NG) <= 63, "more loads in flight than vmcnt can count");
    // The loads are inline asm and the waits are counted by hand: hipcc's own vmcnt bookkeeping gives up at the loop's
    // back edge and drains every load in flight once per trip.  A request is 1 + NG loads; when k-step d is consumed
    // the requests of the DEPTH - 2 k-steps after it are the only ones that may still be under way.
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef typename WgVec<V>::T xvec;
    constexpr int LOADS = 1 + NG;
    f32x2v ga[DEPTH];
    xvec xb[DEPTH][NG];
    bool live[DEPTH];                              // sample of the slot inside [m0, m1); applied when consumed
    const int len = (int)(m1 - m0);                // samples of this workgroup (a chunk is far below 2^31)
    const float* const g_last = a.G + row0 + 2 * r + (size_t)last_row * a.ldg;
    const float* const x_last = a.X + V * r + (size_t)last_row * a.ldx;
    const float* gq = a.G + row0 + 2 * r + (size_t)(m0 + h) * a.ldg;     // row of the k-step to be requested next
    const float* xq = a.X + V * r + (size_t)(m0 + h) * a.ldx;
    const size_t g_step = (size_t)2 * a.ldg, x_step = (size_t)2 * a.ldx;
    int rel = h;                                   // that row's number within the chunk
    auto request = [&](int slot) {
        const bool ok = rel < len;
        const float* gs = ok ? gq : g_last;
        const float* xs = ok ? xq : x_last;
        live[slot] = ok;
        gq += g_step; xq += x_step; rel += 2;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ga[slot]) : "v"(gs) : "memory");
        if constexpr (V == 4) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xb[slot][0]) : "v"(xs) : "memory");
            if constexpr (NG == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(xb[slot][1]) : "v"(xs) : "memory");
        } else if constexpr (V == 2) {
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xb[slot][0]) : "v"(xs) : "memory");
        } else {
            asm volatile("global_load_dword %0, %1, off" : "=v"(xb[slot][0]) : "v"(xs) : "memory");
        }
    };
    auto arrived = [&](int slot) {                 // the slot's registers may be read after this point
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 2) * LOADS) : "memory");
        if constexpr (NG == 2) asm volatile("" : "+v"(ga[slot]), "+v"(xb[slot][0]), "+v"(xb[slot][1]));
        else asm volatile("" : "+v"(ga[slot]), "+v"(xb[slot][0]));
    };
    auto x_of = [&](int slot, int ct) -> float {
        if constexpr (V == 1) return xb[slot][0];
        else return xb[slot][ct / V][ct % V];
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) request(d);
    for (int mb = 0; mb < len; mb += 2 * DEPTH) {
        // one fixed AGPR tuple per accumulator across the back edge (without this hipcc swaps two of them every trip)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+a"(acc[i][ct]));
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            arrived(d);
            // beyond the range only G is zeroed: X then comes from the clamped (real, finite) row and meets a zero
            const float g0 = live[d] ? ga[d][0] : 0.0f, g1 = live[d] ? ga[d][1] : 0.0f;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, x_of(d, 0), acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, x_of(d, 0), acc[1][0], 0, 0, 0);
            request((d + DEPTH - 1) % DEPTH);                      // k-step d + DEPTH - 1, under the MFMAs just issued
#pragma unroll
            for (int ct = 1; ct < CT; ++ct) {
                acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, x_of(d, ct), acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, x_of(d, ct), acc[1][ct], 0, 0, 0);
            }
            bs0 += g0; bs1 += g1;
        }
    }
    // Requests past the end are still under way.  Every register they will write is an operand of the statement that
    // waits for them, so none of these registers can serve as a temporary of the epilogue before the wait is over.
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if constexpr (NG == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ga[d]), "+v"(xb[d][0]), "+v"(xb[d][1]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ga[d]), "+v"(xb[d][0]) :: "memory");
    }
    // element e of an accumulator is MFMA row (e & 3) + 8 (e >> 2) + 4 h, i.e. output feature row0 + 2 * that + i
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = prow0 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * h) + i;
            float* dst = part + (size_t)row * (32 * CT) + V * r;
            if constexpr (V == 4) {
#pragma unroll
                for (int grp = 0; grp < NG; ++grp)
                    *reinterpret_cast<float4*>(dst + 128 * grp) = make_float4(acc[i][4 * grp][e], acc[i][4 * grp + 1][e],
                                                                              acc[i][4 * grp + 2][e], acc[i][4 * grp + 3][e]);
            } else if constexpr (V == 2) {
                *reinterpret_cast<float2*>(dst) = make_float2(acc[i][0][e], acc[i][1][e]);
            } else {
                dst[0] = acc[i][0][e];
            }
        }
    }
    if (a.bpart) {
        bs0 += __shfl_xor(bs0, 32, WAVE);
        bs1 += __shfl_xor(bs1, 32, WAVE);
        if (h == 0) *reinterpret_cast<float2*>(a.bpart + (size_t)blockIdx.x * 256 + prow0 + 2 * r) = make_float2(bs0, bs1);
    }
}

// Heads with up to 5 outputs (alpha_linear, rgb_linear, output_linear): G is a column range of d_raw [M, ldr].  One
// thread per input feature, samples of the block's range in sequence; X rows are read coalesced, G by broadcast.
constexpr int WS_MAX_N = 5;
struct WsmallArgs {
    const float* d_raw; int ldr, c0, N;  // output o <-> d_raw column c0 + o, N <= WS_MAX_N
    const float* X; int K;               // [M, K], K = blockDim.x
    long long M, chunk;
    float* part;                         // [grid][8][K]
    float* bpart;                        // [grid][8]
};
__global__ void wgrad_small_kernel(WsmallArgs a) {
    const int k = threadIdx.x;
    const long long m0 = (long long)blockIdx.x * a.chunk;
    const long long m1 = m0 + a.chunk < a.M ? m0 + a.chunk : a.M;
    float s[WS_MAX_N], b[WS_MAX_N];
#pragma unroll
    for (int o = 0; o < WS_MAX_N; ++o) s[o] = b[o] = 0.f;
    // eight samples' loads are issued before the first of them is used (the sums stay in sample order)
    constexpr int U = 8;
    long long m = m0;
    for (; m + U <= m1; m += U) {
        float x[U], gv[U][WS_MAX_N];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = a.X[(size_t)(m + u) * a.K + k];
            const float* g = a.d_raw + (size_t)(m + u) * a.ldr + a.c0;
#pragma unroll
            for (int o = 0; o < WS_MAX_N; ++o) gv[u][o] = o < a.N ? g[o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int o = 0; o < WS_MAX_N; ++o)
                if (o < a.N) { s[o] = fmaf(gv[u][o], x[u], s[o]); b[o] += gv[u][o]; }
    }
    for (; m < m1; ++m) {
        const float x = a.X[(size_t)m * a.K + k];
        const float* g = a.d_raw + (size_t)m * a.ldr + a.c0;
#pragma unroll
        for (int o = 0; o < WS_MAX_N; ++o)
            if (o < a.N) { const float gv = g[o]; s[o] = fmaf(gv, x, s[o]); b[o] += gv; }
    }
    for (int o = 0; o < a.N; ++o) a.part[((size_t)blockIdx.x * 8 + o) * a.K + k] = s[o];
    if (k == 0) for (int o = 0; o < a.N; ++o) a.bpart[(size_t)blockIdx.x * 8 + o] = b[o];
}

// dst[row * ld + col_off + map(col)] = sum over parts, in part order.  map: 0 identity, 1 encoded-point
// k-slot -> reference column, 2 encoded-view k-slot -> reference column (slots without a feature are dropped);
// 3, 4: the same for the bf16 operand order, where column 16 s + f of the partial tile is element
// j = 4 (f >> 3) + (f & 3) of lane half (f >> 2) & 1 in k-step s, i.e. slot 8 s + j of that half (mlp_layout.h).
// x_scale (device, may be null): the inputs were stored multiplied by *x_scale, a power of two; undone here.
struct ReduceArgs {
    const float* part; int n_parts; int prow; int pcol;   // [n_parts][prow][pcol]
    int N, K;                                             // rows / columns to emit
    float* dst; int ld; int col_off; int map;
    const float* x_scale;
    const float* bpart; float* db; int bstride;           // optional: block row N adds [n_parts][bstride] into db[N]
};
// Block: 64 columns x RED_SPLIT slices of the parts.  A thread adds its slice in part order (eight interleaved chains
// keep eight loads in flight), the slices are combined in slice order through LDS: the order of additions is fixed.
constexpr int RED_SPLIT = 4;
__global__ __launch_bounds__(64 * RED_SPLIT) void wgrad_reduce_kernel(ReduceArgs a) {
    __shared__ float slice_sum[RED_SPLIT][64];
    const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx, row = blockIdx.y;
    const bool bias_row = row == a.N && a.db;
    bool on = !(bias_row ? col >= a.N : (col >= a.K || row >= a.N));
    int dc = col;
    if (a.map == 1) dc = pe_slot_feature(col >> 5, col & 31);
    else if (a.map == 2) dc = view_slot_feature(col >> 4, col & 15);
    else if (a.map >= 3) {
        const int f = col & 15, hh = (f >> 2) & 1, q = 8 * (col >> 4) + 4 * (f >> 3) + (f & 3);
        dc = a.map == 3 ? pe_slot_feature(hh, q) : view_slot_feature(hh, q);
    }
    if (dc < 0 && !bias_row) on = false;
    float sum = 0.f;
    if (on) {
        const float* p = bias_row ? a.bpart + col : a.part + (size_t)row * a.pcol + col;
        const size_t stride = bias_row ? (size_t)a.bstride : (size_t)a.prow * a.pcol;
        const int per = (a.n_parts + RED_SPLIT - 1) / RED_SPLIT;
        int i = sl * per;
        const int end = i + per < a.n_parts ? i + per : a.n_parts;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (; i + 8 <= end; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += p[(size_t)(i + j) * stride];
        }
        for (int j = 0; i < end; ++i, ++j) s[j] += p[(size_t)i * stride];
        sum = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
    slice_sum[sl][tx] = sum;
    __syncthreads();
    if (sl != 0 || !on) return;
#pragma unroll
    for (int q = 1; q < RED_SPLIT; ++q) sum += slice_sum[q][tx];
    if (bias_row) { a.db[col] = sum; return; }
    if (a.x_scale) sum *= 1.0f / *a.x_scale;
    a.dst[(size_t)row * a.ld + a.col_off + dc] = sum;
}

// NERF_WG_MIN_CHUNK (environment, read once): tuning experiments only
static long long wg_min_chunk() {
    static long long v = 0;
    if (v == 0) {
        const char* e = getenv("NERF_WG_MIN_CHUNK");
        long long x = e ? atoll(e) : 128;
        v = x >= 2 ? x : 128;
    }
    return v;
}

int launch_wgrad_reduce(const float* ws, int parts, int pcol, int pcol0, int n_rows, int k_emit, float* dW, int ld,
                        int col_off, int map, const float* x_scale, float* db, hipStream_t st) {
    ReduceArgs r{ws + pcol0, parts, 256, pcol, n_rows, k_emit, dW, ld, col_off, map, x_scale,
                 ws + WG_BIAS_OFFSET, db, 256};
    const int cols = k_emit > n_rows ? k_emit : n_rows;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((cols + 63) / 64, n_rows + (db ? 1 : 0)), dim3(64 * RED_SPLIT), 0, st, r);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

int launch_wgrad(const float* G, int ldg, int N, const float* X, int ldx, int K, long long M,
                        float* ws, float* dW, int ld, int col_off, int map, float* db, hipStream_t st) {
    const int ct = (K + 31) / 32;
    NERF_REQUIRE(N % 64 == 0 && N <= 256 && (K == 32 || K == 64 || K == 128 || K == 256),
                 "weight gradient: unsupported shape %d x %d", N, K);
    NERF_REQUIRE(ldg % 2 == 0 && ldx % 4 == 0 && ((uintptr_t)G & 7) == 0 && ((uintptr_t)X & 15) == 0,
                 "weight gradient: rows of G / X must be 8- / 16-byte aligned (ldg %d, ldx %d)", ldg, ldx);
    // Samples per workgroup: at least wg_min_chunk().  Fewer workgroups write fewer partial tiles (256 KB each), but a
    // 512-ray shard of a data-parallel batch (32 768 coarse samples) then keeps only 64 of 256 compute units busy.
    long long parts = (M + wg_min_chunk() - 1) / wg_min_chunk();
    if (parts > WG_MAX_PARTS) parts = WG_MAX_PARTS;
    if (parts > cu_count()) parts = cu_count();
    if (parts < 1) parts = 1;
    long long chunk = (M + parts - 1) / parts;
    chunk += chunk & 1;
    parts = (M + chunk - 1) / chunk;
    WgradArgs a{G, ldg, N, X, ldx, K, M, chunk, ws, db ? ws + WG_BIAS_OFFSET : nullptr};
    const int ctp = ct <= 1 ? 1 : ct <= 2 ? 2 : ct <= 4 ? 4 : 8;
    dim3 grid((unsigned)parts), block(256);
    switch (ctp) {
        case 1: hipLaunchKernelGGL(wgrad_kernel<1>, grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL(wgrad_kernel<2>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(wgrad_kernel<4>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(wgrad_kernel<8>, grid, block, 0, st, a); break;
    }
    NERF_LAUNCH_CHECK();
    // N = 128: every slot holds two partial tiles of 128 rows (see the kernel), i.e. 2 x parts tiles of 128 rows
    const int tiles = N == 128 ? 2 * (int)parts : (int)parts, tile_rows = N == 128 ? 128 : 256;
    // one launch: block rows 0 .. N - 1 reduce the weight tile, block row N (if db) the column sums of G
    ReduceArgs r{ws, tiles, tile_rows, 32 * ctp, N, map == 0 ? K : 32 * ct, dW, ld, col_off, map, nullptr,
                 db ? a.bpart : nullptr, db, tile_rows};
    const int cols = (db && N > r.K) ? N : r.K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((cols + 63) / 64, N + (db ? 1 : 0)), dim3(64 * RED_SPLIT), 0, st, r);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

int launch_wgrad_small(const float* d_raw, int ldr, int c0, int N, const float* X, int K, long long M, float* ws,
                       float* dW, float* db, hipStream_t st) {
    NERF_REQUIRE(N >= 1 && N <= WS_MAX_N && K <= 256, "weight gradient (head): unsupported shape %d x %d", N, K);
    long long parts = (M + 255) / 256;
    if (parts > 1024) parts = 1024;
    const long long chunk = (M + parts - 1) / parts;
    parts = (M + chunk - 1) / chunk;
    WsmallArgs a{d_raw, ldr, c0, N, X, K, M, chunk, ws, ws + (size_t)1024 * 8 * 256};
    hipLaunchKernelGGL(wgrad_small_kernel, dim3((unsigned)parts), dim3(K), 0, st, a);
    NERF_LAUNCH_CHECK();
    ReduceArgs r{ws, (int)parts, 8, K, N, K, dW, K, 0, 0, nullptr, a.bpart, db, 8};      // block row N: the bias
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((K + 63) / 64, N + 1), dim3(64 * RED_SPLIT), 0, st, r);
    NERF_LAUNCH_CHECK();
    return NERF_OK;
}

}  // namespace nerf
