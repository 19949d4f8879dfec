// Micro-benchmark: bf16 MFMA shape (32x32x16 vs 16x16x32) with the A operand re-read from LDS, random data,
// one wave per SIMD, same work per LDS fragment.  Build: hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const u32x4* g, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[4096];   // 64 KiB of "weights"
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 b[16];
    for (int i = 0; i < 16; ++i) b[i] = g[4096 + threadIdx.x * 16 + i];
    f32x16 a32[2] = {};
    f32x4 a16[4] = {};
    u32x4 q[4];
    for (int kk = 0; kk < 3; ++kk) q[kk] = lds[kk * 64 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
            q[(kk + 3) & 3] = lds[((kk + 3) & 63) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 w = __builtin_bit_cast(bf16x8, q[kk & 3]);
            if (SHAPE == 32) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    a32[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, __builtin_bit_cast(bf16x8, b[(kk * 2 + c) & 15]), a32[c], 0, 0, 0);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    a16[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, b[(kk * 4 + c) & 15]), a16[c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // keep values bounded
        if ((it & 15) == 15) {
            for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) a32[c][e] *= 1e-3f;
            for (int c = 0; c < 4; ++c) for (int e = 0; e < 4; ++e) a16[c][e] *= 1e-3f;
        }
    }
    float s = 0;
    for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) s += a32[c][e];
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 4; ++e) s += a16[c][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const size_t n = 4096 + 256 * 16;
    std::vector<unsigned> h(n * 4);
    srand(1);
    for (auto& v : h) {   // random bf16 pairs in (-1, 1)
        unsigned lo = (rand() & 0x807f) | ((0x3f80 - ((rand() % 8) << 7)) & 0x7f80);
        unsigned hi = (rand() & 0x807f) | ((0x3f80 - ((rand() % 8) << 7)) & 0x7f80);
        v = lo | (hi << 16);
    }
    u32x4* g; float* out;
    hipMalloc(&g, n * 16); hipMalloc(&out, 256 * 256 * 4);
    hipMemcpy(g, h.data(), n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16}) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, g, out, iters);
            else hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, g, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 256.0 * 4 * iters * 64 * 2 * 32768.0;   // per frag: 2 x (32x32x16) or 4 x (16x16x32) MFMAs
            printf("shape %dx%d: %.2f ms, %.0f TFLOP/s\n", shape, shape, ms, flop / ms / 1e9);
        }
    return 0;
}
