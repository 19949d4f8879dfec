// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 for the instruction mixes of the fused MLP kernel.
// One wave per SIMD (256 threads, 1 block per CU).  Cycles from s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_relu(float a, float b) {
    const f32x2 f = {a, b};
    s16x2 i = __builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2));
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(i, z));
}

// V: 0 = MFMA only; 1 = + LDS fragment queue; 2 = 1 + distinct B operand per k-step (16 x 2 registers sets);
//    3 = 2 + epilogue slices of the previous tile (2 acc regs -> cvt_pk -> pk_max -> next B operand)
template <int V>
__global__ __launch_bounds__(256) void k(const u32x4* g, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 A[16][2], B[16][2];
    for (int i = 0; i < 16; ++i) for (int c = 0; c < 2; ++c) { A[i][c] = g[4096 + (threadIdx.x * 32 + i * 2 + c) % 4096]; B[i][c] = A[i][c]; }
    f32x16 acc[2][2] = {};
    float side[16];
    for (int i = 0; i < 16; ++i) side[i] = (float)(threadIdx.x + i) * 0.001f;
    unsigned sdummy = 0;
    u32x4 q[4];
    for (int kk = 0; kk < 3; ++kk) q[kk] = lds[kk * 64 + lane];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {          // 4 tiles x 16 k-steps x 2 MFMAs
            f32x16 (&cur)[2] = acc[tile & 1];
            f32x16 (&prv)[2] = acc[(tile & 1) ^ 1];
            u32x4 (*src)[2] = (tile & 1) ? B : A;
            u32x4 (*dst)[2] = (tile & 1) ? A : B;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int kk = tile * 16 + s;
                __builtin_amdgcn_sched_barrier(0);
                if (V >= 1) q[(kk + 3) & 3] = lds[((kk + 3) & 63) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 w = __builtin_bit_cast(bf16x8, V >= 1 ? q[kk & 3] : q[0]);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const u32x4 b = V >= 2 ? src[s][c] : src[0][c];
                    cur[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, __builtin_bit_cast(bf16x8, b), cur[c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if ((V == 3 || V == 5 || V == 6) && s >= 1 && s <= 8) {
                        const int sl = (s - 1) * 2 + c, gq = sl >> 2, qd = sl & 3, ct = gq >> 1, half = gq & 1;
                        if (V == 5)        // same VALU work on registers that no MFMA touches
                            dst[2 * (tile & 3) + half][ct][qd] = pack_relu(side[(sl * 2) & 15], side[(sl * 2 + 1) & 15]);
                        else
                            dst[2 * (tile & 3) + half][ct][qd] = pack_relu(prv[ct][8 * half + 2 * qd], prv[ct][8 * half + 2 * qd + 1]);
                        if (V == 6) {      // twice the VALU work
                            const int sl2 = sl + 16, gq2 = sl2 >> 2, ct2 = gq2 >> 1 & 1, half2 = gq2 & 1;
                            dst[8 + 2 * (tile & 3) + half2][ct2][qd] = pack_relu(prv[ct2][8 * half2 + 2 * qd], prv[ct2][8 * half2 + 2 * qd + 1]);
                        }
                    }
                    if (V == 4 && c == 1 && s >= 1 && s <= 8) {   // both slices of the k-step after its second MFMA
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc) {
                            const int sl = (s - 1) * 2 + cc, gq = sl >> 2, qd = sl & 3, ct = gq >> 1, half = gq & 1;
                            dst[2 * (tile & 3) + half][ct][qd] = pack_relu(prv[ct][8 * half + 2 * qd], prv[ct][8 * half + 2 * qd + 1]);
                        }
                    }
                    if (V == 7 && s >= 1 && s <= 8) {             // scalar ALU work instead of vector ALU work
                        asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3" : "+s"(sdummy));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if ((it & 7) == 7) for (int t = 0; t < 2; ++t) for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) acc[t][c][e] *= 1e-3f;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int t = 0; t < 2; ++t) for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) s += acc[t][c][e];
    for (int i = 0; i < 16; ++i) for (int c = 0; c < 2; ++c) s += (float)(A[i][c][0] ^ B[i][c][1]);
    s += side[3] + (float)sdummy;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const size_t n = 8192;
    std::vector<unsigned> h(n * 4);
    srand(1);
    for (auto& v : h) {
        unsigned lo = (rand() & 0x807f) | ((0x3f80 - ((rand() % 8) << 7)) & 0x7f80);
        unsigned hi = (rand() & 0x807f) | ((0x3f80 - ((rand() % 8) << 7)) & 0x7f80);
        v = lo | (hi << 16);
    }
    u32x4* g; float* out; unsigned long long* cyc;
    hipMalloc(&g, n * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemcpy(g, h.data(), n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    for (int rep = 0; rep < 1; ++rep)
        for (int v = 0; v < 8; ++v) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
            if (v == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
            if (v == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
            if (v == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
            if (v == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
            if (v == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[256]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
            double c = 0; for (int i = 0; i < 256; ++i) c += hc[i]; c /= 256;
            const double nm = (double)iters * 128;
            printf("variant %d: %.2f ms, %.1f cycles/MFMA, clock %.2f GHz, %.0f TFLOP/s\n", v, ms, c / nm, c / ms / 1e6,
                   256.0 * 4 * nm * 32768 / ms / 1e9);
        }
    return 0;
}
