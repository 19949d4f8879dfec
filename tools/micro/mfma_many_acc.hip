// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 when one wave cycles through NA distinct accumulator tiles
// (the weight-gradient pattern: 2 A operands x 9 B operands per k-step), one wave per SIMD.
// Build twice: with and without -mllvm -amdgpu-mfma-vgpr-form=1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int RT, int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(const u32x4* g, float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[RT][CT];
    for (int i = 0; i < RT; ++i) for (int c = 0; c < CT; ++c) for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;
    u32x4 a[RT], b[CT];
    for (int i = 0; i < RT; ++i) a[i] = g[threadIdx.x + 256 * i];
    for (int c = 0; c < CT; ++c) b[c] = g[threadIdx.x + 256 * (RT + c)];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int c = 0; c < CT; ++c)
                acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[c]), acc[i][c], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < RT; ++i) for (int c = 0; c < CT; ++c) for (int e = 0; e < 16; ++e) s += acc[i][c][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int RT, int CT>
void run(const u32x4* g, float* out, unsigned long long* cyc, const char* name) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<RT, CT>), dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RT, CT>), dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += h[i];
    c /= 256;
    const double n = (double)iters * RT * CT;
    printf("%s RT=%d CT=%d: %.1f memtime-ticks/MFMA (100 MHz ticks x clock ratio), %.1f ns/MFMA -> %.0f TFLOP/s chip\n", name, RT, CT,
           c / n, ms * 1e6 / n, 32768.0 * n * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    u32x4* g; float* out; unsigned long long* cyc;
    hipMalloc(&g, 256 * 32 * 16); hipMemset(g, 0x3c, 256 * 32 * 16);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<1, 2>(g, out, cyc, "2 accumulators ");
    run<2, 4>(g, out, cyc, "8 accumulators ");
    run<2, 8>(g, out, cyc, "16 accumulators");
    run<2, 9>(g, out, cyc, "18 accumulators");
    return 0;
}
