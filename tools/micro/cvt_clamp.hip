// What does the VOP3 clamp bit do on v_cvt_pk_bf16_f32 (gfx950)?  Prints input -> output pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void k(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(in[i]), "v"(in[i]));
    out[i] = r;
}
int main() {
    float h[] = {-1.5f, -0.0f, 0.0f, 0.3f, 0.998f, 0.9999f, 1.0f, 1.5f, 300.0f, 1e-30f, -1e-30f, 3e-39f, NAN, INFINITY, -INFINITY, 0.00390625f, 7.1e-10f};
    const int n = sizeof(h) / 4;
    float* d; unsigned* o; unsigned ho[32];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4 * n);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, 4 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        unsigned lo = (ho[i] & 0xffff) << 16; float f; memcpy(&f, &lo, 4);
        printf("%14.6g -> 0x%04x (%g)\n", h[i], ho[i] & 0xffff, f);
    }
    return 0;
}
