// Shared host/device helpers of libnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/nerf_hip.h"

namespace nerf {

void set_error(const char* fmt, ...);

#define NERF_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::nerf::set_error(__VA_ARGS__);     \
            return NERF_E_INVALID;              \
        }                                       \
    } while (0)

#define NERF_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ::nerf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                              __FILE__, __LINE__);                                        \
            return NERF_E_HIP;                                                            \
        }                                                                                 \
    } while (0)

#define NERF_LAUNCH_CHECK() NERF_HIP_CHECK(hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// float32 operations that the compiler must not fuse into FMAs or reassociate: the
// reference evaluates every product and sum as a separately rounded ATen op.
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

constexpr int WAVE = 64;

}  // namespace nerf
