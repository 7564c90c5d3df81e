// Shared host/device helpers for libaas_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/aas_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

void aas_set_error(const char* fmt, ...);
int aas_debug_flags_value();
int aas_precision_value();

#define AAS_CHECK(cond, ...)            \
    do {                                \
        if (!(cond)) {                  \
            aas_set_error(__VA_ARGS__); \
            return 1;                   \
        }                               \
    } while (0)

#define AAS_LAUNCH_CHECK(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            aas_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                       \
        }                                                                   \
    } while (0)

#define AAS_HIP(call)                                                       \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) {                                            \
            aas_set_error("%s failed: %s", #call, hipGetErrorString(e__));  \
            return 2;                                                       \
        }                                                                   \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// tanh with full fp32 accuracy (tanhf from ocml)
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
