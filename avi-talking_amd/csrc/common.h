// Shared device helpers for libavi_talking_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/avi_talking.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define AVI_WAVE 64

static inline int avi_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AVI_OK : (int)e;
}

__device__ __forceinline__ float avi_gelu(float x) {
    // exact GELU: 0.5 x (1 + erf(x / sqrt(2)))  (torch.nn.functional.gelu, approximate='none')
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float avi_act(float x, int act) {
    switch (act) {
        case AVI_ACT_GELU: return avi_gelu(x);
        case AVI_ACT_LRELU02: return x > 0.f ? x : 0.2f * x;
        case AVI_ACT_RELU: return x > 0.f ? x : 0.f;
        case AVI_ACT_SILU: return x / (1.0f + __expf(-x));
        default: return x;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
