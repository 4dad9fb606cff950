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

// erf(x) as a branch-free rational x P(x^2) / Q(x^2) on the clamped argument (|error| < 5e-7 over the reals, checked
// against scipy in float32 arithmetic): 17 vector instructions instead of libm erff's two divergent ranges, which
// cost a quarter of the 256x256 GEMM tile time when 128 outputs per lane pass through GELU in the epilogue.
__device__ __forceinline__ float avi_erf(float x) {
    x = __builtin_fminf(__builtin_fmaxf(x, -4.f), 4.f);
    const float x2 = x * x;
    float a = -2.72614225801306e-10f;
    a = fmaf(a, x2, 2.77068142495902e-08f);
    a = fmaf(a, x2, -2.10102402082508e-06f);
    a = fmaf(a, x2, -5.69250639462346e-05f);
    a = fmaf(a, x2, -7.34990630326855e-04f);
    a = fmaf(a, x2, -2.95459980854025e-03f);
    a = fmaf(a, x2, -1.60960333262415e-02f);
    float b = -1.45660718464996e-05f;
    b = fmaf(b, x2, -2.13374055278905e-04f);
    b = fmaf(b, x2, -1.68282697438203e-03f);
    b = fmaf(b, x2, -7.37332916720468e-03f);
    b = fmaf(b, x2, -1.42647390514189e-02f);
    return x * a * __builtin_amdgcn_rcpf(b);
}

__device__ __forceinline__ float avi_gelu(float x) {
    // exact-form GELU: 0.5 x (1 + erf(x / sqrt(2)))  (torch.nn.functional.gelu, approximate='none'); |error| < 1e-6
    return 0.5f * x * (1.0f + avi_erf(x * 0.70710678118654752440f));
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
