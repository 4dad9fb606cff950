// Shared device helpers for libavi_talking_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/avi_talking.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define AVI_WAVE 64

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property of a kernel: one grant per (kernel, device),
// taken lock-free on first use from any thread (a process that drives a second GPU, or first calls from two threads,
// must not launch with more than 64 KB of dynamic LDS before the attribute is set on that device).  Two racing threads
// both set the attribute (idempotent for a fixed size); kernels whose size varies request their maximum.
struct AviLdsGrant {
    static constexpr int MAX_DEV = 16;
    std::atomic<int> bytes[MAX_DEV];
    AviLdsGrant() { for (auto& b : bytes) b.store(0, std::memory_order_relaxed); }
    void ensure(const void* fn, int need) {
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEV;
        if (known && bytes[dev].load(std::memory_order_acquire) >= need) return;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, need);
        if (!known) return;
        int cur = bytes[dev].load(std::memory_order_relaxed);
        while (cur < need && !bytes[dev].compare_exchange_weak(cur, need, std::memory_order_release)) {}
    }
};

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

// Split-plane activation formats (x = hi + lo): two bf16 (AVI_PLANES_BF16, operands of the 3-term bf16 GEMM) or two fp16
// (AVI_PLANES_F16, operands of the opt-in 2-term fp16 GEMM AVI_PREC_F16X2).  Returns the 16-bit patterns.
__device__ __forceinline__ void avi_split_hl(float v, int fmt, uint16_t& h, uint16_t& l) {
    if (fmt == AVI_PLANES_F16) {
        const _Float16 hh = (_Float16)v;
        h = __builtin_bit_cast(uint16_t, hh);
        l = __builtin_bit_cast(uint16_t, (_Float16)(v - (float)hh));
    } else {
        const __bf16 hb = (__bf16)v;
        h = __builtin_bit_cast(uint16_t, hb);
        l = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)hb));
    }
}

// ---- fp16-plane range guard (avi_talking.h "Status words").  A producer of AVI_PLANES_F16 planes feeds every value it
// splits to see() (two vector instructions: and, max on the bit pattern - NaN > inf > finite as unsigned) and calls commit()
// once per wave where the wave has reconverged: a wave-wide max, then at most two system-scope stores, and only when
// something is wrong.  status == nullptr: commit() does nothing.
unsigned* avi_status_ptr();                        // api.hip: the process-wide status word (host side)
int avi_fault_injected();                          // api.hip: AVI_FAULT_* bits set by avi_debug_fault_inject
struct AviF16Range {
    unsigned m = 0;
    __device__ __forceinline__ void see(float v) {
        const unsigned b = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
        m = b > m ? b : m;
    }
    __device__ __forceinline__ void commit(unsigned* __restrict__ status) {
        if (!status) return;
        unsigned w = m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned other = (unsigned)__shfl_xor((int)w, o, 64);
            w = other > w ? other : w;
        }
        if ((threadIdx.x & 63) != 0) return;
        if (w >= 0x477FF000u)                          // 65520.0f: rounds to inf in fp16 (NaN / inf patterns are larger still)
            __hip_atomic_store(status + AVI_STATUS_F16_OVERFLOW, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (w != 0u && w < 0x39800000u)                // 2^-12
            __hip_atomic_store(status + AVI_STATUS_F16_TINY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
};

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
        case AVI_ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        default: return x;
    }
}

// Wave-wide sum / max over DPP lanes (quad xor 1, xor 2, half-row mirror, row mirror, row broadcasts), result taken
// from lane 63 with v_readlane: ~7 vector instructions instead of six ds_bpermute round trips (~100 cycles each).
// The result is wave-uniform (an SGPR), which the small latency-bound phases of the sampler rely on.
template <bool MAX>
__device__ __forceinline__ float wave_reduce_dpp(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
    const float id = MAX ? -3.0e38f : 0.f;
#define AVI_DPP(x, ctrl, rm) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, id), \
                                     __builtin_bit_cast(int, x), ctrl, rm, 0xF, false))
    v = op(v, AVI_DPP(v, 0xB1, 0xF));    // quad_perm [1,0,3,2]
    v = op(v, AVI_DPP(v, 0x4E, 0xF));    // quad_perm [2,3,0,1]
    v = op(v, AVI_DPP(v, 0x141, 0xF));   // row_half_mirror
    v = op(v, AVI_DPP(v, 0x140, 0xF));   // row_mirror: every lane of a 16-lane row holds the row's result
    v = op(v, AVI_DPP(v, 0x142, 0xA));   // row_bcast15 into rows 1 and 3
    v = op(v, AVI_DPP(v, 0x143, 0xC));   // row_bcast31 into rows 2 and 3: row 3 holds the wave's result
#undef AVI_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_u(float v) { return wave_reduce_dpp<false>(v); }
__device__ __forceinline__ float wave_max_u(float v) { return wave_reduce_dpp<true>(v); }

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
// Sum / max over a 256-thread workgroup (4 waves) through 4 LDS words; every thread gets the result.  `slot` tells
// consecutive reductions of one kernel apart (no barrier is needed between them: each uses its own 4 words).
template <bool MAX>
__device__ __forceinline__ float block256_reduce(float v, float (*red)[4], int slot) {
    v = MAX ? wave_max(v) : wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[slot][threadIdx.x >> 6] = v;
    __syncthreads();
    return MAX ? fmaxf(fmaxf(red[slot][0], red[slot][1]), fmaxf(red[slot][2], red[slot][3]))
               : (red[slot][0] + red[slot][1]) + (red[slot][2] + red[slot][3]);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
