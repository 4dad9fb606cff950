// GELU from a piecewise-cubic table kept in LDS (scripts/gen_gelu_table.py writes gelu_table.inc).
//
//   gelu(x) = max(x, 0) + e(|x|),   e(u) = -u Phi(-u): smooth, even in x, < 7e-9 beyond u = 6
//
// e is tabulated as one cubic per segment of [0, 6) (128 segments, 2 KiB; 2.4e-7 against float64 = fp32 rounding; the first
// segment passes through the origin exactly, so tiny inputs keep their RELATIVE accuracy, gelu(x) -> x / 2 to 5e-7):
// 10 vector instructions + one 16-byte LDS gather per value, against 22 for the rational erf of common.h.  Kernels whose
// time is the vector pipe's (128 outputs per lane in the 256 x 256 GEMM's epilogue; conv layer 0, 524 M outputs at ~37
// instructions each) copy the table into LDS once per workgroup and call avi_gelu_lds.
// A NaN input does not propagate through max(x, 0): it yields e(6) (inputs are sums of finite products here).
#pragma once
#include "common.h"

static __device__ __attribute__((aligned(16))) const float avi_gelu_tab[512] = {
#include "gelu_table.inc"
};
constexpr int AVI_GELU_TAB_BYTES = 2048;

// tab: the table's copy in LDS (16-byte aligned)
__device__ __forceinline__ float avi_gelu_lds(float x, const char* tab) {
    const float t = __builtin_fminf(__builtin_fabsf(x) * (128.f / 6.f), 127.99999f);
    const int i = (int)t;
    const float f = __builtin_amdgcn_fractf(t);
    const f32x4 c = *reinterpret_cast<const f32x4*>(tab + (i << 4));
    float r = fmaf(c[3], f, c[2]);
    r = fmaf(r, f, c[1]);
    r = fmaf(r, f, c[0]);
    return __builtin_fmaxf(x, 0.f) + r;
}
