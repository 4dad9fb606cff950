// wav2vec2's positional conv embedding (HF Wav2Vec2PositionalConvEmbedding behind models/lib/wav2vec.py:142-148):
//     h[b][t][:] = x[b][t][:] + gelu(conv1d(x, k = 128, groups = 16, padding = 64)[t] + bias)      (the extra last frame dropped)
// for hidden size 768: per (clip, group) a [T x 6144] x [6144 x 48] product whose left matrix is TOEPLITZ - row t is the flat
// window x[t - 64 .. t + 63][48 ch], so consecutive rows overlap in all but 48 values:
//
//     A[t][k] = xflat[(t - 64) * 48 + k],   k = tap * 48 + ch
//
// As an overlapping-row GEMM on the fp32-operand kernel (gemm.hip, 128 x 64 tiles) every workgroup staged 128 rows x 6144
// values = 3 MB of mostly repeated data and split them in its loop: 0.41 ms at 0.56 PF issued, the least efficient
// matrix-core launch of the pass.  Here a workgroup keeps the 255 x 48 window of its 128 rows RESIDENT in LDS as bf16 hi / lo
// planes (49 KB, split once, zero padding at the clip's ends written in place - no regrouping launch) and reads every A
// fragment straight from it: lane (row r, 8 k g) needs 16 contiguous bytes at (r * 48 + 32 ks + 8 g) * 2, and with a row
// stride of 96 B the 16 rows of a fragment fall on 16 different 16-B slots of the 256-B bank row (slot = 6 r + g mod 16):
// conflict-free without padding.  Only the 48 x 6144 weights stream: chunks of 64 k (two k-steps, 12 KB of hi + lo) through
// a double-buffered LDS ring by LDS-DMA (source-side swizzle), one barrier per chunk.
//   4 waves x (32 rows x 48 columns): 6 accumulator tiles, 18 MFMAs per k-step (3-term bf16 split), 2 workgroups per CU.
#include "common.h"

namespace {

constexpr int CG = 48, TAPS = 128, PAD = 64, KTOT = CG * TAPS;          // 6144
constexpr int BMP = 128, NTHR = 256;
constexpr int WIN_ROWS = BMP + TAPS - 1;                                 // 255 input positions
constexpr int WIN_ELEMS = WIN_ROWS * CG;                                 // 12240
constexpr int WIN_BYTES = WIN_ELEMS * 2;                                 // per plane: 24480 (16-byte multiple)
constexpr int CHUNK_K = 64;                                              // weights per ring stage: 48 rows x 128 B per plane
constexpr int WPL_BYTES = CG * CHUNK_K * 2;                              // 6144
constexpr int STAGE_BYTES = 2 * WPL_BYTES;                               // hi | lo
constexpr int SMEM_BYTES = 2 * WIN_BYTES + 2 * STAGE_BYTES;              // 73536

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__global__ __launch_bounds__(NTHR, 2) void posconv_kernel(const float* __restrict__ x, int B, int T, int C,
                                                           const uint16_t* __restrict__ whi,
                                                           const uint16_t* __restrict__ wlo, int wrows,
                                                           const float* __restrict__ bias, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const win_hi = smem;
    char* const win_lo = smem + WIN_BYTES;
    char* const ring = smem + 2 * WIN_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * BMP, grp = blockIdx.y, b = blockIdx.z;
    const float* xb = x + (long long)b * T * C + grp * CG;

    // ---- weight ring: stage s <- K chunk q.  One wave-instruction = 1 KiB = 8 rows x 128 B of one plane; lane -> row
    //      8 j + lane / 8, LDS chunk lane % 8 <- source chunk (lane % 8) ^ (row & 7).  12 pieces per stage, 3 per wave.
    const uint16_t* wsrc[3];
    int wdst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int piece = wave * 3 + i;                  // 0..5: hi rows 8 piece.., 6..11: lo
        const int pl = piece / 6, j = piece - pl * 6;
        const int row = 8 * j + (lane >> 3), c = (lane & 7) ^ (row & 7);
        wsrc[i] = (pl ? wlo : whi) + ((long long)grp * wrows + row) * KTOT + c * 8;
        wdst[i] = pl * WPL_BYTES + j * 1024;
    }
    auto issue_w = [&](int q, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + (long long)q * CHUNK_K),
                                             (lds_void*)(ring + s * STAGE_BYTES + wdst[i]), 16, 0, 0);
    };
    issue_w(0, 0);

    // ---- the window: positions p = 0..254 <-> frames t0 - 64 + p (zero outside the clip), 12 float4 per position
    for (int i = tid; i < WIN_ROWS * (CG / 4); i += NTHR) {
        const int p = i / (CG / 4), c4 = i - p * (CG / 4);
        const int t = t0 - PAD + p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T) v = *reinterpret_cast<const float4*>(xb + (long long)t * C + c4 * 4);
        uint16_t h[4], l[4];
        avi_split_hl(v.x, AVI_PLANES_BF16, h[0], l[0]);
        avi_split_hl(v.y, AVI_PLANES_BF16, h[1], l[1]);
        avi_split_hl(v.z, AVI_PLANES_BF16, h[2], l[2]);
        avi_split_hl(v.w, AVI_PLANES_BF16, h[3], l[3]);
        const int o = (p * CG + c4 * 4) * 2;
        *reinterpret_cast<uint2*>(win_hi + o) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
        *reinterpret_cast<uint2*>(win_lo + o) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
    }

    // ---- fragments.  A: row (32 wave + 16 mt + fr) of the tile, 8 k at 32 ks + 8 fq of the chunk's k range.
    //      W: row 16 nt + fr of the stage, LDS chunk (4 ks + fq) ^ (row & 7).
    const int fr = lane & 15, fq = lane >> 4;
    const int a_off = ((wave * 32 + fr) * CG + fq * 8) * 2;               // + mt * 16 rows, + k0 * 2
    f32x4 acc[3][2];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NQ = KTOT / CHUNK_K;                                    // 96
    for (int q = 0; q < NQ; ++q) {
        const int s = q & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // my pieces of chunk q have landed ...
        __syncthreads();                                                  // ... and everybody's; stage s ^ 1 is no longer read
        if (q + 1 < NQ) issue_w(q + 1, s ^ 1);
        const char* st = ring + s * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], wh[3], wl[3];
            const int k0 = q * CHUNK_K + ks * 32;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int o = a_off + (mt * 16 * CG + k0) * 2;
                ah[mt] = *reinterpret_cast<const bf16x8*>(win_hi + o);
                al[mt] = *reinterpret_cast<const bf16x8*>(win_lo + o);
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int row = nt * 16 + fr;
                const int o = row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4);
                wh[nt] = *reinterpret_cast<const bf16x8*>(st + o);
                wl[nt] = *reinterpret_cast<const bf16x8*>(st + WPL_BYTES + o);
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 c = acc[nt][mt];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nt], ah[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nt], al[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nt], ah[mt], c, 0, 0, 0);
                    acc[nt][mt] = c;
                }
        }
    }

    // ---- epilogue: lane (fr, fq) holds frame t0 + 32 wave + 16 mt + fr, channels 16 nt + 4 fq .. + 3 of the group
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int t = t0 + wave * 32 + mt * 16 + fr;
        if (t >= T) continue;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int n = nt * 16 + fq * 4;
            const float4 bv = *reinterpret_cast<const float4*>(bias + grp * CG + n);
            const float4 r = *reinterpret_cast<const float4*>(xb + (long long)t * C + n);
            const f32x4 a = acc[nt][mt];
            float4 y;
            y.x = r.x + avi_gelu(a[0] + bv.x);
            y.y = r.y + avi_gelu(a[1] + bv.y);
            y.z = r.z + avi_gelu(a[2] + bv.z);
            y.w = r.w + avi_gelu(a[3] + bv.w);
            *reinterpret_cast<float4*>(out + ((long long)b * T + t) * C + grp * CG + n) = y;
        }
    }
}

}  // namespace

extern "C" int avi_posconv_gelu_residual(const float* x, int B, int T, int C, int groups, int taps, const uint16_t* w_hi,
                                         const uint16_t* w_lo, int rows_per_group, const float* bias, float* out,
                                         void* stream) {
    if (!x || !w_hi || !w_lo || !bias || !out || B <= 0 || T <= 0) return AVI_EINVAL;
    if (taps != TAPS || groups <= 0 || C != groups * CG || rows_per_group < CG || B > 65535 || groups > 65535) return AVI_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias) |
         reinterpret_cast<uintptr_t>(w_hi) | reinterpret_cast<uintptr_t>(w_lo)) & 15)
        return AVI_EINVAL;
    {   // not in place: a workgroup reads a +/-64-frame halo of x that its neighbours are storing as out
        const uintptr_t xa = reinterpret_cast<uintptr_t>(x), oa = reinterpret_cast<uintptr_t>(out);
        const uintptr_t nbytes = (uintptr_t)B * T * C * sizeof(float);
        if (xa < oa + nbytes && oa < xa + nbytes) return AVI_EINVAL;
    }
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(posconv_kernel), SMEM_BYTES);
    hipLaunchKernelGGL(posconv_kernel, dim3((T + BMP - 1) / BMP, groups, B), dim3(NTHR), SMEM_BYTES,
                       static_cast<hipStream_t>(stream), x, B, T, C, w_hi, w_lo, rows_per_group, bias, out);
    return avi_launch_status();
}
