// Second-generation GEMM for the large contractions of the audio path: same contract as gemm.hip
// (C = affine(act(A.W^T + bias)) + R, bf16 hi/lo weight planes) for activations stored as SPLIT PLANES
// (AviGemm.Ahi/Alo: x = hi + lo, two bf16 planes = the 4 bytes/element of fp32), built around LDS-DMA instead of
// register staging: the producer of an activation splits it ONCE in its epilogue and the GEMM loop is
// ds_read + MFMA only (the register-staged kernel spends 2.4 vector instructions per MFMA on fp32->bf16 splitting,
// ds_write and address arithmetic).  Measured: 300 vs 282 TFLOP/s algorithmic on the conv1 shape in bf16x3, slower
// than gemm.hip in single-pass bf16; end to end the two tie, so the host keeps fp32 activations by default
// (AVI_W2V_PLANES=1 selects this path).  Structure:
//
//   * tile 256 (m) x 128 (n), K step 32, one workgroup of 512 threads = 8 waves (4 x 2, 64x64 each) per CU;
//   * BOTH operands travel global -> LDS with `global_load_lds_dwordx4` (no VGPR hop, no ds_write, no conversion):
//     activation and weight tiles land as rows of (hi 64 B | lo 64 B);
//   * three LDS stages (3 x 48 KiB): the DMA of K step k+2 is issued before the MFMAs of step k, waits are
//     COUNTED (`s_waitcnt vmcnt(6)`: one stage stays in flight across the barrier) and there is ONE raw
//     s_barrier per K step (cdna_hip_programming.md section 5, "Pipelining across barriers");
//   * LDS-DMA writes lane-linear, so the bank-conflict swizzle lives on the per-lane SOURCE address and the
//     same involution is applied when reading (rule 21): chunk c of a row sits at c ^ (row&7), checked
//     conflict-free for ds_read_b128's 16-lane groups (hi chunk g and lo chunk 4+g);
//   * the epilogue can emit the result as split planes too (AviGemm.Chi/Clo), so conv -> conv chains never pass
//     through fp32.
// Selected by avi_gemm whenever Ahi/Alo are given; gemm.hip's register-staged 128x128 kernel serves fp32 A.
#include "common.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 32, NTHR = 512, STAGES = 3;
constexpr int A_BYTES = BM * 128;           // 32 KiB: rows of 128 B (32 bf16 hi | 32 bf16 lo)
constexpr int W_BYTES = BN * 128;           // 16 KiB: rows of 128 B (32 bf16 hi | 32 bf16 lo)
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int SMEM_BYTES = STAGES * STAGE_BYTES;   // 147456

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ int fW(int row) { return row & 7; }

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    // one 1-KiB piece per wave-instruction: LDS destination = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

template <int NS>
__global__ __launch_bounds__(NTHR, 2) void gemm_dma_kernel(const AviGemm g, const int tilesM, const int tilesN) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = tilesM * tilesN;
    int t = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = t / tilesN, tn = t - tm * tilesN;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = blockIdx.y;
    const int zo = z / g.z_inner, zi = z - zo * g.z_inner;
    const uint16_t* __restrict__ Ahi = g.Ahi + zo * g.sAo + zi * g.sAi;
    const uint16_t* __restrict__ Alo = (NS == 2) ? g.Alo + zo * g.sAo + zi * g.sAi : nullptr;
    const uint16_t* __restrict__ Whi = g.Whi + zo * g.sWo + zi * g.sWi;
    const uint16_t* __restrict__ Wlo = (NS == 2) ? g.Wlo + zo * g.sWo + zi * g.sWi : nullptr;

    // ---- DMA source pointers.  A: 32 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces w, w+8, w+16, w+24.
    //      lane -> (row r = 8*piece + lane/8, LDS chunk c' = lane%8) holds source chunk c = c' ^ fA(r).
    const char* asrc[4];
    bool aact[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        int m = m0 + r;
        m = m < g.M ? m : g.M - 1;
        const int c = (lane & 7) ^ fW(r & 15);
        const uint16_t* base = (c < 4 || NS == 1) ? Ahi : Alo;
        asrc[i] = reinterpret_cast<const char*>(base + (long long)m * g.lda) + (c & 3) * 16;
        aact[i] = (NS == 2) || (c < 4);
    }
    //      W: 16 pieces; wave w issues pieces w, w+8.  chunk c < 4: hi plane k 8c..8c+7; c >= 4: lo plane.
    const char* wsrc[2];
    bool wact[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ fW(r & 15);
        const uint16_t* base = (c < 4 || NS == 1) ? Whi : Wlo;
        wsrc[i] = reinterpret_cast<const char*>(base + (long long)(n0 + r) * (g.ldw ? g.ldw : g.K)) + (c & 3) * 16;
        wact[i] = (NS == 2) || (c < 4);
    }

    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sw = sa + A_BYTES;
        const long long kofsA = (long long)kt * BK * 2, kofsW = (long long)kt * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (aact[i]) glds16(asrc[i] + kofsA, sa + (i * 8 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (wact[i]) glds16(wsrc[i] + kofsW, sw + (i * 8 + wave) * 1024);
    };

    constexpr int MT = 4, NT = 4;
    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    const int nk = g.K / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most the 6 DMA instructions of stage kt+1 are still outstanding
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // The 6 DMA pieces of stage kt+2 are issued BETWEEN groups of MFMAs (one piece costs 100-185 issue cycles next
        // to LDS reads; all six up front left the matrix pipe idle ~0.8 us per step on every wave at once).
        int nb = buf + 2;
        nb = nb >= STAGES ? nb - STAGES : nb;
        const bool pre = kt + 2 < nk;
        char* nsa = smem + nb * STAGE_BYTES;
        char* nsw = nsa + A_BYTES;
        const long long kofs = (long long)(kt + 2) * BK * 2;
        const char* sa = smem + buf * STAGE_BYTES;
        const char* sw = sa + A_BYTES;
        bf16x8 xh[MT], xl[MT], wh[NT], wl[NT];
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int row = wm * 64 + b * 16 + fr;
            const int sz = fW(row & 15);
            xh[b] = *reinterpret_cast<const bf16x8*>(sa + row * 128 + ((fq ^ sz) << 4));
            if (NS == 2) xl[b] = *reinterpret_cast<const bf16x8*>(sa + row * 128 + (((4 + fq) ^ sz) << 4));
        }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int row = wn * 64 + a * 16 + fr;
            const int sz = fW(row & 15);
            wh[a] = *reinterpret_cast<const bf16x8*>(sw + row * 128 + ((fq ^ sz) << 4));
            if (NS == 2) wl[a] = *reinterpret_cast<const bf16x8*>(sw + row * 128 + (((4 + fq) ^ sz) << 4));
        }
        // 48 (x3) or 16 (x1) MFMAs in 6 groups, one DMA piece after each group
        auto piece = [&](int p) __attribute__((always_inline)) {
            if (!pre) return;
            if (p < 4) {
                if (aact[p]) glds16(asrc[p] + kofs, nsa + (p * 8 + wave) * 1024);
            } else {
                if (wact[p - 4]) glds16(wsrc[p - 4] + kofs, nsw + ((p - 4) * 8 + wave) * 1024);
            }
        };
        constexpr int TERMS = NS == 2 ? 3 : 1;
        constexpr int TOTAL = TERMS * NT * MT;          // 48 or 16
        constexpr int PER = (TOTAL + 5) / 6;            // MFMAs per group
#pragma unroll
        for (int i = 0; i < TOTAL; ++i) {
            const int term = i / (NT * MT), ab = i - term * NT * MT, a = ab / MT, b = ab - a * MT;
            if (NS == 2 && term == 0)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[a], xh[b], acc[a][b], 0, 0, 0);
            else if (NS == 2 && term == 1)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[a], xl[b], acc[a][b], 0, 0, 0);
            else
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[a], xh[b], acc[a][b], 0, 0, 0);
            if ((i + 1) % PER == 0 && (i + 1) / PER <= 6) {
                __builtin_amdgcn_sched_barrier(0);
                piece((i + 1) / PER - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (TOTAL / PER < 6) {   // groups did not cover all six pieces (x1: 16 MFMAs, PER = 3 -> 5 groups)
#pragma unroll
            for (int p = TOTAL / PER; p < 6; ++p) piece(p);
        }
        buf = buf + 1 >= STAGES ? 0 : buf + 1;
    }

    // ---- epilogue (same contract as gemm.hip): lane holds C[m][n .. n+3]
    float* __restrict__ C = g.C ? g.C + zo * g.sCo + zi * g.sCi : nullptr;
    uint16_t* __restrict__ Chi = g.Chi ? g.Chi + zo * g.sCo + zi * g.sCi : nullptr;
    uint16_t* __restrict__ Clo = g.Chi ? g.Clo + zo * g.sCo + zi * g.sCi : nullptr;
    const float* __restrict__ bias = g.bias ? g.bias + zo * g.sBo + zi * g.sBi : nullptr;
    const float* __restrict__ R = g.R ? g.R + zo * g.sRo + zi * g.sRi : nullptr;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (!R || (((g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0)));
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int n = n0 + wn * 64 + a * 16 + fq * 4;
        if (n >= g.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < g.N) {
                if (bias) bv[j] = bias[n + j];
                if (g.scale) { sc[j] = g.scale[n + j]; sh[j] = g.shift[n + j]; }
            }
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int m = m0 + wm * 64 + b * 16 + fr;
            if (m >= g.M) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = avi_act(acc[a][b][j] + bv[j], g.act) * sc[j] + sh[j];
            if (vec_ok && n + 3 < g.N) {
                if (R) {
                    const float4 rv = *reinterpret_cast<const float4*>(R + (long long)m * g.ldr + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (C) *reinterpret_cast<float4*>(C + (long long)m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                if (Chi) {   // split once here so the consumer GEMM never converts (ldc % 4 == 0: 8-byte stores)
                    uint16_t h[4], l[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const __bf16 hb = (__bf16)v[j];
                        const __bf16 lb = (__bf16)(v[j] - (float)hb);
                        h[j] = __builtin_bit_cast(uint16_t, hb);
                        l[j] = __builtin_bit_cast(uint16_t, lb);
                    }
                    const long long o = (long long)m * g.ldc + n;
                    *reinterpret_cast<uint2*>(Chi + o) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
                    *reinterpret_cast<uint2*>(Clo + o) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < g.N) {
                        const float y = v[j] + (R ? R[(long long)m * g.ldr + n + j] : 0.f);
                        if (C) C[(long long)m * g.ldc + n + j] = y;
                        if (Chi) {
                            const __bf16 hb = (__bf16)y;
                            Chi[(long long)m * g.ldc + n + j] = __builtin_bit_cast(uint16_t, hb);
                            Clo[(long long)m * g.ldc + n + j] = __builtin_bit_cast(uint16_t, (__bf16)(y - (float)hb));
                        }
                    }
            }
        }
    }
}

template <int NS>
int launch(const AviGemm& g, hipStream_t s) {
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(gemm_dma_kernel<NS>), SMEM_BYTES);
    hipLaunchKernelGGL((gemm_dma_kernel<NS>), dim3(tilesM * tilesN, g.batch), dim3(NTHR), SMEM_BYTES, s, g, tilesM,
                       tilesN);
    return avi_launch_status();
}

}  // namespace

// Called by avi_gemm (gemm.hip) after argument validation.
int avi_gemm_dma_launch(const AviGemm& g, hipStream_t s) {
    return (g.prec & 0xff) == AVI_PREC_BF16X3 ? launch<2>(g, s) : launch<1>(g, s);
}
