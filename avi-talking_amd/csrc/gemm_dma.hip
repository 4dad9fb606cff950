// Second-generation GEMM for the large contractions of the audio path: same contract as gemm.hip
// (C = affine(act(A.W^T + bias)) + R, fp32 activations in HBM, bf16 hi/lo weight planes), built around
// LDS-DMA instead of register staging:
//
//   * tile 256 (m) x 128 (n), K step 32, one workgroup of 512 threads = 8 waves (4 x 2, 64x64 each) per CU;
//   * BOTH operands travel global -> LDS with `global_load_lds_dwordx4` (no VGPR hop, no ds_write): the fp32
//     activation tile lands raw (256 rows x 128 B), the weight tile as 128 rows x (hi 64 B | lo 64 B);
//   * three LDS stages (3 x 48 KiB): the DMA of K step k+2 is issued before the MFMAs of step k, waits are
//     COUNTED (`s_waitcnt vmcnt(6)`: one stage stays in flight across the barrier) and there is ONE raw
//     s_barrier per K step (cdna_hip_programming.md section 5, "Pipelining across barriers");
//   * LDS-DMA writes lane-linear, so the bank-conflict swizzle lives on the per-lane SOURCE address and the
//     same involution is applied when reading (rule 21): activation rows use c ^ (((row>>1)&3)*2 | (row>>3)),
//     weight rows use c ^ (row&7); both were checked conflict-free for ds_read_b128's 16-lane groups;
//   * the fp32 -> bf16 hi/lo split of the activation happens when the MFMA fragment is built (8 floats per
//     lane), under the matrix pipe of the SIMD's other wave.
// Selected by avi_gemm when the grid fills the chip (see launch heuristics there); gemm.hip's register-staged
// 128x128 kernel serves narrow N and small grids.
#include "common.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 32, NTHR = 512, STAGES = 3;
constexpr int A_BYTES = BM * BK * 4;        // 32 KiB: rows of 128 B (32 fp32)
constexpr int W_BYTES = BN * 128;           // 16 KiB: rows of 128 B (32 bf16 hi | 32 bf16 lo)
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int SMEM_BYTES = STAGES * STAGE_BYTES;   // 147456

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ int fA(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }
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
    const float* __restrict__ A = g.A + zo * g.sAo + zi * g.sAi;
    const uint16_t* __restrict__ Whi = g.Whi + zo * g.sWo + zi * g.sWi;
    const uint16_t* __restrict__ Wlo = (NS == 2) ? g.Wlo + zo * g.sWo + zi * g.sWi : nullptr;

    // ---- DMA source pointers.  A: 32 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces w, w+8, w+16, w+24.
    //      lane -> (row r = 8*piece + lane/8, LDS chunk c' = lane%8) holds source chunk c = c' ^ fA(r).
    const char* asrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        int m = m0 + r;
        m = m < g.M ? m : g.M - 1;
        const int c = (lane & 7) ^ fA(r & 15);
        asrc[i] = reinterpret_cast<const char*>(A + (long long)m * g.lda) + c * 16;
    }
    //      W: 16 pieces; wave w issues pieces w, w+8.  chunk c < 4: hi plane k 8c..8c+7; c >= 4: lo plane.
    const char* wsrc[2];
    bool wact[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ fW(r & 15);
        const uint16_t* base = (c < 4 || NS == 1) ? Whi : Wlo;
        wsrc[i] = reinterpret_cast<const char*>(base + (long long)(n0 + r) * g.K) + (c & 3) * 16;
        wact[i] = (NS == 2) || (c < 4);
    }

    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sw = sa + A_BYTES;
        const long long kofsA = (long long)kt * BK * 4, kofsW = (long long)kt * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(asrc[i] + kofsA, sa + (i * 8 + wave) * 1024);
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
        if (kt + 2 < nk) {
            int nb = buf + 2;
            nb = nb >= STAGES ? nb - STAGES : nb;
            issue(kt + 2, nb);            // overwrites the buffer every wave finished reading before this barrier
        }
        const char* sa = smem + buf * STAGE_BYTES;
        const char* sw = sa + A_BYTES;
        bf16x8 xh[MT], xl[MT], wh[NT], wl[NT];
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int row = wm * 64 + b * 16 + fr;
            const int sz = fA(row & 15);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sa + row * 128 + (((2 * fq) ^ sz) << 4));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sa + row * 128 + (((2 * fq + 1) ^ sz) << 4));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = j < 4 ? v0[j] : v1[j - 4];
                const __bf16 hi = (__bf16)xv;
                xh[b][j] = hi;
                if (NS == 2) xl[b][j] = (__bf16)(xv - (float)hi);
            }
        }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int row = wn * 64 + a * 16 + fr;
            const int sz = fW(row & 15);
            wh[a] = *reinterpret_cast<const bf16x8*>(sw + row * 128 + ((fq ^ sz) << 4));
            if (NS == 2) wl[a] = *reinterpret_cast<const bf16x8*>(sw + row * 128 + (((4 + fq) ^ sz) << 4));
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                if (NS == 2) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[a], xh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[a], xl[b], acc[a][b], 0, 0, 0);
                }
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[a], xh[b], acc[a][b], 0, 0, 0);
            }
        buf = buf + 1 >= STAGES ? 0 : buf + 1;
    }

    // ---- epilogue (same contract as gemm.hip): lane holds C[m][n .. n+3]
    float* __restrict__ C = g.C + zo * g.sCo + zi * g.sCi;
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
            float* cp = C + (long long)m * g.ldc + n;
            if (vec_ok && n + 3 < g.N) {
                if (R) {
                    const float4 rv = *reinterpret_cast<const float4*>(R + (long long)m * g.ldr + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < g.N) cp[j] = v[j] + (R ? R[(long long)m * g.ldr + n + j] : 0.f);
            }
        }
    }
}

template <int NS>
int launch(const AviGemm& g, hipStream_t s) {
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<NS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_dma_kernel<NS>), dim3(tilesM * tilesN, g.batch), dim3(NTHR), SMEM_BYTES, s, g, tilesM,
                       tilesN);
    return avi_launch_status();
}

}  // namespace

// Called by avi_gemm (gemm.hip) after argument validation.
int avi_gemm_dma_launch(const AviGemm& g, hipStream_t s) {
    return (g.prec & 0xff) == AVI_PREC_BF16X3 ? launch<2>(g, s) : launch<1>(g, s);
}
