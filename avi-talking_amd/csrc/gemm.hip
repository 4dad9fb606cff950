// Batched strided GEMM on the bf16 matrix cores with fp32 operands in HBM (gfx950).
//
//   C[z][m][n] = affine(act(sum_k A[z][m*lda+k] * W[z][n][k] + bias[n])) + R[z][m][n]
//
// * One workgroup = 256 threads = 4 waves (2 x 2), output tile 128 or 64 (m) x BN = 128 or 64 (n), K step 64; the
//   smaller shapes serve grids that would not fill the chip (avi_gemm / launch_gemm below).
// * A (activations, fp32) is staged through registers: 16-B global loads, split on the fly into
//   bf16 hi (+ lo) parts, written to LDS as 128-B rows with the (row&7)<<4 XOR swizzle so the
//   ds_read_b128 fragment reads are bank-conflict free (cdna_hip_programming.md T2).
// * W (weights) is pre-split bf16 hi/lo [N_pad][K]; same LDS image.
// * MFMA v_mfma_f32_16x16x32_bf16 with the operands swapped (W as the "A" operand, activations as
//   "B"), so each lane's 4 accumulator registers are 4 consecutive n of one row m and the epilogue
//   stores 16 B per lane.
// * prec = BF16X3: three MFMAs per product (hi*hi + hi*lo + lo*hi), fp32 accumulate: ~1e-5
//   relative error, the mode the <=1e-3 parity gate runs in.  prec = BF16: one MFMA.
// * Next K tile's global loads are issued before the current tile's MFMAs (register prefetch,
//   T14 split), one LDS buffer, two barriers per K tile.
// * blockIdx -> tile map is XCD-aware: the N tiles that share an A panel run on one XCD (T1).
#include <cstdlib>

#include "common.h"

namespace {

constexpr int BM = 128;   // rows per tile of the standard instantiation; gemm_kernel<..., 64> halves it
constexpr int BK = 64;
constexpr int ROWB = BK * 2;  // bytes per LDS row (bf16)

template <int BN, int NS, int BMT = BM>
struct GemmSmem {
    static constexpr int A_BYTES = BMT * ROWB;
    static constexpr int W_BYTES = BN * ROWB;
    static constexpr int TOTAL = NS * (A_BYTES + W_BYTES);
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }

template <int BN, int NS, int BMT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const AviGemm g, const int tilesM, const int tilesN) {
    constexpr int AI = BMT / 32;     // staging rows per thread (row r0 + 32 i)
    constexpr int WROWS = BMT / 2;   // rows of a wave (2 x 2 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA0 = smem;
    char* const sA1 = smem + GemmSmem<BN, NS, BMT>::A_BYTES;
    char* const sW0 = smem + NS * GemmSmem<BN, NS, BMT>::A_BYTES;
    char* const sW1 = sW0 + GemmSmem<BN, NS, BMT>::W_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware, bijective block -> tile map (blocks b and b+8 share an XCD)
    const int nwg = tilesM * tilesN;
    int t = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = t / tilesN, tn = t - tm * tilesN;
    const int m0 = tm * BMT, n0 = tn * BN;

    const int z = blockIdx.y;
    const int zo = z / g.z_inner, zi = z - zo * g.z_inner;
    const float* __restrict__ A = g.A + zo * g.sAo + zi * g.sAi;
    const uint16_t* __restrict__ Whi = g.Whi + zo * g.sWo + zi * g.sWi;
    const uint16_t* __restrict__ Wlo = (NS == 2) ? g.Wlo + zo * g.sWo + zi * g.sWi : nullptr;

    // ---- staging assignment: thread -> (row r0 + 32 i, 16-B chunk c) of a 64-wide K slice
    const int r0 = tid >> 3, c = tid & 7;
    const float* aptr[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        int m = m0 + r0 + 32 * i;
        m = m < g.M ? m : g.M - 1;  // clamp: rows past M are computed and discarded
        aptr[i] = A + (long long)m * g.lda + c * 8;
    }
    constexpr int WI = BN / 32;
    const uint16_t* whp[WI];
    const uint16_t* wlp[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const long long off = (long long)(n0 + r0 + 32 * i) * (g.ldw ? g.ldw : g.K) + c * 8;
        whp[i] = Whi + off;
        wlp[i] = (NS == 2) ? Wlo + off : nullptr;
    }

    f32x4 ra[AI][2];
    u32x4 rwh[WI], rwl[WI];

    auto load_tile = [&](const int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            ra[i][0] = *reinterpret_cast<const f32x4*>(aptr[i] + k0);
            ra[i][1] = *reinterpret_cast<const f32x4*>(aptr[i] + k0 + 4);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            rwh[i] = *reinterpret_cast<const u32x4*>(whp[i] + k0);
            if (NS == 2) rwl[i] = *reinterpret_cast<const u32x4*>(wlp[i] + k0);
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int r = r0 + 32 * i;
            bf16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = ra[i][j >> 2][j & 3];
                hi[j] = (__bf16)xv;
                if (NS == 2) lo[j] = (__bf16)(xv - (float)hi[j]);
            }
            *reinterpret_cast<bf16x8*>(sA0 + swz(r, c)) = hi;
            if (NS == 2) *reinterpret_cast<bf16x8*>(sA1 + swz(r, c)) = lo;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<u32x4*>(sW0 + swz(r, c)) = rwh[i];
            if (NS == 2) *reinterpret_cast<u32x4*>(sW1 + swz(r, c)) = rwl[i];
        }
    };

    constexpr int MT = WROWS / 16;   // 16-row m tiles per wave (64 or 32 rows)
    constexpr int NT = BN / 32;  // 16-col n tiles per wave (BN/2 cols)
    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;

    const int nk = g.K / BK;
    const bool dbg_noload = (g.prec & 0x100) != 0, dbg_nomfma = (g.prec & 0x200) != 0;   // timing diagnostics only
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk && !dbg_noload;
        if (more) load_tile((kt + 1) * BK);
        if (!dbg_nomfma)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ks * 4 + fq;
            bf16x8 xh[MT], xl[MT], wh[NT], wl[NT];
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = wm * WROWS + b * 16 + fr;
                xh[b] = *reinterpret_cast<const bf16x8*>(sA0 + swz(row, ch));
                if (NS == 2) xl[b] = *reinterpret_cast<const bf16x8*>(sA1 + swz(row, ch));
            }
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int row = wn * (BN / 2) + a * 16 + fr;
                wh[a] = *reinterpret_cast<const bf16x8*>(sW0 + swz(row, ch));
                if (NS == 2) wl[a] = *reinterpret_cast<const bf16x8*>(sW1 + swz(row, ch));
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
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds C[m = m0 + wm*WROWS + b*16 + fr][n = n0 + wn*BN/2 + a*16 + fq*4 + 0..3]
    float* __restrict__ C = g.C ? g.C + zo * g.sCo + zi * g.sCi : nullptr;
    uint16_t* __restrict__ C16 = g.C16 ? g.C16 + zo * g.sCo + zi * g.sCi : nullptr;   // IEEE half copy of the result
    const float* __restrict__ bias = g.bias ? g.bias + zo * g.sBo + zi * g.sBi : nullptr;
    const float* __restrict__ R = g.R ? g.R + zo * g.sRo + zi * g.sRi : nullptr;
    const bool vec_ok = C && !C16 && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (!R || (((g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0)));
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int n = n0 + wn * (BN / 2) + a * 16 + fq * 4;
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
            const int m = m0 + wm * WROWS + b * 16 + fr;
            if (m >= g.M) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = avi_act(acc[a][b][j] + bv[j], g.act) * sc[j] + sh[j];
            const long long co = (long long)m * g.ldc + n;
            float* cp = C + co;
            if (vec_ok && n + 3 < g.N) {
                if (R) {
                    const float4 rv = *reinterpret_cast<const float4*>(R + (long long)m * g.ldr + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < g.N) {
                        const float y = v[j] + (R ? R[(long long)m * g.ldr + n + j] : 0.f);
                        if (C) cp[j] = y;
                        if (C16) C16[co + j] = __builtin_bit_cast(uint16_t, (_Float16)y);
                    }
            }
        }
    }
}


template <int BN, int NS, int BMT>
int launch_gemm_tile(const AviGemm& g, hipStream_t s) {
    const int tilesM = (g.M + BMT - 1) / BMT, tilesN = (g.N + BN - 1) / BN;
    constexpr int smem = GemmSmem<BN, NS, BMT>::TOTAL;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(gemm_kernel<BN, NS, BMT>), smem);
    dim3 grid(tilesM * tilesN, g.batch, 1);
    hipLaunchKernelGGL((gemm_kernel<BN, NS, BMT>), grid, dim3(256), smem, s, g, tilesM, tilesN);
    return avi_launch_status();
}

// 64 x 64 tiles when 128 x 64 tiles still come to at most two workgroups per CU: such launches are bound by the MFMA
// rate of the workgroups they have, and halving the tile again took 25-40 % off them (squasher 48 -> 29 us, the
// training step 3.5 -> 3.2 ms).  AVI_GEMM_ROWS64 overrides the tile-count threshold (0 = never).
template <int BN, int NS>
int launch_gemm(const AviGemm& g, hipStream_t s) {
    static const int rows64 = [] { const char* e = getenv("AVI_GEMM_ROWS64"); return e ? atoi(e) : 512; }();
    const long long tiles = (long long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.batch;
    // (and whenever the problem has at most 64 rows: the upper half of a 128-row tile would be padding)
    if (BN == 64 && g.M <= 32 && rows64 > 0) return launch_gemm_tile<BN, NS, 32>(g, s);   // the aligner at 32 rows
    if (BN == 64 && (tiles <= rows64 || (g.M <= 64 && rows64 > 0))) return launch_gemm_tile<BN, NS, 64>(g, s);
    return launch_gemm_tile<BN, NS, BM>(g, s);
}

__global__ void pack_split_kernel(const float* __restrict__ W, int N, int K, int N_pad, uint16_t* __restrict__ hi,
                                  uint16_t* __restrict__ lo) {
    const long long total = (long long)N_pad * K;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / K);
        const float x = n < N ? W[i] : 0.f;
        const __bf16 h = (__bf16)x;
        hi[i] = __builtin_bit_cast(uint16_t, h);
        if (lo) {
            const __bf16 l = (__bf16)(x - (float)h);
            lo[i] = __builtin_bit_cast(uint16_t, l);
        }
    }
}

}  // namespace

// gemm_dma.hip: LDS-DMA 256x128 kernel for grids that fill the chip
int avi_gemm_dma_launch(const AviGemm& g, hipStream_t s);
// gemm_pp.hip: 256x256 ping-pong kernel (two wave groups alternate memory and matrix sections)
bool avi_gemm_pp_ok(const AviGemm& g);
int avi_gemm_pp_launch(const AviGemm& g, hipStream_t s);
// gemm_pp192.hip: the same schedule on 128x192 tiles (transformer projections: M = 8000, N = 768 / 2304 / 3072)
bool avi_gemm_pp192_ok(const AviGemm& g);
int avi_gemm_pp192_launch(const AviGemm& g, int bn, hipStream_t s);

// Fraction of the launched tile area that is useful work when `cus` workgroups run per round (1 per CU), times a
// per-tile efficiency (the smaller tile moves more LDS-DMA pieces per MFMA): picks the tile shape for a problem.
static double tile_score(const AviGemm& g, int bm, int bn, double eff) {
    const int cus = g.cus > 0 && g.cus <= 256 ? g.cus : 256;
    const long long tiles = (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.batch;
    const long long rounds = (tiles + cus - 1) / cus;
    return eff * (double)g.M * g.N * g.batch / ((double)rounds * cus * bm * bn);
}

static int gemm_kernel_choice() {   // AVI_GEMM_KERNEL = 2 / 4 / 5 / 6: A/B switches of the plane-operand dispatch below
    const char* e = getenv("AVI_GEMM_KERNEL");   // read per launch (tests switch it inside one process)
    return e ? atoi(e) : 0;
}

extern "C" int avi_gemm(const AviGemm* gp, void* stream) {
    if (!gp) return AVI_EINVAL;
    const AviGemm& g = *gp;
    const bool planes = g.Ahi != nullptr;
    if ((!g.A && !planes) || !g.Whi || (!g.C && !g.Chi && !g.C16) || g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % BK) != 0)
        return AVI_EINVAL;
    if (g.ldw && (g.ldw < g.K || (g.ldw & 7))) return AVI_EINVAL;
    if (g.batch < 1 || g.z_inner < 1 || (g.batch % g.z_inner) != 0 || g.batch > 65535) return AVI_EINVAL;
    if ((g.Chi == nullptr) != (g.Clo == nullptr)) return AVI_EINVAL;
    if (g.Chi && ((g.ldc & 3) || (g.sCo & 3) || (g.sCi & 3) || (reinterpret_cast<uintptr_t>(g.Chi) & 7) ||
                  (reinterpret_cast<uintptr_t>(g.Clo) & 7)))
        return AVI_EINVAL;
    if (g.C16 && (planes || (reinterpret_cast<uintptr_t>(g.C16) & 1))) return AVI_EINVAL;   // fp32-operand kernel only
    if (planes) {
        if ((g.lda & 7) || (g.sAo & 7) || (g.sAi & 7) || (reinterpret_cast<uintptr_t>(g.Ahi) & 15)) return AVI_EINVAL;
        if ((g.prec & 0xff) != AVI_PREC_BF16 && (!g.Alo || (reinterpret_cast<uintptr_t>(g.Alo) & 15))) return AVI_EINVAL;
        if (g.N <= 64) return AVI_EINVAL;          // the LDS-DMA kernel's weight tile is 128 rows
    } else {
        if (g.Chi) return AVI_EINVAL;               // plane output is implemented by the LDS-DMA kernel only
        if ((g.lda & 3) || (g.sAo & 3) || (g.sAi & 3) || (reinterpret_cast<uintptr_t>(g.A) & 15)) return AVI_EINVAL;
    }
    if ((g.sWo & 7) || (g.sWi & 7) || (reinterpret_cast<uintptr_t>(g.Whi) & 15)) return AVI_EINVAL;
    if ((g.prec & 0xff) == AVI_PREC_BF16X3 && (!g.Wlo || (reinterpret_cast<uintptr_t>(g.Wlo) & 15))) return AVI_EINVAL;
    const int prec = g.prec & 0xff;   // bits 8,9: timing diagnostics (skip loads / skip MFMAs), results then invalid
    if (prec != AVI_PREC_BF16 && prec != AVI_PREC_BF16X3 && prec != AVI_PREC_F16X2) return AVI_EINVAL;
    if (prec == AVI_PREC_F16X2 && !planes) return AVI_EINVAL;      // the 2-term fp16 mode exists on the plane-operand kernels
    if ((g.scale == nullptr) != (g.shift == nullptr)) return AVI_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (planes) {   // AVI_GEMM_KERNEL=2 keeps the one-phase LDS-DMA kernel, 4 / 5 force one ping-pong tile shape
        const int choice = gemm_kernel_choice();
        const bool ok256 = avi_gemm_pp_ok(g) && choice != 2 && choice != 5 && choice != 6;
        const bool ok128 = avi_gemm_pp192_ok(g) && choice != 2 && choice != 4;
        // candidates: 256x256 (eff 1.0), 128x256 (0.88), 128x192 (0.85); 5 / 6 force 128x192 / 128x256
        double best = -1.0;
        int pick = 0;
        if (ok256) { best = tile_score(g, 256, 256, 1.0); pick = 1; }
        if (ok128 && choice != 6) { const double v = tile_score(g, 128, 192, 0.85); if (v > best) { best = v; pick = 2; } }
        if (ok128 && choice != 5) { const double v = tile_score(g, 128, 256, 0.88); if (v > best) { best = v; pick = 3; } }
        if (pick == 1) return avi_gemm_pp_launch(g, s);
        if (pick == 2) return avi_gemm_pp192_launch(g, 192, s);
        if (pick == 3) return avi_gemm_pp192_launch(g, 256, s);
        if (prec == AVI_PREC_F16X2) return AVI_EINVAL;             // K does not tile for the ping-pong kernels
        return avi_gemm_dma_launch(g, s);
    }
    // 64-column tiles for narrow outputs and for grids of up to two 128-column workgroups per CU: these launches are
    // bound by the MFMA rate of the workgroups they have (0.75 us per 128x128x64 step on one CU), so twice the
    // workgroups of half the width finish 15-35 % sooner (decoder head, time-embedding MLP, the training step's GEMMs)
    const long long tiles128 = (long long)((g.M + BM - 1) / BM) * ((g.N + 127) / 128) * g.batch;
    const bool narrow = g.N <= 64 || tiles128 <= 512;
    if (prec == AVI_PREC_BF16X3) return narrow ? launch_gemm<64, 2>(g, s) : launch_gemm<128, 2>(g, s);
    return narrow ? launch_gemm<64, 1>(g, s) : launch_gemm<128, 1>(g, s);
}

extern "C" int avi_pack_weight_split(const float* W, int N, int K, int N_pad, uint16_t* hi, uint16_t* lo,
                                     void* stream) {
    if (!W || !hi || N <= 0 || K <= 0 || N_pad < N) return AVI_EINVAL;
    const long long total = (long long)N_pad * K;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_split_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), W, N, K,
                       N_pad, hi, lo);
    return avi_launch_status();
}
