// 128 x 192 tile variant of the ping-pong GEMM (gemm_pp.hip) for the transformer projections of the audio encoder
// (M = 8000 rows, N = 768 / 2304 / 3072: 63 x 4 / 12 / 16 tiles = 0.98 / 2.95 / 3.94 rounds of 256 CUs, where the
// 256 x 256 tile gives 0.38 / 1.13 / 1.5).  Same contract and the same idea - the two waves of a SIMD alternate between
// memory work (ds_read_b128 fragments, LDS-DMA pieces, a counted vmcnt wait) and a matrix section, one s_barrier
// per phase, group A reads -> issue -> MFMAs -> wait and group B (wr = 1) MFMAs -> wait -> reads -> issue inside a barrier
// interval - with the
// schedule re-derived for the smaller tile:
//
//   * 8 waves = 2 (m) x 4 (n), 64 x 48 outputs each (4 x 3 accumulator tiles);
//   * a K tile is staged as THREE areas X0 | X1 | W (64 + 64 + 192 rows of 128 B = 40 KiB), three stages (120 KiB);
//   * two phases per K tile T (phase p = 2T + h, stage T % 3):
//       h = 0   reads X0(T), W(T)     multiplies m-tiles 0,1 x n-tiles 0..2 (18 MFMA in bf16x3)   issues W(T+2)
//       h = 1   reads X1(T)           multiplies m-tiles 2,3 (W fragments stay in registers)      issues X0(T+2), X1(T+2)
//     WAR  an area is restaged >= 2 phases after its last read (W(T-1), X0(T-1) read in phase 2T-2, X1(T-1) in 2T-1;
//          restaged in phases 2T and 2T+1);
//     RAW  every wave's wait for an area sits a barrier ahead of its first reader (group B reads the fragments of phase
//          q+1 at the end of interval q).  Group B waits at the top of its memory work, before that phase's own issues:
//          `vmcnt(4)` before an odd phase's (X0, W of the next K tile have landed; X1 of it and the 3 W pieces of the
//          tile after may fly), `vmcnt(5)` before an even phase's (X1 of this K tile; 3 W + 2 X pieces may fly); group A
//          ends its interval with the same counts, after its issue (derived in front of the K loop).
//   * the K loop is unrolled by three K tiles so every LDS offset is an immediate (K tiles must come in threes:
//     K % 96 == 0 in bf16x3, K % 192 == 0 in bf16); past the end of K the issue slots reload the last K tile.
//   * epilogue: per-wave LDS transposition (64 rows x 48 columns, row stride 208 B), then 16-B stores along rows.
#include <cstdlib>

#include "common.h"
#include "gelu_table.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// NT = n-tiles per wave: 3 -> 128 x 192 tile, 4 -> 128 x 256 tile (189 tiles at M = 8000, N = 768: ONE round when the
// sampler holds 32 of the 256 CUs, where 252 tiles of 128 x 192 need two)
constexpr int BM = 128, NTHR = 512;
constexpr int X_BYTES = 64 * 128;
constexpr int NSTAGE = 3;
template <int NT> struct Geo {
    static constexpr int BN = 64 * NT, W_BYTES = BN * 128;
    static constexpr int STAGE_BYTES = 2 * X_BYTES + W_BYTES;          // 40 / 48 KiB
    static constexpr int EP_STRIDE = 64 * NT + 16;                     // epilogue slab row: 16 NT fp32 + 16 B pad
    static constexpr int EP_SLAB = 64 * EP_STRIDE;
    static constexpr int WORK_BYTES = NSTAGE * STAGE_BYTES;            // 120 / 144 KiB (>= 8 slabs = 104 / 136 KiB)
    static constexpr int SMEM_BYTES = WORK_BYTES + AVI_GELU_TAB_BYTES; // + the GELU table of the epilogue (gelu_table.h)
};
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 4 && N <= 6, "vmcnt literal");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

// F16 (with NS = 2): the opt-in 2-term fp16 mode AVI_PREC_F16X2, as in gemm_pp.hip.
// STREAM-K (sk_total > 0; batch 1): the launch is `gridDim.x` workgroups, one per compute unit the caller can count on,
// and the K loops of ALL tiles - sk_total groups of three K tiles, tile after tile - are cut into gridDim.x equal
// contiguous shares.  A workgroup walks its share: at most one tile tail, whole tiles, one tile head.  A tile computed
// by one workgroup takes the ordinary epilogue; a tile shared by several is finished by whichever contributor arrives
// LAST (no workgroup ever waits for another, so nothing depends on dispatch order or co-residency): every contributor
// stores its 128 x BN fp32 partial to its own slot of the workspace, releases it (agent scope) and bumps the tile's
// counter; the one that reads count - 1 acquires, adds the other slots to its registers, runs the epilogue and puts the
// counter back to zero for the next launch.  M = 8000 on the 224 CUs the sampler leaves gives 189 tiles of 128 x 256
// (N = 768: 84 % of one round) or 567 (N = 2304: 2.53 rounds = 84 % of three): with equal shares every CU computes
// 0.84 / 2.53 tiles' worth instead of 1 / 3.
constexpr int SK_MAXC = 4;     // contributors per tile the workspace has slots for (host: share >= tile K groups / 2)

template <int NS, int NT, bool F16 = false>
__global__ __launch_bounds__(NTHR) void gemm_pp192_kernel(const AviGemm g, const int tilesM, const int tilesN,
                                                          const int sk_total, float* __restrict__ sk_ws,
                                                          int* __restrict__ sk_cnt, unsigned* __restrict__ status) {
    constexpr int BN = Geo<NT>::BN, STAGE_BYTES = Geo<NT>::STAGE_BYTES, EP_STRIDE = Geo<NT>::EP_STRIDE,
                  EP_SLAB = Geo<NT>::EP_SLAB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int sk_last;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nwg = sk_total ? (int)gridDim.x : tilesM * tilesN;
    int wg = blockIdx.x;      // XCD-aware, bijective: workgroups b and b + 8 (one XCD) take neighbouring tiles / shares
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nk_all = g.K / (NS == 2 ? 32 : 64);   // multiple of 3 (checked by the launcher)
    const int G = nk_all / 3;                       // K groups per tile
    int sk_it = sk_total ? (int)((long long)wg * sk_total / nwg) : 0;
    const int it_end = sk_total ? (int)((long long)(wg + 1) * sk_total / nwg) : 1;
  for (;;) {                                        // stream-K: the items of this workgroup's share; else one tile
    int t, kt0, nk;
    if (sk_total) {
        if (sk_it >= it_end) break;
        t = sk_it / G;
        const int g0 = sk_it - t * G, g1 = (g0 + it_end - sk_it) < G ? (g0 + it_end - sk_it) : G;
        kt0 = 3 * g0;
        nk = 3 * (g1 - g0);
        sk_it += g1 - g0;
    } else {
        t = wg;
        kt0 = 0;
        nk = nk_all;
    }
    const int tm = t / tilesN, tn = t - tm * tilesN;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = blockIdx.y;
    const int zo = z / g.z_inner, zi = z - zo * g.z_inner;
    const uint16_t* __restrict__ Ahi = g.Ahi + zo * g.sAo + zi * g.sAi;
    const uint16_t* __restrict__ Alo = (NS == 2) ? g.Alo + zo * g.sAo + zi * g.sAi : Ahi;
    const uint16_t* __restrict__ Whi = g.Whi + zo * g.sWo + zi * g.sWi;
    const uint16_t* __restrict__ Wlo = (NS == 2 && !F16) ? g.Wlo + zo * g.sWo + zi * g.sWi : Whi;

    // ---- LDS-DMA source pointers (1 KiB pieces = 8 rows x 128 B; lane -> row 8 piece + lane/8, source chunk
    //      c = lane%8 ^ (row&7)).  X0 / X1: piece = wave (64 rows: wave-row R/32 owns rows R%32 of its m-half);
    //      W: pieces wave, wave + 8, wave + 16 (192 rows in column order).
    constexpr int KB = NS == 2 ? 64 : 128;   // bytes one K tile advances along a row of a plane
    const long long ldw = g.ldw ? g.ldw : g.K;
    const char* xsrc[2];
    const char* wsrc[NT];
    {
        const int R = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (R & 7);
        const int off = NS == 2 ? (c & 3) * 16 : c * 16;
        const uint16_t* base = (NS == 2 && c >= 4) ? Alo : Ahi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + (R >> 5) * 64 + h * 32 + (R & 31);
            m = m < g.M ? m : g.M - 1;
            xsrc[h] = reinterpret_cast<const char*>(base + (long long)m * g.lda) + off + (long long)kt0 * KB;
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int R = (wave + 8 * i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (R & 7);
        const int off = NS == 2 ? (c & 3) * 16 : c * 16;
        const uint16_t* base = (NS == 2 && c >= 4) ? Wlo : Whi;
        int n = n0 + R;
        n = n < g.N ? n : g.N - 1;
        wsrc[i] = reinterpret_cast<const char*>(base + n * ldw) + off + (long long)kt0 * KB;
    }

    auto issue_w = [&](int T, int stage) __attribute__((always_inline)) {
        const long long kofs = (long long)(T < nk ? T : nk - 1) * KB;
        char* dst = smem + stage * STAGE_BYTES + 2 * X_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < NT; ++i) glds16(wsrc[i] + kofs, dst + i * 8 * 1024);
    };
    auto issue_x = [&](int T, int stage) __attribute__((always_inline)) {
        const long long kofs = (long long)(T < nk ? T : nk - 1) * KB;
        char* dst = smem + stage * STAGE_BYTES + wave * 1024;
        glds16(xsrc[0] + kofs, dst);
        glds16(xsrc[1] + kofs, dst + X_BYTES);
    };

    // ---- fragment addresses: lane (fr, fq) reads row (16-row tile base) + fr, chunks fq and 4 + fq
    const int fr = lane & 15, fq = lane >> 4;
    const int pos = (fq ^ (fr & 7)) << 4;
    const int xoff = (wr * 32 + fr) * 128 + pos;                   // + stage, + h * X_BYTES, + bl * 2048
    const int woff = 2 * X_BYTES + (wc * (16 * NT) + fr) * 128 + pos;   // + stage, + a * 2048
    const char* xhi_p = smem + xoff;
    const char* xlo_p = smem + (xoff ^ 64);
    const char* whi_p = smem + woff;
    const char* wlo_p = smem + (woff ^ 64);

    bf16x8 xh[2], xl[2], wh[NT], wl[NT];
    auto read_x = [&](int stage, int h) __attribute__((always_inline)) {
        const int o = stage * STAGE_BYTES + h * X_BYTES;
#pragma unroll
        for (int bl = 0; bl < 2; ++bl) {
            xh[bl] = *reinterpret_cast<const bf16x8*>(xhi_p + o + bl * 2048);
            xl[bl] = *reinterpret_cast<const bf16x8*>(xlo_p + o + bl * 2048);
        }
    };
    auto read_w = [&](int stage) __attribute__((always_inline)) {
        const int o = stage * STAGE_BYTES;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            wh[a] = *reinterpret_cast<const bf16x8*>(whi_p + o + a * 2048);
            wl[a] = *reinterpret_cast<const bf16x8*>(wlo_p + o + a * 2048);
        }
    };

    f32x4 acc[NT][4];   // [n tile][m tile]
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto multiply = [&](int mh) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int bl = 0; bl < 2; ++bl) {
                f32x4 c = acc[a][mh * 2 + bl];
                if (F16) {
                    const f16x8 wv = __builtin_bit_cast(f16x8, wh[a]);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, __builtin_bit_cast(f16x8, xl[bl]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, __builtin_bit_cast(f16x8, xh[bl]), c, 0, 0, 0);
                } else if (NS == 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[a], xh[bl], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[a], xl[bl], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[a], xh[bl], c, 0, 0, 0);
                } else {   // "hi"/"lo" positions are k 0..31 / 32..63 of the 64-wide K tile
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[a], xh[bl], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[a], xl[bl], c, 0, 0, 0);
                }
                acc[a][mh * 2 + bl] = c;
            }
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: what the steady state has issued before phase 0 (10 pieces per wave)
    issue_w(0, 0);
    issue_x(0, 0);
    issue_w(1, 1);
    issue_x(1, 1);
    wait_vmcnt<NT + 2>();                                // W(0), X0(0), X1(0) have landed
    bar();
    // ONE barrier per phase: group A (wr = 0) runs memory -> matrix inside a barrier interval, group B matrix (on the
    // fragments it read in the previous interval) -> memory, so the two waves of a SIMD alternate (gemm_pp.hip).
    // Fragment reads before the LDS-DMA issue (gemm_pp.hip: the issue blocks the wave on the CU's vector-memory path; the
    // reads' latency then runs under it instead of after it).  Group A keeps its issue in front of its matrix section
    // here (X0 of a K tile is issued only three phases before its first read: issued after the MFMAs it would have one
    // interval to land), and ENDS its interval with the counted wait: group B reads the fragments of phase q+1 at the
    // end of interval q, so group A's wait for them has to sit before barrier q-1 (it used to open interval q, ordered
    // before B's read only by B's MFMAs in between).  Counts: after the even phase's issue X1 of the next K tile and
    // the NT W pieces just issued may fly (NT + 1), after the odd phase's the NT W and 2 X pieces of the tile after
    // (NT + 2); group B waits at the top of its memory section, one phase ahead, with the counts of the header.
    auto mem_even = [&](int Tu, int u) __attribute__((always_inline)) {   // group B: K tile Tu in stage u
        wait_vmcnt<NT + 2>();
        read_w(u);
        read_x(u, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue_w(Tu + 2, u == 0 ? 2 : u - 1);
    };
    auto mem_odd = [&](int Tu, int u) __attribute__((always_inline)) {
        wait_vmcnt<NT + 1>();
        read_x(u, 1);
        __builtin_amdgcn_sched_barrier(0);
        issue_x(Tu + 2, u == 0 ? 2 : u - 1);
    };
    if (wr == 0) {
        for (int T = 0; T < nk; T += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                read_w(u);
                read_x(u, 0);
                __builtin_amdgcn_sched_barrier(0);
                issue_w(T + u + 2, u == 0 ? 2 : u - 1);
                multiply(0);
                wait_vmcnt<NT + 1>();
                bar();
                read_x(u, 1);
                __builtin_amdgcn_sched_barrier(0);
                issue_x(T + u + 2, u == 0 ? 2 : u - 1);
                multiply(1);
                wait_vmcnt<NT + 2>();
                bar();
            }
        }
    } else {
        mem_even(0, 0);
        for (int T = 0; T < nk; T += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                multiply(0);
                mem_odd(T + u, u);
                bar();
                multiply(1);
                mem_even(T + u + 1, u == 2 ? 0 : u + 1);   // even phase of the next K tile (a dummy after the last)
                bar();
            }
        }
    }
    // GELU in the epilogue comes from the same LDS table as in the 256 x 256 kernel (gemm_pp.hip), so that a projection's
    // result does not depend on which tile shape the dispatcher picked for its M and CU budget; fetched here, as the
    // youngest load in front of the drain below (the tail DMAs are still landing: it costs no wait of its own)
    if (g.act == AVI_ACT_GELU && wave < 2)
        glds16(reinterpret_cast<const char*>(avi_gelu_tab) + wave * 1024 + lane * 16,
               smem + Geo<NT>::WORK_BYTES + wave * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();                                               // every wave's tail DMA has landed: the stages are dead

    if (sk_total && nk != nk_all) {                      // ---- stream-K: this workgroup holds a PARTIAL sum of tile t
        // contributors of tile t = the workgroups whose share holds one of its G groups: first .. last, this one is c
        const long long lo = (long long)t * G, hi = lo + G - 1;
        const int w_first = (int)(((lo + 1) * nwg - 1) / sk_total), w_last = (int)(((hi + 1) * nwg - 1) / sk_total);
        const int c = wg - w_first, ncontrib = w_last - w_first + 1;
        constexpr int SLOT = BM * BN;                    // floats per slot; lane-major: [wave][a][b][lane] x f32x4
        f32x4* slot = reinterpret_cast<f32x4*>(sk_ws + ((long long)t * SK_MAXC + c) * SLOT);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) slot[((wave * NT + a) * 4 + b) * 64 + lane] = acc[a][b];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, before the barrier the release follows
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the write-back has completed
            const int old = __hip_atomic_fetch_add(&sk_cnt[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == ncontrib - 1;
            if (last) {
                __hip_atomic_store(&sk_cnt[t], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // this CU's L1 forgets the other slots
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            sk_last = last;
        }
        __syncthreads();
        if (!sk_last) continue;                          // somebody else finishes the tile (wave-uniform: LDS word)
        for (int cc = 0; cc < ncontrib; ++cc) {
            if (cc == c) continue;
            const f32x4* other = reinterpret_cast<const f32x4*>(sk_ws + ((long long)t * SK_MAXC + cc) * SLOT);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += other[((wave * NT + a) * 4 + b) * 64 + lane];
        }
    }

    // ---- epilogue (same contract as gemm.hip) through a per-wave LDS slab: 64 rows x 48 columns
    float* __restrict__ C = g.C ? g.C + zo * g.sCo + zi * g.sCi : nullptr;
    uint16_t* __restrict__ Chi = g.Chi ? g.Chi + zo * g.sCo + zi * g.sCi : nullptr;
    uint16_t* __restrict__ Clo = g.Chi ? g.Clo + zo * g.sCo + zi * g.sCi : nullptr;
    const float* __restrict__ bias = g.bias ? g.bias + zo * g.sBo + zi * g.sBi : nullptr;
    const float* __restrict__ R = g.R ? g.R + zo * g.sRo + zi * g.sRi : nullptr;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (!R || (((g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0))) &&
                        (!Chi || (((reinterpret_cast<uintptr_t>(Chi) | reinterpret_cast<uintptr_t>(Clo)) & 7) == 0));
    char* slab = smem + wave * EP_SLAB;
    AviF16Range rng;                                    // fp16 planes only: range guard of what this wave splits
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            *reinterpret_cast<f32x4*>(slab + (b * 16 + fr) * EP_STRIDE + (a * 16 + fq * 4) * 4) = acc[a][b];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave, other lanes' rows
    // 64 rows x 4 NT float4: item i = lane + 64 it -> row i / (4 NT), columns 4 (i % (4 NT)) .. +3
#pragma unroll
    for (int it = 0; it < 4 * NT; ++it) {
        const int i = lane + 64 * it;
        const int row = i / (4 * NT), c4 = i - row * (4 * NT);
        const int m = m0 + wr * 64 + row, n = n0 + wc * (16 * NT) + c4 * 4;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(slab + row * EP_STRIDE + c4 * 16);
        if (m >= g.M || n >= g.N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = r0[j] + ((bias && n + j < g.N) ? bias[n + j] : 0.f);
        if (g.act == AVI_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = avi_gelu_lds(v[j], smem + Geo<NT>::WORK_BYTES);
        } else if (g.act != AVI_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = avi_act(v[j], g.act);
        }
        if (g.scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < g.N) v[j] = v[j] * g.scale[n + j] + g.shift[n + j];
        }
        if (vec_ok && n + 3 < g.N) {
            if (R) {
                const float4 q = *reinterpret_cast<const float4*>(R + (long long)m * g.ldr + n);
                v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
            }
            if (C) *reinterpret_cast<float4*>(C + (long long)m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            if (Chi) {
                uint32_t h[2], l[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint16_t h0, h1, l0, l1;
                    if (F16) {
                        rng.see(v[2 * j]);
                        rng.see(v[2 * j + 1]);
                    }
                    avi_split_hl(v[2 * j], F16 ? AVI_PLANES_F16 : AVI_PLANES_BF16, h0, l0);
                    avi_split_hl(v[2 * j + 1], F16 ? AVI_PLANES_F16 : AVI_PLANES_BF16, h1, l1);
                    h[j] = h0 | ((uint32_t)h1 << 16);
                    l[j] = l0 | ((uint32_t)l1 << 16);
                }
                const long long o = (long long)m * g.ldc + n;
                *reinterpret_cast<uint2*>(Chi + o) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2*>(Clo + o) = make_uint2(l[0], l[1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < g.N) {
                    const float y = v[j] + (R ? R[(long long)m * g.ldr + n + j] : 0.f);
                    if (C) C[(long long)m * g.ldc + n + j] = y;
                    if (Chi) {
                        if (F16) rng.see(y);
                        avi_split_hl(y, F16 ? AVI_PLANES_F16 : AVI_PLANES_BF16, Chi[(long long)m * g.ldc + n + j],
                                     Clo[(long long)m * g.ldc + n + j]);
                    }
                }
        }
    }
    if (F16 && Chi) rng.commit(status);
    if (!sk_total) break;
    bar();            // the slabs overlay the stages: every wave has read its slab before the next item's LDS-DMA lands
  }
}

// MEASURED (round 3, beside the sampler, 224 CUs, M = 8000, bf16x3; scripts/bench_enc_gemm.py): stream-K LOSES on every
// encoder projection - qkv 121 -> 147 us, out 49 -> 71 us, ffn2 124 -> 182 us.  With 189 tiles on 224 CUs nearly every
// workgroup's share is a tile tail plus a tile head, i.e. two shared tiles: two 128 KB partials stored at the CU's ~14
// B/clk store rate (~5 us each), two agent-scope releases (each writes back its XCD's whole L2, other workgroups' output
// tiles included) and one or two partials read back cost 25-55 us against the 16 % (20 us) of a tile the equal shares save.
// It is therefore OFF unless AVI_GEMM_STREAMK asks for it (1: by the plan below, 2: whenever legal); the path stays
// tested (tests/test_gpu_ops.py::test_stream_k_gemm).  What would pay is a fix-up that does not go through HBM-side
// stores (none exists between CUs on this chip) or problems with K >> 3072.
// Plan (mode 1): shares of at least 4 K groups (12 K tiles), at most SK_MAXC contributors per tile, a last round filled
// to less than 90 %.  Returns the number of workgroups to launch (0 = plain data-parallel launch).
template <int NT>
int stream_k_plan(const AviGemm& g, int tiles, int nk) {
    const char* e_ = getenv("AVI_GEMM_STREAMK");                // 0 off (default), 1 by the plan below, 2 whenever legal
    const int mode = e_ ? atoi(e_) : 0;
    if (!mode || !g.sk_ws || g.batch != 1 || g.cus <= 0 || g.cus > 256) return 0;
    const int G = nk / 3, cus = g.cus;
    const long long total = (long long)tiles * G;
    const long long need = (long long)tiles * SK_MAXC * BM * Geo<NT>::BN + tiles + 64;   // floats: slots + counters
    if (g.sk_ws_floats < need) return 0;
    const long long share = total / cus;                      // K groups per workgroup (floor)
    if (share < 4 || (long long)(SK_MAXC - 2) * share < G) return 0;      // a tile spans <= G / share + 2 shares
    const int rounds = (tiles + cus - 1) / cus;
    const double fill = (double)tiles / ((double)rounds * cus);
    if (mode != 2 && fill > 0.90) return 0;
    return cus;
}

template <int NS, int NT, bool F16 = false>
int launch(const AviGemm& g, hipStream_t s) {
    constexpr int BN = Geo<NT>::BN, SMEM_BYTES = Geo<NT>::SMEM_BYTES;
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(gemm_pp192_kernel<NS, NT, F16>), SMEM_BYTES);
    const int nk = g.K / (NS == 2 ? 32 : 64);
    const int sk_wgs = stream_k_plan<NT>(g, tilesM * tilesN, nk);
    if (sk_wgs) {
        float* ws = g.sk_ws;
        int* cnt = reinterpret_cast<int*>(ws + (long long)tilesM * tilesN * SK_MAXC * BM * BN);
        hipLaunchKernelGGL((gemm_pp192_kernel<NS, NT, F16>), dim3(sk_wgs, 1), dim3(NTHR), SMEM_BYTES, s, g, tilesM, tilesN,
                           tilesM * tilesN * (nk / 3), ws, cnt, avi_status_ptr());
        return avi_launch_status();
    }
    hipLaunchKernelGGL((gemm_pp192_kernel<NS, NT, F16>), dim3(tilesM * tilesN, g.batch), dim3(NTHR), SMEM_BYTES, s, g, tilesM,
                       tilesN, 0, nullptr, nullptr, avi_status_ptr());
    return avi_launch_status();
}

}  // namespace

// true when the 128-row ping-pong kernels can take the problem (K tiles come in threes)
bool avi_gemm_pp192_ok(const AviGemm& g) {
    const int kt = (g.prec & 0xff) == AVI_PREC_BF16 ? 64 : 32;
    return g.Ahi && g.K % (3 * kt) == 0;
}

// bn = 192 or 256
int avi_gemm_pp192_launch(const AviGemm& g, int bn, hipStream_t s) {
    if ((g.prec & 0xff) == AVI_PREC_F16X2) return bn == 256 ? launch<2, 4, true>(g, s) : launch<2, 3, true>(g, s);
    const bool x3 = (g.prec & 0xff) == AVI_PREC_BF16X3;
    if (bn == 256) return x3 ? launch<2, 4>(g, s) : launch<1, 4>(g, s);
    return x3 ? launch<2, 3>(g, s) : launch<1, 3>(g, s);
}
