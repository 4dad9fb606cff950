// Third-generation GEMM for the large contractions of the audio path (conv stack, qkv, ffn): a 256 x 256 tile
// "ping-pong" kernel.  Same contract as gemm.hip / gemm_dma.hip (C = affine(act(A.W^T + bias)) + R on split-plane
// activations AviGemm.Ahi/Alo and bf16 hi/lo weight planes), different schedule.
//
// Why: the 128x128 (gemm.hip) and one-phase 256x128 LDS-DMA (gemm_dma.hip) structures both stop at ~0.86-0.9 PFLOP/s
// of issued MFMA work: every wave does everything (stage, read fragments, multiply), so on each SIMD both resident
// waves sit in their memory section (one LDS-DMA piece costs 100-185 issue cycles next to ds_reads) at the same time
// and the matrix pipe idles ~55 % of the cycles.  Here the two waves of a SIMD ALTERNATE:
//
//   * one workgroup of 8 waves (2 along m x 4 along n, 128 x 64 outputs each) per CU, 128 KiB of LDS for the stages;
//   * a phase = [memory work: 0..12 ds_read_b128 of fragments, 2 LDS-DMA pieces, a counted vmcnt wait] + [matrix section:
//     the MFMAs of one quadrant of the wave's tile (24 in bf16x3, 16 in bf16)] + ONE s_barrier.  The 4 waves with wr = 0
//     ("group A") run  reads -> MFMAs -> issue -> wait  inside a barrier interval, the 4 with wr = 1 ("group B")
//     MFMAs -> wait -> reads -> issue  (their fragments were read in the previous interval), so on every SIMD one wave
//     multiplies while the other loads (the order inside the sections is derived in front of the K loop below);
//   * a K tile (32 k in bf16x3: rows of [hi 64 B | lo 64 B]; 64 k in bf16: rows of 128 B) is staged as FOUR half
//     tiles (X0 X1 W0 W1: the rows the quadrant m-half / n-half of every wave needs), two stages, one half tile per
//     phase, each issued FIVE or six phases before its first read.  Hazards, in phases (cdna_hip_programming.md
//     section 5, "Read a staged buffer one phase after the wait that retires it"):
//       RAW  every wave's wait for a half tile sits a BARRIER ahead of its first reader.  Group B reads the fragments of
//            phase q+1 at the end of interval q: group B's own wait (in its memory work of interval q-1) and group A's
//            (it ends group A's interval q-1, in front of the barrier) both precede barrier q-1;
//       WAR  a slot is restaged >= 2 phases after its last ds_read (group B's read of phase q retires before its
//            matrix section in interval q+1, i.e. before the barrier group A passes on its way to phase q+2).
//     Per K tile T (stage s = T & 1):   reads            matrix quadrant   LDS-DMA issue
//       P1                               X0(T), W0(T)     (m0, n0)          W1(T+1) -> stage s^1
//       P2                               W1(T)            (m0, n1)          X1(T+1) -> stage s^1
//       P3                               X1(T)            (m1, n1)          X0(T+2) -> stage s
//       P4                               -                (m1, n0)          W0(T+2) -> stage s
//     The wait is always `s_waitcnt vmcnt(6)`: the 3 half tiles (6 pieces per wave) issued most recently may stay in
//     flight - 48-64 KiB per CU is always on its way.
//   * past the end of K the issue slots load the last K tile again into slots nobody reads any more, which keeps
//     the vmcnt arithmetic uniform (3 % extra L2 reads at K = 1536).
//   * LDS image and fragment addressing as in gemm_dma.hip: lane-linear LDS-DMA with the bank swizzle applied on the
//     SOURCE address (chunk c of row R sits at c ^ (R & 7)); fragments by ds_read_b128, conflict-free.
#include <cstdlib>

#include "common.h"
#include "gelu_table.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int BM = 256, BN = 256, NTHR = 512;
constexpr int HALF_BYTES = 128 * 128;          // 128 rows x 128 B
constexpr int STAGE_BYTES = 4 * HALF_BYTES;    // X0 X1 W0 W1
constexpr int EP_STRIDE = 272;                 // epilogue slab: 64 fp32 + 16 B pad per row
constexpr int EP_SLAB = 64 * EP_STRIDE;        // per wave
constexpr int WORK_BYTES = 8 * EP_SLAB > 2 * STAGE_BYTES ? 8 * EP_SLAB : 2 * STAGE_BYTES;   // 136 KiB
constexpr int TAB_BYTES = 2048;                // GELU table (gelu_table.h), behind the stages / slabs for the whole kernel
constexpr int DUMMY_BYTES = 8 * 2 * 256;       // landing pad of the WS mode's dummy pieces (one 4-byte load per lane)
constexpr int SMEM_BYTES = WORK_BYTES + TAB_BYTES + DUMMY_BYTES;
constexpr int KX0 = 0, KX1 = 1, KW0 = 2, KW1 = 3;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

// GELU in the epilogue comes from the LDS table of gelu_table.h: with two waves per SIMD the activation's arithmetic (not
// the 256 KB of stores) bounded the epilogue: 28-37 k cycles of a 232 k tile with the rational erf, 25-30 k with the
// table, 19-21 k without an activation (in-kernel stamps, conv layer 2).

// F16 (with NS = 2): the opt-in 2-term fp16 mode AVI_PREC_F16X2 - activation planes are fp16 hi/lo, the weight is ONE fp16
// plane (the "lo" half of a staged weight row is a second copy nobody multiplies), y = w.xh + w.xl: two MFMAs per
// product instead of three; plane outputs are split into fp16 hi/lo.
// WS (F16 only, K tiles in fours): WEIGHT SUPER-TILES.  In F16 mode the "lo" half of a staged weight row used to be a second
// copy of the same 64 bytes; a 256 x 256 tile then stages 64 KB per 32-deep K tile = 2 430 cycles at the CU's 27 B/clk against
// 2 048 cycles of MFMAs (two per product): the 2-term tile is INGEST-bound, which is why it gained only 17-21 % from a third
// fewer MFMAs.  With WS the lo half holds the NEXT K tile's 64 bytes: weights are staged for even K tiles only (into the
// W areas of stage (T / 2) & 1), an odd K tile reads its fragments from the lo position of the previous tile's rows, and the
// issue slots that used to stage an odd tile's weights issue DUMMY pieces (4 bytes per lane into a landing pad) so that
// every wave still issues two pieces per phase and the counted `vmcnt(6)` waits keep their meaning unchanged.  Hazards:
//   WAR  the W areas of stage (j + 1) & 1 were last read by super-tile j - 1 (its odd tile: W0 in P1, W1 in P2); super-tile
//        j + 1 is issued into them in P4 of tile 2j (W0) and P1 of tile 2j + 1 (W1): at least six phases later;
//   RAW  the real issues sit in exactly the slots, 5 phases ahead of the first read, that staged the even tiles before;
//        the odd tile reads the same rows 4 phases later still.
template <int NS, bool F16 = false, bool WS = false>
__global__ __launch_bounds__(NTHR) void gemm_pp_kernel(const AviGemm g, const int tilesM, const int tilesN,
                                                       unsigned* __restrict__ status) {
    static_assert(!WS || (F16 && NS == 2), "weight super-tiles exist in the 2-term fp16 mode only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

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
    const uint16_t* __restrict__ Alo = (NS == 2) ? g.Alo + zo * g.sAo + zi * g.sAi : Ahi;
    const uint16_t* __restrict__ Whi = g.Whi + zo * g.sWo + zi * g.sWi;
    const uint16_t* __restrict__ Wlo = (NS == 2 && !F16) ? g.Wlo + zo * g.sWo + zi * g.sWi : Whi;

    // ---- LDS-DMA source pointers: half tile `kind`, pieces wave and wave + 8 (1 KiB = 8 rows x 128 B each).
    //      lane -> LDS row R = 8 piece + lane/8 of the half tile, LDS chunk lane%8 <- source chunk c = lane%8 ^ (R&7).
    constexpr int KB = NS == 2 ? 64 : 128;   // bytes one K tile advances along a row of a plane
    const char* src[4][2];
#pragma unroll
    for (int kind = 0; kind < 4; ++kind)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int R = (wave + 8 * i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (R & 7);
            const int off = (NS == 2 && !(WS && kind >= 2)) ? (c & 3) * 16 : c * 16;   // WS: 128 contiguous bytes of the fp16 row
            if (kind < 2) {   // activation rows: wave-row wr' = R / 64 owns rows wr' * 128 + h * 64 + R % 64
                int m = m0 + (R >> 6) * 128 + kind * 64 + (R & 63);
                m = m < g.M ? m : g.M - 1;
                const uint16_t* base = (NS == 2 && c >= 4) ? Alo : Ahi;
                src[kind][i] = reinterpret_cast<const char*>(base + (long long)m * g.lda) + off;
            } else {          // weight rows: wave-column wc' = R / 32 owns rows wc' * 64 + h * 32 + R % 32
                int n = n0 + (R >> 5) * 64 + (kind - 2) * 32 + (R & 31);
                n = n < g.N ? n : g.N - 1;
                const uint16_t* base = (NS == 2 && c >= 4) ? Wlo : Whi;
                src[kind][i] = reinterpret_cast<const char*>(base + (long long)n * (g.ldw ? g.ldw : g.K)) + off;
            }
        }
    const int nk = g.K / (NS == 2 ? 32 : 64);   // even, >= 2 (checked by the launcher)

    const bool diag_noload = g.prec & 0x100, diag_nomfma = g.prec & 0x200;   // timing diagnostics (results invalid)
    const char* dummy_src = reinterpret_cast<const char*>(Whi) + lane * 4;      // 256 contiguous bytes, always mapped
    auto dummy = [&]() __attribute__((always_inline)) {                           // two pieces, like every issue slot
        char* dst = smem + WORK_BYTES + TAB_BYTES + wave * 512;
        __builtin_amdgcn_global_load_lds((gbl_void*)dummy_src, (lds_void*)dst, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)dummy_src, (lds_void*)(dst + 256), 4, 0, 0);
    };
    auto issue = [&](int kind, int T, int stage) __attribute__((always_inline)) {
        if (diag_noload && T >= 2) return;
        const int kt = T < nk ? T : (WS && kind >= 2 ? nk - 2 : nk - 1);
        const long long kofs = (long long)kt * KB;
        char* dst = smem + stage * STAGE_BYTES + kind * HALF_BYTES + wave * 1024;
        glds16(src[kind][0] + kofs, dst);
        glds16(src[kind][1] + kofs, dst + 8 * 1024);
    };

    // ---- fragment addresses: lane (fr, fq) reads row (16-row tile base) + fr, chunks fq ("hi" position) and 4 + fq
    const int fr = lane & 15, fq = lane >> 4;
    const int pos = (fq ^ (fr & 7)) << 4;
    const int xoff = (wr * 64 + fr) * 128 + pos;                     // + stage, + h * HALF, + bl * 2048
    const int woff = 2 * HALF_BYTES + (wc * 32 + fr) * 128 + pos;    // + stage, + h * HALF, + al * 2048
    const char* xhi_p = smem + xoff;
    const char* xlo_p = smem + (xoff ^ 64);                          // chunk 4 + fq sits at pos ^ 64
    const char* whi_p = smem + woff;
    const char* wlo_p = smem + (woff ^ 64);

    auto read_x = [&](int stage, int h, bf16x8 (&xh)[4], bf16x8 (&xl)[4]) __attribute__((always_inline)) {
        const int o = stage * STAGE_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int bl = 0; bl < 4; ++bl) {
            xh[bl] = *reinterpret_cast<const bf16x8*>(xhi_p + o + bl * 2048);
            xl[bl] = *reinterpret_cast<const bf16x8*>(xlo_p + o + bl * 2048);
        }
    };
    auto read_w = [&](int stage, int h, bf16x8 (&wh)[2], bf16x8 (&wl)[2]) __attribute__((always_inline)) {
        const int o = stage * STAGE_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int al = 0; al < 2; ++al) {
            wh[al] = *reinterpret_cast<const bf16x8*>(whi_p + o + al * 2048);
            wl[al] = *reinterpret_cast<const bf16x8*>(wlo_p + o + al * 2048);
        }
    };
    // WS: the fragments of K tile `pos` (0 = even tile: hi position, 1 = odd tile: lo position) of the super-tile in `stage`
    auto read_ws = [&](int stage, int h, int pos, bf16x8 (&wh)[2]) __attribute__((always_inline)) {
        const int o = stage * STAGE_BYTES + h * HALF_BYTES;
        const char* base = pos ? wlo_p : whi_p;
#pragma unroll
        for (int al = 0; al < 2; ++al) wh[al] = *reinterpret_cast<const bf16x8*>(base + o + al * 2048);
    };

    f32x4 acc[4][8];   // [n tile][m tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto quadrant = [&](int mh, int nh, const bf16x8 (&xh)[4], const bf16x8 (&xl)[4], const bf16x8 (&wh)[2],
                        const bf16x8 (&wl)[2]) __attribute__((always_inline)) {
        if (diag_nomfma) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int al = 0; al < 2; ++al)
#pragma unroll
            for (int bl = 0; bl < 4; ++bl) {
                f32x4 c = acc[nh * 2 + al][mh * 4 + bl];
                if (F16) {
                    const f16x8 wv = __builtin_bit_cast(f16x8, wh[al]);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, __builtin_bit_cast(f16x8, xl[bl]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, __builtin_bit_cast(f16x8, xh[bl]), c, 0, 0, 0);
                } else if (NS == 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[al], xh[bl], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[al], xl[bl], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[al], xh[bl], c, 0, 0, 0);
                } else {   // "hi"/"lo" positions are k 0..31 / 32..63 of the 64-wide K tile
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[al], xh[bl], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[al], xl[bl], c, 0, 0, 0);
                }
                acc[nh * 2 + al][mh * 4 + bl] = c;
            }
        __builtin_amdgcn_s_setprio(0);
    };
    auto mem_top = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

#ifdef AVI_PP_STAMPS   // diagnostic build only (scripts/pp_stamps.py): g.shift is a uint64 stamp buffer when bit 0x800 is set
    unsigned long long st[6];
    st[0] = __builtin_amdgcn_s_memtime();
    st[4] = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- prologue: the 6 half tiles the steady-state schedule has in flight before phase 1 of K tile 0.  Waves 0 and 1
    //      then fetch the GELU table as their youngest load (as the oldest it put a cold 0.9 us read in front of the
    //      first K tile): the counted waits below only get stricter by one piece for them, and the `vmcnt(0)` in front of
    //      the epilogue covers the table.
    issue(KX0, 0, 0);
    issue(KW0, 0, 0);
    issue(KW1, 0, 0);
    issue(KX1, 0, 0);
    issue(KX0, 1, 1);
    if (WS) dummy(); else issue(KW0, 1, 1);        // WS: tile 1's weights came with tile 0's rows
    if (g.act == AVI_ACT_GELU && wave < 2)
        glds16(reinterpret_cast<const char*>(avi_gelu_tab) + wave * 1024 + lane * 16, smem + WORK_BYTES + wave * 1024);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // X0(0), W0(0), W1(0) have landed
    bar();

#ifdef AVI_PP_STAMPS
    st[1] = __builtin_amdgcn_s_memtime();
#endif
    // ONE barrier per phase.  Group A (wr = 0) runs  mem(p) ; mfma(p) ; barrier,  group B (wr = 1) runs
    // mfma(p-1) ; mem(p) ; barrier : inside a barrier interval one wave of each SIMD is in its memory section while
    // the other multiplies, then they swap.  In phases the hazards are those of the header: every ds_read of phase q
    // has retired before the barrier that ends phase q+1 (WAR: restage >= 2 phases after the last read), and every
    // wave's vmcnt wait for what phase q reads sits in phase q-1 (RAW).
    bf16x8 xh[4], xl[4], w0h[2], w0l[2], w1h[2], w1l[2];
    // Where the LDS-DMA issue and the counted wait sit.
    //  * A piece's issue blocks the wave until the CU's vector-memory path takes it (16 pieces per phase per CU at ~37
    //    cycles each: with the MFMAs switched off a phase takes 600 cycles, with the loads switched off 768), and a wave
    //    issues in order: with  wait ; issue ; reads ; MFMAs  group A's matrix section started ~600 cycles into the
    //    phase although the pipe is free after group B's 384, and a phase cost 600 + 384 (measured 1008).  So: fragment
    //    reads first in every memory section (their LDS latency runs under whatever follows), and group A issues its
    //    pieces AFTER its matrix section, when the path has drained group B's: its MFMAs follow B's directly (968).
    //  * Group B reads the fragments of phase q+1 at the END of interval q, so every wave's wait for that data has to
    //    sit before barrier q-1.  Group B's own wait (top of its memory section of interval q-1, 3 half tiles in flight:
    //    everything issued <= q-4 has landed, phase q+1 reads what was issued <= q-4) does; group A's used to sit at
    //    the TOP of interval q - ordered before B's read only by B's 24 MFMAs in between.  It now ends group A's
    //    interval q-1 (after its issue, before the barrier): same count, same moment, the other side of the barrier.
    auto rd1 = [&](int u) __attribute__((always_inline)) {
        read_w(u, 0, w0h, w0l);
        read_x(u, 0, xh, xl);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto rd2 = [&](int u) __attribute__((always_inline)) {
        read_w(u, 1, w1h, w1l);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto rd3 = [&](int u) __attribute__((always_inline)) {
        read_x(u, 1, xh, xl);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (WS) {
        // K tiles in fours: tile T + v has its X halves in stage v & 1 and its weights in the W areas of stage v >> 1 at
        // position v & 1.  issue_w(kind, v2): the slot that stages tile T + v2's weights - real for an even tile (into the
        // W areas of stage (v2 >> 1) & 1), a dummy for an odd one.
        auto issue_w = [&](int kind, int T, int v2) __attribute__((always_inline)) {
            if (v2 & 1) dummy(); else issue(kind, T + v2, (v2 >> 1) & 1);
        };
        auto rdw1 = [&](int v) __attribute__((always_inline)) {          // W0 + X0 of tile v (mod 4)
            read_ws((v >> 1) & 1, 0, v & 1, w0h);
            read_x(v & 1, 0, xh, xl);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto rdw2 = [&](int v) __attribute__((always_inline)) {
            read_ws((v >> 1) & 1, 1, v & 1, w1h);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (wr == 0) {
            for (int T = 0; T < nk; T += 4) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {               // K tile T + v
                    const int u = v & 1;
                    rdw1(v);
                    quadrant(0, 0, xh, xl, w0h, w0l);
                    issue_w(KW1, T, v + 1);
                    mem_top();
                    bar();
                    rdw2(v);
                    quadrant(0, 1, xh, xl, w1h, w1l);
                    issue(KX1, T + v + 1, u ^ 1);
                    mem_top();
                    bar();
                    rd3(u);
                    quadrant(1, 1, xh, xl, w1h, w1l);
                    issue(KX0, T + v + 2, u);
                    mem_top();
                    bar();
                    quadrant(1, 0, xh, xl, w0h, w0l);
                    issue_w(KW0, T, v + 2);
                    mem_top();
                    bar();
                }
            }
        } else {
            rdw1(0);
            dummy();                                        // the slot of KW1(1)
            for (int T = 0; T < nk; T += 4) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int u = v & 1;
                    quadrant(0, 0, xh, xl, w0h, w0l);
                    mem_top();
                    rdw2(v);
                    issue(KX1, T + v + 1, u ^ 1);
                    bar();
                    quadrant(0, 1, xh, xl, w1h, w1l);
                    mem_top();
                    rd3(u);
                    issue(KX0, T + v + 2, u);
                    bar();
                    quadrant(1, 1, xh, xl, w1h, w1l);
                    mem_top();
                    issue_w(KW0, T, v + 2);
                    bar();
                    quadrant(1, 0, xh, xl, w0h, w0l);
                    mem_top();
                    rdw1((v + 1) & 3);                      // first phase of the next K tile (a dummy after the last)
                    issue_w(KW1, T, v + 2);
                    bar();
                }
            }
        }
    } else
    if (wr == 0) {
        for (int T = 0; T < nk; T += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {               // K tile T + u in stage u
                rd1(u);
                quadrant(0, 0, xh, xl, w0h, w0l);
                issue(KW1, T + u + 1, u ^ 1);
                mem_top();
                bar();
                rd2(u);
                quadrant(0, 1, xh, xl, w1h, w1l);
                issue(KX1, T + u + 1, u ^ 1);
                mem_top();
                bar();
                rd3(u);
                quadrant(1, 1, xh, xl, w1h, w1l);
                issue(KX0, T + u + 2, u);
                mem_top();
                bar();
                quadrant(1, 0, xh, xl, w0h, w0l);
                issue(KW0, T + u + 2, u);
                mem_top();
                bar();
            }
        }
    } else {
        rd1(0);
        issue(KW1, 1, 1);
        for (int T = 0; T < nk; T += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                quadrant(0, 0, xh, xl, w0h, w0l);
                mem_top();
                rd2(u);
                issue(KX1, T + u + 1, u ^ 1);
                bar();
                quadrant(0, 1, xh, xl, w1h, w1l);
                mem_top();
                rd3(u);
                issue(KX0, T + u + 2, u);
                bar();
                quadrant(1, 1, xh, xl, w1h, w1l);
                mem_top();
                issue(KW0, T + u + 2, u);
                bar();
                quadrant(1, 0, xh, xl, w0h, w0l);
                mem_top();
                rd1(u ^ 1);                             // first phase of the next K tile (a dummy after the last)
                issue(KW1, T + u + 2, u);
                bar();
            }
        }
    }

#ifdef AVI_PP_STAMPS
    st[2] = __builtin_amdgcn_s_memtime();
    const bool stamps = g.prec & 0x800;
    unsigned long long* stamp_out = reinterpret_cast<unsigned long long*>(const_cast<float*>(g.shift));
    const float* gscale = stamps ? nullptr : g.scale;
    const float* gshift = stamps ? nullptr : g.shift;
#else
    const float* gscale = g.scale;
    const float* gshift = g.shift;
#endif
    // ---- epilogue (same contract as gemm.hip).  The accumulator layout (lane = row fr, 4 columns) would store 64-B
    //      row segments: 256 CUs finishing their tiles together then write at ~2 TB/s and the epilogue costs a third
    //      of the tile (measured with in-kernel stamps).  Each wave transposes its 128 x 64 block through a private
    //      LDS slab instead (two passes of 64 rows, row stride 272 B: conflict-free b128 writes), so that 8 lanes
    //      hold one row's 64 columns and every store instruction writes whole 128-B (plane) / 256-B (fp32) lines.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();                                              // every wave's tail DMA has landed: the stages are dead
    float* __restrict__ C = g.C ? g.C + zo * g.sCo + zi * g.sCi : nullptr;
    uint16_t* __restrict__ Chi = g.Chi ? g.Chi + zo * g.sCo + zi * g.sCi : nullptr;
    uint16_t* __restrict__ Clo = g.Chi ? g.Clo + zo * g.sCo + zi * g.sCi : nullptr;
    const float* __restrict__ bias = g.bias ? g.bias + zo * g.sBo + zi * g.sBi : nullptr;
    const float* __restrict__ R = g.R ? g.R + zo * g.sRo + zi * g.sRi : nullptr;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (!R || (((g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0))) &&
                        (!Chi || (((g.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(Chi) & 15) == 0) &&
                                  ((reinterpret_cast<uintptr_t>(Clo) & 15) == 0)));
    char* slab = smem + wave * EP_SLAB;
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const int n = n0 + wc * 64 + ecol;
    float bv[8], sc[8], sh[8];
    AviF16Range rng;                                    // fp16 planes only: range guard of what this wave splits
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool in = n + j < g.N;
        bv[j] = (bias && in) ? bias[n + j] : 0.f;
        sc[j] = (gscale && in) ? gscale[n + j] : 1.f;
        sh[j] = (gscale && in) ? gshift[n + j] : 0.f;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bl = 0; bl < 4; ++bl)
                *reinterpret_cast<f32x4*>(slab + (bl * 16 + fr) * EP_STRIDE + (a * 16 + fq * 4) * 4) =
                    pass == 0 ? acc[a][bl] : acc[a][4 + bl];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave, other lanes' rows
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + erow;
            const int m = m0 + wr * 128 + pass * 64 + row;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(slab + row * EP_STRIDE + ecol * 4);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(slab + row * EP_STRIDE + ecol * 4 + 16);
            if (m >= g.M || n >= g.N) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = r0[j] + bv[j];
                v[4 + j] = r1[j] + bv[4 + j];
            }
            if (g.act == AVI_ACT_GELU) {   // one uniform branch per 8 values, straight-line math inside
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = avi_gelu_lds(v[j], smem + WORK_BYTES);
            } else if (g.act != AVI_ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = avi_act(v[j], g.act);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
            if (vec_ok && n + 7 < g.N) {
                if (R) {
                    const float4 q0 = *reinterpret_cast<const float4*>(R + (long long)m * g.ldr + n);
                    const float4 q1 = *reinterpret_cast<const float4*>(R + (long long)m * g.ldr + n + 4);
                    v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w;
                    v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
                }
                if (C) {
                    float* cp = C + (long long)m * g.ldc + n;
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (Chi) {   // split once here so the consumer GEMM never converts
                    uint32_t h[4], l[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
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
                    *reinterpret_cast<uint4*>(Chi + o) = make_uint4(h[0], h[1], h[2], h[3]);
                    *reinterpret_cast<uint4*>(Clo + o) = make_uint4(l[0], l[1], l[2], l[3]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next pass overwrites the slab
    }
    if (F16 && Chi) rng.commit(status);
#ifdef AVI_PP_STAMPS
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[3] = __builtin_amdgcn_s_memtime();
        st[5] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 || tid == 256) {
            unsigned long long* o = stamp_out + ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + (tid >> 8)) * 6;
            for (int i = 0; i < 6; ++i) o[i] = st[i];
        }
    }
#endif
}

template <int NS, bool F16 = false, bool WS = false>
int launch(const AviGemm& g, hipStream_t s) {
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(gemm_pp_kernel<NS, F16, WS>), SMEM_BYTES);
    hipLaunchKernelGGL((gemm_pp_kernel<NS, F16, WS>), dim3(tilesM * tilesN, g.batch), dim3(NTHR), SMEM_BYTES, s, g, tilesM,
                       tilesN, avi_status_ptr());
    return avi_launch_status();
}

}  // namespace

// true when the ping-pong kernel can take the problem (K tiles come in pairs)
bool avi_gemm_pp_ok(const AviGemm& g) {
    const int kt = (g.prec & 0xff) == AVI_PREC_BF16 ? 64 : 32;
    return g.Ahi && g.K % (2 * kt) == 0 && g.K >= 2 * kt;
}

int avi_gemm_pp_launch(const AviGemm& g, hipStream_t s) {
    if ((g.prec & 0xff) == AVI_PREC_F16X2) {
        // weight super-tiles when the K tiles come in fours (every GEMM of the audio path); AVI_GEMM_WS=0: the older staging
        const char* e = getenv("AVI_GEMM_WS");
        const bool ws = (g.K % 128) == 0 && !(e && atoi(e) == 0) && !(g.prec & 0x300);
        return ws ? launch<2, true, true>(g, s) : launch<2, true>(g, s);
    }
    return (g.prec & 0xff) == AVI_PREC_BF16X3 ? launch<2>(g, s) : launch<1>(g, s);
}
