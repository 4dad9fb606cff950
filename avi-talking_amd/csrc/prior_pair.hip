// PAIRED variant of the batched matrix-core DDPM sampler (prior_mfma.inc): two samples share TWO workgroups on two CUs, and
// each of the two streams only HALF of every layer's weights.
//
// Why: the sampler is bound by what one CU can take in from its XCD's L2 (983 KB per layer-step at 30 B/clk = the 32 k cycles a
// layer takes; in-kernel stamps) - not by arithmetic (3 or 6 token rows of a 16-row MFMA tile) and not by
// the chip's bandwidth.  One workgroup per sample makes every CU stream every matrix.  Here the pair splits the matrices
// Megatron-style, so a CU streams 458 KB per layer for TWO samples, on the same 32 CUs for a batch of 32:
//
//   half h of the pair            streams                                   computes (for both samples, 6 token rows)
//   to_q | to_kv  (N-split)       its 4 heads' 256 q columns + all of k, v  q of heads 4h .. 4h+3, k, v
//   attention                     -                                         its 4 heads (wave = (head, sample))
//   to_out        (K-split)       the 256 input columns of its heads        a PARTIAL sum of the 128 outputs
//   >>> exchange 1: partial sums (6 x 128 fp32) to the partner, its partial sums back, both add
//   ff1           (N-split)       256 value + the 256 matching gate rows    half of the SwiGLU inputs
//   SwiGLU                        -                                         its 256 features
//   ff2           (K-split)       its 256 input columns                     a PARTIAL sum of the 128 outputs
//   >>> exchange 2
//   LayerNorms, residual stream, DDPM update: computed redundantly by both halves (a few hundred cycles)
//
// The exchange (MI355X_MICROARCH.md, "handoff-1to1": data-tagged 8-byte granules, no flag, no fence): every value travels as
// one naturally aligned {fp32 value, 32-bit tag} written by ONE relaxed agent-scope store (global_store_dwordx2 sc1: write-
// through to L2) and polled with relaxed agent-scope loads (sc1: L1 bypassed).  tag = launch epoch (16 bits) << 16 | exchange
// number (16 bits; the entry point refuses loops with more than 65534 exchanges), so a granule of an earlier exchange or an
// earlier launch never matches; two slots per half (exchange parity): a slot is rewritten
// at exchange n + 2, which the writer can only reach after it has RECEIVED the partner's n + 1, which the partner sent after a
// workgroup barrier behind its poll of n.  vmcnt is in order per WAVE, so a poll issued behind a prefetch would wait for the
// prefetched weights: waves 4-7 poll (ring empty) while waves 0-3 issue the next matrix's prefetch and keep the CU's memory
// path busy; waves 4-7 prefetch theirs after the sums are in.  Partners are blocks b and b + 8 (one XCD under the observed
// round-robin placement: speed only).  No workgroup waits for anything but its partner's granules; spins are bounded and a
// timeout raises the workspace's error word instead of hanging (the result is then garbage and the host raises).
//
// Same arithmetic as the unpaired kernel except for the order of the two partial sums of to_out / ff2 (fp32, a few ulps).
#include "prior_mfma.inc"

namespace {

constexpr int PAIR_S = 2;                 // samples per pair
constexpr int XCH_VALS = 3 * PAIR_S * DIM; // fp32 values per exchange and direction
constexpr int XCH_HDR = 8;                // header words (u64): [0] epoch, [1] error
constexpr unsigned SPIN_LIMIT = 1u << 22; // ~0.5 s of polling: a partner that never answers is a bug, not a wait

// A wave's share of a linear: NTW column tiles (any tile indices) x KCN consecutive 128-wide K chunks starting at kc0.
// Every unit fits the ring (U <= DEPTH): prefetch loads the whole share, run consumes it.
template <int K, int N, int MODE, int NTW, int KCN>
struct LinP {
    static constexpr bool F16 = MODE != 0;
    static constexpr int KCALL = K / (32 * KCH);
    static constexpr int U = NTW * KCN;
    static_assert(U <= DEPTH, "the share must fit the register ring");

    static __device__ __forceinline__ void prefetch(const uint16_t* __restrict__ Whi, const uint16_t* __restrict__ Wlo,
                                                    const int (&tiles)[NTW], int kc0, WRing& r) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = u / KCN, kc = kc0 + (u - t * KCN);
            const long long o = (((long long)tiles[t] * (K / 32) + kc * KCH) * 64 + lane) * 8;
#pragma unroll
            for (int ks = 0; ks < KCH; ++ks) {
                r.h[u][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Whi + o + ks * 512));
                if (MODE == 0) r.l[u][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wlo + o + ks * 512));
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // as Lin::prefetch: the loads stay in front of the small phase
    }

    template <bool ROTARY>
    static __device__ __forceinline__ void run(WRing& r, SmemS& s, const int (&tiles)[NTW], int kc0) {
        const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
        f32x4 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 xh[KCH], xl[KCH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = u / KCN, kc = kc0 + (u - t * KCN);
            if (KCN > 1 || u == 0) {   // operand fragments of this K chunk from the planes its producer wrote
#pragma unroll
                for (int ks = 0; ks < KCH; ++ks) {
                    const int k0 = kc * 32 * KCH + ks * 32 + g * 8;
                    xh[ks] = *reinterpret_cast<const bf16x8*>(&s.xh[fr][k0]);
                    xl[ks] = *reinterpret_cast<const bf16x8*>(&s.xl[fr][k0]);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KCH; ++ks) {
                if (F16) {
                    const f16x8 wv = __builtin_bit_cast(f16x8, r.h[u][ks]);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, __builtin_bit_cast(f16x8, xl[ks]), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, __builtin_bit_cast(f16x8, xh[ks]), acc[t], 0, 0, 0);
                } else {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.l[u][ks], xh[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.h[u][ks], xl[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.h[u][ks], xh[ks], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {          // D[row = 4g + r][col = fr]: 4 consecutive output columns of token row fr
            const int col = tiles[t] * 16 + g * 4;
            f32x4 v = acc[t];
            if constexpr (ROTARY) {
                if (col < INNER + DH && (col & 63) < ROT) {
                    const int pos = fr % 3, d = col & 63;
                    const float c0 = s.rc[pos * ROT + d], s0 = s.rs[pos * ROT + d];
                    const float c1 = s.rc[pos * ROT + d + 2], s1 = s.rs[pos * ROT + d + 2];
                    v = (f32x4){v[0] * c0 - v[1] * s0, v[1] * c0 + v[0] * s0, v[2] * c1 - v[3] * s1, v[3] * c1 + v[2] * s1};
                }
            }
            *reinterpret_cast<f32x4*>(&s.y[fr][col]) = v;
        }
    }
};

struct Xch {
    unsigned long long* mine;          // [2 parities][XCH_VALS] granules this half writes
    const unsigned long long* theirs;  // the partner's
    unsigned long long* err;           // header word 1
    unsigned* status;                  // the process's status word (may be null)
    unsigned epoch;                    // low 16 bits of the launch epoch
    unsigned seq;                      // exchanges done so far in this launch
    bool dead;                         // this thread has given up on the partner: NaN from now on, no more spinning
};

// the partial sums s.y[0 .. R)[0 .. 128) out as granules (all threads)
__device__ __forceinline__ void xch_send(const SmemS& s, int R, const Xch& x) {
    const unsigned tag = (x.epoch << 16) | (x.seq + 1);
    unsigned long long* dst = x.mine + (x.seq & 1) * XCH_VALS;
    for (int i = threadIdx.x; i < R * DIM; i += NT) {
        const float v = s.y[i >> 7][i & 127];
        const unsigned long long gr = ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, v);
        __hip_atomic_store(dst + i, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// waves 4-7: the partner's partial sums into registers (3 per lane); bounded spin.  The three granules of a lane are looked at
// TOGETHER in every round (one after the other each can cost its own round trip through the fabric; in the persistent
// FaceFormer decode, where a thread waits for 4-16 granules, that was 4 us of a 26-us frame - here, with 3 granules that are
// mostly there already, the launch stayed at 7.7-7.8 ms).
__device__ __forceinline__ void xch_poll(int R, Xch& x, float (&v)[3]) {
    const unsigned tag = (x.epoch << 16) | (x.seq + 1);
    const unsigned long long* src = x.theirs + (x.seq & 1) * XCH_VALS;
    const int j = threadIdx.x - NT / 2;
    unsigned long long gr[3];
    unsigned pending = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = j + k * (NT / 2);
        gr[k] = 0ull;
        if (i < R * DIM && !x.dead) {
            gr[k] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pending |= 1u << k;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (((pending >> k) & 1u) && (unsigned)(gr[k] >> 32) == tag) pending &= ~(1u << k);
    unsigned spins = 0;
    while (pending && !x.dead) {
        if (++spins > SPIN_LIMIT) {
            __hip_atomic_store(x.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x.status) __hip_atomic_store(x.status + AVI_STATUS_PAIR_TIMEOUT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            x.dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if ((pending >> k) & 1u) gr[k] = __hip_atomic_load(src + j + k * (NT / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (((pending >> k) & 1u) && (unsigned)(gr[k] >> 32) == tag) pending &= ~(1u << k);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = j + k * (NT / 2);
        v[k] = i < R * DIM ? (x.dead ? __builtin_nanf("") : __builtin_bit_cast(float, (unsigned)gr[k])) : 0.f;
    }
}
__device__ __forceinline__ void xch_add(SmemS& s, int R, const float (&v)[3]) {
    const int j = threadIdx.x - NT / 2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = j + k * (NT / 2);
        if (i < R * DIM) s.y[i >> 7][i & 127] += v[k];
    }
}

// One exchange, then the prefetch of the next matrix.  PF(): the calling wave's prefetch.
// AVI_PAIR_OVERLAP (diagnostic): waves 0-3 prefetch WHILE waves 4-7 poll.  Measured slower: the polls then queue behind the
// prefetch's 128 KB in the CU's shared vector-memory path (5.2-6.6 k cycles per exchange, the "both endpoints streaming" price
// of MI355X_MICROARCH.md's handoff-1to1 row) - with the path idle a granule round trip costs about half of that.
#ifdef AVI_PAIR_OVERLAP
#define AVI_PAIR_EXCHANGE(PF)                                                                     \
    do {                                                                                          \
        float xv_[3];                                                                             \
        xch_send(s, R, x);                                                                        \
        if (wave < 4) { PF; } else { xch_poll(R, x, xv_); }                                       \
        __syncthreads();                 /* every sender has read s.y */                          \
        if (wave >= 4) { xch_add(s, R, xv_); PF; }                                                \
        ++x.seq;                                                                                  \
        __syncthreads();                                                                          \
    } while (0)
#else
#define AVI_PAIR_EXCHANGE(PF)                                                                     \
    do {                                                                                          \
        float xv_[3];                                                                             \
        xch_send(s, R, x);                                                                        \
        if (wave >= 4) xch_poll(R, x, xv_);                                                       \
        __syncthreads();                 /* every sender has read s.y */                          \
        if (wave >= 4) xch_add(s, R, xv_);                                                        \
        ++x.seq;                                                                                  \
        PF;                                                                                       \
        __syncthreads();                                                                          \
    } while (0)
#endif

template <int FF16>
__device__ __forceinline__ void denoise_pair(const PriorArgs& a, SmemS& s, int S, int hf, Xch& x STAMP_PARAMS) {
    constexpr int ATT = FF16 == 2 ? 1 : 0, FFM = FF16 > 0 ? 1 : 0;
    const AviPriorWeights& w = a.w;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = 3 * S;
    // this wave's column tiles (16 columns each) of the N-split matrices; the K-split ones take chunks 2 hf, 2 hf + 1
    const int t_qkv[3] = {16 * hf + wave, 16 * hf + 8 + wave, 32 + wave};            // 2 q tiles of its heads + 1 of k | v
    const int t_ff1[4] = {16 * hf + wave, 16 * hf + 8 + wave, 32 + 16 * hf + wave, 32 + 16 * hf + 8 + wave};
    const int t_one[1] = {wave};
    const int kc0 = 2 * hf;
    typedef LinP<DIM, NQKV, ATT, 3, 1> LQkv;
    typedef LinP<INNER, DIM, ATT, 1, 2> LOut;
    typedef LinP<DIM, 2 * FFI, FFM, 4, 1> LFf1;
    typedef LinP<FFI, DIM, FFM, 1, 2> LFf2;
    typedef LinP<DIM, DIM, ATT, 1, 1> LProj;
    bool pending = false;
    WRing ring;
    LQkv::prefetch(a.p.layer[0].qkv_hi, a.p.layer[0].qkv_lo, t_qkv, 0, ring);
    for (int l = 0; l < w.depth; ++l) {
        const AviPriorLayerPlanes& P = a.p.layer[l];
        // ---- A: residual += previous FF output (complete: exchange 2 of the previous layer); attention pre-LN
        if (wave < R) {
            const int r = wave;
            float va = s.tok[r][lane], vb = s.tok[r][lane + 64];
            if (pending) {
                va += s.y[r][lane];
                vb += s.y[r][lane + 64];
                s.tok[r][lane] = va;
                s.tok[r][lane + 64] = vb;
            }
            ln_row(va, vb, s.gain[l][0], lane, false);
            put_x<ATT>(s, r, lane, va);
            put_x<ATT>(s, r, lane + 64, vb);
        }
        __syncthreads();
        STAMP(0);
        // ---- B: q of this half's heads | k | v
        LQkv::template run<true>(ring, s, t_qkv, 0);
        __syncthreads();
        STAMP(1);
        LOut::prefetch(P.out_hi, P.out_lo, t_one, kc0, ring);
        // ---- C: attention, wave = (head of this half, sample)
        {
            const int h = 4 * hf + (wave & 3), sm = wave >> 2;
            if (sm < S) {
                const float nk = s.nkv[l][lane], nv = s.nkv[l][DH + lane];
                const float ik0 = s.nkinv[l];
                float kd[3], vd[3], ik[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    kd[j] = s.y[3 * sm + j][INNER + lane];
                    vd[j] = s.y[3 * sm + j][INNER + DH + lane];
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) ik[j] = 1.f / fmaxf(sqrtf(wave_sum_u(kd[j] * kd[j])), 1e-12f);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float q = s.y[3 * sm + i][h * DH + lane];
                    const float iq = 16.0f / fmaxf(sqrtf(wave_sum_u(q * q)), 1e-12f);
                    const float* rb = &s.relb[(h * 3 + i) * 4];
                    const float s0 = wave_sum_u(q * nk) * iq * ik0 + rb[0];
                    const float s1 = wave_sum_u(q * kd[0]) * iq * ik[0] + rb[1];
                    const float s2 = wave_sum_u(q * kd[1]) * iq * ik[1] + rb[2];
                    const float s3 = wave_sum_u(q * kd[2]) * iq * ik[2] + rb[3];
                    const float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
                    const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx), e2 = __expf(s2 - mx), e3 = __expf(s3 - mx);
                    put_x<ATT>(s, 3 * sm + i, h * DH + lane, (e0 * nv + e1 * vd[0] + e2 * vd[1] + e3 * vd[2]) / (e0 + e1 + e2 + e3));
                }
            }
        }
        __syncthreads();
        STAMP(2);
        // ---- D: to_out.0 over this half's 256 input columns -> partial sums in s.y[.][0..127]
        LOut::template run<false>(ring, s, t_one, kc0);
        __syncthreads();
        STAMP(3);
        AVI_PAIR_EXCHANGE(LFf1::prefetch(P.w1_hi, P.w1_lo, t_ff1, 0, ring));
        STAMP(10);
        // ---- E: to_out.1 LayerNorm, residual, FF pre-LN
        if (wave < R) {
            const int r = wave;
            float va = s.y[r][lane], vb = s.y[r][lane + 64];
            ln_row(va, vb, s.gain[l][1], lane, false);
            va += s.tok[r][lane];
            vb += s.tok[r][lane + 64];
            s.tok[r][lane] = va;
            s.tok[r][lane + 64] = vb;
            ln_row(va, vb, s.gain[l][2], lane, false);
            put_x<FFM>(s, r, lane, va);
            put_x<FFM>(s, r, lane + 64, vb);
        }
        __syncthreads();
        STAMP(4);
        // ---- F: this half's 256 value columns and their 256 gate columns
        LFf1::template run<false>(ring, s, t_ff1, 0);
        __syncthreads();
        STAMP(5);
        LFf2::prefetch(P.w2_hi, P.w2_lo, t_one, kc0, ring);
        // ---- G: SwiGLU on this half's features (columns 256 hf .. of the ff2 input)
        for (int o = tid; o < R * (FFI / 2); o += NT) {
            const int m = o / (FFI / 2), c = (FFI / 2) * hf + (o - m * (FFI / 2));
            put_x<FFM>(s, m, c, s.y[m][c] * silu(s.y[m][FFI + c]));
        }
        __syncthreads();
        STAMP(6);
        // ---- H: FF out over this half's 256 input columns -> partial sums
        LFf2::template run<false>(ring, s, t_one, kc0);
        __syncthreads();
        STAMP(7);
        if (l + 1 < w.depth) {
            const AviPriorLayerPlanes& Pn = a.p.layer[l + 1];
            AVI_PAIR_EXCHANGE(LQkv::prefetch(Pn.qkv_hi, Pn.qkv_lo, t_qkv, 0, ring));
        } else {
            AVI_PAIR_EXCHANGE(LProj::prefetch(a.p.proj_hi, a.p.proj_lo, t_one, 0, ring));
        }
        STAMP(11);
        pending = true;
    }
    // final stable LayerNorm + project_out: both halves in full (128 x 128, 64 KB)
    if (wave < R) {
        const int r = wave;
        float va = s.tok[r][lane] + s.y[r][lane], vb = s.tok[r][lane + 64] + s.y[r][lane + 64];
        ln_row(va, vb, s.fin_g, lane, true);
        put_x<ATT>(s, r, lane, va);
        put_x<ATT>(s, r, lane + 64, vb);
    }
    __syncthreads();
    LProj::template run<false>(ring, s, t_one, 0);
    __syncthreads();
    STAMP(8);
}

template <int FF16>
__global__ __launch_bounds__(NT, 2) void prior_sample_pair_kernel(const PriorArgs args_by_value,
                                                                  const float* __restrict__ text_embed,
                                                                  const float* __restrict__ noise,
                                                                  const float* __restrict__ temb, int B, float inv_scale,
                                                                  float* __restrict__ out,
                                                                  unsigned long long* __restrict__ xch_ws,
                                                                  unsigned* __restrict__ status, int fault) {
    const PriorArgs& a = kernarg();
    const AviPriorWeights& w = a.w;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SmemS& s = *reinterpret_cast<SmemS*>(smem_raw);
    const int tid = threadIdx.x;
    // blocks b and b + 8 are partners (one XCD under round-robin placement): 16 blocks = 8 pairs x 2 halves
    const int grp = blockIdx.x >> 4, within = blockIdx.x & 15;
    const int pair = grp * 8 + (within & 7), hf = within >> 3;
    const int npairs = (B + PAIR_S - 1) / PAIR_S;
    if (pair >= npairs) return;                    // both halves of a pair that does not exist leave together
    if (fault && hf == 1) return;                  // avi_debug_fault_inject: the partner that never answers
    const int b0 = pair * PAIR_S;
    const int Sg = min(PAIR_S, B - b0);
    Xch x;
    x.mine = xch_ws + XCH_HDR + ((long long)pair * 2 + hf) * 2 * XCH_VALS;
    x.theirs = xch_ws + XCH_HDR + ((long long)pair * 2 + (hf ^ 1)) * 2 * XCH_VALS;
    x.err = xch_ws + 1;
    x.status = status;
    x.epoch = (unsigned)(__hip_atomic_load(xch_ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu);
    x.seq = 0;
    x.dead = false;
    for (int i = tid; i < MR * XPS; i += NT) (&s.xh[0][0])[i] = (&s.xl[0][0])[i] = 0;
    for (int i = tid; i < MR * DIM; i += NT) (&s.tok[0][0])[i] = 0.f;
    for (int i = tid; i < w.depth * 3 * DIM; i += NT) {
        const int l = i / (3 * DIM), r = i - l * 3 * DIM, k = r / DIM, d = r - k * DIM;
        const AviPriorLayer& Ly = w.layer[l];
        s.gain[l][k][d] = (k == 0 ? Ly.norm_g : k == 1 ? Ly.out_g : Ly.ff_g)[d];
    }
    for (int i = tid; i < w.depth * 2 * DH; i += NT) s.nkv[i / (2 * DH)][i % (2 * DH)] = w.layer[i / (2 * DH)].null_kv[i % (2 * DH)];
    for (int i = tid; i < DIM; i += NT) {
        s.fin_g[i] = w.final_g[i];
        s.lq[i] = w.learned_query[i];
    }
    for (int i = tid; i < 96; i += NT) {
        s.relb[i] = w.rel_bias[i];
        s.rc[i] = w.rot_cos[i];
        s.rs[i] = w.rot_sin[i];
    }
    if (tid < w.depth) {
        float q2 = 0.f;
        for (int d = 0; d < DH; ++d) q2 = fmaf(w.layer[tid].null_kv[d], w.layer[tid].null_kv[d], q2);
        s.nkinv[tid] = 1.f / fmaxf(sqrtf(q2), 1e-12f);
    }
    for (int i = tid; i < Sg * DIM; i += NT) s.xcur[i / DIM][i % DIM] = noise[(long long)(b0 + i / DIM) * DIM + i % DIM];
    __syncthreads();
    const int T = w.timesteps;
#ifdef AVI_PRIOR_STAMPS
    unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    for (int step = 0; step < T; ++step) {
        const int t = T - 1 - step;
        for (int i = tid; i < Sg * DIM; i += NT) {
            const int sm = i / DIM, d = i - sm * DIM;
            s.tok[3 * sm + 0][d] = text_embed[(long long)(b0 + sm) * DIM + d];
            s.tok[3 * sm + 1][d] = temb[t * DIM + d];
            s.tok[3 * sm + 2][d] = s.xcur[sm][d] + s.lq[d];
        }
        __syncthreads();
        STAMP(9);
        denoise_pair<FF16>(a, s, Sg, hf, x STAMP_ARGS);
        for (int i = tid; i < Sg * DIM; i += NT) {
            const int sm = i / DIM, d = i - sm * DIM;
            const float x0 = s.y[3 * sm + 2][d];
            float xn = w.coef1[t] * x0 + w.coef2[t] * s.xcur[sm][d];
            if (t > 0) xn += __expf(0.5f * w.logvar[t]) * noise[((long long)(1 + step) * B + b0 + sm) * DIM + d];
            s.xcur[sm][d] = xn;
        }
        __syncthreads();
    }
    if (hf == 0)
        for (int i = tid; i < Sg * DIM; i += NT) out[(long long)(b0 + i / DIM) * DIM + i % DIM] = s.xcur[i / DIM][i % DIM] * inv_scale;
#ifdef AVI_PRIOR_STAMPS
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 12; ++i) out[i] = (float)stamp_acc[i];   // diagnostic build: the result is overwritten
#endif
}

// after the sampler, in stream order: the next launch's epoch; every 2^16 launches the tag space wraps, so the slots are
// cleared (no granule of the previous cycle can match a new tag)
__global__ void prior_pair_epoch_kernel(unsigned long long* __restrict__ ws, long long slot_words) {
    __shared__ int wrap;
    if (threadIdx.x == 0) {
        const unsigned long long e = ws[0] + 1;
        if (blockIdx.x == 0) ws[0] = e;
        wrap = (e & 0xFFFFull) == 0;
    }
    __syncthreads();
    if (wrap)
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < slot_words; i += (long long)gridDim.x * blockDim.x)
            ws[XCH_HDR + i] = 0ull;
}

template <int FF16>
int launch_pair(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed, const float* noise, int B,
                float inv_scale, float* out, const float* temb, unsigned long long* ws, hipStream_t s) {
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(prior_sample_pair_kernel<FF16>), (int)sizeof(SmemS));
    PriorArgs args;
    args.w = *w;
    args.p = *p;
    const int npairs = (B + PAIR_S - 1) / PAIR_S;
    const int blocks = (npairs + 7) / 8 * 16;
    hipLaunchKernelGGL(prior_sample_pair_kernel<FF16>, dim3(blocks), dim3(NT), sizeof(SmemS), s, args, text_embed, noise, temb,
                       B, inv_scale, out, ws, avi_status_ptr(), avi_fault_injected() & AVI_FAULT_PAIR_PARTNER_ABSENT);
    // ONE block: the epoch word is read-modify-written once; the (rare) wrap clear is a loop of that block
    hipLaunchKernelGGL(prior_pair_epoch_kernel, dim3(1), dim3(256), 0, s, ws, (long long)npairs * 2 * 2 * XCH_VALS);
    return avi_launch_status();
}

}  // namespace

extern "C" long long avi_prior_pair_workspace_bytes(int B) {
    if (B <= 0) return 0;
    const long long npairs = (B + PAIR_S - 1) / PAIR_S;
    return 8 * (XCH_HDR + npairs * 2 * 2 * XCH_VALS);
}

// Paired sampler: same contract as avi_prior_sample_batched_tab (time table built by avi_prior_time_table) plus `workspace`:
// avi_prior_pair_workspace_bytes(B) bytes of device memory, zero-filled ONCE by the caller and then left to the library (it
// carries the launch epoch); one launch at a time may use it (stream order).  A partner that never answers (bounded spin)
// turns the pair's outputs into NaN and raises workspace[1] (u64) and AVI_STATUS_PAIR_TIMEOUT (avi_talking.h).
extern "C" int avi_prior_sample_paired(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                       const float* noise, int B, float inv_scale, float* out, const float* temb_table,
                                       void* workspace, void* stream) {
    if (!w || !p || !text_embed || !noise || !out || !temb_table || !workspace || B <= 0) return AVI_EINVAL;
    if (reinterpret_cast<uintptr_t>(workspace) & 7) return AVI_EINVAL;
    if (w->depth < 1 || w->depth > AVI_PRIOR_MAX_DEPTH || !p->proj_hi) return AVI_EINVAL;
    // two exchanges per layer and step, numbered from 1 in 16 bits of the granule tag
    if (w->timesteps < 1 || 2ll * w->depth * w->timesteps >= 65535) return AVI_EINVAL;
    const bool attn16 = p->proj_lo == nullptr, ff16 = p->layer[0].w1_lo == nullptr;
    if (attn16 && !ff16) return AVI_EINVAL;
    for (int l = 0; l < w->depth; ++l) {
        const AviPriorLayerPlanes& P = p->layer[l];
        if (!P.qkv_hi || !P.out_hi || !P.w1_hi || !P.w2_hi) return AVI_EINVAL;
        if ((P.qkv_lo == nullptr) != attn16 || (P.out_lo == nullptr) != attn16) return AVI_EINVAL;
        if ((P.w1_lo == nullptr) != ff16 || (P.w2_lo == nullptr) != ff16) return AVI_EINVAL;
    }
    // ONE instantiation in this code object (two sampler kernels in one code object ran 1.5x slower each, prior_mfma.inc):
    // the default plane formats - feed-forward matrices one fp16 plane, attention matrices bf16 hi / lo
    if (attn16 || !ff16) return AVI_EINVAL;
    return launch_pair<1>(w, p, text_embed, noise, B, inv_scale, out, temb_table, static_cast<unsigned long long*>(workspace),
                          static_cast<hipStream_t>(stream));
}
