// Batched DDPM sampler of the diffusion prior on the matrix cores.
//
// prior.hip gives every sample its own workgroup and streams the 8.3 MB of denoiser weights once per step PER SAMPLE
// (32 CUs and ~1.7 TB/s of fabric traffic for a batch of 32, which measurably slows the GEMMs running beside it).
// Here one workgroup (512 threads) carries up to 5 samples = 15 token rows = one 16-row MFMA tile, so the weights are
// streamed once per step per GROUP: 4-5x less traffic and 4-5x fewer CUs for the same latency.
//
//   * linears: y[16][N] = x[16][K] . W^T on v_mfma_f32_16x16x32_bf16, 3-term split (fp32-grade).  Weights are bf16
//     hi/lo planes re-laid out FRAGMENT-MAJOR on the host, [N/16][K/32][64 lanes][8]: lane (c = l&15, g = l>>4) of
//     block (tile, kstep) holds W[16 tile + c][32 kstep + 8 g .. +7], so each wave-instruction is one contiguous
//     1-KiB read; activations are split when the fragment is built from LDS.  W is the "A" operand so a lane ends up with
//     4 consecutive output columns of one token row (one ds_write_b128).
//   * small phases (LayerNorms, the 3x4 cosine-sim attention with rotary / null kv / T5 bias, SwiGLU, DDPM update)
//     are the fp32 code of prior.hip looped over the samples of the group.
// Same numerics contract as prior.hip (tests compare both with the oracle).
#include "common.h"

namespace {

constexpr int DIM = 128, DH = 64, INNER = 512, FFI = 512, ROT = 32, NQKV = 640;   // 8 heads = 8 waves
constexpr int NT = 512, MR = 16, SMAX = 5;
constexpr int XS = 516;    // row stride of the linear-input buffer (floats): 512 + 4 keeps ds_read_b128 conflict-free
constexpr int YS = 1028;   // row stride of the linear-output buffer

struct PriorArgs {
    AviPriorWeights w;
    AviPriorPlanes p;
};

struct Smem {
    float tok[MR][DIM];   // residual stream of the group
    float x[MR][XS];      // input of the next linear
    float y[MR][YS];      // output of the last linear
    float xcur[SMAX][DIM];
    // small per-layer vectors cached once per launch: a global load in a small phase would have to wait (vmcnt is
    // in-order) for the weight fragments prefetched just before it
    float gain[AVI_PRIOR_MAX_DEPTH][3][DIM];   // norm.g, to_out.1.g, ff 0.g
    float nkv[AVI_PRIOR_MAX_DEPTH][2 * DH];
    float fin_g[DIM], lq[DIM];
    float relb[96], rc[96], rs[96];
    float nkinv[AVI_PRIOR_MAX_DEPTH];          // 1/|null key| per layer
    float inv[MR][9];                          // 1/|q_h| (8 heads) and 1/|k| of every token row
    float sc[SMAX][8][3][4];                   // attention scores / probabilities
};

__device__ __forceinline__ const PriorArgs& kernarg() {
    return *(const PriorArgs*)__builtin_amdgcn_kernarg_segment_ptr();
}

// y[m][n] = sum_k x[m][k] * W[n][k]   (m < 16).  N/16 column tiles are dealt round-robin to the 8 waves; a wave's work is
// a list of UNITS = (column tile, 128-wide K chunk), each 8 fragment loads (4 k-steps x hi/lo, 16 B per lane) and 12
// MFMAs.  DEPTH units of loads are kept in flight in registers; `prefetch` issues the first DEPTH units BEFORE the small
// phase that precedes the linear, so their latency hides behind it, and `run` consumes units while refilling the ring.
constexpr int DEPTH = 4, KCH = 4;
struct WRing {
    bf16x8 h[DEPTH][KCH], l[DEPTH][KCH];
};

template <int K, int N>
struct Lin {
    static constexpr int NTW = N / 16 / 8;     // column tiles per wave: 5 (640), 8 (1024), 1 (128)
    static constexpr int KC = K / (32 * KCH);  // K chunks: 1 (K=128) or 4 (K=512)
    static constexpr int U = NTW * KC;         // units per wave

    static __device__ __forceinline__ void load_unit(const uint16_t* __restrict__ Whi, const uint16_t* __restrict__ Wlo,
                                                     int u, bf16x8 (&h)[KCH], bf16x8 (&l)[KCH]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
        const int t = u / KC, kc = u - t * KC;
        // fragment-major planes: [column tile][k-step][lane][8]  -> every wave-instruction reads 1 KiB contiguous
        const long long o = (((long long)(wave + 8 * t) * (K / 32) + kc * KCH) * 64 + lane) * 8;
        (void)fr; (void)g;
#pragma unroll
        for (int ks = 0; ks < KCH; ++ks) {
            h[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Whi + o + ks * 512));
            l[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wlo + o + ks * 512));
        }
    }
    static __device__ __forceinline__ void prefetch(const uint16_t* __restrict__ Whi, const uint16_t* __restrict__ Wlo,
                                                    WRing& r) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            if (u < U) load_unit(Whi, Wlo, u, r.h[u], r.l[u]);
        // hipcc's scheduler otherwise sinks these loads down to their first use (after the small phase), which
        // turns the ring into a load -> wait -> MFMA chain; nothing may cross this point
        __builtin_amdgcn_sched_barrier(0);
    }
    // ROTARY: apply the rotary embedding (first 32 dims of every 64-wide q head and of k, interleaved pairs,
    // position = token index row % 3) to the outputs before they are written: a lane holds 4 consecutive columns
    // of one row, i.e. two whole pairs.
    template <bool ROTARY = false>
    static __device__ __forceinline__ void run(const uint16_t* __restrict__ Whi, const uint16_t* __restrict__ Wlo,
                                               WRing& r, Smem& s) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
        f32x4 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 xh[KCH], xl[KCH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = u / KC, kc = u - t * KC;
            if (KC > 1 || u == 0) {   // activation fragments of this K chunk (K = 128: built once)
#pragma unroll
                for (int ks = 0; ks < KCH; ++ks) {
                    const float* xp = &s.x[fr][kc * 32 * KCH + ks * 32 + g * 8];
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xp), v1 = *reinterpret_cast<const f32x4*>(xp + 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xv = j < 4 ? v0[j] : v1[j - 4];
                        const __bf16 hi = (__bf16)xv;
                        xh[ks][j] = hi;
                        xl[ks][j] = (__bf16)(xv - (float)hi);
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < KCH; ++ks) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.l[u % DEPTH][ks], xh[ks], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.h[u % DEPTH][ks], xl[ks], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.h[u % DEPTH][ks], xh[ks], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (u + DEPTH < U) load_unit(Whi, Wlo, u + DEPTH, r.h[u % DEPTH], r.l[u % DEPTH]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // D[row = n_local = 4g + r][col = m = fr]
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = (wave + 8 * t) * 16 + g * 4;
            f32x4 v = acc[t];
            if (ROTARY && col < INNER + DH && (col & 63) < ROT) {
                const int pos = fr % 3, d = col & 63;
                const float c0 = s.rc[pos * ROT + d], s0 = s.rs[pos * ROT + d];
                const float c1 = s.rc[pos * ROT + d + 2], s1 = s.rs[pos * ROT + d + 2];
                v = (f32x4){v[0] * c0 - v[1] * s0, v[1] * c0 + v[0] * s0, v[2] * c1 - v[3] * s1, v[3] * c1 + v[2] * s1};
            }
            *reinterpret_cast<f32x4*>(&s.y[fr][col]) = v;
        }
    }
};

__device__ __forceinline__ void ln_row(float& a, float& b, const float* gm, int lane, bool stable) {
    if (stable) {
        const float mx = wave_max(fmaxf(a, b));
        a /= mx;
        b /= mx;
    }
    const float mean = wave_sum(a + b) * (1.f / DIM);
    const float da = a - mean, db = b - mean;
    const float r = rsqrtf(wave_sum(da * da + db * db) * (1.f / DIM) + 1e-5f);
    a = da * r * gm[lane];
    b = db * r * gm[lane + 64];
}
__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float rotary64(float x, int pos, int lane, const float* rc, const float* rs) {
    const float partner = __shfl_xor(x, 1, 64);
    if (lane < ROT) x = x * rc[pos * ROT + lane] + ((lane & 1) ? partner : -partner) * rs[pos * ROT + lane];
    return x;
}
__device__ __forceinline__ float l2n4(float x) { return x / fmaxf(sqrtf(wave_sum(x * x)), 1e-12f) * 4.0f; }

// denoiser over the group's R = 3*S token rows held in s.tok; prediction of sample i lands in s.y[3i+2][0..127]
__device__ __forceinline__ void denoise(const PriorArgs& a, Smem& s, int S) {
    const AviPriorWeights& w = a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = 3 * S;
    bool pending = false;
    WRing ring;
    for (int l = 0; l < w.depth; ++l) {
        const AviPriorLayerPlanes& P = a.p.layer[l];
        Lin<DIM, NQKV>::prefetch(P.qkv_hi, P.qkv_lo, ring);
        // ---- A: residual += previous FF output; attention pre-LN  (row r -> wave r % 8)
        for (int r = wave; r < R; r += 8) {
            float va = s.tok[r][lane], vb = s.tok[r][lane + 64];
            if (pending) {
                va += s.y[r][lane];
                vb += s.y[r][lane + 64];
                s.tok[r][lane] = va;
                s.tok[r][lane + 64] = vb;
            }
            ln_row(va, vb, s.gain[l][0], lane, false);
            s.x[r][lane] = va;
            s.x[r][lane + 64] = vb;
        }
        __syncthreads();
        // ---- B: q | k | v
        Lin<DIM, NQKV>::template run<true>(P.qkv_hi, P.qkv_lo, ring, s);
        __syncthreads();
        Lin<INNER, DIM>::prefetch(P.out_hi, P.out_lo, ring);
        // ---- C: (rotary already applied by the qkv epilogue) the 3x4 cosine-sim attention of every (sample, head)
        //      at once, one thread per dot product and no
        //      cross-lane reductions (a wave-shuffle reduction per score made this phase cost 4.5 us per sample).
        //      Vector reads are skewed by the thread index so that 64-float rows do not collide on one LDS bank.
        // C2: 1/|.| of every q (R x 8) and k (R) vector; the null key's comes from the cache
        for (int o = tid; o < R * 9; o += NT) {
            const int r = o / 9, v = o - r * 9;
            const float* p = &s.y[r][v < 8 ? v * DH : INNER];
            float q2 = 0.f;
#pragma unroll 8
            for (int d = 0; d < DH; ++d) {
                const float t = p[(d + tid) & 63];
                q2 = fmaf(t, t, q2);
            }
            s.inv[r][v] = 1.f / fmaxf(sqrtf(q2), 1e-12f);
        }
        __syncthreads();
        // C3: scores.  thread = (sample, head, query i, key j); q was multiplied by 16 before l2norm in the reference,
        //     which cancels; both unit vectors are scaled by sqrt(16) -> x16 on the cosine.
        for (int o = tid; o < S * 96; o += NT) {
            const int sm = o / 96, rem = o - sm * 96, h = rem / 12, i = (rem % 12) >> 2, j = rem & 3;
            const float* qp = &s.y[3 * sm + i][h * DH];
            const float* kp = j == 0 ? s.nkv[l] : &s.y[3 * sm + j - 1][INNER];
            float dot = 0.f;
#pragma unroll 8
            for (int d = 0; d < DH; ++d) {
                const int dd = (d + tid) & 63;
                dot = fmaf(qp[dd], kp[dd], dot);
            }
            const float ik = j == 0 ? s.nkinv[l] : s.inv[3 * sm + j - 1][8];
            s.sc[sm][h][i][j] = dot * s.inv[3 * sm + i][h] * ik * 16.0f + s.relb[(h * 3 + i) * 4 + j];
        }
        __syncthreads();
        // C4: softmax (recomputed by every output thread: 4 exps) and P.V -> s.x (input of to_out)
        for (int o = tid; o < R * INNER; o += NT) {
            const int r = o >> 9, c = o & 511, h = c >> 6, d = c & 63, sm = r / 3, i = r - 3 * sm;
            const float* r4 = s.sc[sm][h][i];
            const float mx = fmaxf(fmaxf(r4[0], r4[1]), fmaxf(r4[2], r4[3]));
            const float e0 = __expf(r4[0] - mx), e1 = __expf(r4[1] - mx), e2 = __expf(r4[2] - mx), e3 = __expf(r4[3] - mx);
            s.x[r][c] = (e0 * s.nkv[l][DH + d] + e1 * s.y[3 * sm][INNER + DH + d] + e2 * s.y[3 * sm + 1][INNER + DH + d] +
                         e3 * s.y[3 * sm + 2][INNER + DH + d]) / (e0 + e1 + e2 + e3);
        }
        __syncthreads();
        // ---- D: to_out.0
        Lin<INNER, DIM>::run(P.out_hi, P.out_lo, ring, s);
        __syncthreads();
        Lin<DIM, 2 * FFI>::prefetch(P.w1_hi, P.w1_lo, ring);
        // ---- E: to_out.1 LayerNorm, residual, FF pre-LN
        for (int r = wave; r < R; r += 8) {
            float va = s.y[r][lane], vb = s.y[r][lane + 64];
            ln_row(va, vb, s.gain[l][1], lane, false);
            va += s.tok[r][lane];
            vb += s.tok[r][lane + 64];
            s.tok[r][lane] = va;
            s.tok[r][lane + 64] = vb;
            ln_row(va, vb, s.gain[l][2], lane, false);
            s.x[r][lane] = va;
            s.x[r][lane + 64] = vb;
        }
        __syncthreads();
        // ---- F: FF in (value | gate)
        Lin<DIM, 2 * FFI>::run(P.w1_hi, P.w1_lo, ring, s);
        __syncthreads();
        Lin<FFI, DIM>::prefetch(P.w2_hi, P.w2_lo, ring);
        // ---- G: SwiGLU
        for (int o = tid; o < R * FFI; o += NT) {
            const int m = o / FFI, c = o - m * FFI;
            s.x[m][c] = s.y[m][c] * silu(s.y[m][FFI + c]);
        }
        __syncthreads();
        // ---- H: FF out
        Lin<FFI, DIM>::run(P.w2_hi, P.w2_lo, ring, s);
        __syncthreads();
        pending = true;
    }
    Lin<DIM, DIM>::prefetch(a.p.proj_hi, a.p.proj_lo, ring);
    for (int r = wave; r < R; r += 8) {
        float va = s.tok[r][lane] + s.y[r][lane], vb = s.tok[r][lane + 64] + s.y[r][lane + 64];
        ln_row(va, vb, s.fin_g, lane, true);
        s.x[r][lane] = va;
        s.x[r][lane + 64] = vb;
    }
    __syncthreads();
    Lin<DIM, DIM>::run(a.p.proj_hi, a.p.proj_lo, ring, s);
    __syncthreads();
}

__global__ __launch_bounds__(NT, 2) void prior_sample_mfma_kernel(const PriorArgs args_by_value,
                                                                  const float* __restrict__ text_embed,
                                                                  const float* __restrict__ noise,
                                                                  const float* __restrict__ temb, int B, int S,
                                                                  float inv_scale, float* __restrict__ out) {
    const PriorArgs& a = kernarg();
    const AviPriorWeights& w = a.w;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * S;
    const int Sg = min(S, B - b0);                 // samples in this group
    // zero the padded rows once: they feed the MFMA as ordinary (ignored) rows
    for (int i = tid; i < MR * XS; i += NT) (&s.x[0][0])[i] = 0.f;
    for (int i = tid; i < MR * DIM; i += NT) (&s.tok[0][0])[i] = 0.f;
    for (int i = tid; i < w.depth * 3 * DIM; i += NT) {
        const int l = i / (3 * DIM), r = i - l * 3 * DIM, k = r / DIM, d = r - k * DIM;
        const AviPriorLayer& L = w.layer[l];
        s.gain[l][k][d] = (k == 0 ? L.norm_g : k == 1 ? L.out_g : L.ff_g)[d];
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
    for (int step = 0; step < T; ++step) {
        const int t = T - 1 - step;
        for (int i = tid; i < Sg * DIM; i += NT) {
            const int sm = i / DIM, d = i - sm * DIM;
            s.tok[3 * sm + 0][d] = text_embed[(long long)(b0 + sm) * DIM + d];
            s.tok[3 * sm + 1][d] = temb[t * DIM + d];
            s.tok[3 * sm + 2][d] = s.xcur[sm][d] + s.lq[d];
        }
        __syncthreads();
        denoise(a, s, Sg);
        for (int i = tid; i < Sg * DIM; i += NT) {
            const int sm = i / DIM, d = i - sm * DIM;
            const float x0 = s.y[3 * sm + 2][d];
            float xn = w.coef1[t] * x0 + w.coef2[t] * s.xcur[sm][d];
            if (t > 0) xn += __expf(0.5f * w.logvar[t]) * noise[((long long)(1 + step) * B + b0 + sm) * DIM + d];
            s.xcur[sm][d] = xn;
        }
        __syncthreads();
    }
    for (int i = tid; i < Sg * DIM; i += NT) out[(long long)(b0 + i / DIM) * DIM + i % DIM] = s.xcur[i / DIM][i % DIM] * inv_scale;
}

}  // namespace

// prior.hip
int avi_prior_time_table_launch(const AviPriorWeights* w, float* temb, hipStream_t s);

extern "C" int avi_prior_sample_batched(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                        const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                                        float* temb_scratch, void* stream) {
    if (!w || !p || !text_embed || !noise || !out || !temb_scratch || B <= 0) return AVI_EINVAL;
    if (samples_per_group < 1 || samples_per_group > SMAX) return AVI_EINVAL;
    if (w->depth < 1 || w->depth > AVI_PRIOR_MAX_DEPTH || !p->proj_hi || !p->proj_lo) return AVI_EINVAL;
    for (int l = 0; l < w->depth; ++l) {
        const AviPriorLayerPlanes& P = p->layer[l];
        if (!P.qkv_hi || !P.qkv_lo || !P.out_hi || !P.out_lo || !P.w1_hi || !P.w1_lo || !P.w2_hi || !P.w2_lo)
            return AVI_EINVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = avi_prior_time_table_launch(w, temb_scratch, s);
    if (rc != AVI_OK) return rc;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(prior_sample_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
        attr = true;
    }
    PriorArgs args;
    args.w = *w;
    args.p = *p;
    const int groups = (B + samples_per_group - 1) / samples_per_group;
    hipLaunchKernelGGL(prior_sample_mfma_kernel, dim3(groups), dim3(NT), sizeof(Smem), s, args, text_embed, noise,
                       temb_scratch, B, samples_per_group, inv_scale, out);
    return avi_launch_status();
}
