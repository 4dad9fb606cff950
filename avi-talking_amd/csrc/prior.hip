// Diffusion-prior denoiser (VersatileDiffusionPriorNetwork, models/diffusion_prior.py:169-313, over the
// dalle2 Attention / FeedForward / LayerNorm blocks) and the DDPM sampling loop
// (InstructDiffusionPrior.p_sample_loop_ddpm, models/diffusion_prior.py:329-367), fp32.
//
// The prior diffuses ONE 128-d style token per utterance; the transformer sees 3 tokens
// [text, time, noisy style] (+1 learned null key/value).  Samples are independent, so one workgroup
// (512 threads = 8 waves) owns one sample for the WHOLE loop: no launch or host round trip between the 100
// steps (the reference runs ~150 tiny kernels per step from Python).
//
// Per layer the kernel alternates 4 weight-streaming phases with 4 small phases, 8 barriers in all:
//   A  waves 0-2 (one per token row): fold the previous layer's FF partials into the residual, pre-LN
//   B  q|k|v = LN(x) . Wqkv        -- stream 320 KB
//   C  wave h = head h: reduce its q and the shared k/v, rotary, l2-norm, 3x4 scores + T5 bias, softmax, P.V
//   D  attn . Wout                 -- stream 256 KB
//   E  waves 0-2: reduce, LayerNorm (to_out.1), residual add, FF pre-LN
//   F  LN(x) . W1                  -- stream 512 KB
//   G  reduce + SwiGLU
//   H  h . W2                      -- stream 256 KB
// Weight matrices are fp32 [K][N]; a streaming phase assigns thread = (4 output columns, K slice) and issues up
// to 32 independent 16-byte loads per thread back to back (8 waves x 32 x 1 KiB = 256 KiB in flight per CU), then
// writes split-K partials to LDS; the next small phase reduces exactly the partials it needs.
#include "common.h"

namespace {

constexpr int DIM = 128, NTOK = 3, HEADS = 8, DH = 64, INNER = HEADS * DH, FFI = 512, ROT = 32;
constexpr int NQKV = INNER + 2 * DH;   // 640
constexpr int NT = 512;
#ifndef PRIOR_UB
#define PRIOR_UB 16
#endif
#ifndef PRIOR_WAVES_PER_SIMD
#define PRIOR_WAVES_PER_SIMD 2
#endif
constexpr int PART = 6144;             // max KS*3*N over the four linears

struct Smem {
    float tok[NTOK][DIM];     // residual stream
    float xn[NTOK][DIM];      // normed input of the next linear
    float att[NTOK][INNER];   // attention output / SwiGLU output (3 x 512)
    float part[PART];         // split-K partials [KS][3][N]
    float tmp[2 * DIM];
    float te[2][256];         // time-MLP hidden
};

template <int N> struct Split {
    static constexpr int NV = N / 4;
    static constexpr int KS = NT / NV;          // 3 (N=640), 16 (N=128), 2 (N=1024)
};

// part[ks][m][n] = sum_{k in slice ks} x[m][k] * Wt[k][n]
#ifdef AVI_PRIOR_DIAG
__device__ int g_diag;   // bit 0: skip weight loads (timing-only diagnostic build)
#endif
template <int K, int N>
__device__ __forceinline__ void linear3_partial(const float* __restrict__ Wt, const float* x, int xs, Smem& s) {
    constexpr int NV = Split<N>::NV, KS = Split<N>::KS;
    constexpr int KC = (K + KS - 1) / KS;
    static_assert(KS * N * 3 <= PART, "partial buffer too small");
    const int cq = threadIdx.x % NV, ks = threadIdx.x / NV;
    if (ks >= KS) return;
    const int k0 = ks * KC;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    const float4* wp = reinterpret_cast<const float4*>(Wt) + cq;
    constexpr int UB = KC < PRIOR_UB ? KC : PRIOR_UB;   // 16-B loads kept in flight per thread
#pragma unroll 1   // keep ONE batch of UB loads live: hipcc otherwise unrolls this loop and spills the batches
    for (int kb = 0; kb < KC; kb += UB) {
        float4 w[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = k0 + kb + u;
#ifdef AVI_PRIOR_DIAG
            if (g_diag & 1) { w[u] = make_float4(1e-3f, 2e-3f, -1e-3f, 5e-4f); continue; }
#endif
            w[u] = (kb + u < KC && k < K) ? wp[(long long)k * NV] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = k0 + kb + u;
            if (kb + u < KC && k < K) {
                const float x0 = x[k], x1 = x[xs + k], x2 = x[2 * xs + k];
                a0.x = fmaf(x0, w[u].x, a0.x); a0.y = fmaf(x0, w[u].y, a0.y); a0.z = fmaf(x0, w[u].z, a0.z); a0.w = fmaf(x0, w[u].w, a0.w);
                a1.x = fmaf(x1, w[u].x, a1.x); a1.y = fmaf(x1, w[u].y, a1.y); a1.z = fmaf(x1, w[u].z, a1.z); a1.w = fmaf(x1, w[u].w, a1.w);
                a2.x = fmaf(x2, w[u].x, a2.x); a2.y = fmaf(x2, w[u].y, a2.y); a2.z = fmaf(x2, w[u].z, a2.z); a2.w = fmaf(x2, w[u].w, a2.w);
            }
        }
    }
    float4* p = reinterpret_cast<float4*>(s.part + ks * 3 * N) + cq;
    p[0] = a0;
    p[NV] = a1;
    p[2 * NV] = a2;
}

template <int N>
__device__ __forceinline__ float part_sum(const Smem& s, int m, int n) {
    float a = 0.f;
#pragma unroll
    for (int ks = 0; ks < Split<N>::KS; ++ks) a += s.part[(ks * 3 + m) * N + n];
    return a;
}

// dalle2 LayerNorm of one row held as (a = col lane, b = col lane+64) by one wave; gain only, eps 1e-5.
__device__ __forceinline__ void ln_row(float& a, float& b, const float* __restrict__ g, int lane, bool stable) {
    if (stable) {
        const float mx = wave_max(fmaxf(a, b));
        a /= mx;
        b /= mx;
    }
    const float mean = wave_sum(a + b) * (1.f / DIM);
    const float da = a - mean, db = b - mean;
    const float r = rsqrtf(wave_sum(da * da + db * db) * (1.f / DIM) + 1e-5f);
    a = da * r * g[lane];
    b = db * r * g[lane + 64];
}

__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }

// rotary on the first 32 dims (interleaved pairs) of a 64-vector held one element per lane
__device__ __forceinline__ float rotary64(float x, int pos, int lane, const float* __restrict__ rc,
                                          const float* __restrict__ rs) {
    const float partner = __shfl_xor(x, 1, 64);
    if (lane < ROT) x = x * rc[pos * ROT + lane] + ((lane & 1) ? partner : -partner) * rs[pos * ROT + lane];
    return x;
}
__device__ __forceinline__ float l2n4(float x) {   // 4 * x / max(|x|, 1e-12)  (l2norm then * sqrt(cosine_sim_scale))
    return x / fmaxf(sqrtf(wave_sum(x * x)), 1e-12f) * 4.0f;
}

// One denoiser evaluation: s.tok holds [text, time, noisy+query]; on return s.tmp[0..127] = prediction.
__device__ __forceinline__ void denoise(const AviPriorWeights& w, Smem& s) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool pending = false;   // FF partials of the previous layer still to be folded into the residual
    for (int l = 0; l < w.depth; ++l) {
        const AviPriorLayer& L = w.layer[l];
        // ---- A: residual += previous FF output; attention pre-LN
        if (wave < NTOK) {
            float a = s.tok[wave][lane], b = s.tok[wave][lane + 64];
            if (pending) {
                a += part_sum<DIM>(s, wave, lane);
                b += part_sum<DIM>(s, wave, lane + 64);
                s.tok[wave][lane] = a;
                s.tok[wave][lane + 64] = b;
            }
            ln_row(a, b, L.norm_g, lane, false);
            s.xn[wave][lane] = a;
            s.xn[wave][lane + 64] = b;
        }
        __syncthreads();
        // ---- B: q | k | v
        linear3_partial<DIM, NQKV>(L.wqkv, &s.xn[0][0], DIM, s);
        __syncthreads();
        // ---- C: wave = head (multi-query: the single K/V head is recomputed by every wave)
        {
            const int h = wave;
            float kn[4], vv[4];
            kn[0] = l2n4(L.null_kv[lane]);
            vv[0] = L.null_kv[DH + lane];
#pragma unroll
            for (int i = 0; i < NTOK; ++i) {
                kn[1 + i] = l2n4(rotary64(part_sum<NQKV>(s, i, INNER + lane), i, lane, w.rot_cos, w.rot_sin));
                vv[1 + i] = part_sum<NQKV>(s, i, INNER + DH + lane);
            }
#pragma unroll
            for (int i = 0; i < NTOK; ++i) {
                // q * cosine_sim_scale, rotary, l2norm, * sqrt(scale)   (dalle2 Attention)
                const float q = l2n4(rotary64(part_sum<NQKV>(s, i, h * DH + lane) * 16.0f, i, lane, w.rot_cos, w.rot_sin));
                float sc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[j] = wave_sum(q * kn[j]) + w.rel_bias[(h * NTOK + i) * 4 + j];
                const float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
                const float e0 = __expf(sc[0] - mx), e1 = __expf(sc[1] - mx), e2 = __expf(sc[2] - mx), e3 = __expf(sc[3] - mx);
                const float inv = 1.f / (e0 + e1 + e2 + e3);
                s.att[i][h * DH + lane] = (e0 * vv[0] + e1 * vv[1] + e2 * vv[2] + e3 * vv[3]) * inv;
            }
        }
        __syncthreads();
        // ---- D: to_out.0
        linear3_partial<INNER, DIM>(L.wout, &s.att[0][0], INNER, s);
        __syncthreads();
        // ---- E: to_out.1 LayerNorm, residual, FF pre-LN
        if (wave < NTOK) {
            float a = part_sum<DIM>(s, wave, lane), b = part_sum<DIM>(s, wave, lane + 64);
            ln_row(a, b, L.out_g, lane, false);
            a += s.tok[wave][lane];
            b += s.tok[wave][lane + 64];
            s.tok[wave][lane] = a;
            s.tok[wave][lane + 64] = b;
            ln_row(a, b, L.ff_g, lane, false);
            s.xn[wave][lane] = a;
            s.xn[wave][lane + 64] = b;
        }
        __syncthreads();
        // ---- F: FF in (value | gate)
        linear3_partial<DIM, 2 * FFI>(L.w1, &s.xn[0][0], DIM, s);
        __syncthreads();
        // ---- G: reduce + SwiGLU
        for (int o = tid; o < NTOK * FFI; o += NT) {
            const int m = o / FFI, c = o - m * FFI;
            s.att[m][c] = part_sum<2 * FFI>(s, m, c) * silu(part_sum<2 * FFI>(s, m, FFI + c));
        }
        __syncthreads();
        // ---- H: FF out (partials folded into the residual by the next phase A / the epilogue)
        linear3_partial<FFI, DIM>(L.w2, &s.att[0][0], INNER, s);
        __syncthreads();
        pending = true;
    }
    // epilogue: residual, stable LayerNorm, project_out, keep the last token
    if (wave < NTOK) {
        float a = s.tok[wave][lane], b = s.tok[wave][lane + 64];
        if (pending) {
            a += part_sum<DIM>(s, wave, lane);
            b += part_sum<DIM>(s, wave, lane + 64);
        }
        ln_row(a, b, w.final_g, lane, true);
        s.xn[wave][lane] = a;
        s.xn[wave][lane + 64] = b;
    }
    __syncthreads();
    linear3_partial<DIM, DIM>(w.wproj, &s.xn[0][0], DIM, s);
    __syncthreads();
    if (tid < DIM) s.tmp[tid] = part_sum<DIM>(s, 2, tid);
    __syncthreads();
}

// time embedding: SinusoidalPosEmb table row t -> MLP 128 -> 256 -> 256 -> 128 (SiLU); result in dst[0..127] (LDS or global)
__device__ __forceinline__ void time_embed(const AviPriorWeights& w, int t, Smem& s, float* dst) {
    const int tid = threadIdx.x;
    if (tid < DIM) s.tmp[tid] = w.time_table[t * DIM + tid];
    __syncthreads();
    if (tid < 256) {
        float a = w.t_b0[tid];
#pragma unroll 16
        for (int k = 0; k < DIM; ++k) a = fmaf(s.tmp[k], w.t_w0[k * 256 + tid], a);
        s.te[0][tid] = silu(a);
    }
    __syncthreads();
    if (tid < 256) {
        float a = w.t_b1[tid];
#pragma unroll 16
        for (int k = 0; k < 256; ++k) a = fmaf(s.te[0][k], w.t_w1[k * 256 + tid], a);
        s.te[1][tid] = silu(a);
    }
    __syncthreads();
    if (tid < DIM) {
        float a = w.t_b2[tid];
#pragma unroll 16
        for (int k = 0; k < 256; ++k) a = fmaf(s.te[1][k], w.t_w2[k * DIM + tid], a);
        dst[tid] = a;
    }
    __syncthreads();
}

// The time embedding depends on t and the weights only: one block per timestep fills temb[T][128] once per launch
// of the sampler instead of once per step per sample.
// The weight table is the FIRST kernel argument.  Indexing `w.layer[l]` with a runtime l on the by-value copy makes
// hipcc spill the whole struct to scratch (1.2 KB per lane); reading it through the kernarg segment pointer keeps
// it in constant memory (scalar loads).
__device__ __forceinline__ const AviPriorWeights& kernarg_weights() {
    return *(const AviPriorWeights*)__builtin_amdgcn_kernarg_segment_ptr();
}

__global__ __launch_bounds__(NT) void prior_time_table_kernel(const AviPriorWeights w_arg, float* __restrict__ temb) {
    const AviPriorWeights& w = kernarg_weights();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    time_embed(w, blockIdx.x, s, temb + (long long)blockIdx.x * DIM);
}

// mode 0: one forward at per-sample timestep t[b] with optional cond-drop masks -> pred[b]
// mode 1: full DDPM loop t = T-1..0 with noise[0] = x_T, noise[1+k] = z of the k-th step -> out[b] = x_0 * inv_scale
__global__ __launch_bounds__(NT, PRIOR_WAVES_PER_SIMD) void prior_kernel(const AviPriorWeights w_arg, const float* __restrict__ text_embed,
                                                    const float* __restrict__ x_in, const int* __restrict__ t_in,
                                                    const unsigned char* __restrict__ brain_keep,
                                                    const unsigned char* __restrict__ image_keep,
                                                    const float* __restrict__ noise, const float* __restrict__ temb,
                                                    int B, int mode, float inv_scale, float* __restrict__ out) {
    const AviPriorWeights& w = kernarg_weights();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ float xcur[DIM];
    const bool keep_b = !brain_keep || brain_keep[b];
    if (mode == 0) {
        const bool keep_i = !image_keep || image_keep[b];
        if (tid < DIM) xcur[tid] = keep_i ? x_in[(long long)b * DIM + tid] : w.null_image[tid];
        __syncthreads();
        time_embed(w, t_in[b], s, s.tok[1]);
        if (tid < DIM) {
            s.tok[0][tid] = keep_b ? text_embed[(long long)b * DIM + tid] : w.null_brain[tid];
            s.tok[2][tid] = xcur[tid] + w.learned_query[tid];
        }
        __syncthreads();
        denoise(w, s);
        if (tid < DIM) out[(long long)b * DIM + tid] = s.tmp[tid];
        return;
    }
    const int T = w.timesteps;
    if (tid < DIM) xcur[tid] = noise[(long long)b * DIM + tid];
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = T - 1 - step;
        if (tid < DIM) {
            s.tok[0][tid] = text_embed[(long long)b * DIM + tid];
            s.tok[1][tid] = temb[t * DIM + tid];
            s.tok[2][tid] = xcur[tid] + w.learned_query[tid];
        }
        __syncthreads();
        denoise(w, s);
        if (tid < DIM) {
            // q_posterior mean + sigma * z  (models/diffusion_prior.py:331-341; predict_x_start)
            const float x0 = s.tmp[tid];
            float xn = w.coef1[t] * x0 + w.coef2[t] * xcur[tid];
            if (t > 0) xn += __expf(0.5f * w.logvar[t]) * noise[((long long)(1 + step) * B + b) * DIM + tid];
            xcur[tid] = xn;
        }
        __syncthreads();
    }
    if (tid < DIM) out[(long long)b * DIM + tid] = xcur[tid] * inv_scale;
}

int check_weights(const AviPriorWeights* w) {
    if (!w || w->depth < 1 || w->depth > AVI_PRIOR_MAX_DEPTH || w->timesteps < 1) return AVI_EINVAL;
    if (!w->time_table || !w->t_w0 || !w->t_b0 || !w->t_w1 || !w->t_b1 || !w->t_w2 || !w->t_b2) return AVI_EINVAL;
    if (!w->learned_query || !w->null_brain || !w->null_image || !w->rel_bias || !w->rot_cos || !w->rot_sin)
        return AVI_EINVAL;
    if (!w->final_g || !w->wproj || !w->coef1 || !w->coef2 || !w->logvar) return AVI_EINVAL;
    for (int l = 0; l < w->depth; ++l) {
        const AviPriorLayer& L = w->layer[l];
        if (!L.norm_g || !L.wqkv || !L.null_kv || !L.wout || !L.out_g || !L.ff_g || !L.w1 || !L.w2) return AVI_EINVAL;
    }
    return AVI_OK;
}

void set_attr() {
    static AviLdsGrant grant_prior, grant_table;
    grant_prior.ensure(reinterpret_cast<const void*>(prior_kernel), (int)sizeof(Smem));
    grant_table.ensure(reinterpret_cast<const void*>(prior_time_table_kernel), (int)sizeof(Smem));
}

}  // namespace

// shared with prior_mfma.hip
int avi_prior_time_table_launch(const AviPriorWeights* w, float* temb, hipStream_t s) {
    if (check_weights(w) != AVI_OK || !temb) return AVI_EINVAL;
    set_attr();
    hipLaunchKernelGGL(prior_time_table_kernel, dim3(w->timesteps), dim3(NT), sizeof(Smem), s, *w, temb);
    return avi_launch_status();
}

extern "C" int avi_prior_forward(const AviPriorWeights* w, const float* x_t, const int* t, const float* text_embed,
                                 const unsigned char* brain_keep, const unsigned char* image_keep, int B,
                                 float* pred, void* stream) {
    if (check_weights(w) != AVI_OK || !x_t || !t || !text_embed || !pred || B <= 0) return AVI_EINVAL;
    set_attr();
    hipLaunchKernelGGL(prior_kernel, dim3(B), dim3(NT), sizeof(Smem), static_cast<hipStream_t>(stream), *w,
                       text_embed, x_t, t, brain_keep, image_keep, nullptr, nullptr, B, 0, 1.0f, pred);
    return avi_launch_status();
}

extern "C" int avi_prior_sample(const AviPriorWeights* w, const float* text_embed, const float* noise, int B,
                                float inv_scale, float* out, float* temb_scratch, void* stream) {
    if (check_weights(w) != AVI_OK || !text_embed || !noise || !out || !temb_scratch || B <= 0) return AVI_EINVAL;
    set_attr();
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(prior_time_table_kernel, dim3(w->timesteps), dim3(NT), sizeof(Smem), s, *w, temb_scratch);
    hipLaunchKernelGGL(prior_kernel, dim3(B), dim3(NT), sizeof(Smem), s, *w, text_embed, nullptr, nullptr, nullptr,
                       nullptr, noise, temb_scratch, B, 1, inv_scale, out);
    return avi_launch_status();
}
