// Diffusion-prior denoiser (VersatileDiffusionPriorNetwork, models/diffusion_prior.py:169-313, over the
// dalle2 Attention / FeedForward / LayerNorm blocks) and the DDPM sampling loop
// (InstructDiffusionPrior.p_sample_loop_ddpm, models/diffusion_prior.py:329-367), fp32.
//
// The prior diffuses ONE 128-d style token per utterance; the transformer sees 3 tokens
// [text, time, noisy style] (+1 learned null key/value).  Samples are independent, so one workgroup
// owns one sample for the WHOLE loop: the 3x128 residual stream lives in LDS, weights (2.07 M fp32,
// L2-resident, stored [K][N] so lane n reads consecutive addresses) are streamed once per step, and
// there is no launch or host round trip between the 100 steps.  The reference runs ~150 tiny
// kernels per step from Python.
#include "common.h"

namespace {

constexpr int DIM = 128, NTOK = 3, HEADS = 8, DH = 64, INNER = HEADS * DH, FFI = 512, ROT = 32;
constexpr int NT = 1024;   // 16 waves: enough 16-B weight loads in flight to stream from L2/MALL at rate

struct Smem {
    float tok[NTOK][DIM];      // residual stream
    float xn[NTOK][DIM];       // normed input / scratch
    float big[NTOK][2 * FFI];  // q|kv (640 used) or FF hidden (1024)
    float att[NTOK][INNER];    // attention output / swiglu output
    float part[12288];         // split-K partials [KS][3][N] of the streamed linears (KS*N <= 4096)
    float kn[4][DH];           // [null, k0, k1, k2] normalised keys
    float vv[4][DH];           // [null, v0, v1, v2]
    float sim[HEADS][NTOK][4];
    float tmp[2 * DIM];
};

// out[m][n] = sum_k x[m][k] * Wt[k][n] (m < 3) with the weight matrix streamed ONCE as 16-B loads:
// thread = (column quad cq, K slice ks); NV = N/4 quads, KS = NT/NV slices.  Each thread issues all of its
// slice's loads back to back (no dependence between them), so a workgroup keeps hundreds of KB in flight.
// Partials go to s.part[ks][m][n]; reduce3() sums them.
template <int K, int N>
__device__ __forceinline__ void linear3_partial(const float* __restrict__ Wt, const float* x, int xs, Smem& s) {
    constexpr int NV = N / 4;
    constexpr int KS = NT / NV;                 // 4 (N=1024), 6 (N=640), 32 (N=128)
    constexpr int KC = (K + KS - 1) / KS;       // k per slice
    static_assert(KS * N * 3 <= 12288, "partial buffer too small");
    const int cq = threadIdx.x % NV, ks = threadIdx.x / NV;
    if (ks >= KS) return;
    const int k0 = ks * KC;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    const float4* wp = reinterpret_cast<const float4*>(Wt) + cq;
#pragma unroll 4   // 4 x 16 B per lane in flight per batch (x 16 waves = 64 KB per CU); deeper unrolls spill at 128 VGPRs
    for (int kk = 0; kk < KC; ++kk) {
        const int k = k0 + kk;
        if (k < K) {
            const float4 w = wp[(long long)k * NV];
            const float x0 = x[k], x1 = x[xs + k], x2 = x[2 * xs + k];
            a0.x = fmaf(x0, w.x, a0.x); a0.y = fmaf(x0, w.y, a0.y); a0.z = fmaf(x0, w.z, a0.z); a0.w = fmaf(x0, w.w, a0.w);
            a1.x = fmaf(x1, w.x, a1.x); a1.y = fmaf(x1, w.y, a1.y); a1.z = fmaf(x1, w.z, a1.z); a1.w = fmaf(x1, w.w, a1.w);
            a2.x = fmaf(x2, w.x, a2.x); a2.y = fmaf(x2, w.y, a2.y); a2.z = fmaf(x2, w.z, a2.z); a2.w = fmaf(x2, w.w, a2.w);
        }
    }
    float4* p = reinterpret_cast<float4*>(s.part + (long long)ks * 3 * N) + cq;
    p[0] = a0;
    p[NV] = a1;
    p[2 * NV] = a2;
}

// out[m*os + n] (+)= sum_ks part[ks][m][n]
template <int K, int N, bool ACCUM>
__device__ __forceinline__ void reduce3(Smem& s, float* out, int os) {
    constexpr int NV = N / 4;
    constexpr int KS = NT / NV;
    for (int o = threadIdx.x; o < NTOK * N; o += NT) {
        const int m = o / N, n = o - m * N;
        float a = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a += s.part[(ks * 3 + m) * N + n];
        if (ACCUM) out[m * os + n] += a; else out[m * os + n] = a;
    }
}

// dalle2 LayerNorm (gain only, biased variance, eps 1e-5; `stable` divides by the row max first) of the
// 3 token rows: wave m < 3 owns row m, 2 elements per lane.
__device__ __forceinline__ void layernorm3(const float (*in)[DIM], const float* __restrict__ g, bool stable,
                                           float (*out)[DIM]) {
    const int m = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (m < NTOK) {
        float a = in[m][lane], b = in[m][lane + 64];
        if (stable) {
            const float mx = wave_max(fmaxf(a, b));
            a /= mx;
            b /= mx;
        }
        const float mean = wave_sum(a + b) * (1.f / DIM);
        const float da = a - mean, db = b - mean;
        const float var = wave_sum(da * da + db * db) * (1.f / DIM);
        const float r = rsqrtf(var + 1e-5f);
        out[m][lane] = da * r * g[lane];
        out[m][lane + 64] = db * r * g[lane + 64];
    }
}

__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }

// One denoiser evaluation: s.tok holds [text, time, noisy+query]; on return s.tmp[0..127] = prediction.
__device__ void denoise(const AviPriorWeights& w, Smem& s) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int l = 0; l < w.depth; ++l) {
        const AviPriorLayer& L = w.layer[l];
        // ---- attention (pre-LN, multi-query, cosine-sim, rotary, null kv, T5 bias)
        layernorm3(s.tok, L.norm_g, false, s.xn);
        __syncthreads();
        linear3_partial<DIM, INNER + 2 * DH>(L.wqkv, &s.xn[0][0], DIM, s);
        __syncthreads();
        reduce3<DIM, INNER + 2 * DH, false>(s, &s.big[0][0], 2 * FFI);
        __syncthreads();
        // 24 query vectors + 3 keys + null key: rotary (first 32 dims, interleaved pairs), l2norm, * sqrt(16)
        for (int vix = wave; vix < HEADS * NTOK + 4; vix += NT / 64) {
            float x;
            int pos = -1;
            float* dst;
            if (vix < HEADS * NTOK) {
                const int h = vix / NTOK, i = vix - h * NTOK;
                x = s.big[i][h * DH + lane] * 16.0f;  // q * cosine_sim_scale (dalle2 Attention)
                pos = i;
                dst = &s.att[i][h * DH];             // reuse att as the normalised-q buffer
            } else if (vix < HEADS * NTOK + 3) {
                const int i = vix - HEADS * NTOK;
                x = s.big[i][INNER + lane];
                pos = i;
                dst = s.kn[1 + i];
                s.vv[1 + i][lane] = s.big[i][INNER + DH + lane];
            } else {
                x = L.null_kv[lane];
                dst = s.kn[0];
                s.vv[0][lane] = L.null_kv[DH + lane];
            }
            const float partner = __shfl_xor(x, 1, 64);
            if (pos >= 0 && lane < ROT) {
                const float c = w.rot_cos[pos * ROT + lane], sn = w.rot_sin[pos * ROT + lane];
                x = x * c + ((lane & 1) ? partner : -partner) * sn;
            }
            const float nrm = sqrtf(wave_sum(x * x));
            dst[lane] = x / fmaxf(nrm, 1e-12f) * 4.0f;
        }
        __syncthreads();
        if (tid < HEADS * NTOK * 4) {
            const int h = tid / (NTOK * 4), r = tid - h * NTOK * 4, i = r >> 2, j = r & 3;
            float a = 0.f;
            const float* qv = &s.att[i][h * DH];
            const float* kv = s.kn[j];
#pragma unroll 16
            for (int d = 0; d < DH; ++d) a = fmaf(qv[d], kv[d], a);
            s.sim[h][i][j] = a + w.rel_bias[(h * NTOK + i) * 4 + j];
        }
        __syncthreads();
        if (tid < HEADS * NTOK) {
            float* r = &s.sim[0][0][0] + tid * 4;
            const float mx = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
            const float e0 = __expf(r[0] - mx), e1 = __expf(r[1] - mx), e2 = __expf(r[2] - mx), e3 = __expf(r[3] - mx);
            const float inv = 1.f / (e0 + e1 + e2 + e3);
            r[0] = e0 * inv; r[1] = e1 * inv; r[2] = e2 * inv; r[3] = e3 * inv;
        }
        __syncthreads();
        for (int o = tid; o < NTOK * INNER; o += NT) {
            const int i = o / INNER, c = o - i * INNER, h = c >> 6, d = c & 63;
            const float* p = s.sim[h][i];
            s.big[i][c] = p[0] * s.vv[0][d] + p[1] * s.vv[1][d] + p[2] * s.vv[2][d] + p[3] * s.vv[3][d];
        }
        __syncthreads();
        linear3_partial<INNER, DIM>(L.wout, &s.big[0][0], 2 * FFI, s);
        __syncthreads();
        reduce3<INNER, DIM, false>(s, &s.xn[0][0], DIM);
        __syncthreads();
        layernorm3(s.xn, L.out_g, false, s.xn);   // to_out = Linear -> LayerNorm
        __syncthreads();
        for (int o = tid; o < NTOK * DIM; o += NT) s.tok[o >> 7][o & 127] += s.xn[o >> 7][o & 127];
        __syncthreads();
        // ---- feed-forward (LayerNorm -> Linear 128->1024 -> SwiGLU -> Linear 512->128)
        layernorm3(s.tok, L.ff_g, false, s.xn);
        __syncthreads();
        linear3_partial<DIM, 2 * FFI>(L.w1, &s.xn[0][0], DIM, s);
        __syncthreads();
        reduce3<DIM, 2 * FFI, false>(s, &s.big[0][0], 2 * FFI);
        __syncthreads();
        for (int o = tid; o < NTOK * FFI; o += NT) {
            const int m = o / FFI, c = o - m * FFI;
            s.att[m][c] = s.big[m][c] * silu(s.big[m][FFI + c]);
        }
        __syncthreads();
        linear3_partial<FFI, DIM>(L.w2, &s.att[0][0], INNER, s);
        __syncthreads();
        reduce3<FFI, DIM, true>(s, &s.tok[0][0], DIM);
        __syncthreads();
    }
    layernorm3(s.tok, w.final_g, true, s.xn);
    __syncthreads();
    linear3_partial<DIM, DIM>(w.wproj, &s.xn[0][0], DIM, s);
    __syncthreads();
    reduce3<DIM, DIM, false>(s, &s.xn[0][0], DIM);
    __syncthreads();
    if (tid < DIM) s.tmp[tid] = s.xn[2][tid];  // last token = predicted style
    __syncthreads();
}

// time embedding: SinusoidalPosEmb table row t -> MLP 128 -> 256 -> 256 -> 128 (SiLU); result in tok[1]
__device__ void time_embed(const AviPriorWeights& w, int t, Smem& s) {
    const int tid = threadIdx.x;
    if (tid < DIM) s.tmp[tid] = w.time_table[t * DIM + tid];
    __syncthreads();
    if (tid < 256) {
        float a = w.t_b0[tid];
#pragma unroll 8
        for (int k = 0; k < DIM; ++k) a = fmaf(s.tmp[k], w.t_w0[k * 256 + tid], a);
        s.big[0][tid] = silu(a);
    }
    __syncthreads();
    if (tid < 256) {
        float a = w.t_b1[tid];
#pragma unroll 8
        for (int k = 0; k < 256; ++k) a = fmaf(s.big[0][k], w.t_w1[k * 256 + tid], a);
        s.big[1][tid] = silu(a);
    }
    __syncthreads();
    if (tid < DIM) {
        float a = w.t_b2[tid];
#pragma unroll 8
        for (int k = 0; k < 256; ++k) a = fmaf(s.big[1][k], w.t_w2[k * DIM + tid], a);
        s.tok[1][tid] = a;
    }
    __syncthreads();
}

// mode 0: one forward at per-sample timestep t[b] with optional cond-drop masks -> pred[b]
// mode 1: full DDPM loop t = T-1..0 with noise[0] = x_T, noise[1+k] = z of the k-th step -> out[b] = x_0 * inv_scale
__global__ __launch_bounds__(NT) void prior_kernel(const AviPriorWeights w, const float* __restrict__ text_embed,
                                                    const float* __restrict__ x_in, const int* __restrict__ t_in,
                                                    const unsigned char* __restrict__ brain_keep,
                                                    const unsigned char* __restrict__ image_keep,
                                                    const float* __restrict__ noise, int B, int mode,
                                                    float inv_scale, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ float xcur[DIM];
    const bool keep_b = !brain_keep || brain_keep[b];
    if (mode == 0) {
        const bool keep_i = !image_keep || image_keep[b];
        if (tid < DIM) xcur[tid] = keep_i ? x_in[(long long)b * DIM + tid] : w.null_image[tid];
        __syncthreads();
        time_embed(w, t_in[b], s);
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
        time_embed(w, t, s);
        if (tid < DIM) {
            s.tok[0][tid] = text_embed[(long long)b * DIM + tid];
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
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(prior_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
        done = true;
    }
}

}  // namespace

extern "C" int avi_prior_forward(const AviPriorWeights* w, const float* x_t, const int* t, const float* text_embed,
                                 const unsigned char* brain_keep, const unsigned char* image_keep, int B,
                                 float* pred, void* stream) {
    if (check_weights(w) != AVI_OK || !x_t || !t || !text_embed || !pred || B <= 0) return AVI_EINVAL;
    set_attr();
    hipLaunchKernelGGL(prior_kernel, dim3(B), dim3(NT), sizeof(Smem), static_cast<hipStream_t>(stream), *w,
                       text_embed, x_t, t, brain_keep, image_keep, nullptr, B, 0, 1.0f, pred);
    return avi_launch_status();
}

extern "C" int avi_prior_sample(const AviPriorWeights* w, const float* text_embed, const float* noise, int B,
                                float inv_scale, float* out, void* stream) {
    if (check_weights(w) != AVI_OK || !text_embed || !noise || !out || B <= 0) return AVI_EINVAL;
    set_attr();
    hipLaunchKernelGGL(prior_kernel, dim3(B), dim3(NT), sizeof(Smem), static_cast<hipStream_t>(stream), *w,
                       text_embed, nullptr, nullptr, nullptr, nullptr, noise, B, 1, inv_scale, out);
    return avi_launch_status();
}
