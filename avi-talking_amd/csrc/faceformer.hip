// FaceFormer-style autoregressive coefficient decoder: the whole T-step loop of
// Faceformer.predict (models/faceformer.py:710-729) in ONE launch, one workgroup per utterance.
//
// The reference re-decodes the entire prefix every step (O(T^2) decoder passes, batch forced to 1,
// masks rebuilt on the CPU per step).  Here:
//   * self-attention K/V of step i are appended to a cache (global scratch, L2-resident) -- the causal
//     mask (models/faceformer.py:51-72) means earlier rows never change;
//   * the cross-attention memory mask opens only the diagonal (enc_dec_mask, :75-83), so step i reads
//     memory row i alone and softmax over one key is 1: cross_i = out_proj(v_proj(memory_i)) is
//     precomputed for all frames by two GEMMs and passed in as `cross`;
//   * ALiBi-with-period bias is computed analytically: -slope_h * floor((i-j)/period).
// Wave h of the workgroup owns attention head h (nn.TransformerDecoderLayer(nhead=4), :148).
// Small matrix-vector products split K across thread groups and reduce through LDS.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int NH = 4;

__device__ __forceinline__ float act_relu(float x) { return x > 0.f ? x : 0.f; }

// out[n] = act(bias[n] + sum_k x[k] * Wt[k][n]),  x/out in LDS, Wt [K][N] in global.  part: LDS scratch [256].
__device__ void matvec(const float* __restrict__ Wt, const float* __restrict__ bias, const float* x, int K, int N,
                       float* out, bool relu, float* part) {
    const int tid = threadIdx.x;
    if (N >= NT) {
        for (int n = tid; n < N; n += NT) {
            float a = bias ? bias[n] : 0.f;
#pragma unroll 8
            for (int k = 0; k < K; ++k) a = fmaf(x[k], Wt[(long long)k * N + n], a);
            out[n] = relu ? act_relu(a) : a;
        }
        __syncthreads();
        return;
    }
    const int Nq = (N + 63) & ~63;        // 64, 128 or 192
    const int KG = NT / Nq;               // 4, 2 or 1 K-slices
    const int n = tid % Nq, kg = tid / Nq;
    float a = 0.f;
    if (kg < KG && n < N) {
        const int kc = (K + KG - 1) / KG, k0 = kg * kc, k1 = min(K, k0 + kc);
#pragma unroll 8
        for (int k = k0; k < k1; ++k) a = fmaf(x[k], Wt[(long long)k * N + n], a);
    }
    part[tid] = a;
    __syncthreads();
    if (tid < N) {
        float s = bias ? bias[tid] : 0.f;
        for (int g = 0; g < KG; ++g) s += part[g * Nq + tid];
        out[tid] = relu ? act_relu(s) : s;
    }
    __syncthreads();
}

// x = LayerNorm(x + r) over D (eps 1e-5), all 256 threads; red: LDS scratch [8]
__device__ void add_layernorm(float* x, const float* r, long long rstride_unused, int D, const float* __restrict__ g,
                              const float* __restrict__ b, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int d = tid; d < D; d += NT) {
        const float v = x[d] + r[d];
        x[d] = v;
        s += v;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / D;
    float q = 0.f;
    for (int d = tid; d < D; d += NT) {
        const float v = x[d] - mean;
        q += v * v;
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / D + 1e-5f);
    for (int d = tid; d < D; d += NT) x[d] = (x[d] - mean) * rstd * g[d] + b[d];
    __syncthreads();
}

__global__ __launch_bounds__(NT) void faceformer_decode_kernel(const AviFaceformerWeights w,
                                                                const float* __restrict__ cross, int B, int T, int chunk,
                                                                float* __restrict__ kv, float* __restrict__ out,
                                                                uint16_t* __restrict__ out16) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = w.D, V = w.V, dh = D / NH;
    float* x = sm;              // [D]   decoder stream
    float* emb = x + D;         // [D]   current input embedding (obj_embedding, then vertice_map feedback)
    float* q = emb + D;         // [3D]  q | k | v of this step
    float* att = q + 3 * D;     // [D]
    float* h1 = att + D;        // [2D]
    float* o = h1 + 2 * D;      // [64]  coefficient frame (V <= 64)
    float* part = o + 64;       // [256]
    float* red = part + NT;     // [8]
    float* sc = red + 8;        // [4][chunk] attention scores of the current chunk (chunk = T: the reference's window)

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* kvb = kv + (long long)b * T * 2 * D;
    const float slope = w.slopes[wave];
    for (int d = tid; d < D; d += NT) emb[d] = w.obj_embedding[d];
    __syncthreads();

    for (int i = 0; i < T; ++i) {
        // PPE: x = emb + pe[i mod period]  (models/faceformer.py:87-102,712-716)
        const float* pe = w.pe + (long long)(i % w.period) * D;
        for (int d = tid; d < D; d += NT) x[d] = emb[d] + pe[d];
        __syncthreads();
        matvec(w.wqkv, w.bqkv, x, D, 3 * D, q, false, part);
        for (int d = tid; d < 2 * D; d += NT) kvb[(long long)i * 2 * D + d] = q[D + d];
        __syncthreads();  // K/V of step i visible to the whole workgroup (same CU, workgroup scope)

        // ---- self-attention over keys ks..i (ks = start of the chunk frame i lies in), head = wave
        {
            const int ks = (i / chunk) * chunk;
            const int hoff = wave * dh;
            const float scale = rsqrtf((float)dh);
            float mx = -1.0e30f;
            for (int j = ks + lane; j <= i; j += 64) {
                const float* kr = kvb + (long long)j * 2 * D + hoff;
                float s = 0.f;
                for (int d = 0; d < dh; d += 4) {
                    const float4 kk = *reinterpret_cast<const float4*>(kr + d);
                    s = fmaf(q[hoff + d], kk.x, s);
                    s = fmaf(q[hoff + d + 1], kk.y, s);
                    s = fmaf(q[hoff + d + 2], kk.z, s);
                    s = fmaf(q[hoff + d + 3], kk.w, s);
                }
                s = s * scale - slope * (float)((i - j) / w.period);
                sc[wave * chunk + j - ks] = s;
                mx = fmaxf(mx, s);
            }
            mx = wave_max(mx);
            float sum = 0.f;
            for (int j = ks + lane; j <= i; j += 64) {
                const float p = __expf(sc[wave * chunk + j - ks] - mx);
                sc[wave * chunk + j - ks] = p;
                sum += p;
            }
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
            __builtin_amdgcn_wave_barrier();
            // PV: lanes = (key group jc, dim d); dl dims per pass
            const int dl = dh < 64 ? dh : 64, G = 64 / dl;
            const int dd = lane % dl, jc = lane / dl;
            for (int d0 = 0; d0 < dh; d0 += 64) {
                float a = 0.f;
                for (int j = ks + jc; j <= i; j += G)
                    a = fmaf(sc[wave * chunk + j - ks], kvb[(long long)j * 2 * D + D + hoff + d0 + dd], a);
                for (int off = dl; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
                if (jc == 0) att[hoff + d0 + dd] = a * inv;
            }
        }
        __syncthreads();
        matvec(w.wo, w.bo, att, D, D, q, false, part);             // self_attn.out_proj
        add_layernorm(x, q, 0, D, w.n1g, w.n1b, red);              // norm1(x + sa)
        add_layernorm(x, cross + ((long long)b * T + i) * D, 0, D, w.n2g, w.n2b, red);   // norm2(x + cross_i)
        matvec(w.w1, w.b1, x, D, 2 * D, h1, true, part);           // linear1 + ReLU
        matvec(w.w2, w.b2, h1, 2 * D, D, q, false, part);          // linear2
        add_layernorm(x, q, 0, D, w.n3g, w.n3b, red);              // norm3(x + ff)
        matvec(w.wr, w.br, x, D, V, o, false, part);               // vertice_map_r
        if (tid < V) {
            float v = o[tid];
            if (w.coeff_std) v = v * w.coeff_std[tid] + w.coeff_mean[tid];   // un-normalise (:729)
            if (out16) out16[((long long)b * T + i) * V + tid] = __builtin_bit_cast(uint16_t, (_Float16)v);
            else out[((long long)b * T + i) * V + tid] = v;
        }
        matvec(w.wm, w.bm, o, V, D, emb, false, part);             // vertice_map feedback (on the NORMALISED frame)
    }
}

// Input rows of the teacher-forced pass (models/faceformer.py:382-384): x[b][t] = vertice_map(coeff[b][t-1]) + pe[t % period],
// coeff[b][-1] = 0 (the shift-right start token).  One workgroup per TF_ROWS frames; the 53 x D map streams from L2.
constexpr int TF_ROWS = 8;
__global__ __launch_bounds__(NT) void faceformer_tf_embed_kernel(const float* __restrict__ coeff,
                                                                  const float* __restrict__ wm_t,
                                                                  const float* __restrict__ bm,
                                                                  const float* __restrict__ pe, int B, int T, int V,
                                                                  int D, int period, float* __restrict__ out) {
    __shared__ float c[TF_ROWS][64];
    const long long row0 = (long long)blockIdx.x * TF_ROWS, rows = (long long)B * T;
    for (int e = threadIdx.x; e < TF_ROWS * 64; e += NT) {
        const int r = e >> 6, v = e & 63;
        const long long row = row0 + r;
        float x = 0.f;
        if (row < rows && v < V && row % T != 0) x = coeff[(row - 1) * V + v];
        c[r][v] = x;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += NT) {
        float acc[TF_ROWS];
#pragma unroll
        for (int r = 0; r < TF_ROWS; ++r) acc[r] = 0.f;
        for (int v = 0; v < V; ++v) {
            const float w = wm_t[(long long)v * D + d];
#pragma unroll
            for (int r = 0; r < TF_ROWS; ++r) acc[r] = fmaf(c[r][v], w, acc[r]);
        }
        const float b = bm[d];
#pragma unroll
        for (int r = 0; r < TF_ROWS; ++r) {
            const long long row = row0 + r;
            if (row < rows) out[row * D + d] = acc[r] + b + pe[(long long)((row % T) % period) * D + d];
        }
    }
}

}  // namespace

static int decode_chunked_impl(const AviFaceformerWeights* w, const float* cross, int B, int T, int chunk,
                               float* kv_scratch, float* out, uint16_t* out16, void* stream) {
    if (!w || !cross || !kv_scratch || (!out && !out16) || B <= 0 || T <= 0) return AVI_EINVAL;
    const int dh = w->D / NH;
    if (w->D < 16 || dh * NH != w->D || dh < 4 || (dh & (dh - 1)) != 0) return AVI_EINVAL;
    if (w->V < 1 || w->V > 64 || w->period < 1) return AVI_EINVAL;
    if (!w->wqkv || !w->bqkv || !w->wo || !w->bo || !w->n1g || !w->n1b || !w->n2g || !w->n2b || !w->w1 || !w->b1 ||
        !w->w2 || !w->b2 || !w->n3g || !w->n3b || !w->wr || !w->br || !w->wm || !w->bm || !w->pe || !w->slopes ||
        !w->obj_embedding)
        return AVI_EINVAL;
    if ((w->coeff_mean == nullptr) != (w->coeff_std == nullptr)) return AVI_EINVAL;
    if (chunk <= 0 || chunk > T) chunk = T;
    if (chunk < T && chunk % w->period) return AVI_EINVAL;     // the PPE phase must run on across a chunk boundary
    const size_t smem = sizeof(float) * ((size_t)8 * w->D + 64 + NT + 8 + (size_t)NH * chunk);
    if (smem > 160 * 1024) return AVI_ENOSPC;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(faceformer_decode_kernel), 160 * 1024);
    hipLaunchKernelGGL(faceformer_decode_kernel, dim3(B), dim3(NT), smem, static_cast<hipStream_t>(stream), *w, cross,
                       B, T, chunk, kv_scratch, out, out16);
    return avi_launch_status();
}

extern "C" int avi_faceformer_decode_chunked(const AviFaceformerWeights* w, const float* cross, int B, int T, int chunk,
                                             float* kv_scratch, float* out, void* stream) {
    return decode_chunked_impl(w, cross, B, T, chunk, kv_scratch, out, nullptr, stream);
}
extern "C" int avi_faceformer_decode_chunked_f16(const AviFaceformerWeights* w, const float* cross, int B, int T, int chunk,
                                                 float* kv_scratch, uint16_t* out16, void* stream) {
    return decode_chunked_impl(w, cross, B, T, chunk, kv_scratch, nullptr, out16, stream);
}

extern "C" int avi_faceformer_decode(const AviFaceformerWeights* w, const float* cross, int B, int T, float* kv_scratch,
                                     float* out, void* stream) {
    return avi_faceformer_decode_chunked(w, cross, B, T, T, kv_scratch, out, stream);
}

extern "C" int avi_faceformer_tf_embed(const AviFaceformerWeights* w, const float* coeff, int B, int T, float* out,
                                       void* stream) {
    if (!w || !coeff || !out || B <= 0 || T <= 0) return AVI_EINVAL;
    if (w->D < 1 || w->V < 1 || w->V > 64 || w->period < 1 || !w->wm || !w->bm || !w->pe) return AVI_EINVAL;
    const long long rows = (long long)B * T;
    if ((rows + TF_ROWS - 1) / TF_ROWS > 0x7fffffffLL) return AVI_EINVAL;
    hipLaunchKernelGGL(faceformer_tf_embed_kernel, dim3((unsigned)((rows + TF_ROWS - 1) / TF_ROWS)), dim3(NT), 0,
                       static_cast<hipStream_t>(stream), coeff, w->wm, w->bm, w->pe, B, T, w->V, w->D, w->period, out);
    return avi_launch_status();
}
