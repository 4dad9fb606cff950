// Training-step kernels around the MFMA GEMM (C4-C6 of SURVEY.md 8a): backward of LayerNorm(+act+dropout),
// activations, the 3-token prior attention, token assembly / q_sample, the two losses
// (x0-MSE of models/diffusion_prior.py:391-399 and soft_clip_loss of train_diffusion_prior.py:125-133)
// and a fused multi-tensor AdamW (train_diffusion_prior.py:997-1004).  All fp32.
// Weight gradients are GEMMs over transposed activations (avi_transpose + avi_gemm), bias gradients
// column sums (avi_colsum).
#include "common.h"

namespace {

inline int grid_for(long long total, int block = 256, int cap = 8192) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------ transpose / column sums
__global__ void transpose_kernel(const float* __restrict__ in, int R, int Cc, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        tile[j][tx] = (r < R && c < Cc) ? in[(long long)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (c < Cc && r < R) out[(long long)c * R + r] = tile[tx][j];
    }
}

// out_hi/out_lo [C_pad][R] = bf16 split of in^T (rows >= Cc zero): the transposed operand of a GEMM in one pass
// (the training step used to run a transpose and then a pack kernel for every layer, every step).
__global__ void transpose_pack_kernel(const float* __restrict__ in, int R, int Cc, int C_pad,
                                      uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        tile[j][tx] = (r < R && c < Cc) ? in[(long long)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (c < C_pad && r < R) {
            const float x = tile[tx][j];
            const __bf16 h = (__bf16)x;
            hi[(long long)c * R + r] = __builtin_bit_cast(uint16_t, h);
            lo[(long long)c * R + r] = __builtin_bit_cast(uint16_t, (__bf16)(x - (float)h));
        }
    }
}

// Several transposes in one launch: blockIdx.x runs over the 32x32 blocks of all jobs (first_block = prefix sum).
struct TransposePack { AviTransposeJob j[4]; };
__device__ __forceinline__ void transpose_job_block(const AviTransposeJob& jb, int local) {
    __shared__ float tile[32][33];
    const int Cp = jb.hi ? jb.C_pad : jb.C;
    const int nbx = (Cp + 31) / 32;
    const int bx = (local % nbx) * 32, by = (local / nbx) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        tile[j][tx] = (r < jb.R && c < jb.C) ? jb.in[(long long)r * jb.C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (r >= jb.R) continue;
        const float x = tile[tx][j];
        if (jb.out && c < jb.C) jb.out[(long long)c * jb.R + r] = x;
        if (jb.hi && c < jb.C_pad) {
            const __bf16 h = (__bf16)x;
            jb.hi[(long long)c * jb.R + r] = __builtin_bit_cast(uint16_t, h);
            jb.lo[(long long)c * jb.R + r] = __builtin_bit_cast(uint16_t, (__bf16)(x - (float)h));
        }
    }
}
// column-sum job: block = 16 columns x 16 row groups, partial sums met in LDS in a fixed order (as colsum_kernel)
__device__ __forceinline__ void colsum_job_block(const AviTransposeJob& jb, int local) {
    __shared__ float ps[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = local * 16 + tx;
    float s = 0.f;
    if (c < jb.C) {
#pragma unroll 4
        for (int r = ty; r < jb.R; r += 16) s += jb.in[(long long)r * jb.C + c];
    }
    ps[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < jb.C) {
        s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += ps[j][tx];
        jb.colsum[c] = s;
    }
}
// A job with plane outputs only, R % 64 == 0 and C_pad % 64 == 0 (every weight matrix of the trainer) works on 64 x 64
// tiles: rows are read as 256-B segments and each transposed plane is written in 128-B segments of 8 values per thread (the
// 32 x 32 tile writes 64-B segments of single bf16 values: 2.9 TB/s over the 622 MB of the trainer's weights).
__host__ __device__ __forceinline__ bool transpose_big(const AviTransposeJob& jb) {
    return jb.hi && !jb.out && !jb.colsum && (jb.R & 63) == 0 && (jb.C_pad & 63) == 0;
}
__device__ __forceinline__ void transpose_job_block64(const AviTransposeJob& jb, int local) {
    __shared__ float tile[64][65];
    const int nbx = jb.C_pad >> 6;
    const int bx = (local % nbx) * 64, by = (local / nbx) * 64;
    const int t = threadIdx.x;
    {   // load: thread (row t / 4, float4 column (t % 4) + 4 j), j = 0..3
        const int r = t >> 2;
        const float* src = jb.in + (long long)(by + r) * jb.C + bx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = ((t & 3) + 4 * j) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bx + c + 3 < jb.C && (jb.C & 3) == 0) v = *reinterpret_cast<const float4*>(src + c);
            else {
                if (bx + c < jb.C) v.x = src[c];
                if (bx + c + 1 < jb.C) v.y = src[c + 1];
                if (bx + c + 2 < jb.C) v.z = src[c + 2];
                if (bx + c + 3 < jb.C) v.w = src[c + 3];
            }
            tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {   // store: thread (output row c = t / 8 + 32 pass, rows r8 .. r8 + 7 of the tile)
        const int c = (t >> 3) + 32 * pass, r8 = (t & 7) * 8;
        uint32_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = tile[r8 + 2 * j][c], x1 = tile[r8 + 2 * j + 1][c];
            const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
            h[j] = __builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
            l[j] = __builtin_bit_cast(uint16_t, (__bf16)(x0 - (float)h0)) |
                   ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(x1 - (float)h1)) << 16);
        }
        const long long o = (long long)(bx + c) * jb.R + by + r8;
        *reinterpret_cast<uint4*>(jb.hi + o) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(jb.lo + o) = make_uint4(l[0], l[1], l[2], l[3]);
    }
}
__global__ __launch_bounds__(256) void transpose_jobs_kernel(const TransposePack p, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= p.j[j + 1].first_block) ++j;
    if (p.j[j].colsum) colsum_job_block(p.j[j], blockIdx.x - p.j[j].first_block);   // uniform per block
    else transpose_job_block(p.j[j], blockIdx.x - p.j[j].first_block);
}
__global__ __launch_bounds__(256) void transpose_table_kernel(const AviTransposeJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;            // last job whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AviTransposeJob jb = jobs[lo];
    if (jb.colsum) colsum_job_block(jb, blockIdx.x - jb.first_block);
    else if (transpose_big(jb)) transpose_job_block64(jb, blockIdx.x - jb.first_block);
    else transpose_job_block(jb, blockIdx.x - jb.first_block);
}

// block = 16 columns x 16 row groups: the rows of a column are walked by 16 threads and met in LDS in a fixed order
// (one thread per column walking all rows serially took 15-20 us per call for 64..256 rows: pure load latency)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int R, int Cc,
                                                      float* __restrict__ out, int accumulate) {
    __shared__ float ps[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float s = 0.f;
    if (c < Cc) {
#pragma unroll 4
        for (int r = ty; r < R; r += 16) s += in[(long long)r * Cc + c];
    }
    ps[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < Cc) {
        s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += ps[j][tx];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// ------------------------------------------------------------------ activations
__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case AVI_ACT_GELU: {
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case AVI_ACT_LRELU02: return x > 0.f ? 1.f : 0.2f;
        case AVI_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case AVI_ACT_SILU: {
            const float sg = 1.f / (1.f + __expf(-x));
            return sg * (1.f + x * (1.f - sg));
        }
        default: return 1.f;
    }
}

__global__ void act_fwd_kernel(const float* __restrict__ x, long long n, int act, float* __restrict__ y) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = avi_act(x[i], act);
}
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, long long n, int act,
                               float* __restrict__ dx) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dx[i] = dy[i] * act_grad(x[i], act);
}
// SwiGLU (dalle2 FeedForward): h [R][2F] = value | gate;  y = value * silu(gate)
__global__ void swiglu_fwd_kernel(const float* __restrict__ h, int R, int F, float* __restrict__ y) {
    const long long n = (long long)R * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / F, c = i - r * F;
        const float a = h[r * 2 * F + c], g = h[r * 2 * F + F + c];
        y[i] = a * (g / (1.f + __expf(-g)));
    }
}
__global__ void swiglu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dy, int R, int F,
                                  float* __restrict__ dh) {
    const long long n = (long long)R * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / F, c = i - r * F;
        const float a = h[r * 2 * F + c], g = h[r * 2 * F + F + c];
        const float sg = 1.f / (1.f + __expf(-g));
        dh[r * 2 * F + c] = dy[i] * g * sg;
        dh[r * 2 * F + F + c] = dy[i] * a * sg * (1.f + g * (1.f - sg));
    }
}

// ------------------------------------------------------------------ LayerNorm backward
// forward was  y = act(LN(x*pre)) * mask (+ residual, handled by the caller), LN(u) = (u-mean)*rstd*gamma + beta,
// pre = 1/amax(x) when `stable` (treated as a constant: dalle2 detaches it).  One wave per row.
// Outputs dx and per-row stats (mean, rstd, pre) for the parameter-gradient kernel.
template <int MAXV, bool BR = false>   // BR: one workgroup per row (see layernorm_kernel in elementwise.hip)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mask, int rows, int C, float eps,
                                                      int act, int stable, const float* dx_add, float* dx,
                                                      float* __restrict__ stats) {
    __shared__ float red[5][4];
    constexpr int LN = BR ? 256 : 64;
    const int row = BR ? blockIdx.x : blockIdx.x * 4 + (threadIdx.x >> 6), lane = BR ? threadIdx.x : threadIdx.x & 63;
    if (row >= rows) return;
    auto rsum = [&](float t, int slot) { return BR ? block256_reduce<false>(t, red, slot) : wave_sum(t); };
    auto rmax = [&](float t, int slot) { return BR ? block256_reduce<true>(t, red, slot) : wave_max(t); };
    const int nv = C >> 2;
    const float4* xp = reinterpret_cast<const float4*>(x + (long long)row * C);
    const float4* dyp = reinterpret_cast<const float4*>(dy + (long long)row * C);
    const float4* mp = mask ? reinterpret_cast<const float4*>(mask + (long long)row * C) : nullptr;
    float v[MAXV][4], g1[MAXV][4];
    float pre = 1.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + LN * i;
        const float4 t = idx < nv ? xp[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
    }
    if (stable) {
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (lane + LN * i < nv) mx = fmaxf(mx, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
        pre = 1.f / rmax(mx, 0);
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] *= pre;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean = rsum(s, 1) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + LN * i < nv)
#pragma unroll
            for (int j = 0; j < 4; ++j) q += (v[i][j] - mean) * (v[i][j] - mean);
    const float rstd = rsqrtf(rsum(q, 2) / C + eps);
    // g1 = dL/d(LN output) = dy * mask * act'(u);  then the standard LN backward on gamma*g1
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + LN * i;
        if (idx < nv) {
            const float4 d = dyp[idx];
            const float4 gm = reinterpret_cast<const float4*>(gamma)[idx];
            const float4 bt = beta ? reinterpret_cast<const float4*>(beta)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 mk = mp ? mp[idx] : make_float4(1.f, 1.f, 1.f, 1.f);
            const float dd[4] = {d.x, d.y, d.z, d.w}, gg[4] = {gm.x, gm.y, gm.z, gm.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w},
                        mm[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (v[i][j] - mean) * rstd;
                const float u = xh * gg[j] + bb[j];
                const float t = dd[j] * mm[j] * act_grad(u, act) * gg[j];
                g1[i][j] = t;
                sa += t;
                sb += t * xh;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) g1[i][j] = 0.f;
        }
    }
    sa = rsum(sa, 3) / C;
    sb = rsum(sb, 4) / C;
    float4* dxp = reinterpret_cast<float4*>(dx + (long long)row * C);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + LN * i;
        if (idx < nv) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (v[i][j] - mean) * rstd;
                o[j] = rstd * (g1[i][j] - sa - xh * sb) * pre;
            }
            if (dx_add) {
                const float4 ad = reinterpret_cast<const float4*>(dx_add + (long long)row * C)[idx];
                o[0] += ad.x; o[1] += ad.y; o[2] += ad.z; o[3] += ad.w;
            }
            dxp[idx] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (lane == 0) {
        stats[row * 3 + 0] = mean;
        stats[row * 3 + 1] = rstd;
        stats[row * 3 + 2] = pre;
    }
}

// dgamma[c] (+)= sum_r dy*mask*act'(u)*xhat ;  dbeta[c] (+)= sum_r dy*mask*act'(u)
// block = 16 columns x 16 row groups (a serial walk over the rows by one thread per column cost 50 us per call, 18 % of
// the training step; 64 x 4 still left 64 dependent row visits per thread); partial sums meet in LDS in a fixed order,
// so the result is deterministic.
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mask,
                                                             const float* __restrict__ stats, int rows, int C, int act,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int accumulate) {
    __shared__ float pg[16][17], pb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float sg = 0.f, sb = 0.f;
    if (c < C) {
        const float g = gamma[c], b = beta ? beta[c] : 0.f;
#pragma unroll 4
        for (int r = ty; r < rows; r += 16) {
            const float mean = stats[r * 3], rstd = stats[r * 3 + 1], pre = stats[r * 3 + 2];
            const float xh = (x[(long long)r * C + c] * pre - mean) * rstd;
            const float t =
                dy[(long long)r * C + c] * (mask ? mask[(long long)r * C + c] : 1.f) * act_grad(xh * g + b, act);
            sg += t * xh;
            sb += t;
        }
    }
    pg[ty][tx] = sg;
    pb[ty][tx] = sb;
    __syncthreads();
    if (ty == 0 && c < C) {
        sg = sb = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sg += pg[j][tx];
            sb += pb[j][tx];
        }
        dgamma[c] = accumulate ? dgamma[c] + sg : sg;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sb : sb;
    }
}

// ------------------------------------------------------------------ prior: token assembly + q_sample
// models/diffusion_prior.py:372 (q_sample on x0 = target*scale), :255-303 (cond-drop where(), pos_emb query, concat).
__global__ void prior_tokens_fwd_kernel(const float* __restrict__ target, const float* __restrict__ noise,
                                        const int* __restrict__ t, const float* __restrict__ sqrt_ac,
                                        const float* __restrict__ sqrt_1mac, float scale,
                                        const float* __restrict__ text_embed, const float* __restrict__ time_emb,
                                        const unsigned char* __restrict__ bkeep, const unsigned char* __restrict__ ikeep,
                                        const float* __restrict__ null_brain, const float* __restrict__ null_image,
                                        const float* __restrict__ lq, int B, float* __restrict__ x0,
                                        float* __restrict__ tokens) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 128) return;
    const int b = i >> 7, d = i & 127;
    const float x0v = target[i] * scale;
    x0[i] = x0v;
    const float xt = sqrt_ac[t[b]] * x0v + sqrt_1mac[t[b]] * noise[i];
    float* tk = tokens + (long long)b * 384;
    tk[d] = (!bkeep || bkeep[b]) ? text_embed[i] : null_brain[d];
    tk[128 + d] = time_emb[i];
    tk[256 + d] = ((!ikeep || ikeep[b]) ? xt : null_image[d]) + lq[d];
}
// dtokens [B][3][128] -> dtext [B][128], dtime [B][128], and (accumulated over the batch) dnull_brain, dnull_image, dlq
__global__ void prior_tokens_bwd_kernel(const float* __restrict__ dtok, const unsigned char* __restrict__ bkeep,
                                        const unsigned char* __restrict__ ikeep, int B, float* __restrict__ dtext,
                                        float* __restrict__ dtime, float* __restrict__ dnull_brain,
                                        float* __restrict__ dnull_image, float* __restrict__ dlq) {
    const int d = threadIdx.x;   // 128 threads, one block
    float nb = 0.f, ni = 0.f, q = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* tk = dtok + (long long)b * 384;
        const bool kb = !bkeep || bkeep[b], ki = !ikeep || ikeep[b];
        dtext[b * 128 + d] = kb ? tk[d] : 0.f;
        if (!kb) nb += tk[d];
        dtime[b * 128 + d] = tk[128 + d];
        q += tk[256 + d];
        if (!ki) ni += tk[256 + d];
    }
    dnull_brain[d] = nb;      // the only writer of these three gradients: plain stores, no zeroing needed
    dnull_image[d] = ni;
    dlq[d] = q;
}

// ------------------------------------------------------------------ prior attention (3 tokens + null kv), fwd / bwd
// qkv [B][3][640] = q(512) | k(64) | v(64) per token.  dalle2 Attention with cosine-sim, rotary(32), null kv, T5 bias.
struct AttnCtx {
    float qn[8][3][64];   // normalised, scaled queries
    float kn[4][64];
    float vv[4][64];
    float qinv[8][3];     // 1/|q_rot|
    float kinv[4];
    float p[8][3][4];
};

__device__ void attn_prepare(const float* __restrict__ qkvb, const float* __restrict__ null_kv,
                             const float* __restrict__ rot_cos, const float* __restrict__ rot_sin, AttnCtx& c) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int vix = wave; vix < 28; vix += 4) {
        float x;
        int pos = -1;
        if (vix < 24) {
            const int h = vix / 3, i = vix - h * 3;
            x = qkvb[i * 640 + h * 64 + lane] * 16.0f;
            pos = i;
        } else if (vix < 27) {
            const int i = vix - 24;
            x = qkvb[i * 640 + 512 + lane];
            pos = i;
            c.vv[1 + i][lane] = qkvb[i * 640 + 576 + lane];
        } else {
            x = null_kv[lane];
            c.vv[0][lane] = null_kv[64 + lane];
        }
        const float partner = __shfl_xor(x, 1, 64);
        if (pos >= 0 && lane < 32) {
            const float cs = rot_cos[pos * 32 + lane], sn = rot_sin[pos * 32 + lane];
            x = x * cs + ((lane & 1) ? partner : -partner) * sn;
        }
        const float inv = 1.f / fmaxf(sqrtf(wave_sum(x * x)), 1e-12f);
        if (vix < 24) {
            const int h = vix / 3, i = vix - h * 3;
            c.qn[h][i][lane] = x * inv * 4.0f;
            if (lane == 0) c.qinv[h][i] = inv;
        } else {
            c.kn[vix == 27 ? 0 : vix - 23][lane] = x * inv * 4.0f;
            if (lane == 0) c.kinv[vix == 27 ? 0 : vix - 23] = inv;
        }
    }
    __syncthreads();
}

__device__ void attn_probs(const float* __restrict__ rel_bias, AttnCtx& c) {
    const int tid = threadIdx.x;
    if (tid < 96) {
        const int h = tid / 12, r = tid - h * 12, i = r >> 2, j = r & 3;
        float a = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) a = fmaf(c.qn[h][i][d], c.kn[j][d], a);
        c.p[h][i][j] = a + rel_bias[(h * 3 + i) * 4 + j];
    }
    __syncthreads();
    if (tid < 24) {
        float* r = &c.p[0][0][0] + tid * 4;
        const float mx = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
        const float e0 = __expf(r[0] - mx), e1 = __expf(r[1] - mx), e2 = __expf(r[2] - mx), e3 = __expf(r[3] - mx);
        const float inv = 1.f / (e0 + e1 + e2 + e3);
        r[0] = e0 * inv; r[1] = e1 * inv; r[2] = e2 * inv; r[3] = e3 * inv;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void prior_attn_fwd_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ null_kv,
                                                              const float* __restrict__ rel_bias,
                                                              const float* __restrict__ rot_cos,
                                                              const float* __restrict__ rot_sin,
                                                              float* __restrict__ out) {
    __shared__ AttnCtx c;
    const int b = blockIdx.x;
    attn_prepare(qkv + (long long)b * 1920, null_kv, rot_cos, rot_sin, c);
    attn_probs(rel_bias, c);
    for (int o = threadIdx.x; o < 1536; o += 256) {
        const int i = o / 512, cc = o - i * 512, h = cc >> 6, d = cc & 63;
        const float* p = c.p[h][i];
        out[(long long)b * 1536 + o] = p[0] * c.vv[0][d] + p[1] * c.vv[1][d] + p[2] * c.vv[2][d] + p[3] * c.vv[3][d];
    }
}

// dout [B][3][512] -> dqkv [B][3][640]; dnull_kv [2][64] and drel_bias [8][3][4] accumulated with atomics.
__global__ __launch_bounds__(256) void prior_attn_bwd_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ null_kv,
                                                              const float* __restrict__ rel_bias,
                                                              const float* __restrict__ rot_cos,
                                                              const float* __restrict__ rot_sin,
                                                              const float* __restrict__ dout, float* __restrict__ dqkv,
                                                              float* __restrict__ dnull_kv,
                                                              float* __restrict__ drel_bias) {
    __shared__ AttnCtx c;
    __shared__ float ds[8][3][4];      // dsim
    __shared__ float dkn[4][64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qkvb = qkv + (long long)b * 1920;
    const float* dob = dout + (long long)b * 1536;
    float* dqb = dqkv + (long long)b * 1920;
    attn_prepare(qkvb, null_kv, rot_cos, rot_sin, c);
    attn_probs(rel_bias, c);
    // dp[h][i][j] = sum_d dout[i][h*64+d] * v_j[d]
    if (tid < 96) {
        const int h = tid / 12, r = tid - h * 12, i = r >> 2, j = r & 3;
        float a = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) a = fmaf(dob[i * 512 + h * 64 + d], c.vv[j][d], a);
        ds[h][i][j] = a;
    }
    __syncthreads();
    if (tid < 24) {
        float* dpr = &ds[0][0][0] + tid * 4;
        const float* pr = &c.p[0][0][0] + tid * 4;
        const float dot = dpr[0] * pr[0] + dpr[1] * pr[1] + dpr[2] * pr[2] + dpr[3] * pr[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) dpr[j] = pr[j] * (dpr[j] - dot);
    }
    __syncthreads();
    if (tid < 96) atomicAdd(&drel_bias[tid], (&ds[0][0][0])[tid]);
    // dv_j[d] = sum_{h,i} p[h][i][j] * dout[i][h*64+d]   (wave j, lane d)
    {
        const int j = wave, d = lane;
        float a = 0.f;
        for (int h = 0; h < 8; ++h)
#pragma unroll
            for (int i = 0; i < 3; ++i) a = fmaf(c.p[h][i][j], dob[i * 512 + h * 64 + d], a);
        if (j == 0) atomicAdd(&dnull_kv[64 + d], a);
        else dqb[(j - 1) * 640 + 576 + d] = a;
        // dkn_j[d] = sum_{h,i} dsim[h][i][j] * qn[h][i][d]
        float kk = 0.f;
        for (int h = 0; h < 8; ++h)
#pragma unroll
            for (int i = 0; i < 3; ++i) kk = fmaf(ds[h][i][j], c.qn[h][i][d], kk);
        dkn[j][d] = kk;
    }
    __syncthreads();
    // back through  y = 4 * x_rot / |x_rot|  and the rotary, for 24 queries + 3 keys + null key
    for (int vix = wave; vix < 28; vix += 4) {
        float dy, y, inv;
        int pos = -1;
        if (vix < 24) {
            const int h = vix / 3, i = vix - h * 3;
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) a = fmaf(ds[h][i][j], c.kn[j][lane], a);   // dqn
            dy = a; y = c.qn[h][i][lane]; inv = c.qinv[h][i]; pos = i;
        } else {
            const int j = vix == 27 ? 0 : vix - 23;
            dy = dkn[j][lane]; y = c.kn[j][lane]; inv = c.kinv[j]; pos = vix == 27 ? -1 : vix - 24;
        }
        // y = 4*xhat, xhat = x/|x|:  dx = 4/|x| * (dy - xhat (xhat . dy)) = inv * (4 dy - y (y . dy)/4)
        const float ydy = wave_sum(y * dy);
        float dx = inv * (4.0f * dy - y * ydy * 0.25f);
        if (pos >= 0) {
            // rotary transpose: (a,b) -> (a c - b s, b c + a s)  =>  da = dy0 c + dy1 s ; db = dy1 c - dy0 s
            const float partner = __shfl_xor(dx, 1, 64);
            if (lane < 32) {
                const float cs = rot_cos[pos * 32 + lane], sn = rot_sin[pos * 32 + lane];
                dx = dx * cs + ((lane & 1) ? -partner : partner) * sn;
            }
        }
        if (vix < 24) {
            const int h = vix / 3, i = vix - h * 3;
            dqb[i * 640 + h * 64 + lane] = dx * 16.0f;
        } else if (vix < 27) {
            dqb[(vix - 24) * 640 + 512 + lane] = dx;
        } else {
            atomicAdd(&dnull_kv[lane], dx);
        }
    }
}

// ------------------------------------------------------------------ losses
// x0-MSE: loss = mean((pred - x0)^2);  dpred = weight * 2 (pred - x0) / n.   One block.
__global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ pred, const float* __restrict__ x0,
                                                        int n, float weight, float* __restrict__ loss,
                                                        float* __restrict__ dpred) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = pred[i] - x0[i];
        s += d * d;
        dpred[i] = weight * 2.f * d / n;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / n;
}

// soft_clip_loss on L2-normalised rows.  Step 1: normalise; S1 = tn tn^T / temp, S2 = pn tn^T / temp.
__global__ void clip_norm_kernel(const float* __restrict__ p, const float* __restrict__ t, int B, int D,
                                 float* __restrict__ pn, float* __restrict__ tn, float* __restrict__ pinv) {
    const int row = blockIdx.x, lane = threadIdx.x;   // one wave per row, D = 128
    float a = 0.f, b = 0.f;
    for (int d = lane; d < D; d += 64) {
        a += p[row * D + d] * p[row * D + d];
        b += t[row * D + d] * t[row * D + d];
    }
    const float ia = 1.f / fmaxf(sqrtf(wave_sum(a)), 1e-12f), ib = 1.f / fmaxf(sqrtf(wave_sum(b)), 1e-12f);
    for (int d = lane; d < D; d += 64) {
        pn[row * D + d] = p[row * D + d] * ia;
        tn[row * D + d] = t[row * D + d] * ib;
    }
    if (lane == 0) pinv[row] = ia;
}
__global__ void clip_sim_kernel(const float* __restrict__ pn, const float* __restrict__ tn, int B, int D, float inv_temp,
                                float* __restrict__ S1, float* __restrict__ S2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * B) return;
    const int i = idx / B, j = idx - i * B;
    float a = 0.f, b = 0.f;
    for (int d = 0; d < D; ++d) {
        const float tj = tn[j * D + d];
        a = fmaf(tn[i * D + d], tj, a);
        b = fmaf(pn[i * D + d], tj, b);
    }
    S1[idx] = a * inv_temp;
    S2[idx] = b * inv_temp;
}
// Step 2 (one wave per row/column i): row softmax T = softmax(S1[i,:]), row log-softmax of S2[i,:] (loss1) and column
// log-softmax of S2[:,i] weighted by T[i,:] (loss2); dS2 accumulated: 0.5/B * ((Prow - T)[i,j] + (Pcol[j,i] - T[i,j]) at [j,i]).
__global__ void clip_loss_kernel(const float* __restrict__ S1, const float* __restrict__ S2, int B,
                                 float* __restrict__ dS2, float* __restrict__ loss_parts) {
    const int i = blockIdx.x, lane = threadIdx.x;
    float m1 = -3e38f, m2 = -3e38f, m3 = -3e38f;
    for (int j = lane; j < B; j += 64) {
        m1 = fmaxf(m1, S1[i * B + j]);
        m2 = fmaxf(m2, S2[i * B + j]);
        m3 = fmaxf(m3, S2[j * B + i]);
    }
    m1 = wave_max(m1); m2 = wave_max(m2); m3 = wave_max(m3);
    float z1 = 0.f, z2 = 0.f, z3 = 0.f;
    for (int j = lane; j < B; j += 64) {
        z1 += __expf(S1[i * B + j] - m1);
        z2 += __expf(S2[i * B + j] - m2);
        z3 += __expf(S2[j * B + i] - m3);
    }
    z1 = wave_sum(z1); z2 = wave_sum(z2); z3 = wave_sum(z3);
    const float l2 = logf(z2), l3 = logf(z3);
    float la = 0.f, lb = 0.f;
    const float w = 0.5f / B;
    for (int j = lane; j < B; j += 64) {
        const float T = __expf(S1[i * B + j] - m1) / z1;
        const float ls_row = S2[i * B + j] - m2 - l2;        // log_softmax(S2[i,:])[j]
        const float ls_col = S2[j * B + i] - m3 - l3;        // log_softmax(S2[:,i])[j] = log_softmax(S2^T[i,:])[j]
        la -= ls_row * T;
        lb -= ls_col * T;
        atomicAdd(&dS2[i * B + j], w * (__expf(ls_row) - T));
        atomicAdd(&dS2[j * B + i], w * (__expf(ls_col) - T));
    }
    la = wave_sum(la); lb = wave_sum(lb);
    if (lane == 0) loss_parts[i] = w * (la + lb);
}
// Step 3: dpn = dS2 tn / temp; back through the normalisation: dp = pinv * (dpn - pn (pn . dpn)); loss = sum parts
__global__ void clip_bwd_kernel(const float* __restrict__ dS2, const float* __restrict__ tn, const float* __restrict__ pn,
                                const float* __restrict__ pinv, int B, int D, float inv_temp, float grad_scale,
                                const float* __restrict__ loss_parts, float* __restrict__ dp, float* __restrict__ loss) {
    const int i = blockIdx.x, lane = threadIdx.x;
    float g[4] = {0.f, 0.f, 0.f, 0.f};   // D <= 256
    for (int j = 0; j < B; ++j) {
        const float s = dS2[i * B + j] * inv_temp;
        int q = 0;
        for (int d = lane; d < D; d += 64, ++q) g[q] = fmaf(s, tn[j * D + d], g[q]);
    }
    float dot = 0.f;
    int q = 0;
    for (int d = lane; d < D; d += 64, ++q) dot += g[q] * pn[i * D + d];
    dot = wave_sum(dot);
    q = 0;
    for (int d = lane; d < D; d += 64, ++q) dp[i * D + d] = grad_scale * pinv[i] * (g[q] - pn[i * D + d] * dot);
    if (i == 0) {
        float s = 0.f;
        for (int r = lane; r < B; r += 64) s += loss_parts[r];
        s = wave_sum(s);
        if (lane == 0) loss[0] = s;
    }
}

// ------------------------------------------------------------------ fused AdamW (torch.optim.AdamW semantics)
// p <- p*(1 - lr*wd);  m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;  p <- p - lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// Optionally emits the bf16 hi/lo planes of the updated parameter (the MFMA GEMM's weight format).
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float rsqrt_bc2, float grad_scale, const float* __restrict__ dyn,
                             uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    if (dyn) {  // step-dependent scalars from device memory, so a captured hipGraph can be replayed every step
        lr = dyn[0];
        bc1 = dyn[1];
        rsqrt_bc2 = dyn[2];
        if (dyn[3] > 0.f) b1 = dyn[3];   // this step's beta1 (OneCycleLR cycle_momentum); 0 = the argument's
    }
    const long long nv = n >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {gg.x, gg.y, gg.z, gg.w}, M[4] = {mm.x, mm.y, mm.z, mm.w},
              V[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = G[j] * grad_scale;
            P[j] *= 1.f - lr * wd;
            M[j] = b1 * M[j] + (1.f - b1) * gj;
            V[j] = b2 * V[j] + (1.f - b2) * gj * gj;
            P[j] -= (lr / bc1) * M[j] / (sqrtf(V[j]) * rsqrt_bc2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(P[0], P[1], P[2], P[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(M[0], M[1], M[2], M[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(V[0], V[1], V[2], V[3]);
        if (hi) {
            uint16_t h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const __bf16 hb = (__bf16)P[j];
                const __bf16 lb = (__bf16)(P[j] - (float)hb);
                h[j] = __builtin_bit_cast(uint16_t, hb);
                l[j] = __builtin_bit_cast(uint16_t, lb);
            }
            reinterpret_cast<uint2*>(hi)[i] = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
            reinterpret_cast<uint2*>(lo)[i] = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
        }
    }
}

// ------------------------------------------------------------------ data movement of the step (no torch kernels inside
// the captured training step: see tests/test_gpu_library_only.py)
__global__ void zero_kernel(float4* __restrict__ p, long long n4, float* __restrict__ tail, int ntail) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}

// dst[r][0..C) = src[(idx ? idx[r] : r)][0..C) with row strides in elements (gather of table rows / strided row copy)
__global__ void copy_rows_kernel(const float* __restrict__ src, long long src_stride, const int* __restrict__ idx,
                                 float* __restrict__ dst, long long dst_stride, int rows, int C) {
    const long long total = (long long)rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i - (long long)r * C);
        dst[r * dst_stride + c] = src[(idx ? idx[r] : r) * src_stride + c];
    }
}

// T5 relative-position bias of the 3-token denoiser (dalle2 RelPosBias called with (n, n+1), models/diffusion_prior.py:159):
// bucket of max(i - j, 0) is the distance itself below 16, so bias[h][i][j] = emb[max(i - j, 0)][h]; backward scatters.
__global__ void rel_bias_kernel(const float* __restrict__ emb, float* __restrict__ bias, const float* __restrict__ dbias,
                                float* __restrict__ demb, int heads, int n) {
    const int t = threadIdx.x;
    if (bias) {
        if (t < heads * n * (n + 1)) {
            const int h = t / (n * (n + 1)), r = t - h * n * (n + 1), i = r / (n + 1), j = r - i * (n + 1);
            const int d = i - j > 0 ? i - j : 0;
            bias[t] = emb[d * heads + h];
        }
    } else if (t < 32 * heads) {      // one thread per (bucket d, head) of the WHOLE (32, heads) table: fixed summation
        const int d = t / heads, h = t - d * heads;                      // order, plain store (buckets >= n get zero)
        float a = 0.f;
        if (d < n)
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= n; ++j)
                    if ((i - j > 0 ? i - j : 0) == d) a += dbias[(h * n + i) * (n + 1) + j];
        demb[d * heads + h] = a;
    }
}

}  // namespace

#define S_(x) static_cast<hipStream_t>(x)

extern "C" int avi_transpose(const float* in, int R, int Cc, float* out, void* stream) {
    if (!in || !out || R <= 0 || Cc <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, S_(stream), in, R, Cc, out);
    return avi_launch_status();
}

extern "C" int avi_transpose_pack_split(const float* in, int R, int Cc, int C_pad, uint16_t* hi, uint16_t* lo,
                                        void* stream) {
    if (!in || !hi || !lo || R <= 0 || Cc <= 0 || C_pad < Cc) return AVI_EINVAL;
    hipLaunchKernelGGL(transpose_pack_kernel, dim3((C_pad + 31) / 32, (R + 31) / 32), dim3(256), 0, S_(stream), in, R,
                       Cc, C_pad, hi, lo);
    return avi_launch_status();
}

static int transpose_job_blocks(const AviTransposeJob& jb) {
    if (jb.colsum) return (jb.C + 15) / 16;
    const int Cp = jb.hi ? jb.C_pad : jb.C;
    return ((Cp + 31) / 32) * ((jb.R + 31) / 32);
}
extern "C" int avi_transpose_jobs(const AviTransposeJob* jobs, int njobs, void* stream) {
    if (!jobs || njobs < 1 || njobs > 4) return AVI_EINVAL;
    TransposePack p{};
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        const AviTransposeJob& jb = jobs[i];
        if (!jb.in || jb.R <= 0 || jb.C <= 0) return AVI_EINVAL;
        if (jb.colsum ? (jb.out || jb.hi) : ((!jb.out && !jb.hi) || (jb.hi && !jb.lo) || (jb.hi && jb.C_pad < jb.C)))
            return AVI_EINVAL;
        p.j[i] = jb;
        p.j[i].first_block = total;
        total += transpose_job_blocks(jb);
    }
    hipLaunchKernelGGL(transpose_jobs_kernel, dim3(total), dim3(256), 0, S_(stream), p, njobs);
    return avi_launch_status();
}
extern "C" int avi_transpose_table(const AviTransposeJob* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || njobs < 1 || total_blocks < 1) return AVI_EINVAL;
    hipLaunchKernelGGL(transpose_table_kernel, dim3(total_blocks), dim3(256), 0, S_(stream), jobs_dev, njobs);
    return avi_launch_status();
}

extern "C" int avi_colsum(const float* in, int R, int Cc, float* out, int accumulate, void* stream) {
    if (!in || !out || R <= 0 || Cc <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(colsum_kernel, dim3((Cc + 15) / 16), dim3(256), 0, S_(stream), in, R, Cc, out, accumulate);
    return avi_launch_status();
}

extern "C" int avi_act_fwd(const float* x, long long n, int act, float* y, void* stream) {
    if (!x || !y || n <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), x, n, act, y);
    return avi_launch_status();
}
extern "C" int avi_act_bwd(const float* x, const float* dy, long long n, int act, float* dx, void* stream) {
    if (!x || !dy || !dx || n <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), x, dy, n, act, dx);
    return avi_launch_status();
}
extern "C" int avi_swiglu_fwd(const float* h, int R, int F, float* y, void* stream) {
    if (!h || !y || R <= 0 || F <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for((long long)R * F)), dim3(256), 0, S_(stream), h, R, F, y);
    return avi_launch_status();
}
extern "C" int avi_swiglu_bwd(const float* h, const float* dy, int R, int F, float* dh, void* stream) {
    if (!h || !dy || !dh || R <= 0 || F <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((long long)R * F)), dim3(256), 0, S_(stream), h, dy, R, F, dh);
    return avi_launch_status();
}

extern "C" int avi_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta,
                                 const float* mask, int rows, int C, float eps, int act, int stable,
                                 const float* dx_add, float* dx, float* dgamma, float* dbeta, int accumulate,
                                 float* stats, void* stream) {
    if (!x || !dy || !gamma || !dx || !dgamma || !stats || rows <= 0 || C <= 0 || (C & 3) || C > 4096) return AVI_EINVAL;
    hipStream_t s = S_(stream);
    dim3 grid((rows + 3) / 4), block(256);
    if (C <= 1024)
        hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, block, 0, s, x, dy, gamma, beta, mask, rows, C, eps, act, stable,
                           dx_add, dx, stats);
    else   // a workgroup per row
        hipLaunchKernelGGL((ln_bwd_kernel<4, true>), dim3(rows), block, 0, s, x, dy, gamma, beta, mask, rows, C, eps, act,
                           stable, dx_add, dx, stats);
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3((C + 15) / 16), dim3(256), 0, s, x, dy, gamma, beta, mask, stats,
                       rows, C, act, dgamma, dbeta, accumulate);
    return avi_launch_status();
}

extern "C" int avi_prior_tokens_fwd(const float* target, const float* noise, const int* t, const float* sqrt_ac,
                                    const float* sqrt_1mac, float scale, const float* text_embed,
                                    const float* time_emb, const unsigned char* bkeep, const unsigned char* ikeep,
                                    const float* null_brain, const float* null_image, const float* learned_query,
                                    int B, float* x0, float* tokens, void* stream) {
    if (!target || !noise || !t || !sqrt_ac || !sqrt_1mac || !text_embed || !time_emb || !null_brain || !null_image ||
        !learned_query || !x0 || !tokens || B <= 0)
        return AVI_EINVAL;
    hipLaunchKernelGGL(prior_tokens_fwd_kernel, dim3((B * 128 + 255) / 256), dim3(256), 0, S_(stream), target, noise, t,
                       sqrt_ac, sqrt_1mac, scale, text_embed, time_emb, bkeep, ikeep, null_brain, null_image,
                       learned_query, B, x0, tokens);
    return avi_launch_status();
}
extern "C" int avi_prior_tokens_bwd(const float* dtokens, const unsigned char* bkeep, const unsigned char* ikeep, int B,
                                    float* dtext, float* dtime, float* dnull_brain, float* dnull_image, float* dlq,
                                    void* stream) {
    if (!dtokens || !dtext || !dtime || !dnull_brain || !dnull_image || !dlq || B <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(prior_tokens_bwd_kernel, dim3(1), dim3(128), 0, S_(stream), dtokens, bkeep, ikeep, B, dtext,
                       dtime, dnull_brain, dnull_image, dlq);
    return avi_launch_status();
}

extern "C" int avi_prior_attn_fwd(const float* qkv, const float* null_kv, const float* rel_bias, const float* rot_cos,
                                  const float* rot_sin, int B, float* out, void* stream) {
    if (!qkv || !null_kv || !rel_bias || !rot_cos || !rot_sin || !out || B <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(prior_attn_fwd_kernel, dim3(B), dim3(256), 0, S_(stream), qkv, null_kv, rel_bias, rot_cos,
                       rot_sin, out);
    return avi_launch_status();
}
extern "C" int avi_prior_attn_bwd(const float* qkv, const float* null_kv, const float* rel_bias, const float* rot_cos,
                                  const float* rot_sin, const float* dout, int B, float* dqkv, float* dnull_kv,
                                  float* drel_bias, void* stream) {
    if (!qkv || !null_kv || !rel_bias || !rot_cos || !rot_sin || !dout || !dqkv || !dnull_kv || !drel_bias || B <= 0)
        return AVI_EINVAL;
    hipLaunchKernelGGL(prior_attn_bwd_kernel, dim3(B), dim3(256), 0, S_(stream), qkv, null_kv, rel_bias, rot_cos,
                       rot_sin, dout, dqkv, dnull_kv, drel_bias);
    return avi_launch_status();
}

extern "C" int avi_mse_loss(const float* pred, const float* x0, int n, float weight, float* loss, float* dpred,
                            void* stream) {
    if (!pred || !x0 || !loss || !dpred || n <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(256), 0, S_(stream), pred, x0, n, weight, loss, dpred);
    return avi_launch_status();
}

// scratch: (2*B*D + B + 3*B*B + B) floats
extern "C" int avi_soft_clip_loss(const float* proj, const float* target, int B, int D, float temp, float grad_scale,
                                  float* loss, float* dproj, float* scratch, void* stream) {
    if (!proj || !target || !loss || !dproj || !scratch || B <= 0 || D <= 0 || D > 256 || temp <= 0.f) return AVI_EINVAL;
    hipStream_t s = S_(stream);
    float* pn = scratch;
    float* tn = pn + (long long)B * D;
    float* pinv = tn + (long long)B * D;
    float* S1 = pinv + B;
    float* S2 = S1 + (long long)B * B;
    float* dS2 = S2 + (long long)B * B;
    float* parts = dS2 + (long long)B * B;
    if (hipMemsetAsync(dS2, 0, sizeof(float) * B * B, s) != hipSuccess) return avi_launch_status();
    hipLaunchKernelGGL(clip_norm_kernel, dim3(B), dim3(64), 0, s, proj, target, B, D, pn, tn, pinv);
    hipLaunchKernelGGL(clip_sim_kernel, dim3((B * B + 255) / 256), dim3(256), 0, s, pn, tn, B, D, 1.f / temp, S1, S2);
    hipLaunchKernelGGL(clip_loss_kernel, dim3(B), dim3(64), 0, s, S1, S2, B, dS2, parts);
    hipLaunchKernelGGL(clip_bwd_kernel, dim3(B), dim3(64), 0, s, dS2, tn, pn, pinv, B, D, 1.f / temp, grad_scale, parts,
                       dproj, loss);
    return avi_launch_status();
}

extern "C" int avi_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, float grad_scale, const float* dyn, uint16_t* hi,
                         uint16_t* lo, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || (n & 3) || step < 1 || ((hi == nullptr) != (lo == nullptr))) return AVI_EINVAL;
    // bias corrections in double from the fp32 betas the kernel multiplies with, rounded once: exactly what a caller puts into
    // `dyn` (host/training.py _set_dyn), so a step driven by arguments and a replayed graph driven by `dyn` agree bit for bit
    // (Adam's m / sqrt(v) amplifies a last-bit difference of a parameter into a visible fraction of lr a few steps later)
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n >> 2, 256, 4096)), dim3(256), 0, S_(stream), p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, rsqrt_bc2, grad_scale, dyn, hi, lo);
    return avi_launch_status();
}

extern "C" int avi_zero(float* p, long long n, void* stream) {
    if (!p || n <= 0 || (reinterpret_cast<uintptr_t>(p) & 15)) return AVI_EINVAL;
    const long long n4 = n >> 2;
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(256), 0, S_(stream),
                       reinterpret_cast<float4*>(p), n4, p + 4 * n4, (int)(n & 3));
    return avi_launch_status();
}
extern "C" int avi_copy_rows(const float* src, long long src_stride, const int* row_index, float* dst,
                             long long dst_stride, int rows, int C, void* stream) {
    if (!src || !dst || rows <= 0 || C <= 0) return AVI_EINVAL;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for((long long)rows * C)), dim3(256), 0, S_(stream), src, src_stride,
                       row_index, dst, dst_stride, rows, C);
    return avi_launch_status();
}
extern "C" int avi_prior_rel_bias(const float* emb, float* bias, const float* dbias, float* demb, int heads, int n,
                                  void* stream) {
    if (!((emb && bias && !dbias && !demb) || (!emb && !bias && dbias && demb))) return AVI_EINVAL;
    if (heads <= 0 || n <= 0 || heads * n * (n + 1) > 256 || 32 * heads > 256) return AVI_EINVAL;
    hipLaunchKernelGGL(rel_bias_kernel, dim3(1), dim3(256), 0, S_(stream), emb, bias, dbias, demb, heads, n);
    return avi_launch_status();
}
