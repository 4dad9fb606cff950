// FaceFormer autoregressive decode for WIDE decoders as a chain of small launches per frame (avi_faceformer_decode_steps).
//
// faceformer.hip gives every utterance one workgroup that streams all 8 D^2 fp32 weights through ONE CU per frame: at
// D = 1024 (config/vocaset/demo.yaml) that is 34 MB per frame at a CU's ~35 GB/s = 1.5 ms per frame, slower than the CPU.
// A frame is a chain of dependent matrix-vector products, each needing the whole previous vector, so the chain is cut
// into launches (a dependent launch costs ~1.5-2 us on this chip, less than a grid barrier) and every launch is spread
// over the chip:
//   attn   (B x 4 heads x S key splits)  q/k/v of the frame from the PREVIOUS coefficient frame through the fused matrix
//                                        in_proj . vertice_map (53 inputs instead of D), K/V appended to the cache,
//                                        split-key softmax partials (m, l, acc) over the cache (flash-decoding)
//   comb   (B x 4)                       merge of the S partials (only when S > 1)
//   out    (D/16)                        [out_proj | vertice_map] . [att | o] on the matrix cores -> s1 = x + sa (x = the
//                                        frame's input embedding + pe) and per-tile LayerNorm-1 statistics
//   ff1    (2D/16)                       x2 = LN2(LN1(s1) + cross_i) rebuilt by every workgroup in LDS, h = relu(linear1 x2)
//   ff2    (D/16)                        s3 = x2 + linear2 h and per-tile LayerNorm-3 statistics
//   mapr   (4)                           o = vertice_map_r LN3(s3) (LayerNorm applied while the operand is loaded), frame out
// A trivial launch costs ~3 us in a replayed graph, so LayerNorms are folded into their consumers instead of launched.
// Linear layers: a workgroup owns 16 output columns for all B rows (<= 32 = two MFMA row tiles), its 4 waves split K;
// weights are bf16 hi/lo planes in fragment-major order (one 1-KiB read per wave-instruction), activations are split
// into hi/lo when the fragment is built: 3 MFMAs per product, fp32-grade like the rest of the path.
#include "common.h"

namespace {

constexpr int NT = 256, NH = 4, VP = 64, SMAX = 32, MAXB = 32;

struct Chain {
    AviFaceformerWeights w;
    AviFaceformerPlanes p;
    const float* cross;
    float* kv;
    float* out;
    uint16_t* out16;      // IEEE half output instead of `out` (long-form: half the coefficient bytes), or NULL
    float *o, *part, *att, *s1, *x2, *h, *s3, *st1, *st3;
    int B, T, D, chunk;
};

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------------------- attention
// grid (S, NH, B).  Keys kstart + s*c .. of the chunk; frame i's own key/value are computed here (by the split that
// owns position i) and appended to the cache.
__global__ __launch_bounds__(NT) void ff_attn_kernel(const Chain c, const int i, const int S) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = c.D, dh = D / NH, tid = threadIdx.x;
    const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int kstart = (i / c.chunk) * c.chunk, nkeys = i - kstart + 1;
    const int per = (nkeys + S - 1) / S;
    const int j0 = kstart + s * per, j1 = min(i + 1, j0 + per);
    float* qs = sm;                 // [dh]
    float* ks = qs + dh;            // [dh]  key of frame i
    float* vs = ks + dh;            // [dh]  value of frame i
    float* ov = vs + dh;            // [64]  previous coefficient frame
    float* red = ov + VP;           // [8]
    float* accr = red + 8;          // [NT * 4] cross-group reduction of P.V
    float* sc = accr + NT * 4;      // [per]
    float* kvb = c.kv + (long long)b * c.T * 2 * D;
    const bool owns_i = j1 == i + 1 && j0 <= i;
    const int hoff = h * dh;
    // scores: LPK = dh/4 lanes per key, each a float4 of the head dimension.  The cache rows do not depend on q: the
    // first PF keys of every thread (all of them when a split is <= 32 KB) are requested before anything else.
    const int LPK = dh >> 2, groups = NT / LPK, g = tid / LPK, l = tid - g * LPK;
    constexpr int PF = 4;
    float4 kpre[PF], vpre[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int j = j0 + g + u * groups;
        kpre[u] = vpre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < j1 && j != i) {
            kpre[u] = *reinterpret_cast<const float4*>(kvb + (long long)j * 2 * D + hoff + 4 * l);
            vpre[u] = *reinterpret_cast<const float4*>(kvb + (long long)j * 2 * D + D + hoff + 4 * l);
        }
    }
    if (i > 0 && tid < VP) ov[tid] = c.o[b * VP + tid];
    // q (every split), k and v (the owning split): 64-long dot products against the fused matrix, coalesced over outputs.
    // The matrix columns of a thread's FIRST output do not depend on the previous frame: they are requested before the
    // barrier that publishes it (one memory round trip less on the frame's critical path: 10.1 -> 9.2 us per launch).
    const int nq = (owns_i ? 3 : 1) * dh;
    {
        const int idx0 = tid < nq ? tid : nq - 1;
        const int which0 = idx0 / dh, d0 = idx0 - which0 * dh, col0 = which0 * D + h * dh + d0;
        float a0, wv0[VP];
        if (i == 0) {
            a0 = c.p.qkv0[col0];
        } else {
            a0 = c.p.bf[(long long)(i % c.w.period) * 3 * D + col0];
            const float* wcol = c.p.wf_t + col0;
#pragma unroll
            for (int v = 0; v < VP; ++v) wv0[v] = wcol[(long long)v * 3 * D];
        }
        __syncthreads();
        if (i > 0) {
#pragma unroll
            for (int v = 0; v < VP; ++v) a0 = fmaf(wv0[v], ov[v], a0);
        }
        if (tid < nq) {
            (which0 == 0 ? qs : which0 == 1 ? ks : vs)[d0] = a0;
            if (which0 > 0) kvb[(long long)i * 2 * D + (which0 - 1) * D + h * dh + d0] = a0;
        }
    }
    for (int idx = tid + NT; idx < nq; idx += NT) {
        const int which = idx / dh, d = idx - which * dh, col = which * D + h * dh + d;
        float a;
        if (i == 0) {
            a = c.p.qkv0[col];
        } else {
            a = c.p.bf[(long long)(i % c.w.period) * 3 * D + col];
            const float* wcol = c.p.wf_t + col;
            float wv[VP];                      // all 64 loads in flight before the first is used
#pragma unroll
            for (int v = 0; v < VP; ++v) wv[v] = wcol[(long long)v * 3 * D];
#pragma unroll
            for (int v = 0; v < VP; ++v) a = fmaf(wv[v], ov[v], a);
        }
        (which == 0 ? qs : which == 1 ? ks : vs)[d] = a;
        if (which > 0) kvb[(long long)i * 2 * D + (which - 1) * D + h * dh + d] = a;
    }
    __syncthreads();
    const float scale = rsqrtf((float)dh), slope = c.w.slopes[h];
    const float4 q4 = *reinterpret_cast<const float4*>(qs + 4 * l);
    float mx = -3.0e38f;
    auto score = [&](int j, const float4 k4) {
        float d = q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
        for (int off = 1; off < LPK; off <<= 1) d += __shfl_xor(d, off, 64);
        d = d * scale - slope * (float)((i - j) / c.w.period);
        if (l == 0) sc[j - j0] = d;
        mx = fmaxf(mx, d);
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int j = j0 + g + u * groups;
        if (j < j1) score(j, j == i ? *reinterpret_cast<const float4*>(ks + 4 * l) : kpre[u]);
    }
#pragma unroll 4
    for (int j = j0 + g + PF * groups; j < j1; j += groups)
        score(j, j == i ? *reinterpret_cast<const float4*>(ks + 4 * l)
                        : *reinterpret_cast<const float4*>(kvb + (long long)j * 2 * D + hoff + 4 * l));
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
    for (int j = j0 + tid; j < j1; j += NT) {
        const float pj = __expf(sc[j - j0] - mx);
        sc[j - j0] = pj;
        sum += pj;
    }
    sum = block_reduce(sum, red, false);       // also publishes sc[] to every thread
    // P.V: thread (key group g, float4 column l)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    auto pv = [&](int j, const float4 v4) {
        const float pj = sc[j - j0];
        a.x = fmaf(pj, v4.x, a.x);
        a.y = fmaf(pj, v4.y, a.y);
        a.z = fmaf(pj, v4.z, a.z);
        a.w = fmaf(pj, v4.w, a.w);
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int j = j0 + g + u * groups;
        if (j < j1) pv(j, j == i ? *reinterpret_cast<const float4*>(vs + 4 * l) : vpre[u]);
    }
#pragma unroll 4
    for (int j = j0 + g + PF * groups; j < j1; j += groups)
        pv(j, j == i ? *reinterpret_cast<const float4*>(vs + 4 * l)
                     : *reinterpret_cast<const float4*>(kvb + (long long)j * 2 * D + D + hoff + 4 * l));
    *reinterpret_cast<float4*>(accr + 4 * tid) = a;
    __syncthreads();
    if (tid < dh) {
        const int lq = tid >> 2, comp = tid & 3;
        float r = 0.f;
        for (int gg = 0; gg < groups; ++gg) r += accr[4 * (gg * LPK + lq) + comp];
        if (S == 1) {
            c.att[(long long)b * D + hoff + tid] = r / sum;
        } else {
            float* pp = c.part + ((long long)(b * NH + h) * SMAX + s) * (dh + 2);
            pp[2 + tid] = r;
            if (tid == 0) {
                pp[0] = j1 > j0 ? mx : -3.0e38f;
                pp[1] = j1 > j0 ? sum : 0.f;
            }
        }
    }
}

// merge of the split-key partials: grid (NH, B)
__global__ __launch_bounds__(NT) void ff_comb_kernel(const Chain c, const int S) {
    const int D = c.D, dh = D / NH, h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const float* pp = c.part + (long long)(b * NH + h) * SMAX * (dh + 2);
    // every wave: lane s holds (m_s, l_s)
    const float2 ml = lane < S ? *reinterpret_cast<const float2*>(pp + lane * (dh + 2)) : make_float2(-3.0e38f, 0.f);
    const float M = wave_max(ml.x);
    const float wgt = __expf(ml.x - M);
    const float L = wave_sum(wgt * ml.y);
    if (tid < dh) {
        float r = 0.f;
#pragma unroll 8
        for (int s = 0; s < S; ++s) r += __shfl(wgt, s, 64) * pp[s * (dh + 2) + 2 + tid];
        c.att[(long long)b * D + h * dh + tid] = r / L;
    }
}

// ---------------------------------------------------------------------------------------------------------- linears
enum { EPI_OUT = 0, EPI_FF1 = 1, EPI_FF2 = 2, EPI_MAPR = 3 };

// Row statistics of a LayerNorm input that the PRODUCING launch leaves as per-tile partials (mean and M2 of the 16
// columns a workgroup owns, for every row): merged here by Chan's formula, one wave per row, lane = tile.
__device__ __forceinline__ void merge_row_stats(const float* __restrict__ part, int ntiles, int B, float* mean, float* rstd,
                                                int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = wave; m < B; m += 4) {
        float mu_t = 0.f, m2_t = 0.f;
        if (lane < ntiles) {
            const float2 v = *reinterpret_cast<const float2*>(part + ((long long)lane * MAXB + m) * 2);
            mu_t = v.x, m2_t = v.y;
        }
        const float mu = wave_sum(mu_t) / ntiles;
        const float dlt = lane < ntiles ? mu_t - mu : 0.f;
        const float m2 = wave_sum(m2_t + 16.f * dlt * dlt);
        if (lane == 0) {
            mean[m] = mu;
            rstd[m] = rsqrtf(m2 / D + 1e-5f);
        }
    }
}

// y[m][n] = sum_k a[m][k] W[n][k] for the 16 columns of tile blockIdx.x and rows m < B (RT row tiles of 16).
//   EPI_OUT : a = [att | o_{i-1}] (K = D + 64), W = [out_proj | vertice_map]  ->  s1 = x + self-attention (+ biases, + pe)
//             and the LayerNorm-1 partial statistics of s1
//   EPI_FF1 : prologue x2 = LN2(LN1(s1) + cross_i) for every row into LDS (statistics of LN1 from the partials, of LN2 by
//             a pass over the row kept in registers), a = x2 from LDS -> h = relu(linear1 x2); the workgroup also stores
//             its 8-column share of x2 for the residual of linear2
//   EPI_FF2 : a = h (K = 2D) -> s3 = x2 + linear2 h, and the LayerNorm-3 partial statistics of s3
//   EPI_MAPR: a = LN3(s3) applied while the fragment is loaded -> o = vertice_map_r, the frame written out
template <int EPI, int RT>
__global__ __launch_bounds__(NT) void ff_lin_kernel(const Chain c, const int i) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];          // EPI_FF1: x2 rows [RT*16][D + 4]
    __shared__ __attribute__((aligned(16))) float red[4][2][64][4];
    __shared__ float mean[MAXB], rstd[MAXB];
    constexpr int KB = RT == 1 ? 8 : 4;                                   // k-steps whose loads are issued together
    const int D = c.D, B = c.B, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
    const int fr = lane & 15, g = lane >> 4;
    const int K = EPI == EPI_FF2 ? 2 * D : EPI == EPI_OUT ? D + VP : D;
    const uint16_t* Whi = EPI == EPI_OUT ? c.p.wo_hi : EPI == EPI_FF1 ? c.p.w1_hi : EPI == EPI_FF2 ? c.p.w2_hi : c.p.wr_hi;
    const uint16_t* Wlo = EPI == EPI_OUT ? c.p.wo_lo : EPI == EPI_FF1 ? c.p.w1_lo : EPI == EPI_FF2 ? c.p.w2_lo : c.p.wr_lo;
    const int XS = D + 4;
    if (EPI == EPI_MAPR) {
        merge_row_stats(c.st3, D / 16, B, mean, rstd, D);
        __syncthreads();
    }
    if (EPI == EPI_FF1) {
        merge_row_stats(c.st1, D / 16, B, mean, rstd, D);
        __syncthreads();
        // t = LN1(s1) + cross_i, x2 = LN2(t): one wave per row, the row's D/64 <= 16 values per lane stay in registers
        for (int m = wave; m < B; m += 4) {
            const float* sr = c.s1 + (long long)m * D;
            const float* cr = c.cross + ((long long)m * c.T + i) * D;
            const float mu1 = mean[m], rs1 = rstd[m];
            f32x4 t[4];
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = (e * 64 + lane) * 4;
                t[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (d < D) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(sr + d), cc = *reinterpret_cast<const f32x4*>(cr + d);
                    const f32x4 g1 = *reinterpret_cast<const f32x4*>(c.w.n1g + d), b1 = *reinterpret_cast<const f32x4*>(c.w.n1b + d);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        t[e][j] = (a[j] - mu1) * rs1 * g1[j] + b1[j] + cc[j];
                        s += t[e][j];
                    }
                }
            }
            const float mu2 = wave_sum(s) / D;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = (e * 64 + lane) * 4;
                if (d < D) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) q += (t[e][j] - mu2) * (t[e][j] - mu2);
                }
            }
            const float rs2 = rsqrtf(wave_sum(q) / D + 1e-5f);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = (e * 64 + lane) * 4;
                if (d < D) {
                    const f32x4 g2 = *reinterpret_cast<const f32x4*>(c.w.n2g + d), b2 = *reinterpret_cast<const f32x4*>(c.w.n2b + d);
                    f32x4 x;
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = (t[e][j] - mu2) * rs2 * g2[j] + b2[j];
                    *reinterpret_cast<f32x4*>(dyn + m * XS + d) = x;
                }
            }
        }
        __syncthreads();
        // this workgroup's share of x2 for linear2's residual: D / (2D/16) = 8 columns of every row
        for (int idx = tid; idx < B * 8; idx += NT) {
            const int m = idx >> 3, n = tile * 8 + (idx & 7);
            c.x2[(long long)m * D + n] = dyn[m * XS + n];
        }
    }
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ksteps = K / 32;
    const long long wbase = (long long)tile * ksteps * 512 + lane * 8;
    for (int kb = wave * KB; kb < ksteps; kb += 4 * KB) {       // a wave owns KB consecutive k-steps out of every 4 KB
        u32x4 wh[KB], wl[KB];
        f32x4 av[KB][RT][2];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int ks = kb + u;
            if (ks < ksteps) {
                wh[u] = *reinterpret_cast<const u32x4*>(Whi + wbase + (long long)ks * 512);
                wl[u] = *reinterpret_cast<const u32x4*>(Wlo + wbase + (long long)ks * 512);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int m = rt * 16 + fr, k0 = ks * 32 + g * 8;
                av[u][rt][0] = av[u][rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ks < ksteps && m < B) {
                    const float* ap;
                    if (EPI == EPI_FF1) ap = dyn + m * XS + k0;
                    else if (EPI == EPI_OUT) ap = k0 < D ? c.att + (long long)m * D + k0 : c.o + m * VP + (k0 - D);
                    else if (EPI == EPI_FF2) ap = c.h + (long long)m * K + k0;
                    else ap = c.s3 + (long long)m * D + k0;
                    if (!(EPI == EPI_OUT && i == 0 && k0 >= D)) {       // frame 0 has no previous coefficient frame
                        av[u][rt][0] = *reinterpret_cast<const f32x4*>(ap);
                        av[u][rt][1] = *reinterpret_cast<const f32x4*>(ap + 4);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int ks = kb + u;
            if (ks >= ksteps) break;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, wh[u]), bl = __builtin_bit_cast(bf16x8, wl[u]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 v0 = av[u][rt][0], v1 = av[u][rt][1];
                if (EPI == EPI_MAPR) {
                    const int m = rt * 16 + fr;
                    if (m < B) {
                        const int k0 = ks * 32 + g * 8;
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(c.w.n3g + k0), g1 = *reinterpret_cast<const f32x4*>(c.w.n3g + k0 + 4);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(c.w.n3b + k0), b1 = *reinterpret_cast<const f32x4*>(c.w.n3b + k0 + 4);
                        const float mu = mean[m], rs = rstd[m];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v0[j] = (v0[j] - mu) * rs * g0[j] + b0[j];
                            v1[j] = (v1[j] - mu) * rs * g1[j] + b1[j];
                        }
                    }
                }
                bf16x8 xh, xl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xv = j < 4 ? v0[j] : v1[j - 4];
                    const __bf16 hi = (__bf16)xv;
                    xh[j] = hi;
                    xl[j] = (__bf16)(xv - (float)hi);
                }
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, xh, acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, xl, acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, xh, acc[rt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<f32x4*>(&red[wave][rt][lane][0]) = acc[rt];
    __syncthreads();
    // wave rt finishes row tile rt: lane (row m = lane & 15, columns n0 .. n0 + 3 with n0 = 4 * (lane >> 4))
    if (wave < RT) {
        const int rt = wave, m = rt * 16 + fr, n0 = tile * 16 + 4 * g;
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            y[j] = (red[0][rt][lane][j] + red[1][rt][lane][j]) + (red[2][rt][lane][j] + red[3][rt][lane][j]);
        const bool row = m < B;
        if (EPI == EPI_OUT) {
            // s1 = self-attention + x, x = vertice_map(o_{i-1}) + pe_i (frame 0: obj_embedding + pe_0); the
            // vertice_map product came out of the same accumulation (K = D + 64)
            const f32x4 bo = *reinterpret_cast<const f32x4*>(c.w.bo + n0);
            f32x4 xb;
            if (i == 0) {
                xb = *reinterpret_cast<const f32x4*>(c.p.x0 + n0);
            } else {
                const f32x4 bm = *reinterpret_cast<const f32x4*>(c.w.bm + n0);
                const f32x4 pe = *reinterpret_cast<const f32x4*>(c.w.pe + (long long)(i % c.w.period) * D + n0);
#pragma unroll
                for (int j = 0; j < 4; ++j) xb[j] = bm[j] + pe[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] += bo[j] + xb[j];
            if (row) *reinterpret_cast<f32x4*>(c.s1 + (long long)m * D + n0) = y;
        } else if (EPI == EPI_FF1) {
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(c.w.b1 + n0);
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fmaxf(y[j] + b1[j], 0.f);
            if (row) *reinterpret_cast<f32x4*>(c.h + (long long)m * 2 * D + n0) = y;
        } else if (EPI == EPI_FF2) {
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(c.w.b2 + n0);
            f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (row) r = *reinterpret_cast<const f32x4*>(c.x2 + (long long)m * D + n0);
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] += b2[j] + r[j];
            if (row) *reinterpret_cast<f32x4*>(c.s3 + (long long)m * D + n0) = y;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + j;
                const bool real = n < c.w.V;
                float v = real ? y[j] + c.w.br[n] : 0.f;
                y[j] = v;       // normalised frame: what vertice_map feeds back (models/faceformer.py:722-725)
                if (real && row) {
                    if (c.w.coeff_std) v = v * c.w.coeff_std[n] + c.w.coeff_mean[n];       // :729
                    const long long oi = ((long long)m * c.T + i) * c.w.V + n;
                    if (c.out16) c.out16[oi] = __builtin_bit_cast(uint16_t, (_Float16)v);
                    else c.out[oi] = v;
                }
            }
            if (row) *reinterpret_cast<f32x4*>(c.o + m * VP + n0) = y;
        }
        if (EPI == EPI_OUT || EPI == EPI_FF2) {
            // partial statistics of the next LayerNorm over this tile's 16 columns of row m: lanes m + 16 q, q = 0..3
            float s = (y[0] + y[1]) + (y[2] + y[3]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float mu = s * (1.f / 16.f);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) q += (y[j] - mu) * (y[j] - mu);
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (g == 0 && row)
                *reinterpret_cast<float2*>((EPI == EPI_OUT ? c.st1 : c.st3) + ((long long)tile * MAXB + m) * 2) = make_float2(mu, q);
        }
    }
}

}  // namespace

extern "C" int avi_faceformer_steps_work_floats(int D, int B, long long* floats) {
    if (!floats || D <= 0 || B <= 0) return AVI_EINVAL;
    const long long dh = D / NH;
    *floats = (long long)B * VP + (long long)B * NH * SMAX * (dh + 2) + (long long)B * D * 4 + (long long)B * 2 * D +
              2LL * (D / 16) * MAXB * 2 + 64;
    return AVI_OK;
}

template <int EPI>
static void launch_lin(const Chain& c, int i, int tiles, hipStream_t s) {
    const size_t smem = EPI == EPI_FF1 ? sizeof(float) * (size_t)(c.B > 16 ? 32 : 16) * (c.D + 4) : 0;
    if (c.B > 16) {
        if (EPI == EPI_FF1) {
            static AviLdsGrant grant;
            grant.ensure(reinterpret_cast<const void*>(ff_lin_kernel<EPI, 2>), 150 * 1024);
        }
        hipLaunchKernelGGL((ff_lin_kernel<EPI, 2>), dim3(tiles), dim3(NT), smem, s, c, i);
    } else {
        if (EPI == EPI_FF1) {
            static AviLdsGrant grant;
            grant.ensure(reinterpret_cast<const void*>(ff_lin_kernel<EPI, 1>), 150 * 1024);
        }
        hipLaunchKernelGGL((ff_lin_kernel<EPI, 1>), dim3(tiles), dim3(NT), smem, s, c, i);
    }
}

static int decode_steps_impl(const AviFaceformerWeights* w, const AviFaceformerPlanes* p, const float* cross,
                             int B, int T, int chunk, float* kv_scratch, float* work, float* out, uint16_t* out16,
                             void* stream) {
    if (!w || !p || !cross || !kv_scratch || !work || (!out && !out16) || B <= 0 || B > MAXB || T <= 0) return AVI_EINVAL;
    const int D = w->D, dh = D / NH;
    if (D % 64 || D > 1024 || (dh != 16 && dh != 32 && dh != 64 && dh != 128 && dh != 256)) return AVI_EINVAL;
    if (w->V < 1 || w->V > VP || w->period < 1) return AVI_EINVAL;
    if (chunk <= 0 || chunk > T) chunk = T;
    if (chunk < T && chunk % w->period) return AVI_EINVAL;
    if (!p->wo_hi || !p->wo_lo || !p->w1_hi || !p->w1_lo || !p->w2_hi || !p->w2_lo || !p->wr_hi || !p->wr_lo || !p->wf_t ||
        !p->bf || !p->qkv0 || !p->x0 || !w->bo || !w->b1 || !w->b2 || !w->br || !w->bm || !w->pe || !w->slopes ||
        !w->n1g || !w->n1b || !w->n2g || !w->n2b || !w->n3g || !w->n3b)
        return AVI_EINVAL;
    if ((w->coeff_mean == nullptr) != (w->coeff_std == nullptr)) return AVI_EINVAL;
    Chain c;
    c.w = *w;
    c.p = *p;
    c.cross = cross;
    c.kv = kv_scratch;
    c.out = out;
    c.out16 = out16;
    c.B = B, c.T = T, c.D = D, c.chunk = chunk;
    float* q = work;
    c.o = q, q += (long long)B * VP;
    c.part = q, q += (long long)B * NH * SMAX * (dh + 2);
    c.att = q, q += (long long)B * D;
    c.s1 = q, q += (long long)B * D;
    c.x2 = q, q += (long long)B * D;
    c.s3 = q, q += (long long)B * D;
    c.h = q, q += (long long)B * 2 * D;
    c.st1 = q, q += (long long)(D / 16) * MAXB * 2;
    c.st3 = q;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < T; ++i) {
        const int nkeys = i - (i / chunk) * chunk + 1;
        // one split per ~32 KB of K/V a head has to read, so that no workgroup streams more than a CU takes in ~0.5 us
        const long long bytes = (long long)nkeys * 2 * dh * 4;
        int S = (int)((bytes + 32 * 1024 - 1) / (32 * 1024));
        S = S < 1 ? 1 : S > SMAX ? SMAX : S;
        const int per = (nkeys + S - 1) / S;
        const size_t smem = sizeof(float) * (3 * dh + VP + 8 + NT * 4 + per);
        hipLaunchKernelGGL(ff_attn_kernel, dim3(S, NH, B), dim3(NT), smem, s, c, i, S);
        if (S > 1) hipLaunchKernelGGL(ff_comb_kernel, dim3(NH, B), dim3(NT), 0, s, c, S);
        launch_lin<EPI_OUT>(c, i, D / 16, s);
        launch_lin<EPI_FF1>(c, i, 2 * D / 16, s);
        launch_lin<EPI_FF2>(c, i, D / 16, s);
        launch_lin<EPI_MAPR>(c, i, VP / 16, s);
    }
    return avi_launch_status();
}

extern "C" int avi_faceformer_decode_steps(const AviFaceformerWeights* w, const AviFaceformerPlanes* p, const float* cross,
                                           int B, int T, int chunk, float* kv_scratch, float* work, float* out,
                                           void* stream) {
    return decode_steps_impl(w, p, cross, B, T, chunk, kv_scratch, work, out, nullptr, stream);
}
extern "C" int avi_faceformer_decode_steps_f16(const AviFaceformerWeights* w, const AviFaceformerPlanes* p,
                                               const float* cross, int B, int T, int chunk, float* kv_scratch, float* work,
                                               uint16_t* out16, void* stream) {
    return decode_steps_impl(w, p, cross, B, T, chunk, kv_scratch, work, nullptr, out16, stream);
}
