// Training forward of the 3-token denoiser as ONE launch (models/diffusion_prior.py:119-313 with the dalle2 blocks, the
// pass p_losses differentiates: train_diffusion_prior.py:449).  The step used to walk the 6 layers as ~55 dependent
// launches of 5-10 us (4 GEMMs, 3 LayerNorms, attention, SwiGLU, 2 split-K epilogues per layer: 0.46 ms for 12 MFLOP per
// sample); here a workgroup carries up to 5 samples = 15 token rows through all layers exactly like the DDPM sampler
// (prior_mfma.inc: residual stream in LDS, weights streamed as fragment-major bf16 hi/lo planes, 3-term split) and stores
// every intermediate the backward pass needs.  The planes are re-packed from the flat bf16 hi/lo parameter buffers by
// one table-driven launch per step (avi_pack_fragment_planes: the weights change every step).
#include "prior_mfma.inc"

namespace {

struct TrainArgs {
    AviPriorWeights w;
    AviPriorPlanes p;
    AviPriorTrainDump d;
};

__device__ __forceinline__ const TrainArgs& targs() { return *(const TrainArgs*)__builtin_amdgcn_kernarg_segment_ptr(); }

// rows [0, R) x C floats from an LDS image (row stride ld floats) to global (row stride C)
__device__ __forceinline__ void dump_rows(const float* lds, int ld, float* __restrict__ dst, int C, int R) {
    const int c4 = C >> 2;
    for (int i = threadIdx.x; i < R * c4; i += NT) {
        const int r = i / c4, c = (i - r * c4) * 4;
        *reinterpret_cast<f32x4*>(dst + (long long)r * C + c) = *reinterpret_cast<const f32x4*>(lds + r * ld + c);
    }
}

__global__ __launch_bounds__(NT, 2) void prior_train_fwd_kernel(const TrainArgs args_by_value, int B, int S) {
    const TrainArgs& a = targs();
    const AviPriorWeights& w = a.w;
    const AviPriorTrainDump& d = a.d;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SmemS& s = *reinterpret_cast<SmemS*>(smem_raw);   // inputs of the linears as split planes (prior_mfma.inc)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * S;
    const int Sg = min(S, B - b0), R = 3 * Sg;
    const long long row0 = 3LL * b0;                       // first token row of this group in the [3B][C] arrays
    const long long RT = 3LL * B;                          // rows per layer in the stacked dumps
    for (int i = tid; i < MR * XPS; i += NT) (&s.xh[0][0])[i] = (&s.xl[0][0])[i] = 0;
    for (int i = tid; i < MR * DIM; i += NT) (&s.tok[0][0])[i] = 0.f;
    for (int i = tid; i < w.depth * 3 * DIM; i += NT) {
        const int l = i / (3 * DIM), r = i - l * 3 * DIM, k = r / DIM, dd = r - k * DIM;
        const AviPriorLayer& L = w.layer[l];
        s.gain[l][k][dd] = (k == 0 ? L.norm_g : k == 1 ? L.out_g : L.ff_g)[dd];
    }
    for (int i = tid; i < w.depth * 2 * DH; i += NT) s.nkv[i / (2 * DH)][i % (2 * DH)] = w.layer[i / (2 * DH)].null_kv[i % (2 * DH)];
    for (int i = tid; i < DIM; i += NT) s.fin_g[i] = w.final_g[i];
    for (int i = tid; i < 96; i += NT) {
        s.relb[i] = w.rel_bias[i];
        s.rc[i] = w.rot_cos[i];
        s.rs[i] = w.rot_sin[i];
    }
    if (tid < w.depth) {
        float q2 = 0.f;
        for (int dd = 0; dd < DH; ++dd) q2 = fmaf(w.layer[tid].null_kv[dd], w.layer[tid].null_kv[dd], q2);
        s.nkinv[tid] = 1.f / fmaxf(sqrtf(q2), 1e-12f);
    }
    __syncthreads();
    for (int i = tid; i < R * DIM; i += NT) s.tok[i / DIM][i % DIM] = d.tok0[(row0 + i / DIM) * DIM + i % DIM];
    __syncthreads();

    bool pending = false;
    WRing ring;
    for (int l = 0; l < w.depth; ++l) {
        const AviPriorLayerPlanes& P = a.p.layer[l];
        Lin<DIM, NQKV>::prefetch(P.qkv_hi, P.qkv_lo, ring);
        // ---- residual += previous FF output; attention pre-LN
        for (int r = wave; r < R; r += 8) {
            float va = s.tok[r][lane], vb = s.tok[r][lane + 64];
            if (pending) {
                va += s.y[r][lane];
                vb += s.y[r][lane + 64];
                s.tok[r][lane] = va;
                s.tok[r][lane + 64] = vb;
            }
            float* ti = d.tok_in + ((long long)l * RT + row0 + r) * DIM;
            ti[lane] = va;
            ti[lane + 64] = vb;
            ln_row(va, vb, s.gain[l][0], lane, false);
            put_x<0>(s, r, lane, va);
            put_x<0>(s, r, lane + 64, vb);
            float* n1 = d.n1 + ((long long)l * RT + row0 + r) * DIM;
            n1[lane] = va;
            n1[lane + 64] = vb;
        }
        __syncthreads();
        // ---- q | k | v: raw outputs to global (the backward pass applies the rotary itself), rotated ones to LDS
        Lin<DIM, NQKV>::template run<true, SmemS, true>(P.qkv_hi, P.qkv_lo, ring, s, d.qkv + ((long long)l * RT + row0) * NQKV, R);
        __syncthreads();
        Lin<INNER, DIM>::prefetch(P.out_hi, P.out_lo, ring);
        {   // ---- attention (prior_mfma.inc phase C): wave = head, lane = dim
            const int h = __builtin_amdgcn_readfirstlane(wave);
            const float nk = s.nkv[l][lane], nv = s.nkv[l][DH + lane];
            const float ik0 = s.nkinv[l];
            for (int sm = 0; sm < Sg; ++sm) {
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
                    const float o = (e0 * nv + e1 * vd[0] + e2 * vd[1] + e3 * vd[2]) / (e0 + e1 + e2 + e3);
                    put_x<0>(s, 3 * sm + i, h * DH + lane, o);
                    d.ao[((long long)l * RT + row0 + 3 * sm + i) * INNER + h * DH + lane] = o;
                }
            }
        }
        __syncthreads();
        // ---- to_out.0
        Lin<INNER, DIM>::template run<false, SmemS, true>(P.out_hi, P.out_lo, ring, s, d.o1 + ((long long)l * RT + row0) * DIM, R);
        __syncthreads();
        Lin<DIM, 2 * FFI>::prefetch(P.w1_hi, P.w1_lo, ring);
        // ---- to_out.1 LayerNorm, residual, FF pre-LN
        for (int r = wave; r < R; r += 8) {
            float va = s.y[r][lane], vb = s.y[r][lane + 64];
            ln_row(va, vb, s.gain[l][1], lane, false);
            va += s.tok[r][lane];
            vb += s.tok[r][lane + 64];
            s.tok[r][lane] = va;
            s.tok[r][lane + 64] = vb;
            float* tm = d.tokm + ((long long)l * RT + row0 + r) * DIM;
            tm[lane] = va;
            tm[lane + 64] = vb;
            ln_row(va, vb, s.gain[l][2], lane, false);
            put_x<0>(s, r, lane, va);
            put_x<0>(s, r, lane + 64, vb);
            float* n2 = d.n2 + ((long long)l * RT + row0 + r) * DIM;
            n2[lane] = va;
            n2[lane + 64] = vb;
        }
        __syncthreads();
        // ---- FF in (value | gate)
        Lin<DIM, 2 * FFI>::template run<false, SmemS, true>(P.w1_hi, P.w1_lo, ring, s, d.hff + ((long long)l * RT + row0) * 2 * FFI, R);
        __syncthreads();
        Lin<FFI, DIM>::prefetch(P.w2_hi, P.w2_lo, ring);
        // ---- SwiGLU
        for (int o = tid; o < R * FFI; o += NT) {
            const int m = o / FFI, c = o - m * FFI;
            const float v = s.y[m][c] * silu(s.y[m][FFI + c]);
            put_x<0>(s, m, c, v);
            d.sw[((long long)l * RT + row0 + m) * FFI + c] = v;
        }
        __syncthreads();
        // ---- FF out
        Lin<FFI, DIM>::template run<false, SmemS, true>(P.w2_hi, P.w2_lo, ring, s);
        __syncthreads();
        pending = true;
    }
    Lin<DIM, DIM>::prefetch(a.p.proj_hi, a.p.proj_lo, ring);
    for (int r = wave; r < R; r += 8) {
        float va = s.tok[r][lane] + s.y[r][lane], vb = s.tok[r][lane + 64] + s.y[r][lane + 64];
        float* to = d.tok_out + (row0 + r) * DIM;
        to[lane] = va;
        to[lane + 64] = vb;
        ln_row(va, vb, s.fin_g, lane, true);
        put_x<0>(s, r, lane, va);
        put_x<0>(s, r, lane + 64, vb);
        float* fn = d.fin + (row0 + r) * DIM;
        fn[lane] = va;
        fn[lane + 64] = vb;
    }
    __syncthreads();
    Lin<DIM, DIM>::template run<false, SmemS, true>(a.p.proj_hi, a.p.proj_lo, ring, s, d.po + row0 * DIM, R);
}

// ============================================================================================== fused backward
// The dX chain of the six layers in ONE launch: from the gradient at the output of the last layer down to the gradient of
// the token rows, storing the output gradient of every matrix (the deferred dW launches read them), per-workgroup
// partial LayerNorm-gain gradients (reduced by one small launch: deterministic) and adding the null-kv / relative-bias
// gradients with atomics as the stand-alone attention backward does.  Linears run on the TRANSPOSED fragment-major
// planes (dX = dY . W = Lin over W^T).
constexpr int XPB = 1032, YSB = 516;   // XPB: 16-bit plane row stride (K up to 1024): 2064 B = 129 x 16 B, conflict-free fragment reads
struct AttnB {
    float qn[8][3][64], kn[4][64], vv[4][64], qinv[8][3], kinv[4], p[8][3][4], ds[8][3][4], dkn[4][64];
};
struct SmemB {
    float tok[MR][DIM];     // gradient on the residual stream
    float tmid[MR][DIM];    // gradient at the middle of the layer (after attention, before the feed-forward)
    uint16_t xh[MR][XPB], xl[MR][XPB];   // input of the next linear, split once by its producer (prior_mfma.inc SmemS)
    float y[MR][YSB];
    float gain[AVI_PRIOR_MAX_DEPTH][3][DIM];
    float relb[96];
    float dgw[8][2 * DIM];  // per-wave LayerNorm-gain partials (two LayerNorms handled in one phase)
    AttnB at[2];
};

struct TrainBwdArgs {
    AviPriorWeights w;
    AviPriorPlanes p;       // planes of the TRANSPOSED matrices
    AviPriorTrainBwd d;
};
__device__ __forceinline__ const TrainBwdArgs& bargs() { return *(const TrainBwdArgs*)__builtin_amdgcn_kernarg_segment_ptr(); }

// LayerNorm backward of one 128-wide row held as (a, b) = columns (lane, lane + 64): returns dx in (da, db) and adds
// dy * xhat to the wave's gain-gradient partial.  dalle2 LayerNorm: y = (x - mean) rsqrt(var + 1e-5) g.
__device__ __forceinline__ void ln_bwd_row(float xa, float xb, float& da, float& db, const float* g, int lane, float* dg) {
    const float mean = wave_sum_u(xa + xb) * (1.f / DIM);
    const float ca = xa - mean, cb = xb - mean;
    const float rstd = rsqrtf(wave_sum_u(ca * ca + cb * cb) * (1.f / DIM) + 1e-5f);
    const float ha = ca * rstd, hb = cb * rstd;
    dg[lane] += da * ha;
    dg[lane + 64] += db * hb;
    const float ta = da * g[lane], tb = db * g[lane + 64];
    const float sa = wave_sum_u(ta + tb) * (1.f / DIM), sb = wave_sum_u(ta * ha + tb * hb) * (1.f / DIM);
    da = rstd * (ta - sa - ha * sb);
    db = rstd * (tb - sa - hb * sb);
}

// prior_attn_bwd_kernel (train.hip) for ONE sample on 256 threads t = 0..255 of a 512-thread workgroup (the two halves
// work on two samples at once; every barrier is reached by both).  qkvb: the sample's raw projections [3][640] (global),
// dout(i, c): gradient of the attention output, row i (LDS), dq(i, c): where the projection gradients go (LDS).
template <class DOUT, class DQ>
__device__ __forceinline__ void attn_bwd_sample(int t, bool live, const float* __restrict__ qkvb, const float* __restrict__ null_kv,
                                                const float* relb, const float* __restrict__ rot_cos,
                                                const float* __restrict__ rot_sin, DOUT dout, DQ dq, float* __restrict__ dnull_kv,
                                                float* __restrict__ drel, float* __restrict__ part, AttnB& c) {
    // part != nullptr: this (layer, sample)'s 96 relative-bias and 128 null-kv gradient contributions are STORED there and
    // summed in sample order by prior_attn_grad_kernel (deterministic); nullptr: float atomics into drel / dnull_kv
    const int lane = t & 63, wave = t >> 6;
    if (live) {
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
    }
    __syncthreads();
    if (live && t < 96) {
        const int h = t / 12, r = t - h * 12, i = r >> 2, j = r & 3;
        float a = 0.f, dp = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) {
            a = fmaf(c.qn[h][i][d], c.kn[j][d], a);
            dp = fmaf(dout(i, h * 64 + d), c.vv[j][d], dp);
        }
        c.p[h][i][j] = a + relb[(h * 3 + i) * 4 + j];
        c.ds[h][i][j] = dp;
    }
    __syncthreads();
    if (live && t < 24) {
        float* r = &c.p[0][0][0] + t * 4;
        const float mx = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
        const float e0 = __expf(r[0] - mx), e1 = __expf(r[1] - mx), e2 = __expf(r[2] - mx), e3 = __expf(r[3] - mx);
        const float inv = 1.f / (e0 + e1 + e2 + e3);
        r[0] = e0 * inv; r[1] = e1 * inv; r[2] = e2 * inv; r[3] = e3 * inv;
        float* dpr = &c.ds[0][0][0] + t * 4;
        const float dot = dpr[0] * r[0] + dpr[1] * r[1] + dpr[2] * r[2] + dpr[3] * r[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) dpr[j] = r[j] * (dpr[j] - dot);
    }
    __syncthreads();
    if (live) {
        if (t < 96) {
            if (part) part[t] = (&c.ds[0][0][0])[t];
            else atomicAdd(&drel[t], (&c.ds[0][0][0])[t]);
        }
        const int j = wave, d = lane;
        float a = 0.f, kk = 0.f;
        for (int h = 0; h < 8; ++h)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a = fmaf(c.p[h][i][j], dout(i, h * 64 + d), a);
                kk = fmaf(c.ds[h][i][j], c.qn[h][i][d], kk);
            }
        if (j == 0) {
            if (part) part[96 + 64 + d] = a;
            else atomicAdd(&dnull_kv[64 + d], a);
        } else {
            dq(j - 1, 576 + d, a);
        }
        c.dkn[j][d] = kk;
    }
    __syncthreads();
    if (live) {
        for (int vix = wave; vix < 28; vix += 4) {
            float dy, y, inv;
            int pos = -1;
            if (vix < 24) {
                const int h = vix / 3, i = vix - h * 3;
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) a = fmaf(c.ds[h][i][j], c.kn[j][lane], a);
                dy = a; y = c.qn[h][i][lane]; inv = c.qinv[h][i]; pos = i;
            } else {
                const int j = vix == 27 ? 0 : vix - 23;
                dy = c.dkn[j][lane]; y = c.kn[j][lane]; inv = c.kinv[j]; pos = vix == 27 ? -1 : vix - 24;
            }
            const float ydy = wave_sum(y * dy);
            float dx = inv * (4.0f * dy - y * ydy * 0.25f);
            if (pos >= 0) {
                const float partner = __shfl_xor(dx, 1, 64);
                if (lane < 32) {
                    const float cs = rot_cos[pos * 32 + lane], sn = rot_sin[pos * 32 + lane];
                    dx = dx * cs + ((lane & 1) ? -partner : partner) * sn;
                }
            }
            if (vix < 24) {
                const int h = vix / 3, i = vix - h * 3;
                dq(i, h * 64 + lane, dx * 16.0f);
            } else if (vix < 27) {
                dq(vix - 24, 512 + lane, dx);
            } else if (part) {
                part[96 + lane] = dx;
            } else {
                atomicAdd(&dnull_kv[lane], dx);
            }
        }
    }
    __syncthreads();
}

constexpr int ATTN_PART = 96 + 128;     // floats per (layer, sample) of AviPriorTrainBwd.attn_part

__global__ __launch_bounds__(NT, 2) void prior_train_bwd_kernel(const TrainBwdArgs args_by_value, int B, int S) {
    const TrainBwdArgs& a = bargs();
    const AviPriorWeights& w = a.w;
    const AviPriorTrainBwd& d = a.d;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SmemB& s = *reinterpret_cast<SmemB*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * S;
    const int Sg = min(S, B - b0), R = 3 * Sg;
    const long long row0 = 3LL * b0, RT = 3LL * B;
    for (int i = tid; i < MR * XPB; i += NT) (&s.xh[0][0])[i] = (&s.xl[0][0])[i] = 0;
    for (int i = tid; i < MR * DIM; i += NT) {
        (&s.tok[0][0])[i] = 0.f;
        (&s.tmid[0][0])[i] = 0.f;
    }
    for (int i = tid; i < w.depth * 3 * DIM; i += NT) {
        const int l = i / (3 * DIM), r = i - l * 3 * DIM, k = r / DIM, dd = r - k * DIM;
        const AviPriorLayer& L = w.layer[l];
        s.gain[l][k][dd] = (k == 0 ? L.norm_g : k == 1 ? L.out_g : L.ff_g)[dd];
    }
    for (int i = tid; i < 96; i += NT) s.relb[i] = w.rel_bias[i];
    __syncthreads();
    for (int i = tid; i < R * DIM; i += NT) s.tok[i / DIM][i % DIM] = d.dtok_top[(row0 + i / DIM) * DIM + i % DIM];
    __syncthreads();
    WRing ring;
    float* dgp = d.dgamma_part + (long long)blockIdx.x * w.depth * 3 * DIM;
    for (int l = w.depth - 1; l >= 0; --l) {
        const AviPriorLayerPlanes& P = a.p.layer[l];
        Lin<DIM, FFI>::prefetch(P.w2_hi, P.w2_lo, ring);
        // ---- dy of linear2 = the gradient arriving at the layer; its rows are the input of dX = dY . W2
        for (int i = tid; i < R * DIM; i += NT) {
            const int r = i / DIM, c = i - r * DIM;
            const float v = s.tok[r][c];
            put_x<0>(s, r, c, v);
            d.dy_w2[((long long)l * RT + row0 + r) * DIM + c] = v;
        }
        __syncthreads();
        Lin<DIM, FFI>::template run<false, SmemB, true>(P.w2_hi, P.w2_lo, ring, s);                       // dsw [R][512]
        __syncthreads();
        Lin<2 * FFI, DIM>::prefetch(P.w1_hi, P.w1_lo, ring);
        // ---- SwiGLU backward: hff = (value | gate) from the forward dump
        for (int o = tid; o < R * FFI; o += NT) {
            const int m = o / FFI, c = o - m * FFI;
            const float* hr = d.hff + ((long long)l * RT + row0 + m) * 2 * FFI;
            const float av = hr[c], g = hr[FFI + c], dy = s.y[m][c];
            const float sg = 1.f / (1.f + __expf(-g));
            const float da = dy * g * sg, dgt = dy * av * sg * (1.f + g * (1.f - sg));
            put_x<0>(s, m, c, da);
            put_x<0>(s, m, FFI + c, dgt);
            float* o1 = d.dy_w1 + ((long long)l * RT + row0 + m) * 2 * FFI;
            o1[c] = da;
            o1[FFI + c] = dgt;
        }
        __syncthreads();
        Lin<2 * FFI, DIM>::template run<false, SmemB, true>(P.w1_hi, P.w1_lo, ring, s);                   // dn2 [R][128]
        __syncthreads();
        Lin<DIM, INNER>::prefetch(P.out_hi, P.out_lo, ring);
        // ---- LayerNorm (ff 0.g) backward on tokm, + residual -> dtokm; LayerNorm (to_out.1.g) backward on o1 -> do1
        {
            float* dg = s.dgw[wave];
            dg[lane] = dg[lane + 64] = dg[DIM + lane] = dg[DIM + lane + 64] = 0.f;
            for (int r = wave; r < R; r += 8) {
                const float* tm = d.tokm + ((long long)l * RT + row0 + r) * DIM;
                float da = s.y[r][lane], db = s.y[r][lane + 64];
                ln_bwd_row(tm[lane], tm[lane + 64], da, db, s.gain[l][2], lane, dg);
                da += s.tok[r][lane];
                db += s.tok[r][lane + 64];
                s.tmid[r][lane] = da;
                s.tmid[r][lane + 64] = db;
                const float* o1 = d.o1 + ((long long)l * RT + row0 + r) * DIM;
                ln_bwd_row(o1[lane], o1[lane + 64], da, db, s.gain[l][1], lane, dg + DIM);
                put_x<0>(s, r, lane, da);
                put_x<0>(s, r, lane + 64, db);
                float* dyo = d.dy_out + ((long long)l * RT + row0 + r) * DIM;
                dyo[lane] = da;
                dyo[lane + 64] = db;
            }
        }
        __syncthreads();
        if (tid < 2 * DIM) {                       // gain gradients of this workgroup: ff 0.g (slot 2), to_out.1.g (slot 1)
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) v += s.dgw[wv][tid];
            dgp[(l * 3 + (tid < DIM ? 2 : 1)) * DIM + (tid & (DIM - 1))] = v;
        }
        Lin<DIM, INNER>::template run<false, SmemB, true>(P.out_hi, P.out_lo, ring, s);                   // dao [R][512]
        __syncthreads();
        Lin<NQKV, DIM>::prefetch(P.qkv_hi, P.qkv_lo, ring);
        // ---- attention backward, two samples at a time (threads 0..255 / 256..511)
        for (int sm0 = 0; sm0 < Sg; sm0 += 2) {
            const int half = tid >> 8, sm = sm0 + half;
            const bool live = sm < Sg;
            const int rb = 3 * (live ? sm : 0);
            const float* qkvb = d.qkv + ((long long)l * RT + row0 + rb) * NQKV;
            float* dyq = d.dy_qkv + ((long long)l * RT + row0 + rb) * NQKV;
            auto dout = [&](int i, int c) { return s.y[rb + i][c]; };
            auto dq = [&](int i, int c, float v) {
                put_x<0>(s, rb + i, c, v);
                dyq[i * NQKV + c] = v;
            };
            float* part = d.attn_part ? d.attn_part + ((long long)l * B + b0 + (live ? sm : 0)) * ATTN_PART : nullptr;
            attn_bwd_sample(tid & 255, live, qkvb, w.layer[l].null_kv, s.relb, w.rot_cos, w.rot_sin, dout, dq,
                            d.dnull_kv[l], d.drel, part, s.at[half]);
        }
        Lin<NQKV, DIM>::template run<false, SmemB, true>(P.qkv_hi, P.qkv_lo, ring, s);                    // dn1 [R][128]
        __syncthreads();
        // ---- LayerNorm (norm.g) backward on the layer's input, + dtokm -> gradient leaving the layer
        {
            float* dg = s.dgw[wave];
            dg[lane] = dg[lane + 64] = 0.f;
            for (int r = wave; r < R; r += 8) {
                const float* ti = d.tok_in + ((long long)l * RT + row0 + r) * DIM;
                float da = s.y[r][lane], db = s.y[r][lane + 64];
                ln_bwd_row(ti[lane], ti[lane + 64], da, db, s.gain[l][0], lane, dg);
                s.tok[r][lane] = da + s.tmid[r][lane];
                s.tok[r][lane + 64] = db + s.tmid[r][lane + 64];
            }
        }
        __syncthreads();
        if (tid < DIM) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) v += s.dgw[wv][tid];
            dgp[(l * 3 + 0) * DIM + tid] = v;
        }
        __syncthreads();
    }
    for (int i = tid; i < R * DIM; i += NT) d.dtok0[(row0 + i / DIM) * DIM + i % DIM] = s.tok[i / DIM][i % DIM];
}

// the null-kv gradients are accumulated with atomics: zeroed by this launch first, so that the caller need not clear the
// gradient buffer (grid depth, 128 threads)
__global__ __launch_bounds__(128) void prior_zero_null_kv_kernel(AviPriorTrainBwd d) { d.dnull_kv[blockIdx.x][threadIdx.x] = 0.f; }

// null-kv gradient of layer l = sum over the samples, in sample order, of the stored contributions (block l < depth, 128
// threads); block `depth`: the relative-bias gradient = sum over layers, then samples (96 threads; the bias table is shared
// by the layers).  Plain stores: run-to-run identical, unlike the atomics they replace.
__global__ __launch_bounds__(128) void prior_attn_grad_kernel(AviPriorTrainBwd d, int B, int depth) {
    const int l = blockIdx.x, t = threadIdx.x;
    if (l < depth) {
        float v = 0.f;
        for (int b = 0; b < B; ++b) v += d.attn_part[((long long)l * B + b) * ATTN_PART + 96 + t];
        d.dnull_kv[l][t] = v;
    } else if (t < 96) {
        float v = 0.f;
        for (int ll = 0; ll < depth; ++ll)
            for (int b = 0; b < B; ++b) v += d.attn_part[((long long)ll * B + b) * ATTN_PART + t];
        d.drel[t] = v;
    }
}

// gain gradient = sum of the workgroups' partials, in workgroup order: grid (depth * 3), 128 threads
__global__ __launch_bounds__(128) void prior_gain_grad_kernel(const float* __restrict__ part, int groups, int depth,
                                                               AviPriorGainGrads out) {
    const int li = blockIdx.x, l = li / 3, k = li - 3 * l, c = threadIdx.x;
    float v = 0.f;
    for (int g = 0; g < groups; ++g) v += part[((long long)g * depth * 3 + li) * DIM + c];
    out.g[l][k][c] = v;
}

// dst plane in fragment-major order [N/16][K/32][64 lanes][8] from a row-major [N][K] plane (or, transpose != 0, from the
// row-major [K][N] plane of the transposed matrix): one thread = one lane slot of 8 values
__global__ __launch_bounds__(256) void pack_fragment_planes_kernel(const AviPlaneJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AviPlaneJob jb = jobs[lo];
    const long long slot = (long long)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    const long long slots = (long long)jb.N * jb.K / 8;
    if (slot >= slots) return;
    const int ln = (int)(slot & 63);
    const long long blk = slot >> 6;
    const int ksteps = jb.K / 32;
    const int tile = (int)(blk / ksteps), ks = (int)(blk - (long long)tile * ksteps);
    const int n = tile * 16 + (ln & 15), k0 = ks * 32 + (ln >> 4) * 8;
    if (!jb.transpose) {
        const long long so = (long long)n * jb.K + k0;
        *reinterpret_cast<u32x4*>(jb.dst_hi + slot * 8) = *reinterpret_cast<const u32x4*>(jb.src_hi + so);
        *reinterpret_cast<u32x4*>(jb.dst_lo + slot * 8) = *reinterpret_cast<const u32x4*>(jb.src_lo + so);
    } else {   // source is W [K][N] row-major; the packed matrix is W^T [N][K]
        uint16_t h[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = jb.src_hi[(long long)(k0 + j) * jb.N + n];
            l[j] = jb.src_lo[(long long)(k0 + j) * jb.N + n];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            jb.dst_hi[slot * 8 + j] = h[j];
            jb.dst_lo[slot * 8 + j] = l[j];
        }
    }
}

}  // namespace

extern "C" int avi_pack_fragment_planes(const AviPlaneJob* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || njobs < 1 || total_blocks < 1) return AVI_EINVAL;
    hipLaunchKernelGGL(pack_fragment_planes_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       jobs_dev, njobs);
    return avi_launch_status();
}

extern "C" int avi_prior_train_forward(const AviPriorWeights* w, const AviPriorPlanes* p, const AviPriorTrainDump* d, int B,
                                       int samples_per_group, void* stream) {
    if (!w || !p || !d || B <= 0 || samples_per_group < 1 || samples_per_group > SMAX) return AVI_EINVAL;
    if (w->depth < 1 || w->depth > AVI_PRIOR_MAX_DEPTH || !p->proj_hi || !p->proj_lo || !w->rel_bias || !w->rot_cos ||
        !w->rot_sin || !w->final_g)
        return AVI_EINVAL;
    for (int l = 0; l < w->depth; ++l) {
        const AviPriorLayerPlanes& P = p->layer[l];
        const AviPriorLayer& L = w->layer[l];
        if (!P.qkv_hi || !P.qkv_lo || !P.out_hi || !P.out_lo || !P.w1_hi || !P.w1_lo || !P.w2_hi || !P.w2_lo) return AVI_EINVAL;
        if (!L.norm_g || !L.out_g || !L.ff_g || !L.null_kv) return AVI_EINVAL;
    }
    if (!d->tok0 || !d->tok_in || !d->n1 || !d->qkv || !d->ao || !d->o1 || !d->tokm || !d->n2 || !d->hff || !d->sw ||
        !d->tok_out || !d->fin || !d->po)
        return AVI_EINVAL;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(prior_train_fwd_kernel), (int)sizeof(SmemS));
    TrainArgs args;
    args.w = *w;
    args.p = *p;
    args.d = *d;
    const int groups = (B + samples_per_group - 1) / samples_per_group;
    hipLaunchKernelGGL(prior_train_fwd_kernel, dim3(groups), dim3(NT), sizeof(SmemS), static_cast<hipStream_t>(stream), args,
                       B, samples_per_group);
    return avi_launch_status();
}

extern "C" int avi_prior_train_backward(const AviPriorWeights* w, const AviPriorPlanes* pT, const AviPriorTrainBwd* d,
                                        const AviPriorGainGrads* gains, int B, int samples_per_group, void* stream) {
    if (!w || !pT || !d || !gains || B <= 0 || samples_per_group < 1 || samples_per_group > SMAX) return AVI_EINVAL;
    if (w->depth < 1 || w->depth > AVI_PRIOR_MAX_DEPTH || !w->rel_bias || !w->rot_cos || !w->rot_sin) return AVI_EINVAL;
    for (int l = 0; l < w->depth; ++l) {
        const AviPriorLayerPlanes& P = pT->layer[l];
        const AviPriorLayer& L = w->layer[l];
        if (!P.qkv_hi || !P.qkv_lo || !P.out_hi || !P.out_lo || !P.w1_hi || !P.w1_lo || !P.w2_hi || !P.w2_lo) return AVI_EINVAL;
        if (!L.norm_g || !L.out_g || !L.ff_g || !L.null_kv || !d->dnull_kv[l]) return AVI_EINVAL;
        for (int k = 0; k < 3; ++k)
            if (!gains->g[l][k]) return AVI_EINVAL;
    }
    if (!d->dtok_top || !d->tok_in || !d->qkv || !d->o1 || !d->tokm || !d->hff || !d->dy_w2 || !d->dy_w1 || !d->dy_out ||
        !d->dy_qkv || !d->dtok0 || !d->dgamma_part || !d->drel)
        return AVI_EINVAL;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(prior_train_bwd_kernel), (int)sizeof(SmemB));
    TrainBwdArgs args;
    args.w = *w;
    args.p = *pT;
    args.d = *d;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int groups = (B + samples_per_group - 1) / samples_per_group;
    if (!d->attn_part) hipLaunchKernelGGL(prior_zero_null_kv_kernel, dim3(w->depth), dim3(128), 0, s, *d);
    hipLaunchKernelGGL(prior_train_bwd_kernel, dim3(groups), dim3(NT), sizeof(SmemB), s, args, B, samples_per_group);
    hipLaunchKernelGGL(prior_gain_grad_kernel, dim3(w->depth * 3), dim3(128), 0, s, d->dgamma_part, groups, w->depth, *gains);
    if (d->attn_part) hipLaunchKernelGGL(prior_attn_grad_kernel, dim3(w->depth + 1), dim3(128), 0, s, *d, B, w->depth);
    return avi_launch_status();
}
