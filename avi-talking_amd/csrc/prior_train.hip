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
    Smem& s = *reinterpret_cast<Smem*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * S;
    const int Sg = min(S, B - b0), R = 3 * Sg;
    const long long row0 = 3LL * b0;                       // first token row of this group in the [3B][C] arrays
    const long long RT = 3LL * B;                          // rows per layer in the stacked dumps
    for (int i = tid; i < MR * XS; i += NT) (&s.x[0][0])[i] = 0.f;
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
            s.x[r][lane] = va;
            s.x[r][lane + 64] = vb;
            float* n1 = d.n1 + ((long long)l * RT + row0 + r) * DIM;
            n1[lane] = va;
            n1[lane + 64] = vb;
        }
        __syncthreads();
        // ---- q | k | v: raw outputs to global (the backward pass applies the rotary itself), rotated ones to LDS
        Lin<DIM, NQKV>::template run<true>(P.qkv_hi, P.qkv_lo, ring, s, d.qkv + ((long long)l * RT + row0) * NQKV, R);
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
                    s.x[3 * sm + i][h * DH + lane] = o;
                    d.ao[((long long)l * RT + row0 + 3 * sm + i) * INNER + h * DH + lane] = o;
                }
            }
        }
        __syncthreads();
        // ---- to_out.0
        Lin<INNER, DIM>::run(P.out_hi, P.out_lo, ring, s, d.o1 + ((long long)l * RT + row0) * DIM, R);
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
            s.x[r][lane] = va;
            s.x[r][lane + 64] = vb;
            float* n2 = d.n2 + ((long long)l * RT + row0 + r) * DIM;
            n2[lane] = va;
            n2[lane + 64] = vb;
        }
        __syncthreads();
        // ---- FF in (value | gate)
        Lin<DIM, 2 * FFI>::run(P.w1_hi, P.w1_lo, ring, s, d.hff + ((long long)l * RT + row0) * 2 * FFI, R);
        __syncthreads();
        Lin<FFI, DIM>::prefetch(P.w2_hi, P.w2_lo, ring);
        // ---- SwiGLU
        for (int o = tid; o < R * FFI; o += NT) {
            const int m = o / FFI, c = o - m * FFI;
            const float v = s.y[m][c] * silu(s.y[m][FFI + c]);
            s.x[m][c] = v;
            d.sw[((long long)l * RT + row0 + m) * FFI + c] = v;
        }
        __syncthreads();
        // ---- FF out
        Lin<FFI, DIM>::run(P.w2_hi, P.w2_lo, ring, s);
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
        s.x[r][lane] = va;
        s.x[r][lane + 64] = vb;
        float* fn = d.fin + (row0 + r) * DIM;
        fn[lane] = va;
        fn[lane + 64] = vb;
    }
    __syncthreads();
    Lin<DIM, DIM>::run(a.p.proj_hi, a.p.proj_lo, ring, s, d.po + row0 * DIM, R);
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
    lds_grant.ensure(reinterpret_cast<const void*>(prior_train_fwd_kernel), (int)sizeof(Smem));
    TrainArgs args;
    args.w = *w;
    args.p = *p;
    args.d = *d;
    const int groups = (B + samples_per_group - 1) / samples_per_group;
    hipLaunchKernelGGL(prior_train_fwd_kernel, dim3(groups), dim3(NT), sizeof(Smem), static_cast<hipStream_t>(stream), args,
                       B, samples_per_group);
    return avi_launch_status();
}
