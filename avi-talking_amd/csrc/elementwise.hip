// HBM-bound kernels around the GEMMs: audio normalisation, conv layer 0 + GroupNorm + GELU,
// resample + LayerNorm, LayerNorm, layout packing.  All fp32, 16-B accesses, channels-last.
#include "common.h"
#include "gelu_table.h"

namespace {

// split-plane helpers: x = hi + lo (two bf16), 4 values at a time
__device__ __forceinline__ void split4_store(uint16_t* hi, uint16_t* lo, long long o, const float v[4], int fmt,
                                             AviF16Range* rng = nullptr) {
    uint16_t h[4], l[4];
    if (rng && fmt == AVI_PLANES_F16) {       // fp16 planes: range guard (common.h)
#pragma unroll
        for (int j = 0; j < 4; ++j) rng->see(v[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) avi_split_hl(v[j], fmt, h[j], l[j]);
    *reinterpret_cast<uint2*>(hi + o) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
    *reinterpret_cast<uint2*>(lo + o) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
}
__device__ __forceinline__ float4 join4_load(const uint16_t* hi, const uint16_t* lo, long long o) {
    const uint2 h = *reinterpret_cast<const uint2*>(hi + o), l = *reinterpret_cast<const uint2*>(lo + o);
    auto f = [](uint32_t b) { return __builtin_bit_cast(float, b); };
    return make_float4(f(h.x << 16) + f(l.x << 16), f(h.x & 0xffff0000u) + f(l.x & 0xffff0000u),
                       f(h.y << 16) + f(l.y << 16), f(h.y & 0xffff0000u) + f(l.y & 0xffff0000u));
}

// Accumulators are zeroed by a kernel of this library rather than hipMemsetAsync (one dependency less on the runtime's
// blit path inside captured graphs).
__global__ void zero_doubles_kernel(double* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}
static inline void zero_doubles(double* p, int n, hipStream_t s) {
    hipLaunchKernelGGL(zero_doubles_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, n);
}

// ------------------------------------------------------------------ audio zero-mean / unit-var
// stats[2*s + {0,1}] = sum, sum of squares (double) of clip s (s = 0 when joint).
template <bool I16>
__global__ __launch_bounds__(256) void audio_stats_kernel(const void* __restrict__ pcm, int N, int joint, int vec16,
                                                           double* __restrict__ stats) {
    __shared__ double red[2][4];
    const int b = blockIdx.y;
    const long long base = (long long)b * N;
    double s = 0.0, q = 0.0;
    const int vec = I16 ? 8 : 4;                              // samples per 16-byte load
    if (vec16) {                   // host-checked: the base pointer and every clip start (N % vec == 0) are 16-byte aligned
        for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * vec; i < N; i += gridDim.x * blockDim.x * vec) {
            float x[8];
            if (I16) {
                const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const int16_t*>(pcm) + base + i);
                const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[2 * j] = (float)(int16_t)(w[j] & 0xffffu);
                    x[2 * j + 1] = (float)(int16_t)(w[j] >> 16);
                }
            } else {
                const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(pcm) + base + i);
                x[0] = r.x, x[1] = r.y, x[2] = r.z, x[3] = r.w;
                x[4] = x[5] = x[6] = x[7] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s += (double)x[j];
                q = fma((double)x[j], (double)x[j], q);
            }
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
            const float x = I16 ? (float)reinterpret_cast<const int16_t*>(pcm)[base + i]
                                : reinterpret_cast<const float*>(pcm)[base + i];
            s += x;
            q += (double)x * x;
        }
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    // one atomic pair per WORKGROUP: in joint mode every wave of the grid used to add into the same two doubles
    // (8192 serialized memory-side atomics = 0.2 ms for a 10 MB read)
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int slot = joint ? 0 : b;
        atomicAdd(&stats[2 * slot], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&stats[2 * slot + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

template <bool I16>
__global__ void audio_apply_kernel(const void* __restrict__ pcm, int N, int B, int joint, float eps,
                                   const double* __restrict__ stats, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int slot = joint ? 0 : b;
    const double cnt = joint ? (double)N * B : (double)N;
    const double mean = stats[2 * slot] / cnt;
    double var = stats[2 * slot + 1] / cnt - mean * mean;  // population variance (numpy .var())
    if (var < 0.0) var = 0.0;
    const float fm = (float)mean, inv = (float)(1.0 / sqrt(var + (double)eps));
    const long long base = (long long)b * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float x = I16 ? (float)reinterpret_cast<const int16_t*>(pcm)[base + i]
                            : reinterpret_cast<const float*>(pcm)[base + i];
        out[base + i] = (x - fm) * inv;
    }
}

// ------------------------------------------------------------------ conv layer 0 + GroupNorm + GELU
// GroupNorm(512 groups over 512 channels) normalises each channel over time.  conv0 is linear in the
// 10-sample window, so the per-channel mean / variance over t follow from the window's first and
// second moments:  mean_c = w_c . S1 / T0,  E[y_c^2] = w_c^T S2 w_c / T0.  One pass over the audio
// (65 sums per clip) replaces a full evaluation of the 512 x T0 activation.
constexpr int C0 = 512, K0 = 10, ST0 = 5, NMOM = 65;  // 10 first + 55 second moments

// First and second window moments of the audio: mom[j] = sum_t x[5t+j], mom[10 + p(j,i)] = sum_t x[5t+j] x[5t+i] (j <= i).
// Every thread OWNS one moment (tid % 65) over a third of the workgroup's 512 window positions and keeps its sum in a
// double: no cross-lane reduction at all.  (One position per lane with 65 wave-wide double reductions per wave cost
// 100 us for 10 MB of audio.)
constexpr int MOM_CH = 512;     // window positions per workgroup
constexpr int MOM_LDS = (MOM_CH + 24) * ST0 + K0 + 1;            // + zero padding read by the last runs, + the constant 1
__global__ __launch_bounds__(256) void conv0_moments_kernel(const float* __restrict__ x, int N, int T0,
                                                             double* __restrict__ mom) {
    __shared__ float sx[MOM_LDS];
    __shared__ double part[3][NMOM + 1];
    const int b = blockIdx.y, t0 = blockIdx.x * MOM_CH, t1 = min(T0, t0 + MOM_CH);
    const float* xb = x + (long long)b * N;
    const int nload = (t1 - t0 - 1) * ST0 + K0;
    {   // branch-free staging: clamped addresses, every load of a thread in flight at once, zeros past the chunk
        constexpr int NL = (MOM_LDS + 255) / 256;
        float v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int i = threadIdx.x + 256 * u;
            v[u] = xb[t0 * ST0 + min(i, nload - 1)];
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < MOM_LDS) sx[i] = i < nload ? v[u] : (i == MOM_LDS - 1 ? 1.f : 0.f);
        }
    }
    __syncthreads();
    const int mi = threadIdx.x % NMOM, sub = threadIdx.x / NMOM;      // 3 x 65 = 195 working threads
    if (sub < 3) {
        int ja = mi, jb = -1;
        if (mi >= K0) {                     // pair index p -> (j, i), rows of lengths 10, 9, ..., 1
            int p = mi - K0, j = 0;
            while (p >= K0 - j) { p -= K0 - j; ++j; }
            ja = j, jb = j + p;
        }
        // first moments multiply by the constant 1 (stride 0): one code path, no branch in the loop; positions past
        // the chunk read zeros.  Runs of 8 positions are summed in fp32 (8 independent LDS reads in flight), the runs
        // in double.
        const float* pa = sx + ja;
        const float* pb = jb < 0 ? sx + MOM_LDS - 1 : sx + jb;
        const int sb = jb < 0 ? 0 : ST0;
        double acc = 0.0;
        const int n = t1 - t0;
        for (int t = sub; t < n; t += 24) {
            float run = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)       // a select, not a branch: positions past the chunk contribute nothing
                run = fmaf(t + 3 * u < n ? pa[(t + 3 * u) * ST0] : 0.f, pb[(t + 3 * u) * sb], run);
            acc += (double)run;
        }
        part[sub][mi] = acc;
    }
    __syncthreads();
    // one slot per (clip, chunk): no atomics, no zeroing; conv0_finalize_kernel adds the chunks in a fixed order
    if (threadIdx.x < NMOM)
        mom[((long long)b * gridDim.x + blockIdx.x) * NMOM + threadIdx.x] =
            (part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x];
}
__global__ void conv0_finalize_kernel(const double* __restrict__ mom, int nchunks, const float* __restrict__ w0,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, int T0,
                                      float eps, float* __restrict__ ss) {
    __shared__ double mb[NMOM];
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < NMOM) {            // the clip's moments = sum over its chunks, in chunk order
        double a = 0.0;
#pragma unroll 8
        for (int k = 0; k < nchunks; ++k) a += mom[((long long)b * nchunks + k) * NMOM + threadIdx.x];
        mb[threadIdx.x] = a;
    }
    __syncthreads();
    if (c >= C0) return;
    double w[K0];
#pragma unroll
    for (int j = 0; j < K0; ++j) w[j] = (double)w0[c * K0 + j];
    double m = 0.0, e2 = 0.0;
    int p = 0;
#pragma unroll
    for (int j = 0; j < K0; ++j) {
        m += w[j] * mb[j];
#pragma unroll
        for (int i = j; i < K0; ++i) {
            const double v = w[j] * w[i] * mb[K0 + p++];
            e2 += (i == j) ? v : 2.0 * v;
        }
    }
    m /= T0;
    double var = e2 / T0 - m * m;
    if (var < 0.0) var = 0.0;
    const double sc = (double)gamma[c] / sqrt(var + (double)eps);
    ss[(long long)b * 2 * C0 + c] = (float)sc;
    ss[(long long)b * 2 * C0 + C0 + c] = (float)((double)beta[c] - m * sc);
}

// y[b][t][c] = gelu(conv0(x)[t][c] * scale[b][c] + shift[b][c]);  block = 64 t x 512 c.
constexpr int C0_TT = 64;
__global__ __launch_bounds__(256) void conv0_apply_kernel(const float* __restrict__ x, int N, int T0,
                                                           const float* __restrict__ w0,
                                                           const float* __restrict__ ss, float* __restrict__ y,
                                                           uint16_t* __restrict__ y_hi, uint16_t* __restrict__ y_lo,
                                                           int fmt, unsigned* __restrict__ status) {
    __shared__ float sx[C0_TT * ST0 + K0 + 2];
    // 524 M outputs per step at ~37 vector instructions each made this kernel the vector pipe's (0.47-0.52 ms for 2.1 GB of
    // stores); GELU from the LDS table (gelu_table.h) is 12 of them fewer
    __shared__ __attribute__((aligned(16))) float gtab[512];
    const int b = blockIdx.y, t0 = blockIdx.x * C0_TT;
    const float* xb = x + (long long)b * N;
    const int nload = C0_TT * ST0 + K0 - ST0;  // samples needed by 64 outputs
    for (int i = threadIdx.x; i < nload; i += blockDim.x) {
        const int s = t0 * ST0 + i;
        sx[i] = s < N ? xb[s] : 0.f;
    }
    for (int i = threadIdx.x; i < 512; i += blockDim.x) gtab[i] = avi_gelu_tab[i];
    const int cg = threadIdx.x & 127;  // 4 channels per thread
    const int tsub = threadIdx.x >> 7; // 2 time rows in flight
    float w[4][K0], sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cg * 4 + q;
#pragma unroll
        for (int j = 0; j < K0; ++j) w[q][j] = w0[c * K0 + j];
        sc[q] = ss[(long long)b * 2 * C0 + c];
        sh[q] = ss[(long long)b * 2 * C0 + C0 + c];
    }
    __syncthreads();
    AviF16Range rng;
    for (int tt = tsub; tt < C0_TT; tt += 2) {
        const int t = t0 + tt;
        if (t >= T0) break;
        float xv[K0];
#pragma unroll
        for (int j = 0; j < K0; ++j) xv[j] = sx[tt * ST0 + j];
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < K0; ++j) a = fmaf(w[q][j], xv[j], a);
            o[q] = avi_gelu_lds(a * sc[q] + sh[q], reinterpret_cast<const char*>(gtab));
        }
        const long long off = ((long long)b * T0 + t) * C0 + cg * 4;
        if (y) *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
        if (y_hi) split4_store(y_hi, y_lo, off, o, fmt, &rng);
    }
    if (y_hi && fmt == AVI_PLANES_F16) rng.commit(status);     // tsub is wave-uniform: the wave left the loop together
}

// ------------------------------------------------------------------ LayerNorm helpers (one wave per row)
// BR = false: one wave per row (4 rows per workgroup), MAXV float4 per lane: C <= 256 MAXV.
// BR = true : one WORKGROUP per row, MAXV float4 per thread: C <= 1024 MAXV.  For a few wide rows (the aligner's 64 x 4096)
//             one wave per row leaves 16 waves on the chip walking 80 KB each (24 us); a workgroup per row takes 6.
template <int MAXV, bool BR = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* in, int rows, int C,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, int act,
                                                         const float* __restrict__ mask, const float* residual,
                                                         int stable, float* out, uint16_t* __restrict__ out_hi,
                                                         uint16_t* __restrict__ out_lo, int fmt,
                                                         unsigned* __restrict__ status) {
    __shared__ float red[3][4];
    constexpr int LN = BR ? 256 : 64;
    const int row = BR ? blockIdx.x : blockIdx.x * 4 + (threadIdx.x >> 6), lane = BR ? threadIdx.x : threadIdx.x & 63;
    if (row >= rows) return;
    auto rsum = [&](float t, int slot) { return BR ? block256_reduce<false>(t, red, slot) : wave_sum(t); };
    auto rmax = [&](float t, int slot) { return BR ? block256_reduce<true>(t, red, slot) : wave_max(t); };
    const float* x = in + (long long)row * C;
    const int nv = C >> 2;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + LN * i;
        v[i] = idx < nv ? reinterpret_cast<const float4*>(x)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (stable) {  // dalle2 LayerNorm(stable=True): x / x.amax(-1) first
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (lane + LN * i < nv) mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        const float inv = 1.f / rmax(mx, 0);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) { v[i].x *= inv; v[i].y *= inv; v[i].z *= inv; v[i].w *= inv; }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = rsum(s, 1) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + LN * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(rsum(q, 2) / C + eps);
    float* o = out ? out + (long long)row * C : nullptr;
    AviF16Range rng;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + LN * i;
        if (idx < nv) {
            float4 g = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gamma) g = reinterpret_cast<const float4*>(gamma)[idx];
            if (beta) bb = reinterpret_cast<const float4*>(beta)[idx];
            float4 r;
            r.x = avi_act((v[i].x - mean) * rstd * g.x + bb.x, act);
            r.y = avi_act((v[i].y - mean) * rstd * g.y + bb.y, act);
            r.z = avi_act((v[i].z - mean) * rstd * g.z + bb.z, act);
            r.w = avi_act((v[i].w - mean) * rstd * g.w + bb.w, act);
            if (mask) {  // dropout keep-mask, already scaled by 1/(1-p)
                const float4 mk = reinterpret_cast<const float4*>(mask + (long long)row * C)[idx];
                r.x *= mk.x; r.y *= mk.y; r.z *= mk.z; r.w *= mk.w;
            }
            if (residual) {
                const float4 rr = reinterpret_cast<const float4*>(residual + (long long)row * C)[idx];
                r.x += rr.x; r.y += rr.y; r.z += rr.z; r.w += rr.w;
            }
            if (o) reinterpret_cast<float4*>(o)[idx] = r;
            if (out_hi) {
                const float rv[4] = {r.x, r.y, r.z, r.w};
                split4_store(out_hi, out_lo, (long long)row * C + idx * 4, rv, fmt, &rng);
            }
        }
    }
    if (out_hi && fmt == AVI_PLANES_F16) rng.commit(status);   // a row belongs to whole waves: no lane has left
}

// ------------------------------------------------------------------ split-K epilogue (skinny GEMMs of the aligner MLP)
// y[r][:] = sum_z parts[z][r][:] + bias, then optionally LayerNorm -> act -> + residual.  One workgroup per row,
// MAXV float4 per thread (C <= 1024 * MAXV); partial sums are added in z order, so the result is deterministic.
template <int MAXV>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ parts, int nparts,
                                                               long long part_stride, int C,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int do_ln,
                                                               int act, const float* residual, float* out) {
    __shared__ float red[2][4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = C >> 2;
    float4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = tid + 256 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < nv) {
            // eight partial sums in flight, added in z order (one dependent load per z cost 24 us for 16 slices)
            const float* pp = parts + (long long)row * C;
            int z = 0;
            for (; z + 8 <= nparts; z += 8) {
                float4 p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = reinterpret_cast<const float4*>(pp + (z + u) * part_stride)[idx];
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[i].x += p[u].x; v[i].y += p[u].y; v[i].z += p[u].z; v[i].w += p[u].w; }
            }
            for (; z < nparts; ++z) {
                const float4 p = reinterpret_cast<const float4*>(pp + z * part_stride)[idx];
                v[i].x += p.x; v[i].y += p.y; v[i].z += p.z; v[i].w += p.w;
            }
            if (bias) {
                const float4 b = reinterpret_cast<const float4*>(bias)[idx];
                v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w;
            }
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (do_ln) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
        s = wave_sum(s);
        if (lane == 0) red[0][wave] = s;
        __syncthreads();
        mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (tid + 256 * i < nv) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += a * a + b * b + c * c + d * d;
            }
        q = wave_sum(q);
        if (lane == 0) red[1][wave] = q;
        __syncthreads();
        rstd = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / C + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = tid + 256 * i;
        if (idx >= nv) continue;
        float4 r = v[i];
        if (do_ln) {
            float4 g = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gamma) g = reinterpret_cast<const float4*>(gamma)[idx];
            if (beta) bb = reinterpret_cast<const float4*>(beta)[idx];
            r.x = (r.x - mean) * rstd * g.x + bb.x;
            r.y = (r.y - mean) * rstd * g.y + bb.y;
            r.z = (r.z - mean) * rstd * g.z + bb.z;
            r.w = (r.w - mean) * rstd * g.w + bb.w;
        }
        r.x = avi_act(r.x, act); r.y = avi_act(r.y, act); r.z = avi_act(r.z, act); r.w = avi_act(r.w, act);
        if (residual) {
            const float4 rr = reinterpret_cast<const float4*>(residual + (long long)row * C)[idx];
            r.x += rr.x; r.y += rr.y; r.z += rr.z; r.w += rr.w;
        }
        reinterpret_cast<float4*>(out + (long long)row * C)[idx] = r;
    }
}

// out[b][t] = LN( lerp(in[b][i0], in[b][i1]) ), align_corners=True index math of
// torch upsample_linear1d: scale = (Tin-1)/(Tout-1), src = scale*t, i0 = (int)src, l1 = src - i0.
template <int MAXV>
__global__ __launch_bounds__(256) void interp_ln_kernel(const float* __restrict__ in, const uint16_t* __restrict__ in_hi,
                                                         const uint16_t* __restrict__ in_lo, int B, int Tin, int C,
                                                         int Tout, const float scale, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * Tout) return;
    const int b = row / Tout, t = row - b * Tout;
    // `scale` = (Tin-1)/(Tout-1) is divided on the host (IEEE).  The empty asm pins the rounded product:
    // hipcc otherwise contracts scale*t - i0 into one v_fma_f32 (even through __fmul_rn/__fsub_rn), which
    // moves the lerp weight by ~1e-5 away from torch's CPU upsample_linear1d.
    float src = scale * (float)t;
    asm volatile("" : "+v"(src));
    int i0 = (int)src;
    if (i0 > Tin - 1) i0 = Tin - 1;
    const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
    const float l1 = src - (float)i0, l0 = 1.f - l1;
    const long long o0 = ((long long)b * Tin + i0) * C, o1 = ((long long)b * Tin + i1) * C;
    const int nv = C >> 2;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 a = in ? *reinterpret_cast<const float4*>(in + o0 + idx * 4) : join4_load(in_hi, in_lo, o0 + idx * 4);
            const float4 c = in ? *reinterpret_cast<const float4*>(in + o1 + idx * 4) : join4_load(in_hi, in_lo, o1 + idx * 4);
            v[i] = make_float4(l0 * a.x + l1 * c.x, l0 * a.y + l1 * c.y, l0 * a.z + l1 * c.z, l0 * a.w + l1 * c.w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    float4* o = reinterpret_cast<float4*>(out + (long long)row * C);
    if (!gamma) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + 64 * i;
            if (idx < nv) o[idx] = v[i];
        }
        return;
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + bb * bb + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[idx];
            const float4 bb = reinterpret_cast<const float4*>(beta)[idx];
            float4 r;
            r.x = (v[i].x - mean) * rstd * g.x + bb.x;
            r.y = (v[i].y - mean) * rstd * g.y + bb.y;
            r.z = (v[i].z - mean) * rstd * g.z + bb.z;
            r.w = (v[i].w - mean) * rstd * g.w + bb.w;
            o[idx] = r;
        }
    }
}

// ------------------------------------------------------------------ layout kernels
__global__ void group_pad_pack_kernel(const float* __restrict__ h, int B, int T, int G, int Cg, int pad,
                                      float* __restrict__ xg) {
    const int Tp = T + 2 * pad, v4 = Cg >> 2;
    const long long total = (long long)B * G * Tp * v4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % v4);
        long long r = i / v4;
        const int tp = (int)(r % Tp);
        r /= Tp;
        const int g = (int)(r % G), b = (int)(r / G);
        const int t = tp - pad;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T)
            val = *reinterpret_cast<const float4*>(h + ((long long)b * T + t) * (G * Cg) + g * Cg + cv * 4);
        reinterpret_cast<float4*>(xg)[i] = val;
    }
}

__global__ void pad_repeat_kernel(const float* __restrict__ in, int B, int T, int C, int rep, int padL, int padR,
                                  int mode, float* __restrict__ out) {
    const int To = padL + T * rep + padR, v4 = C >> 2;
    const long long total = (long long)B * To * v4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % v4);
        long long r = i / v4;
        const int to = (int)(r % To), b = (int)(r / To);
        int src = to - padL;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        bool valid = true;
        if (src < 0) { valid = mode == 1; src = 0; }
        else if (src >= T * rep) { valid = mode == 1; src = T * rep - 1; }
        if (valid) val = *reinterpret_cast<const float4*>(in + ((long long)b * T + src / rep) * C + cv * 4);
        reinterpret_cast<float4*>(out)[i] = val;
    }
}

__global__ void add_rowbcast_kernel(const float* __restrict__ in, const float* __restrict__ add, int B, int T, int C,
                                    float* __restrict__ out) {
    const int v4 = C >> 2;
    const long long total = (long long)B * T * v4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % v4);
        const int b = (int)(i / ((long long)T * v4));
        const float4 a = reinterpret_cast<const float4*>(in)[i];
        const float4 d = reinterpret_cast<const float4*>(add + (long long)b * C)[cv];
        reinterpret_cast<float4*>(out)[i] = make_float4(a.x + d.x, a.y + d.y, a.z + d.z, a.w + d.w);
    }
}

// CLIPTextEmbeddings: out[b][t][:] = table[ids[b][t]][:] + pos[t][:]  (one float4 per thread; an id outside the table
// is clamped - no fault - and counted so that the host can raise what nn.Embedding would have raised)
__global__ void embed_tokens_kernel(const long long* __restrict__ ids, const float* __restrict__ table,
                                    const float* __restrict__ pos, int B, int T, int C, int vocab,
                                    float* __restrict__ out, int* __restrict__ bad_ids) {
    const int v4 = C >> 2;
    const long long total = (long long)B * T * v4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % v4);
        const long long row = i / v4;
        const int t = (int)(row % T);
        long long id = ids[row];
        if (id < 0 || id >= vocab) {
            if (cv == 0 && bad_ids) atomicAdd(bad_ids, 1);
            id = id < 0 ? 0 : vocab - 1;
        }
        const float4 a = reinterpret_cast<const float4*>(table + id * C)[cv];
        const float4 d = reinterpret_cast<const float4*>(pos + (long long)t * C)[cv];
        reinterpret_cast<float4*>(out)[i] = make_float4(a.x + d.x, a.y + d.y, a.z + d.z, a.w + d.w);
    }
}

// out[b][c] = mean over t of in[b][t][c]; one thread per (b, c), consecutive threads on consecutive channels
__global__ void mean_tokens_kernel(const float* __restrict__ in, int B, int T, int C, float* __restrict__ out) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)B * C) return;
    const int b = (int)(i / C), c = (int)(i % C);
    const float* p = in + (long long)b * T * C + c;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += p[(long long)t * C];
    out[i] = s / (float)T;
}

inline int grid_for(long long total, int block = 256, int cap = 8192) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int avi_audio_normalize(const void* pcm, int is_int16, int B, int N, int joint, float eps, float* out,
                                   double* stats, void* stream) {
    if (!pcm || !out || !stats || B <= 0 || N <= 0) return AVI_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    zero_doubles(stats, 2 * B, s);
    dim3 sgrid(grid_for(N / 8, 256, 32), B); // up to 32 workgroups per clip (16-byte loads): 64 atomics per clip
    dim3 grid(grid_for(N, 256, 64), B);
    // 16-byte loads only when the buffer allows them (a pointer taken from an offset view need not be aligned)
    const int vec16 = ((reinterpret_cast<uintptr_t>(pcm) & 15) == 0) && (N % (is_int16 ? 8 : 4)) == 0;
    if (is_int16) {
        hipLaunchKernelGGL(audio_stats_kernel<true>, sgrid, dim3(256), 0, s, pcm, N, joint, vec16, stats);
        hipLaunchKernelGGL(audio_apply_kernel<true>, grid, dim3(256), 0, s, pcm, N, B, joint, eps, stats, out);
    } else {
        hipLaunchKernelGGL(audio_stats_kernel<false>, sgrid, dim3(256), 0, s, pcm, N, joint, vec16, stats);
        hipLaunchKernelGGL(audio_apply_kernel<false>, grid, dim3(256), 0, s, pcm, N, B, joint, eps, stats, out);
    }
    return avi_launch_status();
}

static int conv0_impl(const float* x, int B, int N, const float* w0, const float* gamma, const float* beta, float eps,
                      float* y, uint16_t* y_hi, uint16_t* y_lo, double* moments, float* scale_shift, int fmt, void* stream) {
    if (fmt != AVI_PLANES_BF16 && fmt != AVI_PLANES_F16) return AVI_EINVAL;
    if (!x || !w0 || !gamma || !beta || (!y && !y_hi) || !moments || !scale_shift || B <= 0 || N < K0) return AVI_EINVAL;
    if ((y_hi == nullptr) != (y_lo == nullptr)) return AVI_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T0 = (N - K0) / ST0 + 1;
    const int nchunks = (T0 + MOM_CH - 1) / MOM_CH;
    hipLaunchKernelGGL(conv0_moments_kernel, dim3(nchunks, B), dim3(256), 0, s, x, N, T0, moments);
    hipLaunchKernelGGL(conv0_finalize_kernel, dim3(C0 / 256, B), dim3(256), 0, s, moments, nchunks, w0, gamma, beta, T0,
                       eps, scale_shift);
    hipLaunchKernelGGL(conv0_apply_kernel, dim3((T0 + C0_TT - 1) / C0_TT, B), dim3(256), 0, s, x, N, T0, w0,
                       scale_shift, y, y_hi, y_lo, fmt, avi_status_ptr());
    return avi_launch_status();
}

extern "C" int avi_conv0_gn_gelu(const float* x, int B, int N, const float* w0, const float* gamma,
                                 const float* beta, float eps, float* y, double* moments, float* scale_shift,
                                 void* stream) {
    return conv0_impl(x, B, N, w0, gamma, beta, eps, y, nullptr, nullptr, moments, scale_shift, AVI_PLANES_BF16, stream);
}

extern "C" int avi_conv0_gn_gelu_planes(const float* x, int B, int N, const float* w0, const float* gamma,
                                        const float* beta, float eps, uint16_t* y_hi, uint16_t* y_lo, double* moments,
                                        float* scale_shift, int plane_fmt, void* stream) {
    return conv0_impl(x, B, N, w0, gamma, beta, eps, nullptr, y_hi, y_lo, moments, scale_shift, plane_fmt, stream);
}

extern "C" int avi_interp_layernorm(const float* in, int B, int Tin, int C, int Tout, const float* gamma,
                                    const float* beta, float eps, float* out, void* stream) {
    if (!in || !out || B <= 0 || Tin <= 0 || Tout <= 0 || (C & 3) || C > 1024) return AVI_EINVAL;
    if ((gamma == nullptr) != (beta == nullptr)) return AVI_EINVAL;
    const int rows = B * Tout;
    const float scale = Tout > 1 ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f;
    hipLaunchKernelGGL(interp_ln_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                       (const uint16_t*)nullptr, (const uint16_t*)nullptr, B, Tin, C, Tout, scale, gamma, beta, eps, out);
    return avi_launch_status();
}

extern "C" int avi_interp_layernorm_planes(const uint16_t* in_hi, const uint16_t* in_lo, int B, int Tin, int C, int Tout,
                                           const float* gamma, const float* beta, float eps, float* out, void* stream) {
    if (!in_hi || !in_lo || !out || B <= 0 || Tin <= 0 || Tout <= 0 || (C & 3) || C > 1024) return AVI_EINVAL;
    if ((gamma == nullptr) != (beta == nullptr)) return AVI_EINVAL;
    const int rows = B * Tout;
    const float scale = Tout > 1 ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f;
    hipLaunchKernelGGL(interp_ln_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const float*)nullptr, in_hi, in_lo, B, Tin, C, Tout, scale, gamma, beta, eps, out);
    return avi_launch_status();
}

extern "C" int avi_layernorm_ex(const float* in, int rows, int C, const float* gamma, const float* beta, float eps,
                                int act, const float* mask, const float* residual, int stable, float* out,
                                void* stream) {
    if (!in || !out || rows <= 0 || C <= 0 || (C & 3) || C > 4096) return AVI_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((rows + 3) / 4), block(256);
    if (C <= 1024)
        hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, in, rows, C, gamma, beta, eps, act, mask, residual,
                           stable, out, (uint16_t*)nullptr, (uint16_t*)nullptr, 0, (unsigned*)nullptr);
    else   // a workgroup per row
        hipLaunchKernelGGL((layernorm_kernel<4, true>), dim3(rows), block, 0, s, in, rows, C, gamma, beta, eps, act, mask,
                           residual, stable, out, (uint16_t*)nullptr, (uint16_t*)nullptr, 0, (unsigned*)nullptr);
    return avi_launch_status();
}

extern "C" int avi_splitk_epilogue(const float* parts, int nparts, long long part_stride, int rows, int C,
                                   const float* bias, const float* gamma, const float* beta, float eps, int do_ln,
                                   int act, const float* residual, float* out, void* stream) {
    if (!parts || !out || nparts < 1 || rows <= 0 || C <= 0 || (C & 3) || C > 4096 || (part_stride & 3))
        return AVI_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (C <= 1024)
        hipLaunchKernelGGL(splitk_epilogue_kernel<1>, dim3(rows), dim3(256), 0, s, parts, nparts, part_stride, C, bias,
                           gamma, beta, eps, do_ln, act, residual, out);
    else
        hipLaunchKernelGGL(splitk_epilogue_kernel<4>, dim3(rows), dim3(256), 0, s, parts, nparts, part_stride, C, bias,
                           gamma, beta, eps, do_ln, act, residual, out);
    return avi_launch_status();
}

extern "C" int avi_layernorm_planes(const float* in, int rows, int C, const float* gamma, const float* beta, float eps,
                                    float* out, uint16_t* out_hi, uint16_t* out_lo, int plane_fmt, void* stream) {
    if (!in || !out_hi || !out_lo || rows <= 0 || C <= 0 || (C & 3) || C > 1024) return AVI_EINVAL;
    if (plane_fmt != AVI_PLANES_BF16 && plane_fmt != AVI_PLANES_F16) return AVI_EINVAL;
    hipLaunchKernelGGL(layernorm_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                       rows, C, gamma, beta, eps, AVI_ACT_NONE, (const float*)nullptr, (const float*)nullptr, 0, out,
                       out_hi, out_lo, plane_fmt, avi_status_ptr());
    return avi_launch_status();
}

extern "C" int avi_layernorm_act(const float* in, int rows, int C, const float* gamma, const float* beta, float eps,
                                 int act, const float* residual, float* out, void* stream) {
    return avi_layernorm_ex(in, rows, C, gamma, beta, eps, act, nullptr, residual, 0, out, stream);
}

extern "C" int avi_layernorm(const float* in, int rows, int C, const float* gamma, const float* beta, float eps,
                             float* out, void* stream) {
    return avi_layernorm_act(in, rows, C, gamma, beta, eps, AVI_ACT_NONE, nullptr, out, stream);
}

extern "C" int avi_group_pad_pack(const float* h, int B, int T, int G, int Cg, int pad, float* xg, void* stream) {
    if (!h || !xg || B <= 0 || T <= 0 || G <= 0 || Cg <= 0 || (Cg & 3) || pad < 0) return AVI_EINVAL;
    const long long total = (long long)B * G * (T + 2 * pad) * (Cg >> 2);
    hipLaunchKernelGGL(group_pad_pack_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h, B, T, G, Cg, pad, xg);
    return avi_launch_status();
}

extern "C" int avi_pad_repeat(const float* in, int B, int T, int C, int rep, int padL, int padR, int mode,
                              float* out, void* stream) {
    if (!in || !out || B <= 0 || T <= 0 || C <= 0 || (C & 3) || rep < 1 || padL < 0 || padR < 0) return AVI_EINVAL;
    const long long total = (long long)B * (padL + (long long)T * rep + padR) * (C >> 2);
    hipLaunchKernelGGL(pad_repeat_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                       B, T, C, rep, padL, padR, mode, out);
    return avi_launch_status();
}

extern "C" int avi_add_rowbcast(const float* in, const float* add, int B, int T, int C, float* out, void* stream) {
    if (!in || !add || !out || B <= 0 || T <= 0 || C <= 0 || (C & 3)) return AVI_EINVAL;
    const long long total = (long long)B * T * (C >> 2);
    hipLaunchKernelGGL(add_rowbcast_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       in, add, B, T, C, out);
    return avi_launch_status();
}

extern "C" int avi_embed_tokens(const long long* ids, const float* table, const float* pos, int B, int T, int C,
                                int vocab, float* out, int* bad_ids, void* stream) {
    if (!ids || !table || !pos || !out || B <= 0 || T <= 0 || C <= 0 || (C & 3) || vocab <= 0) return AVI_EINVAL;
    const long long total = (long long)B * T * (C >> 2);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       ids, table, pos, B, T, C, vocab, out, bad_ids);
    return avi_launch_status();
}

extern "C" int avi_mean_tokens(const float* in, int B, int T, int C, float* out, void* stream) {
    if (!in || !out || B <= 0 || T <= 0 || C <= 0) return AVI_EINVAL;
    const long long total = (long long)B * C;
    hipLaunchKernelGGL(mean_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, B, T, C, out);
    return avi_launch_status();
}
