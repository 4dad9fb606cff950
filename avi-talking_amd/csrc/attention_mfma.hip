// Head-dim-64, unbiased multi-head attention on the bf16 matrix cores with 3-term split operands (fp32-grade):
// the wav2vec2 encoder's attention (HF eager attention via models/lib/wav2vec.py:142-148), 12 layers x 384 (b,h).
//
// Two launches per call:
//  1. attn_prep_kernel   packed fp32 QKV [B][T][3*H*64] -> bf16 hi/lo planes
//                          Q  [B*H][Tp][64]  (pre-multiplied by the softmax scale)
//                          K  [B*H][Tp][64]
//                          V^T[B*H][64][Tp]  (transposed through LDS), Tp = T rounded up to 64, zero padded.
//  2. attn_mfma_kernel   one wave = 16 (or 2 x 16) queries, flash-style loop over 64-key tiles, no LDS and no barriers:
//        S^T = K . Q^T      (A = K rows, B = Q cols)   -> lane (q = l&15, g = l>>4) holds keys 16t + 4g + r
//        online softmax per query (16 own values + 2 xor-shuffles across the 4 lane groups)
//        O^T = V^T . P^T    (A = V^T rows d, B = P from the accumulator registers)
//     The accumulator layout of S^T is already the B-operand layout of the second product once the 32 keys of a
//     k-step are relabelled  key(s,g,j) = 32 s + 16 (j>>2) + 4 g + (j&3);  V^T is read with the same relabelling
//     (two 8-byte loads per fragment), so P never leaves registers and there is no transpose in the loop.
//     Every product is hi*hi + hi*lo + lo*hi (P is split on the fly), fp32 accumulate.
//     Output rows are written as 16-byte stores: lane holds O[q][16 dt + 4g .. +3].
#include "common.h"

namespace {

constexpr int HD = 64;

__device__ __forceinline__ uint16_t bf16_bits(float x) {
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float bf16_val(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

// grid (Tp/64, B*H); block 256.  Each block converts a 64-row time slab of one head.
__global__ __launch_bounds__(256) void attn_prep_kernel(const float* __restrict__ qkv, int T, int Tp, int H, int ld,
                                                         float scale, uint16_t* __restrict__ planes) {
    __shared__ float vt[HD][65];
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H, t0 = blockIdx.x * 64, tid = threadIdx.x;
    const long long plane = (long long)gridDim.y * Tp * HD;
    uint16_t* Qhi = planes;
    uint16_t* Qlo = planes + plane;
    uint16_t* Khi = planes + 2 * plane;
    uint16_t* Klo = planes + 3 * plane;
    uint16_t* Vhi = planes + 4 * plane;
    uint16_t* Vlo = planes + 5 * plane;
    // 64 rows x 64 dims: thread -> (row r = tid/4 + 64*0, 16 dims)
    const int r = tid >> 2, c0 = (tid & 3) * 16;
    const int t = t0 + r;
    const float* src = qkv + ((long long)b * T + t) * ld + h * HD + c0;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), k = q, v = q;
        if (t < T) {
            q = *reinterpret_cast<const float4*>(src + i);
            k = *reinterpret_cast<const float4*>(src + H * HD + i);
            v = *reinterpret_cast<const float4*>(src + 2 * H * HD + i);
        }
        const float qq[4] = {q.x * scale, q.y * scale, q.z * scale, q.w * scale}, kk[4] = {k.x, k.y, k.z, k.w};
        uint16_t qh[4], ql[4], kh[4], kl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qh[j] = bf16_bits(qq[j]);
            ql[j] = bf16_bits(qq[j] - bf16_val(qh[j]));
            kh[j] = bf16_bits(kk[j]);
            kl[j] = bf16_bits(kk[j] - bf16_val(kh[j]));
        }
        const long long o = ((long long)bh * Tp + t) * HD + c0 + i;
        *reinterpret_cast<uint2*>(Qhi + o) = make_uint2(qh[0] | ((uint32_t)qh[1] << 16), qh[2] | ((uint32_t)qh[3] << 16));
        *reinterpret_cast<uint2*>(Qlo + o) = make_uint2(ql[0] | ((uint32_t)ql[1] << 16), ql[2] | ((uint32_t)ql[3] << 16));
        *reinterpret_cast<uint2*>(Khi + o) = make_uint2(kh[0] | ((uint32_t)kh[1] << 16), kh[2] | ((uint32_t)kh[3] << 16));
        *reinterpret_cast<uint2*>(Klo + o) = make_uint2(kl[0] | ((uint32_t)kl[1] << 16), kl[2] | ((uint32_t)kl[3] << 16));
        vt[c0 + i + 0][r] = v.x;
        vt[c0 + i + 1][r] = v.y;
        vt[c0 + i + 2][r] = v.z;
        vt[c0 + i + 3][r] = v.w;
    }
    __syncthreads();
    // V^T: thread -> (dim d = tid/4, 16 consecutive times)
    const int d = tid >> 2, tt0 = (tid & 3) * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        uint16_t vh[4], vl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = vt[d][tt0 + i + j];
            vh[j] = bf16_bits(x);
            vl[j] = bf16_bits(x - bf16_val(vh[j]));
        }
        const long long o = ((long long)bh * HD + d) * Tp + t0 + tt0 + i;
        *reinterpret_cast<uint2*>(Vhi + o) = make_uint2(vh[0] | ((uint32_t)vh[1] << 16), vh[2] | ((uint32_t)vh[3] << 16));
        *reinterpret_cast<uint2*>(Vlo + o) = make_uint2(vl[0] | ((uint32_t)vl[1] << 16), vl[2] | ((uint32_t)vl[3] << 16));
    }
}

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ bf16x8 ld_frag16(const uint16_t* p) {   // 8 consecutive bf16
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ bf16x8 ld_frag8x2(const uint16_t* p0, const uint16_t* p1) {   // 4 + 4 bf16
    const u32x2 a = *reinterpret_cast<const u32x2*>(p0), b = *reinterpret_cast<const u32x2*>(p1);
    const u32x4 v = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// grid (ceil(T/(64 QT)), B*H); block 256 = 4 independent waves of QT x 16 queries.  Every K / V^T fragment a wave
// loads feeds QT query tiles, which halves (QT = 2) the L1 traffic per MFMA of this LDS-free design.
template <int QT>
__global__ __launch_bounds__(256) void attn_mfma_kernel(const uint16_t* __restrict__ planes, int T, int Tp, int H,
                                                         int ldo, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int q0 = (blockIdx.x * 4 + wave) * (16 * QT);
    if (q0 >= T) return;   // wave-uniform; no barriers in this kernel
    const int fr = lane & 15, g = lane >> 4;
    const long long plane = (long long)gridDim.y * Tp * HD;
    const uint16_t* Qhi = planes + (long long)bh * Tp * HD;
    const uint16_t* Qlo = Qhi + plane;
    const uint16_t* Khi = Qhi + 2 * plane;
    const uint16_t* Klo = Qhi + 3 * plane;
    const uint16_t* Vhi = Qhi + 4 * plane;   // [64][Tp]
    const uint16_t* Vlo = Qhi + 5 * plane;

    // Q fragments (B operand of S^T): lane (q = fr, g) holds Q[q0 + 16 qt + fr][32 ks + 8 g .. +7]
    // (rows up to Tp exist, zero padded; q0 + 16 QT <= Tp because Tp is a multiple of 64)
    bf16x8 qh[QT][2], ql[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const long long o = (long long)(q0 + qt * 16 + fr) * HD + ks * 32 + g * 8;
            qh[qt][ks] = ld_frag16(Qhi + o);
            ql[qt][ks] = ld_frag16(Qlo + o);
        }
    f32x4 acc_o[QT][4];
    float m[QT], l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc_o[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m[qt] = -1.0e30f;
        l[qt] = 0.f;
    }

    for (int j0 = 0; j0 < T; j0 += 64) {
        // ---- S^T tile: 4 key tiles x (16 keys x 16 queries) per query tile
        f32x4 s[QT][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[qt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const long long o = (long long)(j0 + t * 16 + fr) * HD + ks * 32 + g * 8;
                const bf16x8 kh = ld_frag16(Khi + o), kl = ld_frag16(Klo + o);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[qt][ks], s[qt][t], 0, 0, 0);
                    s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[qt][ks], s[qt][t], 0, 0, 0);
                    s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[qt][ks], s[qt][t], 0, 0, 0);
                }
            }
        }
        // ---- online softmax for query fr of each tile (keys 16 t + 4 g + r in this lane)
        bf16x8 ph[QT][2], pl[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = -1.0e30f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (j0 + t * 16 + g * 4 + r >= T) s[qt][t][r] = -1.0e30f;
                    mx = fmaxf(mx, s[qt][t][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[qt], mx);
            const float corr = __expf(m[qt] - mn);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = s[qt][t][r] > -1.0e29f ? __expf(s[qt][t][r] - mn) : 0.f;
                    s[qt][t][r] = p;
                    sum += p;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            l[qt] = l[qt] * corr + sum;
            m[qt] = mn;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc_o[qt][dt] *= corr;
            // P fragments (B operand): k-step ks covers key tiles 2ks, 2ks+1
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float p = s[qt][2 * ks + (j >> 2)][j & 3];
                    const __bf16 hi = (__bf16)p;
                    ph[qt][ks][j] = hi;
                    pl[qt][ks][j] = (__bf16)(p - (float)hi);
                }
        }
        // ---- O^T += V^T . P^T : A fragment of d-tile dt, k-step ks = V^T[16 dt + fr][j0 + 32 ks + {4g.., 16+4g..}]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const long long o = (long long)(dt * 16 + fr) * Tp + j0 + ks * 32 + g * 4;
                const bf16x8 vh = ld_frag8x2(Vhi + o, Vhi + o + 16), vl = ld_frag8x2(Vlo + o, Vlo + o + 16);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph[qt][ks], acc_o[qt][dt], 0, 0, 0);
                    acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl[qt][ks], acc_o[qt][dt], 0, 0, 0);
                    acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph[qt][ks], acc_o[qt][dt], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
        if (q0 + qt * 16 + fr < T) {
            const float inv = 1.f / l[qt];
            float* op = out + ((long long)b * T + q0 + qt * 16 + fr) * ldo + h * HD + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<float4*>(op + dt * 16) = make_float4(
                    acc_o[qt][dt][0] * inv, acc_o[qt][dt][1] * inv, acc_o[qt][dt][2] * inv, acc_o[qt][dt][3] * inv);
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused variant (the default): ONE launch, no scratch.  One workgroup of 8 waves per (batch, head, 256-query block):
//   * K and V of the head are converted fp32 -> bf16 hi/lo ONCE per workgroup while they are staged into LDS
//     (4 images of [256 keys][64 dims] bf16 = 128 KiB), so the loop never waits on global memory; the two-launch
//     version above spends 75 us per layer in attn_prep_kernel and is latency-bound on its fragment loads from L2.
//   * both images are ROW-major with one swizzle that serves K's row reads and V's transposed reads: the 32-B slot s
//     of row r sits at slot s ^ ((r >> 1) & 3)  ->  ds_read_b128 (K fragment: row = key, 8 dims) and
//     ds_read_b64_tr_b16 (V^T fragment: 4 keys x 16 dims block delivered dim-major) are both bank-conflict-free.
//   * per wave: 2 x 16 queries (Q scaled and split in registers), same S^T = K.Q^T / O^T = V^T.P^T scheme and the same
//     key relabelling as attn_mfma_kernel: the transposed read of the keys {4g..4g+3} and {16+4g..} of a 32-key
//     step lands directly in the A-operand layout.
constexpr int FCH = 256;                       // keys per LDS chunk = queries per workgroup
constexpr int IMG = FCH * 128;                 // one image: 256 rows x 128 B
constexpr int FUSED_SMEM = 4 * IMG;            // Khi Klo Vhi Vlo

__device__ __forceinline__ int img_off(int row, int chunk) {   // byte offset of 16-B chunk `chunk` (0..7) of `row`
    return row * 128 + (((((chunk >> 1) ^ ((row >> 1) & 3)) << 1) | (chunk & 1)) << 4);
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 ld_tr2(const char* p0, const char* p1) {   // two 4-key transposed reads -> 8 keys
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

// Head dims 16, 32 and 64; additive score biases of the reference restated analytically (attention.hip, modes 1/2).
//   D = 16 uses v_mfma_f32_16x16x16_bf16 for S^T (the whole head dim is one k-step of 16); images keep 128-B rows
//   for every D (the first 2 D bytes hold data), so one swizzle/offset function serves all three.
template <int D>
__global__ __launch_bounds__(512) void attn_fused_kernel(const float* __restrict__ qp, const float* __restrict__ kp,
                                                         const float* __restrict__ vp, int Tq, int Tk, int H, int ldq,
                                                         int ldk, float scale_in, int bias_mode,
                                                         const float* __restrict__ slopes, int period,
                                                         float* __restrict__ out, int ldo,
                                                         uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo,
                                                         int out_fmt, unsigned* __restrict__ status) {
    constexpr int KS = D >= 32 ? D / 32 : 1;      // k-steps of the score product
    constexpr int DT = D / 16;                    // 16-dim tiles of the output
    constexpr int CH = D / 8;                     // 16-B chunks (8 dims) per image row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Khi = smem;
    char* Klo = smem + IMG;
    char* Vhi = smem + 2 * IMG;
    char* Vlo = smem + 3 * IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int fr = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * FCH + wave * 32;
    const bool active = q0 < Tq;                         // wave-uniform
    const float* qbase = qp + (long long)b * Tq * ldq + h * D;
    const float* kbase = kp + (long long)b * Tk * ldk + h * D;
    const float* vbase = vp + (long long)b * Tk * ldk + h * D;
    // scores are kept in the log2 domain (log2 e folded into the Q scale and the bias slope): softmax needs one v_exp_f32
    // per score and no multiply
    constexpr float LOG2E = 1.4426950408889634f;
    const float scale = scale_in * LOG2E;
    const float slope = (bias_mode != 0 && slopes) ? slopes[h] * LOG2E : 0.f;

    // ---- Q fragments (B operand of S^T): lane (q = fr, g), scaled, split
    //      D >= 32: Q[q][32 ks + 8 g .. +7];   D = 16: Q[q][4 g .. +3] in elements 0..3
    bf16x8 qh[2][KS], ql[2][KS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int q = q0 + qt * 16 + fr;
        q = q < Tq ? q : Tq - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (D >= 32) {
                const float* p = qbase + (long long)q * ldq + ks * 32 + g * 8;
                const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
                x[0] = a.x * scale; x[1] = a.y * scale; x[2] = a.z * scale; x[3] = a.w * scale;
                x[4] = c.x * scale; x[5] = c.y * scale; x[6] = c.z * scale; x[7] = c.w * scale;
            } else {
                const float4 a = *reinterpret_cast<const float4*>(qbase + (long long)q * ldq + g * 4);
                x[0] = a.x * scale; x[1] = a.y * scale; x[2] = a.z * scale; x[3] = a.w * scale;
            }
            split8(x, qh[qt][ks], ql[qt][ks]);
        }
    }
    f32x4 acc_o[2][DT];
    float m[2], l[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc_o[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m[qt] = -1.0e30f;
        l[qt] = 0.f;
    }
    // per-lane fragment offsets inside an image (row bases that are multiples of 16 / 32 add as immediates)
    int kofs[KS], vofs[DT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        kofs[ks] = D >= 32 ? img_off(fr, 4 * ks + g) : img_off(fr, g >> 1) + 8 * (g & 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)      // block row 4g + fr/4, dims 16 dt + 4 (fr%4) .. +3 (8 B)
        vofs[dt] = img_off(4 * g + (fr >> 2), 2 * dt + ((fr & 3) >> 1)) + 8 * (fr & 1);

    // causal mode never needs keys past the workgroup's last query row
    int kend = Tk;
    if (bias_mode == 2) {
        const int lastq = blockIdx.x * FCH + FCH - 1 < Tq ? blockIdx.x * FCH + FCH - 1 : Tq - 1;
        kend = lastq + 1 < Tk ? lastq + 1 : Tk;
    }
    for (int k0 = 0; k0 < kend; k0 += FCH) {
        const int kn = (kend - k0) < FCH ? (kend - k0) : FCH;     // keys in this chunk
        const int rows = (kn + 63) & ~63;                         // staged rows (whole 64-key tiles, zero padded)
        if (k0) __syncthreads();                                  // previous chunk fully consumed
        // ---- stage K and V: item = (row, 16-B chunk of 8 dims); CH lanes cover one fp32 row of the head
        for (int it = tid; it < rows * CH; it += 512) {
            const int row = it / CH, c = it - row * CH;
            float kx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, vx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (row < kn) {
                const long long o = (long long)(k0 + row) * ldk + c * 8;
                const float4 k0v = *reinterpret_cast<const float4*>(kbase + o);
                const float4 k1v = *reinterpret_cast<const float4*>(kbase + o + 4);
                const float4 v0v = *reinterpret_cast<const float4*>(vbase + o);
                const float4 v1v = *reinterpret_cast<const float4*>(vbase + o + 4);
                kx[0] = k0v.x; kx[1] = k0v.y; kx[2] = k0v.z; kx[3] = k0v.w;
                kx[4] = k1v.x; kx[5] = k1v.y; kx[6] = k1v.z; kx[7] = k1v.w;
                vx[0] = v0v.x; vx[1] = v0v.y; vx[2] = v0v.z; vx[3] = v0v.w;
                vx[4] = v1v.x; vx[5] = v1v.y; vx[6] = v1v.z; vx[7] = v1v.w;
            }
            bf16x8 hi, lo;
            const int o = img_off(row, c);
            split8(kx, hi, lo);
            *reinterpret_cast<bf16x8*>(Khi + o) = hi;
            *reinterpret_cast<bf16x8*>(Klo + o) = lo;
            split8(vx, hi, lo);
            *reinterpret_cast<bf16x8*>(Vhi + o) = hi;
            *reinterpret_cast<bf16x8*>(Vlo + o) = lo;
        }
        __syncthreads();
        if (!active) continue;                                    // wave-uniform: EXEC stays all ones below
        for (int jl = 0; jl < rows; jl += 64) {
            const int j0 = k0 + jl;
            // ---- S^T tile: 4 key tiles x (16 keys x 16 queries) per query tile
            f32x4 s[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) s[qt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int o = (jl + t * 16) * 128 + kofs[ks];
                    if (D >= 32) {
                        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Khi + o);
                        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Klo + o);
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt) {
                            s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[qt][ks], s[qt][t], 0, 0, 0);
                            s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[qt][ks], s[qt][t], 0, 0, 0);
                            s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[qt][ks], s[qt][t], 0, 0, 0);
                        }
                    } else {   // lane (key = fr, g) holds K[key][4 g .. +3]
                        const s16x4 kh = *reinterpret_cast<const s16x4*>(Khi + o);
                        const s16x4 kl = *reinterpret_cast<const s16x4*>(Klo + o);
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt) {
                            const s16x4 bh_ = __builtin_shufflevector(__builtin_bit_cast(s16x8, qh[qt][0]),
                                                                      __builtin_bit_cast(s16x8, qh[qt][0]), 0, 1, 2, 3);
                            const s16x4 bl_ = __builtin_shufflevector(__builtin_bit_cast(s16x8, ql[qt][0]),
                                                                      __builtin_bit_cast(s16x8, ql[qt][0]), 0, 1, 2, 3);
                            s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kl, bh_, s[qt][t], 0, 0, 0);
                            s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kh, bl_, s[qt][t], 0, 0, 0);
                            s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kh, bh_, s[qt][t], 0, 0, 0);
                        }
                    }
                }
            }
            // ---- online softmax for query fr of each tile (keys 16 t + 4 g + r in this lane)
            bf16x8 ph[2][2], pl[2][2];
            // masks and biases only where the tile needs them (wave-uniform): a key tile that ends past Tk, any bias
            // mode, or - causal - a tile that reaches the wave's first query row
            const bool plain = bias_mode == 0 && j0 + 64 <= Tk;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qi = q0 + qt * 16 + fr;
                float mx = -1.0e30f;
                if (plain) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qt][t][r]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = j0 + t * 16 + g * 4 + r;
                            float sv = s[qt][t][r];
                            bool valid = j < Tk;
                            if (bias_mode == 1) {          // inferno TransformerMasking.py:80-98
                                const int dlt = qi > j ? qi - j : j - qi;
                                sv -= slope * (float)dlt;
                            } else if (bias_mode == 2) {   // models/faceformer.py:51-72
                                valid = valid && (j <= qi);
                                sv -= slope * (float)((qi - j) / period);
                            }
                            sv = valid ? sv : -1.0e30f;     // exp2 of it underflows to exactly 0 below: every query
                            s[qt][t][r] = sv;               // has seen a valid key (key 0) by the end of the first tile
                            mx = fmaxf(mx, sv);
                        }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mn = fmaxf(m[qt], mx);
                const float corr = __builtin_amdgcn_exp2f(m[qt] - mn);
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(s[qt][t][r] - mn);
                        s[qt][t][r] = pv;
                        sum += pv;
                    }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                l[qt] = l[qt] * corr + sum;
                m[qt] = mn;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc_o[qt][dt] *= corr;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pv = s[qt][2 * ks + (j >> 2)][j & 3];
                        const __bf16 hi = (__bf16)pv;
                        ph[qt][ks][j] = hi;
                        pl[qt][ks][j] = (__bf16)(pv - (float)hi);
                    }
            }
            // ---- O^T += V^T . P^T : transposed reads of keys {32 ks + 4 g ..+3} and {32 ks + 16 + 4 g ..+3}
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int o = (jl + ks * 32) * 128 + vofs[dt];
                    const bf16x8 vh = ld_tr2(Vhi + o, Vhi + o + 16 * 128);
                    const bf16x8 vl = ld_tr2(Vlo + o, Vlo + o + 16 * 128);
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph[qt][ks], acc_o[qt][dt], 0, 0, 0);
                        acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl[qt][ks], acc_o[qt][dt], 0, 0, 0);
                        acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph[qt][ks], acc_o[qt][dt], 0, 0, 0);
                    }
                }
        }
    }
    if (!active) return;
    AviF16Range rng;                              // fp16 planes only: range guard (common.h)
    const bool guard = out_hi && out_fmt == AVI_PLANES_F16;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
        if (q0 + qt * 16 + fr < Tq) {
            const float inv = 1.f / l[qt];
            const long long o = ((long long)b * Tq + q0 + qt * 16 + fr) * ldo + h * D + g * 4;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const float v[4] = {acc_o[qt][dt][0] * inv, acc_o[qt][dt][1] * inv, acc_o[qt][dt][2] * inv,
                                    acc_o[qt][dt][3] * inv};
                if (out) *reinterpret_cast<float4*>(out + o + dt * 16) = make_float4(v[0], v[1], v[2], v[3]);
                if (out_hi) {   // split planes for the consumer GEMM (x = hi + lo)
                    uint32_t hh[2], ll[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        uint16_t h0, h1, l0, l1;
                        if (guard) {
                            rng.see(v[2 * j]);
                            rng.see(v[2 * j + 1]);
                        }
                        avi_split_hl(v[2 * j], out_fmt, h0, l0);
                        avi_split_hl(v[2 * j + 1], out_fmt, h1, l1);
                        hh[j] = h0 | ((uint32_t)h1 << 16);
                        ll[j] = l0 | ((uint32_t)l1 << 16);
                    }
                    *reinterpret_cast<uint2*>(out_hi + o + dt * 16) = make_uint2(hh[0], hh[1]);
                    *reinterpret_cast<uint2*>(out_lo + o + dt * 16) = make_uint2(ll[0], ll[1]);
                }
            }
        }
    if (guard) rng.commit(status);
}

template <int D>
int launch_fused(const float* q, const float* k, const float* v, float* out, int B, int H, int Tq, int Tk, int ldq,
                 int ldk, int ldo, float scale, int bias_mode, const float* slopes, int period, hipStream_t s,
                 uint16_t* out_hi = nullptr, uint16_t* out_lo = nullptr, int out_fmt = AVI_PLANES_BF16) {
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(attn_fused_kernel<D>), FUSED_SMEM);
    hipLaunchKernelGGL(attn_fused_kernel<D>, dim3((Tq + FCH - 1) / FCH, B * H), dim3(512), FUSED_SMEM, s, q, k, v, Tq, Tk,
                       H, ldq, ldk, scale, bias_mode, slopes, period, out, ldo, out_hi, out_lo, out_fmt, avi_status_ptr());
    return avi_launch_status();
}

}  // namespace

// scratch: 6 * B*H*Tp*64 bf16 (= 12 * B*H*Tp*64 bytes), Tp = T rounded up to 64.
extern "C" int avi_attention_d64(const float* qkv, int B, int H, int T, int ld, float scale, float* out, int ldo,
                                 uint16_t* scratch, void* stream) {
    if (!qkv || !out || !scratch || B <= 0 || H <= 0 || T <= 0 || (ld & 3) || (ldo & 3)) return AVI_EINVAL;
    if (ld < 3 * H * HD || ldo < H * HD || (long long)B * H > 65535) return AVI_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
        (reinterpret_cast<uintptr_t>(scratch) & 15))
        return AVI_EINVAL;
    const int Tp = (T + 63) / 64 * 64;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const bool two_launch = [] { const char* e = getenv("AVI_ATTN_TWO_LAUNCH"); return e && atoi(e) != 0; }();
    if (!two_launch)     // fused: K/V converted while staged into LDS, no scratch traffic
        return launch_fused<64>(qkv, qkv + H * HD, qkv + 2 * H * HD, out, B, H, T, T, ld, ld, ldo, scale, 0, nullptr, 1, s);
    hipLaunchKernelGGL(attn_prep_kernel, dim3(Tp / 64, B * H), dim3(256), 0, s, qkv, T, Tp, H, ld, scale, scratch);
    // two query tiles per wave once there are enough waves to fill the 1024 SIMDs, else one
    if ((long long)B * H * ((T + 31) / 32) >= 2048)
        hipLaunchKernelGGL(attn_mfma_kernel<2>, dim3((T + 127) / 128, B * H), dim3(256), 0, s, scratch, T, Tp, H, ldo, out);
    else
        hipLaunchKernelGGL(attn_mfma_kernel<1>, dim3(Tp / 64, B * H), dim3(256), 0, s, scratch, T, Tp, H, ldo, out);
    return avi_launch_status();
}

// Matrix-core path of avi_attention (attention.hip) for head dims 16 / 32 / 64; strides and pointers 16-B aligned.
int avi_attention_fused_launch(const float* q, const float* k, const float* v, float* out, int B, int H, int Tq, int Tk,
                               int D, int ldq, int ldk, int ldo, float scale, int bias_mode, const float* slopes,
                               int period, hipStream_t s) {
    switch (D) {
        case 16: return launch_fused<16>(q, k, v, out, B, H, Tq, Tk, ldq, ldk, ldo, scale, bias_mode, slopes, period, s);
        case 32: return launch_fused<32>(q, k, v, out, B, H, Tq, Tk, ldq, ldk, ldo, scale, bias_mode, slopes, period, s);
        case 64: return launch_fused<64>(q, k, v, out, B, H, Tq, Tk, ldq, ldk, ldo, scale, bias_mode, slopes, period, s);
        default: return AVI_EINVAL;
    }
}

// Head-dim-64 attention over a packed QKV projection whose result is written as split bf16 planes (and optionally as
// fp32 too): the operand format of the ping-pong GEMM that consumes it (the encoder's out_proj).
extern "C" int avi_attention_d64_planes(const float* qkv, int B, int H, int T, int ld, float scale, float* out,
                                        uint16_t* out_hi, uint16_t* out_lo, int ldo, int plane_fmt, void* stream) {
    if (!qkv || !out_hi || !out_lo || B <= 0 || H <= 0 || T <= 0 || (ld & 3) || (ldo & 3)) return AVI_EINVAL;
    if (plane_fmt != AVI_PLANES_BF16 && plane_fmt != AVI_PLANES_F16) return AVI_EINVAL;
    if (ld < 3 * H * HD || ldo < H * HD || (long long)B * H > 65535) return AVI_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
        ((reinterpret_cast<uintptr_t>(out_hi) | reinterpret_cast<uintptr_t>(out_lo)) & 7))
        return AVI_EINVAL;
    return launch_fused<64>(qkv, qkv + H * HD, qkv + 2 * H * HD, out, B, H, T, T, ld, ld, ldo, scale, 0, nullptr, 1,
                            static_cast<hipStream_t>(stream), out_hi, out_lo, plane_fmt);
}

// The same with the bias modes of avi_attention (2 with zero slopes = the plain causal mask of the CLIP text model).
extern "C" int avi_attention_d64_planes_biased(const float* qkv, int B, int H, int T, int ld, float scale,
                                               int bias_mode, const float* slopes, int period, float* out,
                                               uint16_t* out_hi, uint16_t* out_lo, int ldo, int plane_fmt,
                                               void* stream) {
    if (!qkv || !out_hi || !out_lo || B <= 0 || H <= 0 || T <= 0 || (ld & 3) || (ldo & 3)) return AVI_EINVAL;
    if (plane_fmt != AVI_PLANES_BF16 && plane_fmt != AVI_PLANES_F16) return AVI_EINVAL;
    if (ld < 3 * H * HD || ldo < H * HD || (long long)B * H > 65535) return AVI_EINVAL;
    if (bias_mode < 0 || bias_mode > 2 || (bias_mode != 0 && !slopes) || period < 1) return AVI_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
        ((reinterpret_cast<uintptr_t>(out_hi) | reinterpret_cast<uintptr_t>(out_lo)) & 7))
        return AVI_EINVAL;
    return launch_fused<64>(qkv, qkv + H * HD, qkv + 2 * H * HD, out, B, H, T, T, ld, ld, ldo, scale, bias_mode, slopes,
                            period, static_cast<hipStream_t>(stream), out_hi, out_lo, plane_fmt);
}
