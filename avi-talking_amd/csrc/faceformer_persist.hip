// FaceFormer autoregressive decode for WIDE decoders, ONE utterance, as ONE persistent launch
// (avi_faceformer_decode_persistent; models/faceformer.py:710-729 = `predict`'s loop, KV-cached as in faceformer_steps.hip).
//
// faceformer_steps.hip cuts a frame into 6 launches.  At B = 1 each of them lasts 5-10 us although it moves a few hundred
// KB: a launch starts with cold L2s (the XCDs' L2s are written back / invalidated at kernel boundaries), so every launch
// re-fetches its weight slice through the fabric, then waits for its inputs, then drains - 40 us per frame for 22 MB of
// weights that never change.  Here the weights never move: 256 workgroups, one per CU (the launch needs all 256 CUs of an
// MI355X free: 157 KB of LDS each at D = 1024), and workgroup g keeps ITS rows of every matrix in LDS for the whole decode, in fp32:
//     q/k/v from LN3's output  3D/256 rows x D  (in_proj . vertice_map . vertice_map_r folded into one matrix: the next
//                              frame's q, k, v straight from this frame's last LayerNorm, not through the 53 coefficients)
//     out_proj | vertice_map   D/256 rows x (D + 64)        linear1  2D/256 rows x D        linear2  D/256 rows x 2D
//     vertice_map_r            1 row (workgroups 64..127: the coefficients themselves, off the critical path)
// (137 KB at D = 1024).  What moves between CUs is the frame's activation vectors, as data-tagged 8-byte granules
// {value, tag} in device memory (one relaxed agent-scope store each, polled with relaxed agent-scope loads; the mechanism of
// prior_pair.hip): tag = launch epoch << 16 | (6 frame + edge + 1), so a granule says by itself whether it is the one
// awaited - no flags, no fences, no barrier.  Five edges on a frame's critical path, a sixth beside it:
//     S3 (frame i-1) -> A  every workgroup: y = LN3(s3) (whole row, redundantly), its 3D/256 columns of q, k, v          -> QKV
//                          workgroups 64..127 then: one coefficient each of frame i-1 (vertice_map_r y), the frame's output -> O
//     QKV           ->  B  16 workgroups (head h, key residue s = j mod 4): softmax partial (m, l, acc) over ITS keys of the
//                          cache (it alone ever reads them: plain loads, its own L2), key i appended by residue i mod 4   -> PART
//     PART, O       ->  C  every workgroup: merge of the 16 partials, its rows of s1 = x + out_proj(att), x from O      -> S1
//     S1            ->  D  every workgroup: x2 = LN2(LN1(s1) + cross_i) (whole rows, redundantly), its rows of relu(linear1) -> H
//     H             ->  E  every workgroup: its rows of s3 = x2 + linear2(h)                                -> S3
// Every spin is bounded: a workgroup that never sees a granule (the launch did not get all its CUs, a foreign kernel holds
// one) gives up ONCE, takes NaN from then on - so does everything downstream, with correct tags, nobody else stalls - and
// raises AVI_STATUS_EXCHANGE_TIMEOUT; the launch always drains.  Arithmetic: fp32 multiply-adds on fp32 weights (the launch
// chain's 3-term bf16 products agree to ~1e-6); LayerNorm statistics from per-thread (n, mean, M2) triples merged in one
// workgroup reduction (row_stats).  Rows of a matrix belong to WAVES (gemv_rows): the matrix-vector products need neither LDS
// traffic for partial sums nor barriers.
#include "common.h"

namespace {

constexpr int NT = 256, NH = 4, VP = 64, PB = 1, NWG = 256, MAPR0 = 64, KSP = 4, NATT = NH * KSP, NEDGE = 6, KEYMAX = 256, PF = 16;
constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr int XCH_HDR = 8;                // header words (u64): [0] launch epoch
enum { E_QKV = 0, E_PART = 1, E_S1 = 2, E_H = 3, E_S3 = 4, E_O = 5 };

struct Geo {
    int D, dh, nq, no, n1, n2;          // rows of each matrix a workgroup owns
    int sq, so, s1, s2;                 // padded row strides (floats): + 32 so that two rows met by one wave sit 32 banks apart
    int oq, oo, o1, o2, orr, ocq, img;  // offsets of the slices in the image, and its size (floats)
    long long xq, xp, xs1, xh, xs3, xo, xpar;   // exchange offsets (granules) inside one parity, size of a parity
    int ps;                             // granules of one partial: [0] m, [1] l, [4 + d] acc
};
__host__ __device__ inline Geo geo(int D) {
    Geo g;
    g.D = D, g.dh = D / NH;
    g.nq = 3 * D / NWG, g.no = D / NWG, g.n1 = 2 * D / NWG, g.n2 = D / NWG;
    g.sq = D + 32, g.so = D + VP + 32, g.s1 = D + 32, g.s2 = 2 * D + 32;
    g.oq = 0;
    g.oo = g.oq + g.nq * g.sq;
    g.o1 = g.oo + g.no * g.so;
    g.o2 = g.o1 + g.n1 * g.s1;
    g.orr = g.o2 + g.n2 * g.s2;
    g.ocq = g.orr + D;                  // [16]: constant part of my q/k/v columns
    g.img = g.ocq + 16;
    g.ps = g.dh + 4;
    g.xq = 0;
    g.xp = g.xq + (long long)PB * 3 * D;
    g.xs1 = g.xp + (long long)PB * NATT * g.ps;
    g.xh = g.xs1 + (long long)PB * D;
    g.xs3 = g.xh + (long long)PB * 2 * D;
    g.xo = g.xs3 + (long long)PB * D;
    g.xpar = g.xo + (long long)PB * VP;
    return g;
}
inline int lds_floats(const Geo& g) {
    return g.img + PB * 2 * g.D + PB * VP + 1024 + 3 * g.dh + KEYMAX + NT * 4;
}

struct Persist {
    AviFaceformerWeights w;
    AviFaceformerPlanes p;
    const float* image;
    const float* cross;
    float* kv;
    float* out;
    uint16_t* out16;
    unsigned long long* xch;
    unsigned* status;
    int B, T, D, chunk;
    int fault;            // avi_debug_fault_inject: the last workgroup leaves at once
};

struct Ex {
    unsigned* status;
    int* wg_dead;       // LDS: some thread of this workgroup has given up
    bool dead;          // this thread (or, from the next stage on, its workgroup) has given up once: NaN from now on, no more
                        // spinning - a launch that lost a workgroup pays ONE bounded spin per workgroup, not one per stage
};

__device__ __forceinline__ void publish(unsigned long long* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long peek(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the value of granule *p once it carries `tag` (gr = what an earlier peek saw); bounded
__device__ __forceinline__ float settle(const unsigned long long* p, unsigned long long gr, unsigned tag, Ex& x) {
    unsigned spins = 0;
    while (!x.dead && (unsigned)(gr >> 32) != tag) {
        if (++spins > SPIN_LIMIT) {
            if (x.status) __hip_atomic_store(x.status + AVI_STATUS_EXCHANGE_TIMEOUT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            *x.wg_dead = 1;
            x.dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);       // 64 clocks between two looks (measured per frame: none 27.2 us, 1: 27.2-27.5,
        gr = peek(p);                      // 4: 29.7, 16: 39.7; none for the first 64 looks only: 28.1)
    }
    return x.dead ? __builtin_nanf("") : __builtin_bit_cast(float, (unsigned)gr);
}
// Several granules of one thread: every one of gr[idx] (idx < N, bit idx of `want` set; gr = what a first look saw, addr(idx) its
// address) is looked at until it carries `tag` - ALL the missing ones together in every round.  One after the other (settle on
// each in turn) a thread pays a full round trip through the fabric for every granule whose FIRST look came too early, and the
// first looks are all taken right after the thread's own publish: up to N round trips per wait instead of one or two.
// (Two generations of looks in flight half a beat apart - to see a granule sooner after it becomes visible - measured
// SLOWER: 23.4 vs 21.9 us per frame at D = 1024, 22.8 vs 19.8 at D = 256: twice the polling traffic on the same fabric.)
template <int N, typename F>
__device__ __forceinline__ void settle_all(F addr, unsigned long long (&gr)[N], unsigned want, unsigned tag, Ex& x) {
    unsigned pending = 0;
#pragma unroll
    for (int idx = 0; idx < N; ++idx)
        if (((want >> idx) & 1u) && (unsigned)(gr[idx] >> 32) != tag) pending |= 1u << idx;
    unsigned spins = 0;
    while (pending && !x.dead) {
        if (++spins > SPIN_LIMIT) {
            if (x.status) __hip_atomic_store(x.status + AVI_STATUS_EXCHANGE_TIMEOUT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            *x.wg_dead = 1;
            x.dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int idx = 0; idx < N; ++idx)
            if ((pending >> idx) & 1u) gr[idx] = peek(addr(idx));
#pragma unroll
        for (int idx = 0; idx < N; ++idx)
            if (((pending >> idx) & 1u) && (unsigned)(gr[idx] >> 32) == tag) pending &= ~(1u << idx);
    }
}
__device__ __forceinline__ float granule_value(unsigned long long gr, const Ex& x) {
    return x.dead ? __builtin_nanf("") : __builtin_bit_cast(float, (unsigned)gr);
}

// ---- small reductions on DPP lanes (a ds_bpermute round trip costs ~100 cycles; these are 2-7 vector instructions) ----
#define FFP_DPP(x, ctrl, rm) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rm, 0xF, false))
// sum over each aligned group of 4 lanes, in all 4
__device__ __forceinline__ float quad_sum(float v) {
    v += FFP_DPP(v, 0xB1, 0xF);      // quad_perm [1,0,3,2]
    v += FFP_DPP(v, 0x4E, 0xF);      // quad_perm [2,3,0,1]
    return v;
}
// sums over lanes 0..31 and 32..63 (wave-uniform)
__device__ __forceinline__ void half_sums(float v, float& lo, float& hi) {
    v = quad_sum(v);
    v += FFP_DPP(v, 0x141, 0xF);     // row_half_mirror
    v += FFP_DPP(v, 0x140, 0xF);     // row_mirror: every lane of a 16-lane row holds the row's sum
    v += FFP_DPP(v, 0x142, 0xA);     // row_bcast15 into rows 1 and 3
    lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// N values per thread summed (or maximised) over the workgroup, one barrier; every thread gets the results.  `red` slots
// [slot .. slot + N) of 4 words each must not be in use by a reduction less than one barrier old.
template <int N, bool MAX>
__device__ __forceinline__ void block_reduce(float (&v)[N], float (*red)[4], int slot) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const float w = MAX ? wave_max_u(v[q]) : wave_sum_u(v[q]);
        if ((threadIdx.x & 63) == 0) red[slot + q][threadIdx.x >> 6] = w;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q)
        v[q] = MAX ? fmaxf(fmaxf(red[slot + q][0], red[slot + q][1]), fmaxf(red[slot + q][2], red[slot + q][3]))
                   : (red[slot + q][0] + red[slot + q][1]) + (red[slot + q][2] + red[slot + q][3]);
}

// LayerNorm statistics of BT rows whose values sit E per thread in registers (columns past D hold 0): ONE workgroup
// reduction instead of two (mean, then squares about it).  Every thread takes the squares about the mean of ITS values and
// the partial (n, mean, M2) triples are merged - sum_i (x_i - m)^2 = sum over threads of [M2_t + n_t (m_t - m)^2] - so the
// E[x^2] - mean^2 cancellation only touches the spread of the thread means (a quarter of the variance at 4 values per thread).
template <int BT, int E>
__device__ __forceinline__ void row_stats(const float (&t)[BT][E], int D, float (*red)[4], int slot, float (&mu)[BT], float (&rs)[BT]) {
    const int tid = threadIdx.x;
    float v[3 * BT];                    // per row: sum, sum_t M2_t, sum_t n_t m_t^2
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        float n = 0.f, sum = 0.f;
#pragma unroll
        for (int u = 0; u < E; ++u)
            if (u * NT + tid < D) n += 1.f, sum += t[b][u];
        const float m = n > 0.f ? sum / n : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int u = 0; u < E; ++u)
            if (u * NT + tid < D) m2 += (t[b][u] - m) * (t[b][u] - m);
        v[3 * b] = sum, v[3 * b + 1] = m2, v[3 * b + 2] = n * m * m;
    }
    block_reduce<3 * BT, false>(v, red, slot);
    const float invD = 1.f / D;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        mu[b] = v[3 * b] * invD;
        // sum_t n_t (m_t - mu)^2 = sum_t n_t m_t^2 - D mu^2   (the thread means m_t are O(sigma) apart, not O(mu): mild)
        const float var = (v[3 * b + 1] + fmaxf(v[3 * b + 2] - v[3 * b] * mu[b], 0.f)) * invD;
        rs[b] = rsqrtf(var + 1e-5f);
    }
}

// element u * NT + tid of row b of a published vector (n <= E * NT values per row, row stride `stride` granules) -> v[b][u]
template <int BT, int E>
__device__ __forceinline__ void poll_regs(const unsigned long long* src, long long stride, int n, unsigned tag, Ex& x,
                                          float (&v)[BT][E]) {
    unsigned long long gr[BT * E];
    unsigned want = 0;
    auto addr = [&](int idx) __attribute__((always_inline)) { return src + (idx / E) * stride + (idx % E) * NT + threadIdx.x; };
#pragma unroll
    for (int idx = 0; idx < BT * E; ++idx) {
        const bool on = (idx % E) * NT + (int)threadIdx.x < n;
        if (on) want |= 1u << idx;
        gr[idx] = (on && !x.dead) ? peek(addr(idx)) : 0ull;
    }
    settle_all<BT * E>(addr, gr, want, tag, x);
#pragma unroll
    for (int idx = 0; idx < BT * E; ++idx) v[idx / E][idx % E] = ((want >> idx) & 1u) ? granule_value(gr[idx], x) : 0.f;
}

// sums[q] = sum_k W[(wave * rpw + q) * ws + k] x[k] for q < rpw <= RPW: wave w owns rows w rpw .. w rpw + rpw - 1 (rows >= nrow
// repeat row nrow - 1), a lane takes every 64th float4 of them.  K % 4 == 0.  The sums are wave-uniform: no LDS, no barrier -
// the lanes q < rpw of each wave publish.
template <int RPW>
__device__ __forceinline__ void gemv_rows(const float* __restrict__ W, int ws, int rpw, int nrow, int K,
                                          const float* __restrict__ x, float (&sums)[RPW]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, K4 = K >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* w4[RPW];
    float a[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = wave * rpw + q;
        w4[q] = reinterpret_cast<const f32x4*>(W + (r < nrow ? r : nrow - 1) * ws);
        a[q] = 0.f;
    }
#pragma unroll 4
    for (int k = lane; k < K4; k += 64) {
        const f32x4 xv = x4[k];
#pragma unroll
        for (int q = 0; q < RPW; ++q)
            if (q < rpw) {
                const f32x4 wv = w4[q][k];
                a[q] = fmaf(wv[0], xv[0], fmaf(wv[1], xv[1], fmaf(wv[2], xv[2], fmaf(wv[3], xv[3], a[q]))));
            }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) sums[q] = q < rpw ? wave_sum_u(a[q]) : 0.f;
}

// AVI_FFP_STAMPS (diagnostic build, scripts/ffp_stamps.py): thread 0 of workgroups 0, 13, 70 and 200 adds up the time (100 MHz
// ticks) between the marks below over all frames; at the end the sums overwrite the first floats of the output.
#ifdef AVI_FFP_STAMPS
#define FFP_STAMP(k)                                         \
    do {                                                     \
        if (tid == 0) {                                      \
            const long long t_ = wall_clock64();             \
            stamp_acc[k] += t_ - stamp_last;                 \
            stamp_last = t_;                                 \
        }                                                    \
    } while (0)
#else
#define FFP_STAMP(k) do { } while (0)
#endif

template <int BT>
__global__ __launch_bounds__(NT) void ff_persist_kernel(const Persist c) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Geo G = geo(c.D);
    constexpr int E = 4;                    // values of a D-long row per thread (D <= 1024)
    const int D = c.D, dh = G.dh, tid = threadIdx.x, g = blockIdx.x;
    const int dsh = 31 - __builtin_clz(dh);                 // dh is a power of two
    float* img = lds;
    float* vec = img + G.img;               // [PB][2D]: the frame's activation rows, one stage after the other
    float* ov = vec + PB * 2 * D;           // [PB][VP]: the previous coefficient frame (normalised)
    float (*red)[4] = reinterpret_cast<float (*)[4]>(ov + PB * VP);   // [32][4] reduction slots
    float* res = ov + PB * VP + 128;        // [PB][16]
    float* ml = res + PB * 16;              // [PB][NATT][2]
    int* wg_dead = reinterpret_cast<int*>(ml + PB * NATT * 2);
    float* qs = ov + PB * VP + 1024;        // [dh]
    float* ks = qs + dh;
    float* vs = ks + dh;
    float* sc = vs + dh;                    // [KEYMAX]
    float* accr = sc + KEYMAX;              // [NT * 4]
    if (c.fault && g == NWG - 1) return;
    for (int i = tid; i < G.img / 4; i += NT)
        reinterpret_cast<f32x4*>(img)[i] = reinterpret_cast<const f32x4*>(c.image + (long long)g * G.img)[i];
    Ex x;
    x.status = c.status;
    x.wg_dead = wg_dead;
    x.dead = false;
    if (tid == 0) *wg_dead = 0;
    const unsigned epoch = (unsigned)(peek(c.xch) & 0xFFFFu);
    unsigned long long* xbase = c.xch + XCH_HDR;
    const float scale = rsqrtf((float)dh);
    // per-thread constants: LayerNorm gains / biases of my E columns, biases of my rows
    float g1[E], b1n[E], g2[E], b2n[E], g3[E], b3n[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int d = u * NT + tid, dd = d < D ? d : 0;
        g1[u] = c.w.n1g[dd], b1n[u] = c.w.n1b[dd], g2[u] = c.w.n2g[dd], b2n[u] = c.w.n2b[dd];
        g3[u] = c.w.n3g[dd], b3n[u] = c.w.n3b[dd];
    }
    static_assert(BT == 1, "the stages below take one utterance");
    // rows of each matrix per wave; lane q < rpw of wave w publishes row w rpw + q (if it exists)
    const int lane = tid & 63, wave = tid >> 6;
    const int rpq = (G.nq + 3) >> 2, rpo = (G.no + 3) >> 2, rp1 = (G.n1 + 3) >> 2, rp2 = (G.n2 + 3) >> 2;
    const int rowq = wave * rpq + lane, rowo = wave * rpo + lane, row1 = wave * rp1 + lane, row2 = wave * rp2 + lane;
    const bool pubq = lane < rpq && rowq < G.nq, pubo = lane < rpo && rowo < G.no, pub1 = lane < rp1 && row1 < G.n1,
               pub2 = lane < rp2 && row2 < G.n2;
    const float bias_o = pubo ? c.w.bo[g * G.no + rowo] : 0.f, bias_m = pubo ? c.w.bm[g * G.no + rowo] : 0.f;
    const float bias_1 = pub1 ? c.w.b1[g * G.n1 + row1] : 0.f, bias_2 = pub2 ? c.w.b2[g * G.n2 + row2] : 0.f;
    const int gm = g - MAPR0;                               // workgroups MAPR0 .. MAPR0 + 63: coefficient gm of every frame
    const bool mapr = gm >= 0 && gm < VP;
    const float bias_r = (mapr && gm < c.w.V) ? c.w.br[gm] : 0.f;
    __syncthreads();
#ifdef AVI_FFP_STAMPS
    long long stamp_acc[16] = {0}, stamp_last = wall_clock64();
#endif

    // y = LN3(s3 of frame f) for my E columns -> vec[0 .. D) (every workgroup, redundantly); ends with a barrier
    auto last_norm = [&](int f) __attribute__((always_inline)) {
        float t[BT][E];
        poll_regs<BT, E>(xbase + (long long)(f & 1) * G.xpar + G.xs3, D, D, ((epoch << 16) | (unsigned)(f * NEDGE + 1)) + E_S3, x, t);
        FFP_STAMP(0);       // waited for s3 of the previous frame
        float mu[BT], rs[BT];
        row_stats<BT, E>(t, D, red, 4 + 8 * BT, mu, rs);
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int u = 0; u < E; ++u)
                if (u * NT + tid < D) vec[b * 2 * D + u * NT + tid] = (t[b][u] - mu[b]) * rs[b] * g3[u] + b3n[u];
        __syncthreads();
    };
    // coefficient gm of frame f from y in vec (workgroups MAPR0 ..): published for frame f + 1's out-projection, and written out
    auto emit_coeff = [&](int f) __attribute__((always_inline)) {
        float ys[1];
        gemv_rows<1>(img + G.orr, D, 1, 1, D, vec, ys);
        if (tid < BT) {
            const bool real = gm < c.w.V;
            float v = real ? ys[0] + bias_r : 0.f;
            // normalised: what vertice_map feeds back (:722-725)
            publish(xbase + (long long)(f & 1) * G.xpar + G.xo + (long long)tid * VP + gm, v, ((epoch << 16) | (unsigned)(f * NEDGE + 1)) + E_O);
#ifdef AVI_FFP_STAMPS
            if (real && f > 1) {       // the first two frames' slots of the output carry the stamps in this build
#else
            if (real) {
#endif
                if (c.w.coeff_std) v = v * c.w.coeff_std[gm] + c.w.coeff_mean[gm];       // :729
                const long long oi2 = ((long long)tid * c.T + f) * c.w.V + gm;
                if (c.out16) c.out16[oi2] = __builtin_bit_cast(uint16_t, (_Float16)v);
                else c.out[oi2] = v;
            }
        }
    };

    for (int i = 0; i < c.T; ++i) {
        const int phase = i % c.w.period;
        if (*reinterpret_cast<volatile int*>(wg_dead)) x.dead = true;
        unsigned long long* X = xbase + (long long)(i & 1) * G.xpar;
        const unsigned tb = (epoch << 16) | (unsigned)(i * NEDGE + 1);      // tag of edge e of this frame: tb + e
        // loads of this frame that depend on nothing: requested before the first wait
        float crs[BT][E];                   // cross_i, my E columns
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int u = 0; u < E; ++u) {
                const int d = u * NT + tid;
                crs[b][u] = d < D ? c.cross[((long long)b * c.T + i) * D + d] : 0.f;
            }
        const float pe_o = !pubo ? 0.f : i == 0 ? c.p.x0[g * G.no + rowo] : bias_m + c.w.pe[(long long)phase * D + g * G.no + rowo];

        // ---- A: my columns of q, k, v, straight from the previous frame's last LayerNorm ---------------------------------
        const float bfq = !pubq ? 0.f : i == 0 ? c.p.qkv0[g * G.nq + rowq] : c.p.bf[(long long)phase * 3 * D + g * G.nq + rowq];
        if (i == 0) {
            if (pubq) publish(X + G.xq + g * G.nq + rowq, bfq, tb + E_QKV);
        } else {
            last_norm(i - 1);
            float qs3[3];
            gemv_rows<3>(img + G.oq, G.sq, rpq, G.nq, D, vec, qs3);
            if (pubq) publish(X + G.xq + g * G.nq + rowq, (lane == 0 ? qs3[0] : lane == 1 ? qs3[1] : qs3[2]) + bfq + img[G.ocq + rowq], tb + E_QKV);
            if (mapr) emit_coeff(i - 1);
        }
        FFP_STAMP(1);       // q/k/v columns

        // ---- B: split-key attention partials (16 workgroups) -----------------------------------------------------------
        if (g < NATT) {
            const int h = g / KSP, s = g - h * KSP, hoff = h * dh;
            const int kstart = (i / c.chunk) * c.chunk;
            const int j0 = kstart + ((s - kstart % KSP) + KSP) % KSP;     // my first key of the window; keys j0, j0 + 4, ...
            const int nk = i >= j0 ? (i - j0) / KSP + 1 : 0;
            const bool owner = (i % KSP) == s;                            // then my LAST key is frame i itself
            const int LPK = dh >> 2, lpsh = dsh - 2, groups = NT >> lpsh, gk = tid >> lpsh, l = tid & (LPK - 1);
            const int nd4 = dh >> 4, part = tid & 3;                      // float4s of a key a score thread covers (<= 16)
            const float slope = c.w.slopes[h];
            for (int b = 0; b < BT; ++b) {
                float* kvb = c.kv + (long long)b * c.T * 2 * D;
                // my rows of the cache do not depend on q: requested before the poll.  Scores: thread (key tid / 4, quarter
                // tid % 4 of the head dimension); P.V: thread (key group gk, float4 column l).
                f32x4 kpre[16], vpre[PF];
                const int n0 = tid >> 2;
                {
                    const int j = j0 + n0 * KSP;
                    const bool on = n0 < nk && j != i;
                    const f32x4* kr = reinterpret_cast<const f32x4*>(kvb + (long long)(on ? j : 0) * 2 * D + hoff) + part * nd4;
#pragma unroll
                    for (int q = 0; q < 16; ++q) kpre[q] = (on && q < nd4) ? kr[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int n = gk + u * groups, j = j0 + n * KSP;
                    vpre[u] = (n < nk && j != i) ? *reinterpret_cast<const f32x4*>(kvb + (long long)j * 2 * D + D + hoff + 4 * l)
                                                 : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                {
                    const unsigned long long* Xq = X + G.xq + (long long)b * 3 * D + hoff;
                    const int nw = owner ? 3 : 1;
                    unsigned long long gr[3];
#pragma unroll
                    for (int w3 = 0; w3 < 3; ++w3) gr[w3] = (w3 < nw && tid < dh && !x.dead) ? peek(Xq + w3 * D + tid) : 0ull;
                    settle_all<3>([&](int w3) __attribute__((always_inline)) { return Xq + w3 * D + tid; }, gr,
                                  tid < dh ? (nw == 3 ? 7u : 1u) : 0u, tb + E_QKV, x);
#pragma unroll
                    for (int w3 = 0; w3 < 3; ++w3)
                        if (w3 < nw && tid < dh) {
                            const float v = granule_value(gr[w3], x);
                            (w3 == 0 ? qs : w3 == 1 ? ks : vs)[tid] = v;
                            if (w3 > 0) kvb[(long long)i * 2 * D + (w3 - 1) * D + hoff + tid] = v;
                        }
                }
                __syncthreads();
                FFP_STAMP(10);
                // scores of keys n = 64 p + tid / 4; pass 0 from the prefetched rows
                float sco[KEYMAX / 64], mx[1] = {-3.0e38f};
#pragma unroll
                for (int p = 0; p < KEYMAX / 64; ++p) {
                    sco[p] = -3.0e38f;
                    if (p * 64 >= nk) continue;                            // workgroup-uniform
                    const int n = p * 64 + n0, j = j0 + n * KSP;
                    const bool on = n < nk;
                    const f32x4* q4 = reinterpret_cast<const f32x4*>(qs) + part * nd4;
                    const f32x4* kl = reinterpret_cast<const f32x4*>(ks) + part * nd4;
                    const f32x4* kg = reinterpret_cast<const f32x4*>(kvb + (long long)(on ? j : 0) * 2 * D + hoff) + part * nd4;
                    float d = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if (q < nd4) {
                            const f32x4 k4 = (on && j == i) ? kl[q] : (p == 0 ? kpre[q] : (on ? kg[q] : (f32x4){0.f, 0.f, 0.f, 0.f}));
                            const f32x4 qq = q4[q];
                            d = fmaf(qq[0], k4[0], fmaf(qq[1], k4[1], fmaf(qq[2], k4[2], fmaf(qq[3], k4[3], d))));
                        }
                    }
                    d = quad_sum(d);
                    if (on) {
                        sco[p] = d * scale - slope * (float)((i - j) / c.w.period);
                        mx[0] = fmaxf(mx[0], sco[p]);
                    }
                }
                // softmax in ONE workgroup exchange: exponentials about the WAVE's maximum, (max, sum) of the four waves
                // through LDS, the weights rescaled to the common maximum where they are used
                const float mw = wave_max_u(mx[0]);
                float sw = 0.f;
#pragma unroll
                for (int p = 0; p < KEYMAX / 64; ++p) {
                    const int n = p * 64 + n0;
                    if (n < nk && part == 0) {
                        const float pj = __expf(sco[p] - mw);
                        sc[n] = pj;
                        sw += pj;
                    }
                }
                sw = wave_sum_u(sw);
                if ((tid & 63) == 0) red[0][tid >> 6] = mw, red[1][tid >> 6] = sw;
                __syncthreads();                          // also publishes sc[] to every thread
                FFP_STAMP(11);
                mx[0] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
                float fw[4], sum[1] = {0.f};              // wave w's keys: weight exp(m_w - M)
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    fw[w4] = __expf(red[0][w4] - mx[0]);
                    sum[0] = fmaf(fw[w4], red[1][w4], sum[0]);
                }
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int n = gk + u * groups;
                    if (n < nk) a += (sc[n] * fw[(n & 63) >> 4]) * (j0 + n * KSP == i ? *reinterpret_cast<const f32x4*>(vs + 4 * l) : vpre[u]);
                }
                for (int n = gk + PF * groups; n < nk; n += groups) {
                    const int j = j0 + n * KSP;
                    a += (sc[n] * fw[(n & 63) >> 4]) * (j == i ? *reinterpret_cast<const f32x4*>(vs + 4 * l)
                                         : *reinterpret_cast<const f32x4*>(kvb + (long long)j * 2 * D + D + hoff + 4 * l));
                }
                *reinterpret_cast<f32x4*>(accr + 4 * tid) = a;
                __syncthreads();
                FFP_STAMP(12);
                unsigned long long* Xp = X + G.xp + (long long)(b * NATT + g) * G.ps;
                if (tid < dh) {
                    const int lq = tid >> 2, comp = tid & 3;
                    float r = 0.f;
                    for (int gg = 0; gg < groups; ++gg) r += accr[4 * (gg * LPK + lq) + comp];
                    publish(Xp + 4 + tid, r, tb + E_PART);
                }
                if (tid == 0) {
                    publish(Xp + 0, nk > 0 ? mx[0] : -3.0e38f, tb + E_PART);
                    publish(Xp + 1, nk > 0 ? sum[0] : 0.f, tb + E_PART);
                }
                if (b + 1 < BT) __syncthreads();       // qs / ks / vs / sc / accr are free for the next row
            }
        }
        FFP_STAMP(2);       // attention (workgroups 0..15)

        // ---- C: merged attention, my rows of s1 = x + out_proj(att) ---------------------------------------------------
        const unsigned long long* po = xbase + (long long)((i - 1) & 1) * G.xpar + G.xo + tid;   // o_{i-1}: requested now,
        const unsigned long long go = (i > 0 && tid < BT * VP && !x.dead) ? peek(po) : 0ull;         // waited for below
        if (tid < BT * NATT * 2) {
            const int bb = tid / (NATT * 2), r = tid - bb * NATT * 2;
            const unsigned long long* p = X + G.xp + (long long)(bb * NATT + (r >> 1)) * G.ps + (r & 1);
            ml[tid] = settle(p, x.dead ? 0ull : peek(p), tb + E_PART, x);
        }
        {
            // the accumulators of my E columns: requested before the (m, l) pairs are waited for
            unsigned long long gr[E * KSP];             // (column u, key residue s) -> u * KSP + s; BT == 1
            auto addr = [&](int idx) __attribute__((always_inline)) {
                const int d = (idx / KSP) * NT + tid, h = d >> dsh, dd = d & (dh - 1);
                return X + G.xp + (long long)(h * KSP + idx % KSP) * G.ps + 4 + dd;
            };
            unsigned want = 0;
#pragma unroll
            for (int idx = 0; idx < E * KSP; ++idx) {
                const bool on = (idx / KSP) * NT + tid < D;
                if (on) want |= 1u << idx;
                gr[idx] = (on && !x.dead) ? peek(addr(idx)) : 0ull;
            }
            settle_all<E * KSP>(addr, gr, want, tb + E_PART, x);
            __syncthreads();
#pragma unroll
            for (int b = 0; b < BT; ++b)
#pragma unroll
                for (int u = 0; u < E; ++u) {
                    const int d = u * NT + tid, h = d >> dsh;
                    if (d < D) {
                        const float* mlh = ml + (b * NATT + h * KSP) * 2;
                        float M = -3.0e38f;
#pragma unroll
                        for (int s = 0; s < KSP; ++s) M = fmaxf(M, mlh[2 * s]);
                        float L = 0.f, r = 0.f;
#pragma unroll
                        for (int s = 0; s < KSP; ++s) {
                            const float wgt = __expf(mlh[2 * s] - M);
                            L = fmaf(wgt, mlh[2 * s + 1], L);
                            r = fmaf(wgt, granule_value(gr[u * KSP + s], x), r);
                        }
                        vec[b * 2 * D + d] = r / L;
                    }
                }
        }
        if (tid < BT * VP) {                                                                     // [att | o_{i-1}]
            const float o = i > 0 ? settle(po, go, tb - NEDGE + E_O, x) : 0.f;
            vec[(tid / VP) * 2 * D + D + (tid % VP)] = o;
        }
        __syncthreads();
        FFP_STAMP(3);       // waited for the partials, merged them
        {
            float os[1];
            gemv_rows<1>(img + G.oo, G.so, rpo, G.no, D + VP, vec, os);
            // x = vertice_map(o_{i-1}) + pe_i (frame 0: obj_embedding + pe_0); the vertice_map product came out of the same rows
            if (pubo) publish(X + G.xs1 + g * G.no + rowo, os[0] + bias_o + pe_o, tb + E_S1);
        }
        FFP_STAMP(4);       // out-projection rows

        // ---- D: x2 = LN2(LN1(s1) + cross_i), my rows of h = relu(linear1 x2) ---------------------------------------------
        {
            float t[BT][E];
            poll_regs<BT, E>(X + G.xs1, D, D, tb + E_S1, x, t);
            FFP_STAMP(5);   // waited for s1
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {          // LayerNorm 1 (+ cross), LayerNorm 2
                float mu[BT], rs[BT];
                row_stats<BT, E>(t, D, red, 4 + pass * 4 * BT, mu, rs);
#pragma unroll
                for (int b = 0; b < BT; ++b)
#pragma unroll
                    for (int u = 0; u < E; ++u)
                        if (u * NT + tid < D)
                            t[b][u] = pass == 0 ? (t[b][u] - mu[b]) * rs[b] * g1[u] + b1n[u] + crs[b][u]
                                                : (t[b][u] - mu[b]) * rs[b] * g2[u] + b2n[u];
            }
#pragma unroll
            for (int b = 0; b < BT; ++b)
#pragma unroll
                for (int u = 0; u < E; ++u)
                    if (u * NT + tid < D) vec[b * 2 * D + u * NT + tid] = t[b][u];
            __syncthreads();
        }
        FFP_STAMP(6);       // two LayerNorms
        const float x2own = pub2 ? vec[g * G.n2 + row2] : 0.f;       // linear2's residual, my rows
        {
            float hs2[2];
            gemv_rows<2>(img + G.o1, G.s1, rp1, G.n1, D, vec, hs2);
            if (pub1) publish(X + G.xh + g * G.n1 + row1, fmaxf((lane == 0 ? hs2[0] : hs2[1]) + bias_1, 0.f), tb + E_H);
        }
        FFP_STAMP(7);       // linear1 rows

        // ---- E: my rows of s3 = x2 + linear2 h ----------------------------------------------------------------------------
        {
            float t[BT][2 * E];
            poll_regs<BT, 2 * E>(X + G.xh, 2 * D, 2 * D, tb + E_H, x, t);
            __syncthreads();            // every wave is done with x2 in vec (linear1 has no barrier of its own)
#pragma unroll
            for (int b = 0; b < BT; ++b)
#pragma unroll
                for (int u = 0; u < 2 * E; ++u)
                    if (u * NT + tid < 2 * D) vec[b * 2 * D + u * NT + tid] = t[b][u];
            __syncthreads();
        }
        FFP_STAMP(8);       // waited for h
        {
            float ss[1];
            gemv_rows<1>(img + G.o2, G.s2, rp2, G.n2, 2 * D, vec, ss);
            if (pub2) publish(X + G.xs3 + g * G.n2 + row2, x2own + bias_2 + ss[0], tb + E_S3);
        }
        FFP_STAMP(9);       // linear2 rows

    }
    if (mapr) {             // the last frame's coefficients
        last_norm(c.T - 1);
        emit_coeff(c.T - 1);
    }
#ifdef AVI_FFP_STAMPS
    __syncthreads();
    if (tid == 0 && (g == 0 || g == 200 || g == 13 || g == 70) && c.out)      // attention, plain, attention, coefficient workgroup
        for (int k = 0; k < 16; ++k) c.out[(g == 0 ? 0 : g == 200 ? 16 : g == 13 ? 32 : 48) + k] = (float)stamp_acc[k];
#endif
}

// after the decode, in stream order: the next launch's epoch; every 2^16 launches the tag space wraps, so the slots are cleared
__global__ void ff_persist_epoch_kernel(unsigned long long* __restrict__ ws, long long slot_words) {
    __shared__ int wrap;
    if (threadIdx.x == 0) {
        const unsigned long long e = ws[0] + 1;
        ws[0] = e;
        wrap = (e & 0xFFFFull) == 0;
    }
    __syncthreads();
    if (wrap)
        for (long long i = threadIdx.x; i < slot_words; i += blockDim.x) ws[XCH_HDR + i] = 0ull;
}

// image[g] = workgroup g's rows of every matrix, fp32, padded strides (see Geo); grid NWG
__global__ __launch_bounds__(NT) void ff_persist_pack_kernel(const AviFaceformerWeights w, const AviFaceformerPlanes p,
                                                              float* __restrict__ image) {
    const Geo G = geo(w.D);
    const int D = w.D, g = blockIdx.x;
    float* im = image + (long long)g * G.img;
    for (int i = threadIdx.x; i < G.img; i += NT) {
        float v = 0.f;
        if (i < G.oo) {                       // q/k/v from LN3's output: row c = g nq + r of (wf_t^T . wr^T), [3D][D]
            const int r = i / G.sq, k = i - r * G.sq, col = g * G.nq + r;
            if (k < D)
                for (int vv = 0; vv < w.V; ++vv) v = fmaf(p.wf_t[(long long)vv * 3 * D + col], w.wr[(long long)k * w.V + vv], v);
        } else if (i < G.o1) {                // [out_proj | vertice_map] row n: wo [K = D][N = D], wm [V][D]
            const int q = i - G.oo, r = q / G.so, k = q - r * G.so, n = g * G.no + r;
            if (k < D) v = w.wo[(long long)k * D + n];
            else if (k - D < w.V) v = w.wm[(long long)(k - D) * D + n];
        } else if (i < G.o2) {                // linear1 row m: w1 [D][2D]
            const int q = i - G.o1, r = q / G.s1, k = q - r * G.s1;
            if (k < D) v = w.w1[(long long)k * 2 * D + g * G.n1 + r];
        } else if (i < G.orr) {               // linear2 row n: w2 [2D][D]
            const int q = i - G.o2, r = q / G.s2, k = q - r * G.s2;
            if (k < 2 * D) v = w.w2[(long long)k * D + g * G.n2 + r];
        } else if (i < G.ocq) {               // vertice_map_r row g - MAPR0: wr [D][V]
            const int k = i - G.orr, gm = g - MAPR0;
            if (gm >= 0 && gm < w.V) v = w.wr[(long long)k * w.V + gm];
        } else {                              // constant part of my q/k/v columns: wf_t^T . br
            const int r = i - G.ocq, col = g * G.nq + r;
            if (r < G.nq)
                for (int vv = 0; vv < w.V; ++vv) v = fmaf(p.wf_t[(long long)vv * 3 * D + col], w.br[vv], v);
        }
        im[i] = v;
    }
}

bool persist_shape_ok(int D) { return D == 256 || D == 512 || D == 1024; }

}  // namespace

// *image_floats: size of the per-model LDS image (avi_faceformer_persist_pack); *xch_bytes: the exchange workspace, zero-filled
// ONCE by the caller and then left to the library (it carries the launch epoch); one launch at a time may use it.
extern "C" int avi_faceformer_persist_sizes(int D, long long* image_floats, long long* xch_bytes) {
    if (!persist_shape_ok(D) || !image_floats || !xch_bytes) return AVI_EINVAL;
    const Geo G = geo(D);
    *image_floats = (long long)NWG * G.img;
    *xch_bytes = 8 * (XCH_HDR + 2 * G.xpar);
    return AVI_OK;
}

extern "C" int avi_faceformer_persist_pack(const AviFaceformerWeights* w, const AviFaceformerPlanes* p, float* image,
                                           void* stream) {
    if (!w || !p || !image || !persist_shape_ok(w->D) || w->V < 1 || w->V > VP) return AVI_EINVAL;
    if (!w->wo || !w->w1 || !w->w2 || !w->wr || !w->wm || !w->br || !p->wf_t) return AVI_EINVAL;
    if (reinterpret_cast<uintptr_t>(image) & 15) return AVI_EINVAL;
    hipLaunchKernelGGL(ff_persist_pack_kernel, dim3(NWG), dim3(NT), 0, static_cast<hipStream_t>(stream), *w, *p, image);
    return avi_launch_status();
}

extern "C" int avi_faceformer_decode_persistent(const AviFaceformerWeights* w, const AviFaceformerPlanes* p, const float* image,
                                                const float* cross, int B, int T, int chunk, float* kv_scratch, void* xch,
                                                float* out, uint16_t* out16, void* stream) {
    if (!w || !p || !image || !cross || !kv_scratch || !xch || (!out && !out16) || B <= 0 || B > PB || T <= 0) return AVI_EINVAL;
    if (!persist_shape_ok(w->D) || w->V < 1 || w->V > VP || w->period < 1) return AVI_EINVAL;
    if ((long long)T * NEDGE + NEDGE >= 65535) return AVI_EINVAL;          // 16 bits of tag per launch
    if (chunk <= 0 || chunk > T) chunk = T;
    if (chunk < T && chunk % w->period) return AVI_EINVAL;
    if (chunk > KEYMAX * KSP) return AVI_EINVAL;                            // a key residue's scores live in LDS
    if (!p->bf || !p->qkv0 || !p->x0 || !w->bo || !w->b1 || !w->b2 || !w->br || !w->bm || !w->pe || !w->slopes || !w->n1g ||
        !w->n1b || !w->n2g || !w->n2b || !w->n3g || !w->n3b)
        return AVI_EINVAL;
    if ((w->coeff_mean == nullptr) != (w->coeff_std == nullptr)) return AVI_EINVAL;
    if ((reinterpret_cast<uintptr_t>(xch) & 7) || (reinterpret_cast<uintptr_t>(image) & 15)) return AVI_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus < NWG)
        return AVI_EINVAL;                                                  // one workgroup per CU, all resident at once
    const Geo G = geo(w->D);
    const int smem = lds_floats(G) * (int)sizeof(float);
    static AviLdsGrant grant1;
    grant1.ensure(reinterpret_cast<const void*>(ff_persist_kernel<1>), 160 * 1024);
    if (smem > 160 * 1024) return AVI_EINVAL;
    Persist c;
    c.w = *w, c.p = *p, c.image = image, c.cross = cross, c.kv = kv_scratch, c.out = out, c.out16 = out16;
    c.xch = static_cast<unsigned long long*>(xch);
    c.status = avi_status_ptr();
    c.B = B, c.T = T, c.D = w->D, c.chunk = chunk;
    c.fault = avi_fault_injected() & AVI_FAULT_EXCHANGE_ABSENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ff_persist_kernel<1>, dim3(NWG), dim3(NT), smem, s, c);
    hipLaunchKernelGGL(ff_persist_epoch_kernel, dim3(1), dim3(256), 0, s, c.xch, 2 * G.xpar);
    return avi_launch_status();
}
