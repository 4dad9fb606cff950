// FLAME vertices from per-frame coefficients: linear blend skinning of the 5-joint head model
// (third_party/inferno/inferno/utils/lbs.py `lbs` :142-235 as called by `FLAME.forward`, DecaFLAME.py:222-244;
// SURVEY.md 8f row 1).  All fp32 on the vector pipe: 26 GFLOP and 0.5 GB of output at config[1] size, 2 % of a step.
//
//   v_shaped  = v_template + S_shape . shape            per CLIP   (flame_shape_kernel)
//   per FRAME (flame_frame_kernel, one wave per frame):
//     joints   = J_regressor . (v_shaped + S_exp . exp) = j_template + j_shape . shape + j_exp . exp   (re-associated:
//                the 5 x V regressor is folded into the bases once per model, 1e-7 relative)
//     R_j      = rodrigues(pose_j)  (the reference's +1e-8 inside the norm kept), pose feature = (R_1..4 - I)
//     A_j      = chain of the kinematic tree [-1,0,1,1,1] made relative to the rest joints (rows 0..2 of each 4x4)
//   per (frame, vertex) (flame_vertices_kernel):
//     v_posed  = v_shaped + sum_k exp_k E_k + sum_p feature_p P_p      (86 basis vectors, staged once per block in LDS)
//     vertex   = (sum_j w_vj A_j) . [v_posed; 1]
// flame_vertices_mfma_kernel (the path taken when the basis planes are present): the 86-vector blend is a
// [frames x 96] . [96 x V*3] product in 3-term split bf16 on the matrix cores, one wave per 16 vertices x all frames
// of one clip with the basis fragments resident in registers; the skinning runs on the accumulators (a lane holds
// x, y, z of 4 consecutive vertices of one frame) and the result leaves as 48 contiguous bytes per lane.
// flame_vertices_kernel (no planes given; up to 106 basis vectors, all held in LDS): all fp32 on the vector pipe,
// block = 128 vertices x all frames of one clip; 512 threads = 128 vertices x 4 frame
// subgroups, 8 frames in flight per thread (24 accumulators); per-frame coefficients are wave-uniform (scalar loads)
// and stored frame-group-major [f/8][k][8] by flame_frame_kernel so that one 32-byte scalar load feeds 8 frames.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int NJ = 5, NPF = 36, VT = 128, FG = 8;   // joints, pose features, vertices per block, frames per group

// ---- preparation: ONE launch, two kinds of workgroups (256 threads) -----------------------------------------------------
// (as three launches - shape, rest joints, frames - the preparation took 77 us of a 346-us pass at config[1] size: each was
// a chain of dependent load rounds a few workgroups deep; the two kinds below do not depend on each other)
//
// "tile" workgroups, one per (clip, 16 frames): everything the vertices kernels need per FRAME.
//   vector-pipe kernel (KP == 0): coefficients fp32 frame-group-major `coef` [F/8][K][8] (K = n_exp + 36), transforms
//     fp32 `xf` [F][5][12].
//   matrix-core kernel (KP = 96 | 160): everything in MFMA FRAGMENT order, one block of bytes per (clip, 16-frame tile)
//     so that the vertices kernel fetches it by LDS-DMA as it lies and every lane reads its 16 bytes conflict-free:
//       coef tile  [plane hi|lo][ks][fq 4][fr 16][8 bf16]: element (frame fr, basis vector ks*32 + fq*8 + u)
//       xf tile    [entry e 12][fq 4][fr 16][8 bf16]: slot s = fq*8 + u = joint*6 + term of the transform blend
//     The blend  T_e = sum_j w_j A_j[e]  is ONE 32-deep bf16 product per entry with fp32-equivalent operands: w and A are
//     split in three bf16 parts each (w = w1 + w2 + w3 to 2^-24) and the six partial products that matter
//     (w1 a1, w1 a2, w1 a3, w2 a1, w2 a2, w3 a1; the rest is below 2^-24 of |w a|) take six slots per joint.
//     Frames past the end of the clip (its last tile) repeat the clip's last frame.
// "shape" workgroups, one per (32 columns of V*3, 32 clips): v_shaped = v_template + shape . shape_basis in exact fp32 on
//   the matrix cores (v_mfma_f32_16x16x4_f32); a wave owns a quarter of the k range (the chain of dependent load rounds is
//   what this small product costs: 300 rows deep on the vector pipe it took 44 us), the four partial sums meet in LDS.
constexpr int XF_TERM_A[6] = {0, 1, 2, 0, 1, 0};    // which part of A a slot holds; the w side holds {0,0,0,1,1,2}
constexpr int XF_TILE = 12 * 1024;                  // bytes of a transform tile
constexpr int MAX_EXP = 124;

__device__ __forceinline__ void split3_bf16(float v, uint16_t out[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const __bf16 h = (__bf16)v;
        out[i] = __builtin_bit_cast(uint16_t, h);
        v -= (float)h;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // rows of V*3 floats are dword aligned only
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

struct PrepTileLds {
    float jes[NJ * 3 * MAX_EXP], es[16][128], Rs[16][NJ * 9 + 3], Js[16][16], Gs[16][NJ * 12 + 4], jc[16];
};
struct PrepShapeLds {
    f32x4 red[4][4][64];
};
union PrepLds {
    PrepTileLds t;
    PrepShapeLds s;
};

__device__ __forceinline__ void prep_tile_block(const AviFlameBasis& fb, const float* __restrict__ shape,
                                                const float* __restrict__ exp, const float* __restrict__ pose, int T, int b,
                                                int ti, float* __restrict__ coef, float* __restrict__ xf, int KP,
                                                PrepTileLds& L) {
    const int tid = threadIdx.x, fr = tid >> 4, sub = tid & 15;
    const int K = fb.n_exp + NPF, nt = T - ti * 16 < 16 ? T - ti * 16 : 16;       // frames of this tile
    const long long fbase = (long long)b * T + ti * 16;
    const long long f = fbase + (fr < nt ? fr : nt - 1);
    // Every global value this workgroup needs is requested before the first one is used (loops of load -> store pairs
    // wait out one memory round trip per iteration: 26 of them made this the longest kernel of the preparation).
    {
        float v[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {       // j_exp: 15 n_exp <= 1 860 values; 16 frames of expression: <= 1 984
            const int i = tid + u * 256, r = i / fb.n_exp, k = i - r * fb.n_exp;
            v[u] = i < NJ * 3 * fb.n_exp ? fb.j_exp[i] : 0.f;
            w[u] = i < 16 * fb.n_exp ? exp[(fbase + (r < nt ? r : nt - 1)) * fb.n_exp + k] : 0.f;
        }
        // rest joints of the clip, j_template + j_shape . shape[b]: a thread takes shape[k] against all 15 rows
        float a[NJ * 3];
#pragma unroll
        for (int c = 0; c < NJ * 3; ++c) a[c] = 0.f;
        for (int k0 = 0; k0 < fb.n_shape; k0 += 256) {
            const int k = k0 + tid, kk = k < fb.n_shape ? k : fb.n_shape - 1;
            const float x = k < fb.n_shape ? shape[(long long)b * fb.n_shape + kk] : 0.f;
            float js[NJ * 3];
#pragma unroll
            for (int c = 0; c < NJ * 3; ++c) js[c] = fb.j_shape[(long long)c * fb.n_shape + kk];
#pragma unroll
            for (int c = 0; c < NJ * 3; ++c) a[c] = fmaf(js[c], x, a[c]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, r = i / fb.n_exp, k = i - r * fb.n_exp;
            if (i < NJ * 3 * fb.n_exp) L.jes[i] = v[u];
            if (i < 16 * fb.n_exp) L.es[r][k] = w[u];
        }
#pragma unroll
        for (int c = 0; c < NJ * 3; ++c) {
            const float t = wave_sum_u(a[c]);
            if ((tid & 63) == 0) L.Js[tid >> 6][c] = t;      // Js is free until the joints are formed below
        }
    }
    if (sub < NJ) {       // lbs.py:304-335
        const float* p = pose + f * (NJ * 3) + sub * 3;
        const float x = p[0], y = p[1], z = p[2];
        const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rx = x / angle, ry = y / angle, rz = z / angle;
        const float sn = sinf(angle), c1 = 1.f - cosf(angle);
        // K = [[0,-rz,ry],[rz,0,-rx],[-ry,rx,0]];  R = I + s K + (1-c) K.K
        const float kk[9] = {-(ry * ry + rz * rz), rx * ry, rx * rz, rx * ry, -(rx * rx + rz * rz), ry * rz,
                             rx * rz, ry * rz, -(rx * rx + ry * ry)};
        const float k1[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) L.Rs[fr][sub * 9 + i] = ((i % 4 == 0) ? 1.f : 0.f) + sn * k1[i] + c1 * kk[i];
    }
    __syncthreads();
    if (tid < NJ * 3) L.jc[tid] = fb.j_template[tid] + ((L.Js[0][tid] + L.Js[1][tid]) + (L.Js[2][tid] + L.Js[3][tid]));
    __syncthreads();
    if (sub < NJ * 3) {   // joints of frame fr: rest joints + j_exp . exp
        float a = L.jc[sub];
        const float* je = L.jes + sub * fb.n_exp;
        for (int k = 0; k < fb.n_exp; ++k) a = fmaf(je[k], L.es[fr][k], a);
        L.Js[fr][sub] = a;
    }
    __syncthreads();
    if (sub == 0) {       // lbs.py:351-410, parents = [-1, 0, 1, 1, 1]
        const float* R = L.Rs[fr];
        const float* J = L.Js[fr];
        float* G = L.Gs[fr];
        float Gc[NJ][12];  // rows 0..2 of the chained transforms
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int par = j == 0 ? -1 : (j == 1 ? 0 : 1);
            float tr[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) tr[c] = J[j * 3 + c] - (par >= 0 ? J[par * 3 + c] : 0.f);
            if (par < 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) Gc[j][r * 4 + c] = R[j * 9 + r * 3 + c];
                    Gc[j][r * 4 + 3] = tr[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        Gc[j][r * 4 + c] = Gc[par][r * 4 + 0] * R[j * 9 + c] + Gc[par][r * 4 + 1] * R[j * 9 + 3 + c] +
                                           Gc[par][r * 4 + 2] * R[j * 9 + 6 + c];
                    Gc[j][r * 4 + 3] = Gc[par][r * 4 + 0] * tr[0] + Gc[par][r * 4 + 1] * tr[1] + Gc[par][r * 4 + 2] * tr[2] +
                                       Gc[par][r * 4 + 3];
                }
            }
        }
        // relative to the rest pose: translation -= G[:3,:3] . joint
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                G[j * 12 + r * 4 + 0] = Gc[j][r * 4 + 0];
                G[j * 12 + r * 4 + 1] = Gc[j][r * 4 + 1];
                G[j * 12 + r * 4 + 2] = Gc[j][r * 4 + 2];
                G[j * 12 + r * 4 + 3] = Gc[j][r * 4 + 3] - (Gc[j][r * 4 + 0] * J[j * 3] + Gc[j][r * 4 + 1] * J[j * 3 + 1] +
                                                             Gc[j][r * 4 + 2] * J[j * 3 + 2]);
            }
    }
    __syncthreads();
    // coefficient k of frame r: expression, then (R[1 + q/9] - I).flat[q % 9]  (lbs.py:210), zero padding
    auto coefficient = [&](int r, int k) __attribute__((always_inline)) {
        if (k < fb.n_exp) return L.es[r][k];
        if (k >= K) return 0.f;
        const int q = k - fb.n_exp;
        return L.Rs[r][9 + q] - ((q % 9) % 4 == 0 ? 1.f : 0.f);
    };
    if (KP) {
        const int KS = KP >> 5, tile = b * ((T + 15) >> 4) + ti;
        char* ct = reinterpret_cast<char*>(coef) + (long long)tile * (2 * KS * 1024);
        for (int ch = tid; ch < KS * 64; ch += 256) {        // chunk ((ks*4 + fq)*16 + r): 8 coefficients of frame r
            const int r = ch & 15, k0 = (ch >> 4) * 8;
            u16x8 hi, lo;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = coefficient(r, k0 + u);
                const __bf16 h = (__bf16)v;
                hi[u] = __builtin_bit_cast(uint16_t, h);
                lo[u] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
            }
            *reinterpret_cast<u16x8*>(ct + ch * 16) = hi;
            *reinterpret_cast<u16x8*>(ct + KS * 1024 + ch * 16) = lo;
        }
        char* xt = reinterpret_cast<char*>(xf) + (long long)tile * XF_TILE;
        for (int ch = tid; ch < 12 * 64; ch += 256) {        // chunk ((e*4 + fq)*16 + r): 8 slots of entry e, frame r
            const int r = ch & 15, fq = (ch >> 4) & 3, e = ch >> 6;
            u16x8 v;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sl = fq * 8 + u, j = sl / 6, term = sl - j * 6;
                uint16_t parts[3] = {0, 0, 0};
                if (j < NJ) split3_bf16(L.Gs[r][j * 12 + e], parts);
                v[u] = j < NJ ? (XF_TERM_A[term] == 0 ? parts[0] : XF_TERM_A[term] == 1 ? parts[1] : parts[2]) : (uint16_t)0;
            }
            *reinterpret_cast<u16x8*>(xt + ch * 16) = v;
        }
    } else {
        for (int i = tid; i < 16 * K; i += 256) {
            const int r = i / K, k = i - r * K;
            const long long ff = fbase + r;
            if (r < nt) coef[((ff / FG) * K + k) * FG + (ff % FG)] = coefficient(r, k);
        }
        for (int i = tid; i < 16 * NJ * 12; i += 256) {
            const int r = i / (NJ * 12), q = i - r * (NJ * 12);
            if (r < nt) xf[(fbase + r) * (NJ * 12) + q] = L.Gs[r][q];
        }
    }
}

__device__ __forceinline__ void prep_shape_block(const AviFlameBasis& fb, const float* __restrict__ shape, int B, int cb,
                                                 int gb, float* __restrict__ v_shaped, PrepShapeLds& L) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4, n = fb.V * 3;
    const int ksteps = (fb.n_shape + 3) >> 2, per = (ksteps + 3) >> 2;
    const int s0 = wave * per, s1 = s0 + per < ksteps ? s0 + per : ksteps;
    const float* bp[2];
    const float* sp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = cb * 32 + q * 16 + li, c = gb * 32 + q * 16 + li;
        bp[q] = fb.shape_basis + (i < n ? i : n - 1);
        sp[q] = shape + (long long)(c < B ? c : B - 1) * fb.n_shape;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q >> 1][q & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < s1; s += 10) {     // ten k steps of loads in flight
        float a[10][2], c[10][2];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int k = (s + u) * 4 + lk, kk = k < fb.n_shape ? k : fb.n_shape - 1;
            const bool on = s + u < s1 && k < fb.n_shape;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a[u][q] = on ? bp[q][(long long)kk * n] : 0.f;
                c[u][q] = sp[q][kk];
            }
        }
#pragma unroll
        for (int u = 0; u < 10; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[q >> 1][q & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][q >> 1], c[u][q & 1], acc[q >> 1][q & 1], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) L.red[wave][q][lane] = acc[q >> 1][q & 1];
    __syncthreads();
    // wave w finishes tile (columns q >> 1, clips q & 1), q = w: lane holds clip li, columns 4 lk .. 4 lk + 3
    const int q = wave, c = gb * 32 + (q & 1) * 16 + li, i = cb * 32 + (q >> 1) * 16 + lk * 4;
    const f32x4 v = (L.red[0][q][lane] + L.red[1][q][lane]) + (L.red[2][q][lane] + L.red[3][q][lane]);
    float t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = fb.v_template[i + r < n ? i + r : n - 1];
    if (c < B) {
        float* o = v_shaped + (long long)c * n + i;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (i + r < n) o[r] = t[r] + v[r];
    }
}

// grid: B * ceil(T/16) tile workgroups, then ceil(V*3/32) * ceil(B/32) shape workgroups
__global__ __launch_bounds__(256) void flame_prep_kernel(const AviFlameBasis fb, const float* __restrict__ shape,
                                                          const float* __restrict__ exp, const float* __restrict__ pose, int B,
                                                          int T, float* __restrict__ v_shaped, float* __restrict__ coef,
                                                          float* __restrict__ xf, int KP, int only) {
    __shared__ PrepLds L;
    const int ntile = (T + 15) >> 4, tiles = B * ntile;
    if ((int)blockIdx.x < tiles) {
        if (only == 2) return;      // diagnostic (AVI_FLAME_ONLY): time one kind of workgroup alone
        const int b = blockIdx.x / ntile;
        prep_tile_block(fb, shape, exp, pose, T, b, blockIdx.x - b * ntile, coef, xf, KP, L.t);
    } else {
        if (only == 1) return;
        const int sb = blockIdx.x - tiles, ncb = (fb.V * 3 + 31) >> 5, gb = sb / ncb;
        prep_shape_block(fb, shape, B, sb - gb * ncb, gb, v_shaped, L.s);
    }
}

// grid (ceil(V/128), B); block 512; dynamic LDS = K * 3 * 128 floats
__global__ __launch_bounds__(512) void flame_vertices_kernel(const AviFlameBasis fb, const float* __restrict__ v_shaped,
                                                              const float* __restrict__ coef,
                                                              const float* __restrict__ xf, int T,
                                                              float* __restrict__ verts) {
    extern __shared__ __attribute__((aligned(16))) float sb[];   // [K][3][VT]
    const int K = fb.n_exp + NPF, n3 = fb.V * 3;
    const int tid = threadIdx.x, vl = tid & (VT - 1);
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 7);    // frame subgroup 0..3 (two waves each)
    const int v0 = blockIdx.x * VT, v = v0 + vl, b = blockIdx.y;
    const bool vok = v < fb.V;
    for (int i = tid; i < K * 3 * VT; i += 512) {
        const int k = i / (3 * VT), r = i - k * 3 * VT, c = r / VT, x = r - c * VT;
        const int vv = v0 + x;
        sb[i] = vv < fb.V ? fb.frame_basis[(long long)k * n3 + vv * 3 + c] : 0.f;
    }
    float w[NJ], vs[3];
#pragma unroll
    for (int j = 0; j < NJ; ++j) w[j] = vok ? fb.lbs_weights[(long long)v * NJ + j] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) vs[c] = vok ? v_shaped[(long long)b * n3 + v * 3 + c] : 0.f;
    __syncthreads();
    // frames of clip b: f = b*T + t; groups of 8 aligned to the GLOBAL frame index (the coef layout)
    const int fbeg = b * T, fend = fbeg + T;
    for (int g0 = (fbeg / FG) + sub; g0 * FG < fend; g0 += 4) {
        float acc[FG][3];
#pragma unroll
        for (int i = 0; i < FG; ++i) { acc[i][0] = vs[0]; acc[i][1] = vs[1]; acc[i][2] = vs[2]; }
        const float* cg = coef + (long long)g0 * K * FG;
        for (int k = 0; k < K; ++k) {
            const float b0 = sb[(k * 3 + 0) * VT + vl], b1 = sb[(k * 3 + 1) * VT + vl], b2 = sb[(k * 3 + 2) * VT + vl];
#pragma unroll
            for (int i = 0; i < FG; ++i) {
                const float c = cg[k * FG + i];   // wave-uniform
                acc[i][0] = fmaf(c, b0, acc[i][0]);
                acc[i][1] = fmaf(c, b1, acc[i][1]);
                acc[i][2] = fmaf(c, b2, acc[i][2]);
            }
        }
#pragma unroll
        for (int i = 0; i < FG; ++i) {
            const int f = g0 * FG + i;
            if (f < fbeg || f >= fend) continue;   // wave-uniform
            const float* a = xf + (long long)f * (NJ * 12);
            float tm[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) s = fmaf(w[j], a[j * 12 + e], s);
                tm[e] = s;
            }
            if (vok) {
                float* o = verts + ((long long)f * fb.V + v) * 3;
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    o[r] = tm[r * 4 + 0] * acc[i][0] + tm[r * 4 + 1] * acc[i][1] + tm[r * 4 + 2] * acc[i][2] + tm[r * 4 + 3];
            }
        }
    }
}


// ---- matrix-core path ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// basis planes [3][Vp][KP]: hi/lo[(c*Vp + v)*KP + k] = split of frame_basis[k][v*3 + c]; zero for v >= V, k >= K
__global__ __launch_bounds__(256) void flame_pack_basis_kernel(const AviFlameBasis fb, int Vp, int KP,
                                                                uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= 3LL * Vp * KP) return;
    const int k = (int)(i % KP), v = (int)((i / KP) % Vp), c = (int)(i / ((long long)KP * Vp));
    const int K = fb.n_exp + NPF;
    const float x = (k < K && v < fb.V) ? fb.frame_basis[(long long)k * fb.V * 3 + v * 3 + c] : 0.f;
    const __bf16 h = (__bf16)x;
    hi[i] = __builtin_bit_cast(uint16_t, h);
    lo[i] = __builtin_bit_cast(uint16_t, (__bf16)(x - (float)h));
}

// One workgroup (256 threads) per (64 vertices, clip): wave w owns vertices [vt*16, vt*16+16), vt = 4*bx + w, and walks
// the clip's frames in tiles of 16.  MFMA operand roles as in gemm.hip: first operand = rows of the "n" matrix (vertices),
// second = rows of the "m" matrix (frames); lane (fr, fq) receives D[frame f0+fr][vertex vt*16 + fq*4 + 0..3].
//   blend      27 (KS = 3) products 16x16x32 in 3-term split bf16 against the basis fragments resident in registers,
//              accumulators started at the clip's shaped template;
//   transforms 12 products 16x16x32 (one per entry of the 3 x 4 blend of the five joint transforms, see
//              flame_frame_kernel) against the wave's skinning-weight fragment - the 240 multiply-adds per lane this
//              replaces were what the kernel's time went on (vector pipe 4 cycles each, 265 -> 223 us with an exact-fp32
//              16x16x4 form, this form halves the matrix cycles again and needs no accumulator set-up);
//   skinning   3 fused multiply-adds per output on the accumulators; the 48 bytes per lane leave through an LDS slab so
//              that 12 consecutive lanes write one frame's 192 bytes.
// The operands of a tile (coefficient planes 2 KS KiB + transform planes 12 KiB, both in fragment order) are fetched once
// per workgroup by LDS-DMA, one tile ahead: requested right after the barrier that opens tile t, they land under its
// arithmetic and stores.
//   vmcnt: the DMA requests of a wave are older than the three output stores of the tile, so `s_waitcnt vmcnt(3)` at
//   the top of the next tile means "my DMA pieces have landed"; the barrier that follows makes that true for all waves
//   and also orders the previous tile's LDS reads before the buffer is refilled (two buffers, one barrier per tile).
// Workgroup numbering is XCD-aware: ids are dealt round-robin to the 8 XCDs (each with its own L2), and a frame's row of
// the output is only 4-byte aligned (V*3 floats), so neighbouring vertex groups share cache lines; numbered naively the
// two halves of such a line are written from two L2s and reach memory as two partial writes - the store pattern ALONE then
// takes 123-142 us for the 482 MB of config[1] against 87-116 us with neighbours on one XCD (`scripts/probe/
// flame_store_probe.hip`; 74-93 us if rows could be padded to 64 bytes, which the reference's contiguous layout forbids).
typedef __attribute__((address_space(3))) void flame_lds_void;
typedef const __attribute__((address_space(1))) void flame_gbl_void;

template <int KS, int OCC>
__global__ __launch_bounds__(256, OCC) void flame_vertices_mfma_kernel(const AviFlameBasis fb,
                                                                   const float* __restrict__ v_shaped,
                                                                   const char* __restrict__ coef_tiles,
                                                                   const char* __restrict__ xf_tiles, int T, int Vp, int nx,
                                                                   int total, float* __restrict__ verts) {
    constexpr int KP = KS * 32, SLAB_ROW = 52;
    constexpr int CT = 2 * KS * 1024;                  // bytes of a coefficient tile (hi | lo)
    constexpr int BUF = CT + XF_TILE;
    constexpr int NCH = 2 * KS + 12;                   // 1-KiB DMA pieces per tile
    __shared__ __attribute__((aligned(16))) char stage[2 * BUF];
    __shared__ __attribute__((aligned(16))) float slabs[4 * 16 * SLAB_ROW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // the launch holds 8 * ceil(total / 8) workgroups: id % 8 is the XCD, which takes a contiguous range of (clip, group)
    const int per = gridDim.x >> 3, L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (L >= total) return;
    const int b = L / nx, bx = L - b * nx;
    const int vt_raw = bx * 4 + wave;
    const bool live = vt_raw * 16 < fb.V;              // wave-uniform; a spare wave still fetches and synchronises
    const int vt = live ? vt_raw : 0;
    const int fbeg = b * T, fend = fbeg + T, ntile = (T + 15) >> 4;
    const char* ctile = coef_tiles + (long long)b * ntile * CT;
    const char* xtile = xf_tiles + (long long)b * ntile * XF_TILE;

    auto issue = [&](int ti, int buf) __attribute__((always_inline)) {
        for (int c = wave; c < NCH; c += 4) {
            const char* src = c < 2 * KS ? ctile + (long long)ti * CT + c * 1024 : xtile + (long long)ti * XF_TILE + (c - 2 * KS) * 1024;
            __builtin_amdgcn_global_load_lds((flame_gbl_void*)(src + lane * 16), (flame_lds_void*)(stage + buf * BUF + c * 1024), 16,
                                             0, 0);
        }
    };
    issue(0, 0);

    bf16x8 bh[3][KS], bl[3][KS];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const long long o = ((long long)c * Vp + vt * 16 + fr) * KP + ks * 32 + fq * 8;
            bh[c][ks] = *reinterpret_cast<const bf16x8*>(fb.basis_hi + o);
            bl[c][ks] = *reinterpret_cast<const bf16x8*>(fb.basis_lo + o);
        }
    // skinning-weight fragment: slot s = fq*8 + u = joint*6 + term holds part {0,0,0,1,1,2}[term] of w[vertex fr][joint]
    bf16x8 wf;
    {
        const float* wr = fb.lbs_weights + (long long)(vt * 16 + fr < fb.V ? vt * 16 + fr : fb.V - 1) * NJ;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sl = fq * 8 + u, j = sl / 6, term = sl - j * 6;
            uint16_t parts[3] = {0, 0, 0};
            if (j < NJ) split3_bf16(wr[j], parts);
            const uint16_t v = j < NJ ? (term < 3 ? parts[0] : term < 5 ? parts[1] : parts[2]) : 0;
            wf[u] = __builtin_bit_cast(__bf16, v);
        }
    }
    const int vb = vt * 16 + fq * 4, n3 = fb.V * 3;
    f32x4 vs[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int vi = vb + j < fb.V ? vb + j : fb.V - 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) vs[c][j] = v_shaped[(long long)b * n3 + vi * 3 + c];
    }
    const bool full = vt * 16 + 16 <= fb.V;            // wave-uniform
    // a zero accumulator the compiler cannot fold: with a literal zero it picks the MFMA form whose result overwrites
    // its accumulator operand and clears 4 registers before each of the 12 transform products (48 moves per tile)
    f32x4 zero4;
    asm("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0"
        : "=v"(zero4[0]), "=v"(zero4[1]), "=v"(zero4[2]), "=v"(zero4[3]));
    int buf = 0;
    for (int ti = 0; ti < ntile; ++ti, buf ^= 1) {
        const int f0 = fbeg + ti * 16;
        // my pieces of this tile have landed (only the previous tile's three output stores may still be in flight)
        if (live && full && ti) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ti + 1 < ntile) issue(ti + 1, buf ^ 1);
        if (!live) continue;
        const char* sb = stage + buf * BUF;
        f32x4 acc[3] = {vs[0], vs[1], vs[2]};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int o = (ks * 4 + fq) * 256 + fr * 16;
            const bf16x8 ch = *reinterpret_cast<const bf16x8*>(sb + o);
            const bf16x8 cl = *reinterpret_cast<const bf16x8*>(sb + KS * 1024 + o);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[c][ks], ch, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[c][ks], cl, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[c][ks], ch, acc[c], 0, 0, 0);
            }
        }
        const char* xb = sb + CT + fq * 256 + fr * 16;
        float o[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            f32x4 t[4];                                   // t[e][j]: entry e of row r of the blended transform, vertex j
#pragma unroll
            for (int e = 0; e < 4; ++e)
                t[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, *reinterpret_cast<const bf16x8*>(xb + (r * 4 + e) * 1024),
                                                               zero4, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j * 3 + r] = fmaf(t[2][j], acc[2][j], fmaf(t[1][j], acc[1][j], fmaf(t[0][j], acc[0][j], t[3][j])));
        }
        if (full) {
            float* slab = slabs + wave * (16 * SLAB_ROW);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                *reinterpret_cast<f32x4*>(slab + fr * SLAB_ROW + fq * 12 + 4 * i) =
                    (f32x4){o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 val[3];
            int fl[3], piece[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int cidx = i * 64 + lane;
                fl[i] = cidx / 12;
                piece[i] = cidx - fl[i] * 12;
                val[i] = *reinterpret_cast<const f32x4*>(slab + fl[i] * SLAB_ROW + piece[i] * 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // exactly three store instructions per lane and tile (the vmcnt(3) above counts on it): frames past the
            // clip are redirected to this lane's last valid destination, which receives the same value again
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int ff = f0 + fl[i] < fend ? f0 + fl[i] : -1;
                float* dst = verts + ((long long)(ff >= 0 ? ff : f0) * fb.V + vt * 16) * 3 + piece[i] * 4;
                if (ff < 0) val[i] = *reinterpret_cast<const f32x4*>(slab + 0 * SLAB_ROW + piece[i] * 4);
                *reinterpret_cast<f32x4u*>(dst) = val[i];
            }
        } else if (f0 + fr < fend) {
            float* op = verts + ((long long)(f0 + fr) * fb.V + vb) * 3;
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (vb + i / 3 < fb.V) op[i] = o[i];
        }
    }
}

}  // namespace

extern "C" int avi_flame_vertices(const AviFlameBasis* fbp, const float* shape, const float* exp, const float* pose,
                                  int B, int T, float* v_shaped, float* coef, float* xf, float* verts, void* stream) {
    if (!fbp || !shape || !exp || !pose || !v_shaped || !coef || !xf || !verts || B <= 0 || T <= 0) return AVI_EINVAL;
    const AviFlameBasis& fb = *fbp;
    if (!fb.v_template || !fb.shape_basis || !fb.frame_basis || !fb.j_template || !fb.j_shape || !fb.j_exp ||
        !fb.lbs_weights || fb.V <= 0 || fb.n_shape <= 0 || fb.n_exp <= 0 || fb.n_exp > MAX_EXP)
        return AVI_EINVAL;
    const int K = fb.n_exp + NPF;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ntile = (T + 15) / 16;
    const long long blocks = (long long)B * ntile + (long long)((fb.V * 3 + 31) / 32) * ((B + 31) / 32);
    if (blocks > (1ll << 30)) return AVI_EINVAL;
    const bool mc = fb.basis_hi && fb.basis_lo && K <= 160;
    const int smem = K * 3 * VT * (int)sizeof(float);      // the vector-pipe kernel keeps the whole basis slice in LDS
    if (!mc && smem > 160 * 1024) return AVI_EINVAL;
    const int KSm = K <= 96 ? 3 : 5;
    if (mc && ((reinterpret_cast<uintptr_t>(fb.basis_hi) | reinterpret_cast<uintptr_t>(fb.basis_lo) |
                reinterpret_cast<uintptr_t>(coef) | reinterpret_cast<uintptr_t>(xf)) & 15))
        return AVI_EINVAL;
    static const int only = getenv("AVI_FLAME_ONLY") ? atoi(getenv("AVI_FLAME_ONLY")) : 0;
    hipLaunchKernelGGL(flame_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, s, fb, shape, exp, pose, B, T, v_shaped, coef,
                       xf, mc ? KSm * 32 : 0, only);
    if (mc) {   // matrix-core path
        const int KS = KSm, Vp = (fb.V + 15) / 16 * 16;
        const int nx = (Vp / 16 + 3) / 4;
        const long long total = (long long)nx * B;
        if (total > (1ll << 30)) return AVI_EINVAL;
        const dim3 grid((unsigned)((total + 7) / 8 * 8));
        const char* ct = reinterpret_cast<const char*>(coef);
        const char* xt = reinterpret_cast<const char*>(xf);
        if (KS == 3)
            hipLaunchKernelGGL((flame_vertices_mfma_kernel<3, 3>), grid, dim3(256), 0, s, fb, v_shaped, ct, xt, T, Vp, nx,
                               (int)total, verts);
        else
            hipLaunchKernelGGL((flame_vertices_mfma_kernel<5, 2>), grid, dim3(256), 0, s, fb, v_shaped, ct, xt, T, Vp, nx,
                               (int)total, verts);
        return avi_launch_status();
    }
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(flame_vertices_kernel), 160 * 1024);
    hipLaunchKernelGGL(flame_vertices_kernel, dim3((fb.V + VT - 1) / VT, B), dim3(512), smem, s, fb, v_shaped, coef, xf,
                       T, verts);
    return avi_launch_status();
}

// Split bf16 planes of the per-frame basis for the matrix-core path: hi/lo [3][Vp][KP] uint16, Vp = V rounded up to 16,
// KP = 96 for n_exp + 36 <= 96, else 160.  Run once per model; the caller stores the pointers in AviFlameBasis.
extern "C" int avi_flame_pack_basis(const AviFlameBasis* fbp, uint16_t* hi, uint16_t* lo, void* stream) {
    if (!fbp || !hi || !lo || !fbp->frame_basis || fbp->V <= 0 || fbp->n_exp <= 0) return AVI_EINVAL;
    const int K = fbp->n_exp + NPF;
    if (K > 160) return AVI_EINVAL;
    const int KP = K <= 96 ? 96 : 160, Vp = (fbp->V + 15) / 16 * 16;
    const long long total = 3LL * Vp * KP;
    hipLaunchKernelGGL(flame_pack_basis_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *fbp, Vp, KP, hi, lo);
    return avi_launch_status();
}
