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
#include "common.h"

namespace {

constexpr int NJ = 5, NPF = 36, VT = 128, FG = 8;   // joints, pose features, vertices per block, frames per group

constexpr int SC = 4;   // clips per thread of the shape pass: one basis load feeds SC accumulators

// grid (ceil(V*3/256), ceil(B/SC)).  The k loop is unrolled by 4 so that four basis loads are in flight per thread.
__global__ __launch_bounds__(256) void flame_shape_kernel(const AviFlameBasis fb, const float* __restrict__ shape, int B,
                                                           float* __restrict__ v_shaped) {
    const int b0 = blockIdx.y * SC, i = blockIdx.x * blockDim.x + threadIdx.x, n = fb.V * 3;
    if (i >= n) return;
    const float* sp[SC];     // wave-uniform rows of `shape` (scalar loads); clips past B repeat the last one
#pragma unroll
    for (int c = 0; c < SC; ++c) sp[c] = shape + (long long)(b0 + c < B ? b0 + c : B - 1) * fb.n_shape;
    float a[SC];
    const float t = fb.v_template[i];
#pragma unroll
    for (int c = 0; c < SC; ++c) a[c] = t;
    const float* bp = fb.shape_basis + i;
    int k = 0;
    for (; k + 4 <= fb.n_shape; k += 4) {
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = bp[(long long)(k + u) * n];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < SC; ++c) a[c] = fmaf(sp[c][k + u], e[u], a[c]);
    }
    for (; k < fb.n_shape; ++k) {
        const float e = bp[(long long)k * n];
#pragma unroll
        for (int c = 0; c < SC; ++c) a[c] = fmaf(sp[c][k], e, a[c]);
    }
#pragma unroll
    for (int c = 0; c < SC; ++c)
        if (b0 + c < B) v_shaped[(long long)(b0 + c) * n + i] = a[c];
}

// jclip[b][15] = j_template + j_shape . shape[b]: the per-clip part of the joint regression, hoisted out of the frames.
// One wave per clip; lanes split k.
__global__ __launch_bounds__(64) void flame_joints_kernel(const AviFlameBasis fb, const float* __restrict__ shape,
                                                           float* __restrict__ jclip) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* sp = shape + (long long)b * fb.n_shape;
    float a[NJ * 3];
#pragma unroll
    for (int j = 0; j < NJ * 3; ++j) a[j] = 0.f;
    for (int k = lane; k < fb.n_shape; k += 64) {
        const float x = sp[k];
#pragma unroll
        for (int j = 0; j < NJ * 3; ++j) a[j] = fmaf(fb.j_shape[(long long)j * fb.n_shape + k], x, a[j]);
    }
#pragma unroll
    for (int j = 0; j < NJ * 3; ++j) {
        const float v = wave_sum_u(a[j]);
        if (lane == 0) jclip[b * 16 + j] = fb.j_template[j] + v;
    }
}

// frame record (floats): coefficients live in `coef` [F/8][K][8] (K = n_exp + 36), transforms in `xf` [F][5][12]
// one wave per frame, four frames per block
__global__ __launch_bounds__(256) void flame_frame_kernel(const AviFlameBasis fb, const float* __restrict__ jclip,
                                                           const float* __restrict__ exp, const float* __restrict__ pose,
                                                           int T, int F, float* __restrict__ coef,
                                                           float* __restrict__ xf, int KP) {
    __shared__ float Js[4][NJ * 3], Rs[4][NJ * 9];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + wv < F ? blockIdx.x * 4 + wv : F - 1;   // a spare wave repeats the last frame
    const int b = f / T;
    float* J = Js[wv];
    float* R = Rs[wv];
    const int K = fb.n_exp + NPF;
    const float* ef = exp + (long long)f * fb.n_exp;
    if (lane < NJ * 3) {   // joint coordinate `lane`
        float a = jclip[b * 16 + lane];
        const float* je = fb.j_exp + (long long)lane * fb.n_exp;
        for (int k = 0; k < fb.n_exp; ++k) a = fmaf(je[k], ef[k], a);
        J[lane] = a;
    }
    if (lane < NJ) {       // lbs.py:304-335
        const float* p = pose + (long long)f * (NJ * 3) + lane * 3;
        const float x = p[0], y = p[1], z = p[2];
        const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rx = x / angle, ry = y / angle, rz = z / angle;
        const float s = sinf(angle), c1 = 1.f - cosf(angle);
        // K = [[0,-rz,ry],[rz,0,-rx],[-ry,rx,0]];  R = I + s K + (1-c) K.K
        const float kk[9] = {-(ry * ry + rz * rz), rx * ry, rx * rz, rx * ry, -(rx * rx + rz * rz), ry * rz,
                             rx * rz, ry * rz, -(rx * rx + ry * ry)};
        const float k1[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) R[lane * 9 + i] = ((i % 4 == 0) ? 1.f : 0.f) + s * k1[i] + c1 * kk[i];
    }
    __syncthreads();
    // coefficients of the 86 basis vectors: fp32 frame-group-major for the vector-pipe kernel (KP == 0), or split
    // bf16 planes [F][KP] hi | [F][KP] lo (zero beyond K) for the matrix-core kernel
    uint16_t* chi = reinterpret_cast<uint16_t*>(coef);
    uint16_t* clo = chi + (long long)F * KP;
    for (int k = lane; k < (KP ? KP : K); k += 64) {
        float v = 0.f;
        if (k < fb.n_exp) v = ef[k];
        else if (k < K) {
            const int q = k - fb.n_exp;   // (R[1 + q/9] - I).flat[q % 9]   (lbs.py:210)
            v = R[9 + q] - ((q % 9) % 4 == 0 ? 1.f : 0.f);
        }
        if (KP) {
            const __bf16 h = (__bf16)v;
            const __bf16 l = (__bf16)(v - (float)h);
            chi[(long long)f * KP + k] = __builtin_bit_cast(uint16_t, h);
            clo[(long long)f * KP + k] = __builtin_bit_cast(uint16_t, l);
        } else {
            coef[((long long)(f / FG) * K + k) * FG + (f % FG)] = v;
        }
    }
    if (lane == 0) {       // lbs.py:351-410, parents = [-1, 0, 1, 1, 1]
        float G[NJ][12];   // rows 0..2 of the chained transforms
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int par = j == 0 ? -1 : (j == 1 ? 0 : 1);
            float t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = J[j * 3 + c] - (par >= 0 ? J[par * 3 + c] : 0.f);
            if (par < 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) G[j][r * 4 + c] = R[j * 9 + r * 3 + c];
                    G[j][r * 4 + 3] = t[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        G[j][r * 4 + c] = G[par][r * 4 + 0] * R[j * 9 + c] + G[par][r * 4 + 1] * R[j * 9 + 3 + c] +
                                          G[par][r * 4 + 2] * R[j * 9 + 6 + c];
                    G[j][r * 4 + 3] = G[par][r * 4 + 0] * t[0] + G[par][r * 4 + 1] * t[1] + G[par][r * 4 + 2] * t[2] +
                                      G[par][r * 4 + 3];
                }
            }
        }
        // relative to the rest pose: translation -= G[:3,:3] . joint
        float* o = xf + (long long)f * (NJ * 12);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                o[j * 12 + r * 4 + 0] = G[j][r * 4 + 0];
                o[j * 12 + r * 4 + 1] = G[j][r * 4 + 1];
                o[j * 12 + r * 4 + 2] = G[j][r * 4 + 2];
                o[j * 12 + r * 4 + 3] = G[j][r * 4 + 3] - (G[j][r * 4 + 0] * J[j * 3] + G[j][r * 4 + 1] * J[j * 3 + 1] +
                                                           G[j][r * 4 + 2] * J[j * 3 + 2]);
            }
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
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 48-byte runs of the output are dword aligned

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

// grid (ceil(V/64), B), block 256: wave w owns vertices [vt*16, vt*16+16), vt = 4*blockIdx.x + w, and walks the frames of
// clip blockIdx.y in tiles of 16.  MFMA operand roles as in gemm.hip: first operand = rows of the "n" matrix (vertices),
// second = rows of the "m" matrix (frames); lane (fr, fq) receives D[frame f0+fr][vertex vt*16 + fq*4 + 0..3] per
// coordinate plane.  The per-tile operands of the four waves (they work on the SAME 16 frames of the clip) are
// fetched once per workgroup by LDS-DMA, one frame tile ahead: the coefficient planes (2 x 32 KP bytes) and the 16 x 240 B
// of transforms of tile t+1 are requested right after the barrier that opens tile t and land under its MFMAs, skinning
// and stores (with per-wave global loads instead, every wave waited out the L2 latency twice per tile: 329 -> 265 us).
//   vmcnt: the DMA requests of a wave are older than the three output stores of the tile, so `s_waitcnt vmcnt(3)` at
//   the top of the next tile means "my DMA pieces have landed"; the barrier that follows makes that true for all waves
//   and also orders the previous tile's LDS reads before the buffer is refilled (two buffers, one barrier per tile).
typedef __attribute__((address_space(3))) void flame_lds_void;
typedef const __attribute__((address_space(1))) void flame_gbl_void;

template <int KS>
__global__ __launch_bounds__(256, KS == 3 ? 2 : 1) void flame_vertices_mfma_kernel(const AviFlameBasis fb,
                                                                   const float* __restrict__ v_shaped,
                                                                   const uint16_t* __restrict__ chi,
                                                                   const uint16_t* __restrict__ clo,
                                                                   const float* __restrict__ xf, int T, int Vp,
                                                                   float* __restrict__ verts) {
    constexpr int KP = KS * 32, SLAB_ROW = 52;
    constexpr int CROW = KP * 2;                       // bytes of one frame's coefficients in a plane (192 / 320)
    constexpr int CPL = 16 * CROW;                     // one plane of a 16-frame tile = KS KiB
    constexpr int XROW = NJ * 12 * 4;                  // 240 B of transforms per frame
    constexpr int BUF = 2 * CPL + 4096;                // [coef hi | coef lo | transforms (3840 B used)]
    constexpr int NCH = 2 * KS + 4;                    // 1-KiB DMA pieces per tile
    __shared__ __attribute__((aligned(16))) char stage[2 * BUF];
    __shared__ __attribute__((aligned(16))) float slabs[4 * 16 * SLAB_ROW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.y;
    const int vt_raw = blockIdx.x * 4 + wave;
    const bool live = vt_raw * 16 < fb.V;              // wave-uniform; a spare wave still fetches and synchronises
    const int vt = live ? vt_raw : 0;
    const int fbeg = b * T, fend = fbeg + T;

    // piece c of a tile: 64 lanes x 16 B of [coef hi | coef lo | transforms]; rows past the clip re-read its last frame
    auto issue = [&](int f0, int buf) __attribute__((always_inline)) {
        for (int c = wave; c < NCH; c += 4) {
            const char* base;
            int row_bytes, o;
            if (c < 2 * KS) {
                base = reinterpret_cast<const char*>(c < KS ? chi : clo);
                row_bytes = CROW;
                o = (c < KS ? c : c - KS) * 1024 + lane * 16;
            } else {
                base = reinterpret_cast<const char*>(xf);
                row_bytes = XROW;
                o = (c - 2 * KS) * 1024 + lane * 16;
            }
            const int rf = o / row_bytes, within = o - rf * row_bytes;
            const int f = f0 + rf < fend ? f0 + rf : fend - 1;
            const char* src = base + (long long)f * row_bytes + within;
            char* dst = stage + buf * BUF + (c < 2 * KS ? c * 1024 : 2 * CPL + (c - 2 * KS) * 1024);
            __builtin_amdgcn_global_load_lds((flame_gbl_void*)src, (flame_lds_void*)dst, 16, 0, 0);
        }
    };
    issue(fbeg, 0);

    bf16x8 bh[3][KS], bl[3][KS];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const long long o = ((long long)c * Vp + vt * 16 + fr) * KP + ks * 32 + fq * 8;
            bh[c][ks] = *reinterpret_cast<const bf16x8*>(fb.basis_hi + o);
            bl[c][ks] = *reinterpret_cast<const bf16x8*>(fb.basis_lo + o);
        }
    const int vb = vt * 16 + fq * 4, n3 = fb.V * 3;
    float w[4][NJ], vs[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int vi = vb + j < fb.V ? vb + j : fb.V - 1;
#pragma unroll
        for (int q = 0; q < NJ; ++q) w[j][q] = fb.lbs_weights[(long long)vi * NJ + q];
#pragma unroll
        for (int c = 0; c < 3; ++c) vs[j][c] = v_shaped[(long long)b * n3 + vi * 3 + c];
    }
    const bool full = vt * 16 + 16 <= fb.V;            // wave-uniform
    int buf = 0;
    for (int f0 = fbeg; f0 < fend; f0 += 16, buf ^= 1) {
        // my pieces of this tile have landed (only the previous tile's three output stores may still be in flight)
        if (live && full && f0 != fbeg) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (f0 + 16 < fend) issue(f0 + 16, buf ^ 1);
        if (!live) continue;
        const char* sb = stage + buf * BUF;
        f32x4 acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int o = fr * CROW + (ks * 4 + fq) * 16;
            const bf16x8 ch = *reinterpret_cast<const bf16x8*>(sb + o);
            const bf16x8 cl = *reinterpret_cast<const bf16x8*>(sb + CPL + o);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[c][ks], ch, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[c][ks], cl, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[c][ks], ch, acc[c], 0, 0, 0);
            }
        }
        const f32x4* ap = reinterpret_cast<const f32x4*>(sb + 2 * CPL + fr * XROW);
        float px[4], py[4], pz[4], o[12];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            px[j] = vs[j][0] + acc[0][j];
            py[j] = vs[j][1] + acc[1][j];
            pz[j] = vs[j][2] + acc[2][j];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            f32x4 a[NJ];
#pragma unroll
            for (int q = 0; q < NJ; ++q) a[q] = ap[q * 3 + r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 t = a[0] * w[j][0];
#pragma unroll
                for (int q = 1; q < NJ; ++q) t += a[q] * w[j][q];
                o[j * 3 + r] = t[0] * px[j] + t[1] * py[j] + t[2] * pz[j] + t[3];
            }
        }
        const int f = f0 + fr;
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
        } else if (f < fend) {
            float* op = verts + ((long long)f * fb.V + vb) * 3;
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
        !fb.lbs_weights || fb.V <= 0 || fb.n_shape <= 0 || fb.n_exp <= 0)
        return AVI_EINVAL;
    const int K = fb.n_exp + NPF;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int F = B * T;
    float* jclip = xf + (long long)F * (NJ * 12);   // [B][16] behind the transforms
    hipLaunchKernelGGL(flame_shape_kernel, dim3((fb.V * 3 + 255) / 256, (B + SC - 1) / SC), dim3(256), 0, s, fb, shape, B,
                       v_shaped);
    hipLaunchKernelGGL(flame_joints_kernel, dim3(B), dim3(64), 0, s, fb, shape, jclip);
    if (fb.basis_hi && fb.basis_lo && K <= 160) {   // matrix-core path
        if ((reinterpret_cast<uintptr_t>(fb.basis_hi) | reinterpret_cast<uintptr_t>(fb.basis_lo) |
             reinterpret_cast<uintptr_t>(coef) | reinterpret_cast<uintptr_t>(xf)) & 15)
            return AVI_EINVAL;
        const int KS = K <= 96 ? 3 : 5, KP = KS * 32, Vp = (fb.V + 15) / 16 * 16;
        hipLaunchKernelGGL(flame_frame_kernel, dim3((F + 3) / 4), dim3(256), 0, s, fb, jclip, exp, pose, T, F, coef, xf, KP);
        const uint16_t* chi = reinterpret_cast<const uint16_t*>(coef);
        const uint16_t* clo = chi + (long long)F * KP;
        const dim3 grid((Vp / 16 + 3) / 4, B);
        if (KS == 3)
            hipLaunchKernelGGL(flame_vertices_mfma_kernel<3>, grid, dim3(256), 0, s, fb, v_shaped, chi, clo, xf, T, Vp, verts);
        else
            hipLaunchKernelGGL(flame_vertices_mfma_kernel<5>, grid, dim3(256), 0, s, fb, v_shaped, chi, clo, xf, T, Vp, verts);
        return avi_launch_status();
    }
    const int smem = K * 3 * VT * (int)sizeof(float);
    if (smem > 160 * 1024) return AVI_EINVAL;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(flame_vertices_kernel), 160 * 1024);
    hipLaunchKernelGGL(flame_frame_kernel, dim3((F + 3) / 4), dim3(256), 0, s, fb, jclip, exp, pose, T, F, coef, xf, 0);
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
