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
// flame_vertices_kernel: block = 128 vertices x all frames of one clip; 512 threads = 128 vertices x 4 frame
// subgroups, 8 frames in flight per thread (24 accumulators); per-frame coefficients are wave-uniform (scalar loads)
// and stored frame-group-major [f/8][k][8] by flame_frame_kernel so that one 32-byte scalar load feeds 8 frames.
#include "common.h"

namespace {

constexpr int NJ = 5, NPF = 36, VT = 128, FG = 8;   // joints, pose features, vertices per block, frames per group

__global__ __launch_bounds__(256) void flame_shape_kernel(const AviFlameBasis fb, const float* __restrict__ shape,
                                                           float* __restrict__ v_shaped) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, n = fb.V * 3;
    if (i >= n) return;
    float a = fb.v_template[i];
    const float* sp = shape + (long long)b * fb.n_shape;
    for (int k = 0; k < fb.n_shape; ++k) a = fmaf(sp[k], fb.shape_basis[(long long)k * n + i], a);
    v_shaped[(long long)b * n + i] = a;
}

// frame record (floats): coefficients live in `coef` [F/8][K][8] (K = n_exp + 36), transforms in `xf` [F][5][12]
__global__ __launch_bounds__(64) void flame_frame_kernel(const AviFlameBasis fb, const float* __restrict__ shape,
                                                          const float* __restrict__ exp, const float* __restrict__ pose,
                                                          int T, int F, float* __restrict__ coef,
                                                          float* __restrict__ xf) {
    __shared__ float J[NJ * 3], R[NJ * 9], A[NJ * 16];
    const int f = blockIdx.x, lane = threadIdx.x, b = f / T;
    const int K = fb.n_exp + NPF;
    const float* ef = exp + (long long)f * fb.n_exp;
    if (lane < NJ * 3) {   // joint coordinate `lane`
        float a = fb.j_template[lane];
        const float* js = fb.j_shape + (long long)lane * fb.n_shape;
        const float* sp = shape + (long long)b * fb.n_shape;
        for (int k = 0; k < fb.n_shape; ++k) a = fmaf(js[k], sp[k], a);
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
    for (int k = lane; k < K; k += 64) {   // coefficients of the 86 basis vectors
        float v;
        if (k < fb.n_exp) v = ef[k];
        else {
            const int q = k - fb.n_exp;   // (R[1 + q/9] - I).flat[q % 9]   (lbs.py:210)
            v = R[9 + q] - ((q % 9) % 4 == 0 ? 1.f : 0.f);
        }
        coef[((long long)(f / FG) * K + k) * FG + (f % FG)] = v;
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
        (void)A;
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

}  // namespace

extern "C" int avi_flame_vertices(const AviFlameBasis* fbp, const float* shape, const float* exp, const float* pose,
                                  int B, int T, float* v_shaped, float* coef, float* xf, float* verts, void* stream) {
    if (!fbp || !shape || !exp || !pose || !v_shaped || !coef || !xf || !verts || B <= 0 || T <= 0) return AVI_EINVAL;
    const AviFlameBasis& fb = *fbp;
    if (!fb.v_template || !fb.shape_basis || !fb.frame_basis || !fb.j_template || !fb.j_shape || !fb.j_exp ||
        !fb.lbs_weights || fb.V <= 0 || fb.n_shape <= 0 || fb.n_exp <= 0)
        return AVI_EINVAL;
    const int K = fb.n_exp + NPF;
    const int smem = K * 3 * VT * (int)sizeof(float);
    if (smem > 160 * 1024) return AVI_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int F = B * T;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(flame_vertices_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(flame_shape_kernel, dim3((fb.V * 3 + 255) / 256, B), dim3(256), 0, s, fb, shape, v_shaped);
    hipLaunchKernelGGL(flame_frame_kernel, dim3(F), dim3(64), 0, s, fb, shape, exp, pose, T, F, coef, xf);
    hipLaunchKernelGGL(flame_vertices_kernel, dim3((fb.V + VT - 1) / VT, B), dim3(512), smem, s, fb, v_shaped, coef, xf,
                       T, verts);
    return avi_launch_status();
}
