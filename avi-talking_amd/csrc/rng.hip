// Counter-based random draws for the captured passes: Philox-4x32-10 (Salmon et al., SC'11) addressed by
// (seed, step offset, subsequence, element), seed and offset read from DEVICE memory so that a replayed hipGraph draws fresh
// numbers every pass (avi_rng_advance, the graph's last node, bumps the offset).
//
// Replaces the draws the reference makes inside its steps with torch generators:
//   * sampling: x_T and one noise tensor per DDPM step (models/diffusion_prior.py:337,349-351);
//   * training: timesteps, q_sample noise, the two cond-drop masks (models/diffusion_prior.py:445,453,255-259 via
//     train_diffusion_prior.py:449) and BrainNetwork's dropout masks (models/diffusion_prior.py:62-75).
// The numbers are not torch's (its offset bookkeeping is no contract: its CPU and CUDA generators already disagree for one
// seed); parity runs keep injecting recorded tensors.  Oracle: oracle/rng.py (pinned by the Random123 known-answer vectors).
//
// HBM-bound streaming kernel: one Philox block (4 words) per thread per iteration, 16-byte stores.
#include "common.h"

namespace {

struct Words { uint32_t w[4]; };

__device__ __forceinline__ Words philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        if (r < 9) { k0 += W0; k1 += W1; }
    }
    return Words{{c0, c1, c2, c3}};
}

__device__ __forceinline__ float u24(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-8f; }   // [0, 1)

// kinds: include/avi_talking.h AVI_RNG_*
template <int KIND>
__global__ __launch_bounds__(256) void rng_fill_kernel(const unsigned long long* __restrict__ state, uint32_t subseq,
                                                        float param, long long n, void* __restrict__ out) {
    const unsigned long long seed = state[0], offset = state[1];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    const long long nblk = (n + 3) >> 2;
    for (long long b = blockIdx.x * (long long)blockDim.x + threadIdx.x; b < nblk; b += (long long)gridDim.x * blockDim.x) {
        const Words r = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32) | (subseq << 16), c2, c3, k0, k1);
        const long long i = b << 2;
        const int cnt = (int)((n - i) < 4 ? (n - i) : 4);
        if (KIND == AVI_RNG_RAW || KIND == AVI_RNG_RANDINT_I32) {
            uint32_t v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = KIND == AVI_RNG_RAW ? r.w[j] : (uint32_t)(((unsigned long long)r.w[j] * (unsigned long long)(uint32_t)param) >> 32);
            uint32_t* o = reinterpret_cast<uint32_t*>(out) + i;
            if (cnt == 4) *reinterpret_cast<uint4*>(o) = make_uint4(v[0], v[1], v[2], v[3]);
            else for (int j = 0; j < cnt; ++j) o[j] = v[j];
        } else if (KIND == AVI_RNG_BERNOULLI_U8) {
            uint8_t* o = reinterpret_cast<uint8_t*>(out) + i;
            uint32_t pack = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) pack |= (uint32_t)(u24(r.w[j]) < param) << (8 * j);
            if (cnt == 4) *reinterpret_cast<uint32_t*>(o) = pack;
            else for (int j = 0; j < cnt; ++j) o[j] = (uint8_t)(pack >> (8 * j));
        } else {
            float v[4];
            if (KIND == AVI_RNG_NORMAL) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {   // Box-Muller: u1 in (0, 1], u2 in [0, 1)
                    const float u1 = (float)((r.w[2 * p] >> 8) + 1u) * 5.9604644775390625e-8f;
                    const float rad = sqrtf(-2.f * logf(u1));
                    float s, c;
                    sincospif(2.f * u24(r.w[2 * p + 1]), &s, &c);
                    v[2 * p] = rad * c;
                    v[2 * p + 1] = rad * s;
                }
            } else {
                const float scale = 1.f / (1.f - param);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float u = u24(r.w[j]);
                    v[j] = KIND == AVI_RNG_UNIFORM ? u : (u >= param ? scale : 0.f);
                }
            }
            float* o = reinterpret_cast<float*>(out) + i;
            if (cnt == 4) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int j = 0; j < cnt; ++j) o[j] = v[j];
        }
    }
}

__global__ void rng_advance_kernel(unsigned long long* state, unsigned long long delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += delta;
}

}  // namespace

extern "C" int avi_rng_fill(const unsigned long long* state, unsigned subsequence, int kind, float param, long long n,
                            void* out, void* stream) {
    if (!state || !out || n <= 0 || subsequence > 0xffffu || (reinterpret_cast<uintptr_t>(state) & 7)) return AVI_EINVAL;
    const int esz = kind == AVI_RNG_BERNOULLI_U8 ? 1 : 4;
    if (reinterpret_cast<uintptr_t>(out) & (4 * esz - 1)) return AVI_EINVAL;       // the 4-element vector store
    if ((kind == AVI_RNG_KEEP_SCALED || kind == AVI_RNG_BERNOULLI_U8) && !(param >= 0.f && param <= 1.f)) return AVI_EINVAL;
    if (kind == AVI_RNG_KEEP_SCALED && param >= 1.f) return AVI_EINVAL;
    if (kind == AVI_RNG_RANDINT_I32 && !(param >= 1.f && param <= 16777216.f)) return AVI_EINVAL;
    const long long nblk = (n + 3) >> 2;
    long long g = (nblk + 255) / 256;
    const dim3 grid((unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g))), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define AVI_RNG_CASE(K) \
    case K: hipLaunchKernelGGL(rng_fill_kernel<K>, grid, block, 0, s, state, (uint32_t)subsequence, param, n, out); break;
    switch (kind) {
        AVI_RNG_CASE(AVI_RNG_RAW)
        AVI_RNG_CASE(AVI_RNG_NORMAL)
        AVI_RNG_CASE(AVI_RNG_KEEP_SCALED)
        AVI_RNG_CASE(AVI_RNG_BERNOULLI_U8)
        AVI_RNG_CASE(AVI_RNG_RANDINT_I32)
        AVI_RNG_CASE(AVI_RNG_UNIFORM)
        default: return AVI_EINVAL;
    }
#undef AVI_RNG_CASE
    return avi_launch_status();
}

extern "C" int avi_rng_advance(unsigned long long* state, unsigned long long delta, void* stream) {
    if (!state || (reinterpret_cast<uintptr_t>(state) & 7)) return AVI_EINVAL;
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state, delta);
    return avi_launch_status();
}
