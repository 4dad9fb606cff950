// Reference kernel (dev tool, not product code) for the NEXT plane format of the plane-operand GEMMs (DESIGN.md section 3 and
// section 8 "next" (0)):   x = x16 + xr,  w = w16 + wr   (x16 = fp16(x), xr = x - x16; likewise w)
//
//     y  =  x16 . w16                          one fp16 MFMA 16x16x32 per 32 k              (the main term, 2^-11 operands)
//        +  q6(x) . q6(wr)                     one block-scaled e2m3 MFMA 16x16x128 per 128 k (weights' rounding residual)
//        +  q6(xr) . q6(w)                     one block-scaled e2m3 MFMA 16x16x128 per 128 k (activations' rounding residual)
//
// q6 = e2m3 (1 sign, 2 exponent, 3 mantissa bits, bias 1, max 7.5) with one power-of-two scale per (row, 32-k block), the
// operand format of v_mfma_scale_f32_16x16x128_f8f6f4.  Per 32 k that is 18 + 2 x 24 / 4 = 30 issue ticks against 54 for the
// three bf16 MFMAs of today's 3-term mode (scripts/probe/mfma_scale_probe.hip measured the rates and the lane layout), and
// 2 + 2 x 0.78 = 3.6 bytes per element pair instead of 4 + 4.
//
// What this program pins on the hardware, with random data of the encoder's statistics:
//   1. the packing: sign bit, subnormals, negative block exponents, two different scales per instruction (checked against an
//      fp64 evaluation of exactly the quantised operands: must agree to fp32 rounding);
//   2. the accuracy of the format against the exact product, beside today's two formats evaluated the same way
//      (3-term bf16 split; 2-term fp16 activations x 1 fp16 weight plane).
// One wave per 16 x 16 output tile, operands read straight from global memory in fragment order: a REFERENCE, not a fast kernel.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/gemm_f16x6_ref.hip -o gpurun_out/gemm_f16x6_ref && gpurun_out/gemm_f16x6_ref
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// One quantised plane: per (row, 32-k block) 24 bytes of codes (element j at bit 6 j, little endian) + the E8M0 scale in
// byte 24, padded to 32 bytes.
struct Q6Block {
    uint8_t code[24];
    uint8_t scale;
    uint8_t pad[7];
};
static_assert(sizeof(Q6Block) == 32, "block");

__device__ __forceinline__ v8i load_codes(const Q6Block* b) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(b);
    v8i v;
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = (int)p[j];
    v[6] = 0;
    v[7] = 0;
    return v;
}

// grid (N / 16, M / 16), block 64.  X16 [M][K], W16 [N][K] fp16; X6, XR6 [M][K/32], W6, WR6 [N][K/32] blocks; C [M][N].
// terms: bit 0 main, bit 1 q6(x).q6(wr), bit 2 q6(xr).q6(w)
__global__ void gemm_f16x6_ref(const _Float16* __restrict__ X16, const _Float16* __restrict__ W16,
                               const Q6Block* __restrict__ X6, const Q6Block* __restrict__ XR6,
                               const Q6Block* __restrict__ W6, const Q6Block* __restrict__ WR6, float* __restrict__ C, int M,
                               int N, int K, int terms) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int KB = K / 32;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 128) {
        if (terms & 1) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {   // lane (r, g): row r, k = k0 + 32 kk + 8 g .. + 7
                const f16x8 a = *reinterpret_cast<const f16x8*>(X16 + (long long)(m0 + r) * K + k0 + 32 * kk + 8 * g);
                const f16x8 b = *reinterpret_cast<const f16x8*>(W16 + (long long)(n0 + r) * K + k0 + 32 * kk + 8 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            }
        }
        const int kb = k0 / 32 + g;            // lane (r, g): row r, the g-th 32-k block of this 128-k step
        if (terms & 2) {
            const Q6Block* a = X6 + (long long)(m0 + r) * KB + kb;
            const Q6Block* b = WR6 + (long long)(n0 + r) * KB + kb;
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(load_codes(a), load_codes(b), acc, 2, 2, 0, (int)a->scale, 0,
                                                                   (int)b->scale);
        }
        if (terms & 4) {
            const Q6Block* a = XR6 + (long long)(m0 + r) * KB + kb;
            const Q6Block* b = W6 + (long long)(n0 + r) * KB + kb;
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(load_codes(a), load_codes(b), acc, 2, 2, 0, (int)a->scale, 0,
                                                                   (int)b->scale);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) C[(long long)(m0 + 4 * g + j) * N + n0 + r] = acc[j];   // lane holds C[4 g + j][r]
}

// ---------------------------------------------------------------- host: quantisers and references
static double e2m3_value(int code) {           // 6-bit code -> value
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const double v = e == 0 ? m * 0.125 : (1.0 + m / 8.0) * std::ldexp(1.0, e - 1);
    return s ? -v : v;
}

static int e2m3_encode(double v) {             // round to nearest (ties to even mantissa), saturate at 7.5
    static double table[32];
    static bool init = false;
    if (!init) {
        for (int c = 0; c < 32; ++c) table[c] = e2m3_value(c);
        init = true;
    }
    const double a = std::fabs(v);
    int best = 0;
    double bd = 1e300;
    for (int c = 0; c < 32; ++c) {
        const double d = std::fabs(table[c] - a);
        if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; }
    }
    return best | (v < 0 ? 32 : 0);
}

// rows x K values -> blocks [rows][K / 32]; deq receives the values the hardware will see
static void quantise(const std::vector<double>& v, int rows, int K, std::vector<Q6Block>& out, std::vector<double>& deq) {
    out.assign((size_t)rows * (K / 32), Q6Block{});
    deq.assign(v.size(), 0.0);
    for (int r = 0; r < rows; ++r)
        for (int b = 0; b < K / 32; ++b) {
            double amax = 0;
            for (int j = 0; j < 32; ++j) amax = std::fmax(amax, std::fabs(v[(size_t)r * K + 32 * b + j]));
            int s = amax > 0 ? (int)std::ceil(std::log2(amax / 7.5)) : -127;     // smallest power of two with amax / 2^s <= 7.5
            if (s < -127) s = -127;
            Q6Block& q = out[(size_t)r * (K / 32) + b];
            q.scale = (uint8_t)(127 + s);
            for (int j = 0; j < 32; ++j) {
                const int c = e2m3_encode(std::ldexp(v[(size_t)r * K + 32 * b + j], -s));
                const int bo = 6 * j;
                for (int t = 0; t < 6; ++t)
                    if ((c >> t) & 1) q.code[(bo + t) >> 3] |= (uint8_t)(1u << ((bo + t) & 7));
                deq[(size_t)r * K + 32 * b + j] = std::ldexp(e2m3_value(c), s);
            }
        }
}

static float bf16_round(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float y;
    std::memcpy(&y, &u, 4);
    return y;
}

int main(int argc, char** argv) {
    const int M = 256, N = 256;
    const int K = argc > 1 ? atoi(argv[1]) : 768;
    if (K % 128) { printf("K must be a multiple of 128\n"); return 1; }
    std::mt19937 rng(12345);
    std::normal_distribution<double> nd(0.0, 1.0);
    // activations like a LayerNorm output with a few outliers per row (wav2vec2's hidden states have them), weights ~ N(0, 1/K)
    std::vector<double> x((size_t)M * K), w((size_t)N * K);
    for (auto& v : x) v = nd(rng);
    for (int r = 0; r < M; ++r)
        for (int j = 0; j < 3; ++j) x[(size_t)r * K + (rng() % K)] *= 12.0;
    for (auto& v : w) v = nd(rng) / std::sqrt((double)K);
    for (auto& v : x) v = (double)(float)v;     // the operands are fp32 numbers
    for (auto& v : w) v = (double)(float)v;

    std::vector<_Float16> x16(x.size()), w16(w.size());
    std::vector<double> x16d(x.size()), w16d(w.size()), xr(x.size()), wr(w.size());
    for (size_t i = 0; i < x.size(); ++i) { x16[i] = (_Float16)(float)x[i]; x16d[i] = (double)(float)x16[i]; xr[i] = x[i] - x16d[i]; }
    for (size_t i = 0; i < w.size(); ++i) { w16[i] = (_Float16)(float)w[i]; w16d[i] = (double)(float)w16[i]; wr[i] = w[i] - w16d[i]; }
    std::vector<Q6Block> X6, XR6, W6, WR6;
    std::vector<double> x6d, xr6d, w6d, wr6d;
    quantise(x, M, K, X6, x6d);
    quantise(xr, M, K, XR6, xr6d);
    quantise(w, N, K, W6, w6d);
    quantise(wr, N, K, WR6, wr6d);

    _Float16 *dX16, *dW16;
    Q6Block *dX6, *dXR6, *dW6, *dWR6;
    float* dC;
    hipMalloc(&dX16, x16.size() * 2); hipMalloc(&dW16, w16.size() * 2);
    hipMalloc(&dX6, X6.size() * 32); hipMalloc(&dXR6, XR6.size() * 32); hipMalloc(&dW6, W6.size() * 32); hipMalloc(&dWR6, WR6.size() * 32);
    hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dX16, x16.data(), x16.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dX6, X6.data(), X6.size() * 32, hipMemcpyHostToDevice);
    hipMemcpy(dXR6, XR6.data(), XR6.size() * 32, hipMemcpyHostToDevice);
    hipMemcpy(dW6, W6.data(), W6.size() * 32, hipMemcpyHostToDevice);
    hipMemcpy(dWR6, WR6.data(), WR6.size() * 32, hipMemcpyHostToDevice);

    // references in fp64
    std::vector<double> exact((size_t)M * N), main_t((size_t)M * N), c1((size_t)M * N), c2((size_t)M * N), bf3((size_t)M * N),
        f16x2((size_t)M * N);
    std::vector<double> xh(x.size()), xl(x.size()), wh(w.size()), wl(w.size()), xr16(x.size());
    for (size_t i = 0; i < x.size(); ++i) { xh[i] = bf16_round((float)x[i]); xl[i] = bf16_round((float)(x[i] - xh[i])); xr16[i] = (double)(float)(_Float16)(float)xr[i]; }
    for (size_t i = 0; i < w.size(); ++i) { wh[i] = bf16_round((float)w[i]); wl[i] = bf16_round((float)(w[i] - wh[i])); }
    double ymax = 0;
    for (int i = 0; i < M; ++i)
        for (int n = 0; n < N; ++n) {
            double e = 0, a = 0, b = 0, c = 0, d3 = 0, d2 = 0;
            for (int k = 0; k < K; ++k) {
                const size_t xi = (size_t)i * K + k, wi = (size_t)n * K + k;
                e += x[xi] * w[wi];
                a += x16d[xi] * w16d[wi];
                b += x6d[xi] * wr6d[wi];
                c += xr6d[xi] * w6d[wi];
                d3 += xh[xi] * wh[wi] + xh[xi] * wl[wi] + xl[xi] * wh[wi];
                d2 += (x16d[xi] + xr16[xi]) * w16d[wi];
            }
            const size_t o = (size_t)i * N + n;
            exact[o] = e; main_t[o] = a; c1[o] = b; c2[o] = c; bf3[o] = d3; f16x2[o] = d2;
            ymax = std::fmax(ymax, std::fabs(e));
        }

    std::vector<float> hc((size_t)M * N);
    auto run = [&](int terms) {
        hipLaunchKernelGGL(gemm_f16x6_ref, dim3(N / 16, M / 16), dim3(64), 0, 0, dX16, dW16, dX6, dXR6, dW6, dWR6, dC, M, N, K, terms);
        hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost);
    };
    auto maxdiff = [&](const std::vector<double>& ref) {
        double d = 0;
        for (size_t i = 0; i < ref.size(); ++i) d = std::fmax(d, std::fabs((double)hc[i] - ref[i]));
        return d;
    };
    printf("M = N = %d, K = %d; max |y| = %.3f\n", M, K, ymax);
    int fail = 0;
    // 1. each term alone against the fp64 evaluation of exactly its quantised operands
    run(1);
    double d = maxdiff(main_t);
    printf("main term   x16.w16          : max |kernel - fp64 of the same operands| = %.3e (fp32 accumulation of %d products)\n", d, K);
    fail += d > 2e-5 * ymax;
    double t1 = 0, t2 = 0;
    for (size_t i = 0; i < c1.size(); ++i) { t1 = std::fmax(t1, std::fabs(c1[i])); t2 = std::fmax(t2, std::fabs(c2[i])); }
    run(2);
    d = maxdiff(c1);
    printf("cross term  q6(x).q6(wr)     : max |kernel - fp64 of the same operands| = %.3e (term magnitude %.3e)\n", d, t1);
    fail += d > 1e-5 * t1 + 1e-9;
    run(4);
    d = maxdiff(c2);
    printf("cross term  q6(xr).q6(w)     : max |kernel - fp64 of the same operands| = %.3e (term magnitude %.3e)\n", d, t2);
    fail += d > 1e-5 * t2 + 1e-9;
    // 2. the format against the exact product, beside today's formats (their fp64 evaluation: the formats' own error)
    run(7);
    const double e_new = maxdiff(exact);
    double e_main = 0, e_bf3 = 0, e_f2 = 0, e_new64 = 0;
    for (size_t i = 0; i < exact.size(); ++i) {
        e_main = std::fmax(e_main, std::fabs(main_t[i] - exact[i]));
        e_bf3 = std::fmax(e_bf3, std::fabs(bf3[i] - exact[i]));
        e_f2 = std::fmax(e_f2, std::fabs(f16x2[i] - exact[i]));
        e_new64 = std::fmax(e_new64, std::fabs(main_t[i] + c1[i] + c2[i] - exact[i]));
    }
    printf("against the exact product (max abs over %d outputs, |y| <= %.2f):\n", M * N, ymax);
    printf("  fp16 main term alone                          %.3e\n", e_main);
    printf("  2-term fp16 (x16 + xr16) . w16   [conv today] %.3e\n", e_f2);
    printf("  fp16 + two e2m3 cross terms, kernel           %.3e   (its fp64 evaluation: %.3e)\n", e_new, e_new64);
    printf("  3-term bf16 split                [rest today] %.3e\n", e_bf3);
    printf("%s\n", fail ? "FAILED" : "OK");
    return fail ? 1 : 0;
}
