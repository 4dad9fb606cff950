// Probe (dev tool, not product code): operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950,
// checked with exactly representable integer data against a CPU reference.  The next precision format (DESIGN.md section 3:
// fp16 main term + block-scaled fp6 cross terms) is built on this instruction; the guides say "check the map with exact
// integer data before relying on it".
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/mfma_scale_probe.hip -o gpurun_out/mfma_scale_probe && gpurun_out/mfma_scale_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

// FMT: 0 = fp8 e4m3, 2 = fp6 e2m3, 4 = fp4 e2m1 (cbsz / blgp codes)
template <int FMT>
__global__ void probe(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ sa,
                      const uint32_t* __restrict__ sb, float* __restrict__ c) {
    const int lane = threadIdx.x;
    v8i A, B;
    for (int j = 0; j < 8; ++j) {
        A[j] = (int)a[lane * 8 + j];
        B[j] = (int)b[lane * 8 + j];
    }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, FMT, FMT, 0, (int)sa[lane], 0, (int)sb[lane]);
    for (int j = 0; j < 4; ++j) c[lane * 4 + j] = acc[j];
}

// issue rate: cycles per instruction with 4 independent accumulators, one wave
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
template <int FMT>
__global__ void rate(float* out, unsigned long long* cyc, int iters) {
    v8i A, B;
    for (int j = 0; j < 8; ++j) { A[j] = 0x3c3c3c3c + threadIdx.x; B[j] = 0x38383838 + j; }
    v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8_t a16, b16;
    for (int j = 0; j < 8; ++j) { a16[j] = (__bf16)(1.0f + j); b16[j] = (__bf16)(0.5f); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (FMT < 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16, b16, acc[k], 0, 0, 0);
            else acc[k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[k], FMT, FMT, 0, 127, 0, 127);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

static uint32_t enc(int n, int fmt) {   // small non-negative integers, exactly representable
    if (n == 0) return 0;
    int e = 0;
    while ((1 << (e + 1)) <= n) ++e;                 // n in [2^e, 2^(e+1))
    if (fmt == 0) {                                  // e4m3: bias 7, 3 mantissa bits
        const int m = (n - (1 << e)) * 8 / (1 << e);
        return (uint32_t)(((e + 7) << 3) | m);
    }
    if (fmt == 2) {                                  // e2m3: bias 1
        const int m = (n - (1 << e)) * 8 / (1 << e);
        return (uint32_t)(((e + 1) << 3) | m);
    }
    const int m = (n - (1 << e)) * 2 / (1 << e);     // e2m1: bias 1
    return (uint32_t)(((e + 1) << 1) | m);
}

int main() {
    const int fmts[3] = {0, 2, 4}, bits[3] = {8, 6, 4}, maxv[3] = {15, 7, 3};
    const char* names[3] = {"fp8 e4m3", "fp6 e2m3", "fp4 e2m1"};
    uint32_t *da, *db, *dsa, *dsb;
    float* dc;
    hipMalloc(&da, 64 * 8 * 4); hipMalloc(&db, 64 * 8 * 4); hipMalloc(&dsa, 64 * 4); hipMalloc(&dsb, 64 * 4);
    hipMalloc(&dc, 64 * 4 * 4);
    for (int f = 0; f < 3; ++f) {
        // logical matrices: A[16][128] (row i, k), B[16][128] (column n, k): C[i][n] = sum_k A[i][k] B[n][k] * 2^(sA[i][k/32]) * 2^(sB[n][k/32])
        std::vector<int> A(16 * 128), B(16 * 128), SA(16 * 4), SB(16 * 4);
        srand(7 + f);
        for (auto& v : A) v = rand() % (maxv[f] + 1);
        for (auto& v : B) v = rand() % (maxv[f] + 1);
        for (auto& v : SA) v = rand() % 4 - 1;       // exponents -1 .. 2
        for (auto& v : SB) v = rand() % 3 - 1;
        // hypothesis H: lane l = (r = l & 15, g = l >> 4) holds row r, k = 32 g .. 32 g + 31, element j at bit offset j * bits
        //               of its 256-bit register group (little endian); its scale byte 0 = E8M0 of (row r, block g)
        std::vector<uint32_t> ha(64 * 8, 0), hb(64 * 8, 0), hsa(64), hsb(64);
        for (int l = 0; l < 64; ++l) {
            const int r = l & 15, g = l >> 4;
            for (int j = 0; j < 32; ++j) {
                const uint64_t ca = enc(A[r * 128 + 32 * g + j], fmts[f]), cb = enc(B[r * 128 + 32 * g + j], fmts[f]);
                const int bo = j * bits[f];
                for (int t = 0; t < bits[f]; ++t) {
                    if ((ca >> t) & 1) ha[l * 8 + (bo + t) / 32] |= 1u << ((bo + t) % 32);
                    if ((cb >> t) & 1) hb[l * 8 + (bo + t) / 32] |= 1u << ((bo + t) % 32);
                }
            }
            hsa[l] = (uint32_t)(127 + SA[r * 4 + g]);
            hsb[l] = (uint32_t)(127 + SB[r * 4 + g]);
        }
        hipMemcpy(da, ha.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa.data(), 64 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dsb, hsb.data(), 64 * 4, hipMemcpyHostToDevice);
        if (f == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        if (f == 1) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        if (f == 2) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        std::vector<float> hc(64 * 4);
        hipMemcpy(hc.data(), dc, 64 * 4 * 4, hipMemcpyDeviceToHost);
        // C/D layout (dtype independent, cdna guide): lane l holds C[row = 4 (l >> 4) + j][col = l & 15]
        int bad = 0, bad_t = 0;
        double worst = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int row = 4 * (l >> 4) + j, col = l & 15;
                double ref = 0, ref_t = 0;
                for (int k = 0; k < 128; ++k) {
                    ref += (double)A[row * 128 + k] * B[col * 128 + k] * std::ldexp(1.0, SA[row * 4 + k / 32] + SB[col * 4 + k / 32]);
                    ref_t += (double)A[col * 128 + k] * B[row * 128 + k] * std::ldexp(1.0, SA[col * 4 + k / 32] + SB[row * 4 + k / 32]);
                }
                if (std::fabs(hc[l * 4 + j] - ref) > 1e-3 * (1 + std::fabs(ref))) ++bad;
                if (std::fabs(hc[l * 4 + j] - ref_t) > 1e-3 * (1 + std::fabs(ref_t))) ++bad_t;
                worst = std::fmax(worst, std::fabs(hc[l * 4 + j] - ref));
            }
        printf("%s: hypothesis (lane = row l&15, k block l>>4, element j at bit j*%d, scale byte 0 per (row, block)): "
               "%d of 256 outputs differ with C[row = 4(l>>4)+j][col = l&15] = A.B^T, %d with the transposed reading; c[0..3] = %g %g %g %g\n",
               names[f], bits[f], bad, bad_t, hc[0], hc[1], hc[2], hc[3]);
    }
    // ---- issue rate (s_memtime ticks per instruction; the ratio to the bf16 16x16x32 form is what matters)
    unsigned long long* dcyc;
    hipMalloc(&dcyc, 8);
    const int iters = 20000;
    unsigned long long ticks[4];
    hipLaunchKernelGGL(rate<-1>, dim3(1), dim3(64), 0, 0, dc, dcyc, iters); hipMemcpy(&ticks[0], dcyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, dc, dcyc, iters); hipMemcpy(&ticks[1], dcyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(rate<2>, dim3(1), dim3(64), 0, 0, dc, dcyc, iters); hipMemcpy(&ticks[2], dcyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(rate<4>, dim3(1), dim3(64), 0, 0, dc, dcyc, iters); hipMemcpy(&ticks[3], dcyc, 8, hipMemcpyDeviceToHost);
    const char* rn[4] = {"bf16 16x16x32", "scaled fp8 16x16x128", "scaled fp6 16x16x128", "scaled fp4 16x16x128"};
    for (int i = 0; i < 4; ++i)
        printf("%-22s %8.2f ticks per instruction = %.2f x the bf16 form (K per instruction: %d)\n", rn[i],
               (double)ticks[i] / (4.0 * iters), (double)ticks[i] / (double)ticks[0], i == 0 ? 32 : 128);
    return 0;
}
