// Probe: how fast can 256 workgroups stream a [4096][4096] bf16 plane pair (64 MB) when every lane reads 16 bytes
//   0  as the MFMA operand of a row-major plane: lane (row n = lane & 15, k group lane >> 4): 16 x 64-B segments per instruction
//   1  as 1 KB contiguous per instruction (fragment-major planes)
//   2  as 0, but 128 B per row and instruction (lanes 0..7 of a row group read consecutive 16 B): 8 rows x 128 B
// (hipcc --offload-arch=gfx950 -O3 scripts/probe/wstream_probe.hip -o /tmp/wsp && /tmp/wsp)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096, K = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned short* __restrict__ hi, const unsigned short* __restrict__ lo,
                                                     unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, z = blockIdx.y;
    const int n0 = blockIdx.x * 64 + wave * 16;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[64];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        long long oh;
        if (MODE == 0) oh = (long long)(n0 + (lane & 15)) * K + z * 1024 + ks * 32 + 8 * (lane >> 4);
        else if (MODE == 1) oh = ((long long)(blockIdx.x * 4 + wave) * 128 + z * 32 + ks) * 512 + lane * 8;
        else oh = (long long)(n0 + (ks & 1) * 8 + (lane >> 3)) * K + z * 1024 + (ks >> 1) * 64 + 8 * (lane & 7);
        r[2 * ks] = *reinterpret_cast<const u32x4*>(hi + oh);
        r[2 * ks + 1] = *reinterpret_cast<const u32x4*>(lo + oh);
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) acc += r[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) out[0] = 1;
}
int main() {
    unsigned short *hi, *lo;
    unsigned* out;
    (void)hipMalloc(&hi, (size_t)N * K * 2);
    (void)hipMalloc(&lo, (size_t)N * K * 2);
    (void)hipMalloc(&out, 64);
    (void)hipMemset(hi, 1, (size_t)N * K * 2);
    (void)hipMemset(lo, 2, (size_t)N * K * 2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(64, 4), dim3(256), 0, 0, hi, lo, out);
            if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(64, 4), dim3(256), 0, 0, hi, lo, out);
            if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(64, 4), dim3(256), 0, 0, hi, lo, out);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("mode %d: %.1f us  (%.2f TB/s of 67 MB)\n", mode, best * 1e3, 2.0 * N * K * 2 / best / 1e9);
    }
    return 0;
}
