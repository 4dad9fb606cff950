// Probe: what does the STORE pattern of flame_vertices_mfma_kernel cost by itself?  (hipcc --offload-arch=gfx950 -O3 ...)
// Output [F = 8000][V = 5023][3] floats (row = 60 276 B: 4-byte aligned only).  A wave owns 16 vertices (192 B per frame)
// and walks 250 frames of a clip in tiles of 16; variants differ in how those 16 x 192 B leave the wave.
//   0  as the kernel: 3 x dwordx4 per lane, 12 consecutive lanes cover one frame's 192 B (slab order)
//   1  accumulator order: lane (fr, fq) writes its own 48 B as 3 x dwordx4
//   2  slab order, but the row pitch padded to 60 288 B and runs 64-B aligned (what alignment alone is worth)
//   3  workgroup order: the 4 waves' 768 B per frame written by 48 consecutive lanes (4 frames per instruction)
//   4  a plain streaming fill of the same bytes (ceiling)
//   5  as 0, with the workgroups renumbered so that neighbouring vertex tiles of a clip run on the SAME XCD (workgroup
//      ids are dealt round-robin to the 8 XCDs, each with its own L2: the cache lines two neighbours share are otherwise
//      written partially from two L2s)
//   6  as 3, renumbered the same way
//   7  as 6, with every dwordx4 store 16-B aligned in the OUTPUT: per frame the run's first (4 - offset % 4) % 4 floats
//      and what is left at its end leave as single dwords, the 47-48 chunks between as aligned dwordx4
//   8  as 6 with 512-thread workgroups (8 waves = 128 vertices = 1536 B per frame: half as many shared lines)
//   9  as 6 with nontemporal stores
//  10  as 2 (padded pitch), renumbered
//  11  workgroup order, one frame's 768 B per wave instruction (48 lanes active, 4 instructions per wave), renumbered
//  12  as 10 (padded pitch, renumbered) with the buffer's base moved by 4 B: every run misaligned, pitch still 64-B
//  13  as 10 with the base moved by 52 B per ... (pitch 60 288, run offset = (frame * 52) % 128 bytes: the real layout's
//      phase walk without its pitch)
//  20  as 5 (slab order, renumbered) with at most two tiles of stores in flight per wave (s_waitcnt vmcnt(3) per tile)
//  21  as 20 at 3 workgroups per CU (50 KB of LDS each)
//  22  as 21 with the kernel's operand traffic: 18 KB of LDS-DMA per workgroup and tile, one tile ahead, one barrier
//  23  as 22 with 10 KB per tile
//  24  as 22 at 2 workgroups per CU
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int V = 5023, T = 250, B = 32;

template <int NT>
__global__ __launch_bounds__(NT) void wide_kernel(float* __restrict__ verts, long long pitch, int nt) {
    constexpr int RUN = NT / 64 * 48;     // floats per frame and workgroup
    const int id = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y, per = total / 8;
    const int L = (id % 8) * per + id / 8;
    const int by = L / gridDim.x, bx = L - by * gridDim.x;
    const int len = V * 3 - bx * RUN < RUN ? V * 3 - bx * RUN : RUN;
    const int fbeg = by * T, fend = fbeg + T;
    for (int f0 = fbeg; f0 < fend; f0 += 16) {
        const f32x4 val = {(float)f0, (float)threadIdx.x, 1.f, 2.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int cidx = i * NT + threadIdx.x, fl = cidx / (RUN / 4), piece = cidx - fl * (RUN / 4);
            const int ff = f0 + fl < fend ? f0 + fl : fend - 1;
            float* dst = verts + (long long)ff * pitch + bx * RUN + piece * 4;
            if (piece * 4 + 4 <= len) {
                if (nt) __builtin_nontemporal_store(val, reinterpret_cast<f32x4u*>(dst));
                else *reinterpret_cast<f32x4u*>(dst) = val;
            }
        }
    }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// DMA: KiB of operands per tile (0 = none); LDSKB: LDS held per workgroup; throttle as the product kernel
template <int DMA, int LDSKB>
__global__ __launch_bounds__(256) void loop_kernel(float* __restrict__ verts, long long pitch, const char* __restrict__ ops) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int id = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y, per = total / 8;
    const int L = (id % 8) * per + id / 8;
    const int by = L / gridDim.x, bx = L - by * gridDim.x;
    const int vt = bx * 4 + wave;
    const bool full = vt * 16 + 16 <= V;
    const int fbeg = by * T, fend = fbeg + T;
    const char* src = ops + (long long)by * 16 * DMA * 1024;
    auto issue = [&](int ti, int buf) {
        for (int c = wave; c < DMA; c += 4)
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + ((long long)ti * DMA + c) * 1024 + lane * 16),
                                             (lds_void*)(lds + (buf * DMA + c) * 1024), 16, 0, 0);
    };
    if (DMA) issue(0, 0);
    int buf = 0, ti = 0;
    for (int f0 = fbeg; f0 < fend; f0 += 16, ++ti, buf ^= 1) {
        if (full && ti) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (DMA) {
            __syncthreads();
            if (f0 + 16 < fend) issue(ti + 1, buf ^ 1);
        }
        if (!full) continue;
        f32x4 val = {(float)f0, (float)lane, 1.f, 2.f};
        if (DMA) val[2] = *reinterpret_cast<const float*>(lds + buf * DMA * 1024 + lane * 4);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int cidx = i * 64 + lane, fl = cidx / 12, piece = cidx - fl * 12;
            const int ff = f0 + fl < fend ? f0 + fl : fend - 1;
            float* dst = verts + (long long)ff * pitch + vt * 48 + piece * 4;
            *reinterpret_cast<f32x4u*>(dst) = val;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ verts, long long pitch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    int bx = blockIdx.x, by = blockIdx.y;
    if (MODE >= 5) {   // also 10
        const int id = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y, per = total / 8;   // total % 8 == 0 here
        const int L = (id % 8) * per + id / 8;
        by = L / gridDim.x;
        bx = L - by * gridDim.x;
    }
    const int vt = bx * 4 + wave;
    const bool full = vt * 16 + 16 <= V;
    if (!full) return;
    const int fbeg = by * T, fend = fbeg + T;
    for (int f0 = fbeg; f0 < fend; f0 += 16) {
        const f32x4 val = {(float)f0, (float)lane, 1.f, 2.f};
        if (MODE == 0 || MODE == 2 || MODE == 5 || MODE == 10 || MODE == 13) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int cidx = i * 64 + lane, fl = cidx / 12, piece = cidx - fl * 12;
                const int ff = f0 + fl < fend ? f0 + fl : fend - 1;
                float* dst = verts + (long long)ff * pitch + vt * 48 + piece * 4 + (MODE == 13 ? (ff * 13) % 32 : 0);
                *reinterpret_cast<f32x4u*>(dst) = val;
            }
        } else if (MODE == 1) {
            const int ff = f0 + fr < fend ? f0 + fr : fend - 1;
            float* dst = verts + (long long)ff * pitch + vt * 48 + fq * 12;
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4u*>(dst + 4 * i) = val;
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int cidx = i * 256 + threadIdx.x, fl = cidx / 48, j = cidx - fl * 48;
                const int ff = f0 + fl < fend ? f0 + fl : fend - 1;
                const long long o = (long long)ff * pitch + bx * 192;
                const int h = (4 - (int)(o & 3)) & 3, start = h + 4 * j;
                float* dst = verts + o;
                if (start + 4 <= 192) {
                    *reinterpret_cast<f32x4*>(dst + start) = val;
                } else {
                    for (int q = start; q < 192; ++q) dst[q] = val[q & 3];
                }
                if (j == 47)
                    for (int q = 0; q < h; ++q) dst[q] = val[q];
            }
        } else if (MODE == 11) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int fl = i * 4 + wave;
                const int ff = f0 + fl < fend ? f0 + fl : fend - 1;
                float* dst = verts + (long long)ff * pitch + bx * 192 + lane * 4;
                if (lane < 48) *reinterpret_cast<f32x4u*>(dst) = val;
            }
        } else if (MODE == 3 || MODE == 6) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int cidx = i * 256 + threadIdx.x, fl = cidx / 48, piece = cidx - fl * 48;
                const int ff = f0 + fl < fend ? f0 + fl : fend - 1;
                float* dst = verts + (long long)ff * pitch + bx * 192 + piece * 4;
                *reinterpret_cast<f32x4u*>(dst) = val;
            }
        }
    }
}

__global__ __launch_bounds__(256) void fill_kernel(f32x4* __restrict__ p, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) p[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
}

int main() {
    const long long F = (long long)B * T;
    float* buf;
    const long long pitch_pad = 60288 / 4;
    (void)hipMalloc(&buf, F * pitch_pad * 4 + 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const dim3 grid((V / 16 + 3) / 4, B);
    char* ops;
    (void)hipMalloc(&ops, 32ll * 16 * 18 * 1024);
    (void)hipMemset(ops, 0, 32ll * 16 * 18 * 1024);
    for (int mode = 0; mode < 25; ++mode) {
        if (mode > 13 && mode < 20) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipEventRecord(e0, 0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(store_kernel<0>, grid, dim3(256), 0, 0, buf, (long long)V * 3); break;
                case 1: hipLaunchKernelGGL(store_kernel<1>, grid, dim3(256), 0, 0, buf, (long long)V * 3); break;
                case 2: hipLaunchKernelGGL(store_kernel<2>, grid, dim3(256), 0, 0, buf, pitch_pad); break;
                case 3: hipLaunchKernelGGL(store_kernel<3>, grid, dim3(256), 0, 0, buf, (long long)V * 3); break;
                case 5: hipLaunchKernelGGL(store_kernel<5>, grid, dim3(256), 0, 0, buf, (long long)V * 3); break;
                case 6: hipLaunchKernelGGL(store_kernel<6>, grid, dim3(256), 0, 0, buf, (long long)V * 3); break;
                case 7: hipLaunchKernelGGL(store_kernel<7>, grid, dim3(256), 0, 0, buf, (long long)V * 3); break;
                case 8: hipLaunchKernelGGL(wide_kernel<512>, dim3(40, B), dim3(512), 0, 0, buf, (long long)V * 3, 0); break;
                case 9: hipLaunchKernelGGL(wide_kernel<256>, dim3(79, B), dim3(256), 0, 0, buf, (long long)V * 3, 1); break;
                case 10: hipLaunchKernelGGL(store_kernel<10>, grid, dim3(256), 0, 0, buf, pitch_pad); break;
                case 11: hipLaunchKernelGGL(store_kernel<11>, grid, dim3(256), 0, 0, buf, (long long)V * 3); break;
                case 12: hipLaunchKernelGGL(store_kernel<10>, grid, dim3(256), 0, 0, buf + 1, pitch_pad); break;
                case 13: hipLaunchKernelGGL(store_kernel<13>, grid, dim3(256), 0, 0, buf, pitch_pad); break;
                case 20: hipLaunchKernelGGL((loop_kernel<0, 0>), grid, dim3(256), 0, 0, buf, (long long)V * 3, ops); break;
                case 21: hipLaunchKernelGGL((loop_kernel<0, 50>), grid, dim3(256), 50 * 1024, 0, buf, (long long)V * 3, ops); break;
                case 22: hipLaunchKernelGGL((loop_kernel<18, 50>), grid, dim3(256), 50 * 1024, 0, buf, (long long)V * 3, ops); break;
                case 23: hipLaunchKernelGGL((loop_kernel<10, 50>), grid, dim3(256), 50 * 1024, 0, buf, (long long)V * 3, ops); break;
                case 24: hipLaunchKernelGGL((loop_kernel<18, 70>), grid, dim3(256), 70 * 1024, 0, buf, (long long)V * 3, ops); break;
                case 4: hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<f32x4*>(buf), F * V * 3 / 4); break;
            }
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("mode %d: %.1f us  (%.2f TB/s of 482 MB)\n", mode, best * 1e3, F * V * 12.0 / best / 1e9);
    }
    return 0;
}
