// Library-level entry points of libavi_talking_hip.so.
#include "common.h"

// A build with packed-FP32 VALU instructions (AVI_PACKED_FP32=1, diagnostics only: build.py explains why the product is
// built without them) says so in its version string; lib.load() refuses such a library unless explicitly allowed.
#ifdef AVI_BUILD_PACKED_FP32
extern "C" const char* avi_version(void) { return "avi_talking_hip 0.2.0 (gfx950, packed-fp32 DIAGNOSTIC build)"; }
#else
extern "C" const char* avi_version(void) { return "avi_talking_hip 0.2.0 (gfx950)"; }
#endif

// Status words (avi_talking.h): one pointer per process, read by every launch site that can report a device-side failure.
static std::atomic<unsigned*> g_status_words{nullptr};
unsigned* avi_status_ptr() { return g_status_words.load(std::memory_order_acquire); }
extern "C" int avi_set_status_words(void* words) {
    if (reinterpret_cast<uintptr_t>(words) & 3) return AVI_EINVAL;
    g_status_words.store(static_cast<unsigned*>(words), std::memory_order_release);
    return AVI_OK;
}
extern "C" void* avi_status_words(void) { return avi_status_ptr(); }

// Fault injection for the tests of the failure paths (avi_talking.h "Diagnostics").
static std::atomic<int> g_fault{0};
int avi_fault_injected() { return g_fault.load(std::memory_order_relaxed); }
extern "C" int avi_debug_fault_inject(int faults) {
    g_fault.store(faults, std::memory_order_relaxed);
    return AVI_OK;
}

__global__ void raise_status_kernel(unsigned* status, int k) {
    if (status) __hip_atomic_store(status + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int avi_debug_raise_status(int k, void* stream) {
    if (k < 0 || k >= AVI_STATUS_WORDS) return AVI_EINVAL;
    hipLaunchKernelGGL(raise_status_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), avi_status_ptr(), k);
    return avi_launch_status();
}

// Diagnostics: where the workgroups of a launch run.  out[b] = XCC_ID << 16 | HW_ID[15:0] of workgroup b (HW_ID: wave, SIMD,
// CU, shader array and shader engine ids within the XCD).  Used to probe CU masks of streams and the round-robin dealing of
// workgroups over the XCDs (speed only: nothing in the library depends on placement for correctness).
__global__ void where_kernel(unsigned* __restrict__ out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);      // keep the slot busy so that later workgroups spread out
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
}
extern "C" int avi_debug_where(unsigned* out, int blocks, int threads, int lds_bytes, int spin, void* stream) {
    if (!out || blocks < 1 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || spin < 0) return AVI_EINVAL;
    static AviLdsGrant lds_grant;
    lds_grant.ensure(reinterpret_cast<const void*>(where_kernel), 160 * 1024);
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(threads), lds_bytes, static_cast<hipStream_t>(stream), out, spin);
    return avi_launch_status();
}
