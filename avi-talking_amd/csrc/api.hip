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
