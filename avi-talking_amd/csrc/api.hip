// Library-level entry points of libavi_talking_hip.so.
#include "common.h"

// A build with packed-FP32 VALU instructions (AVI_PACKED_FP32=1, diagnostics only: build.py explains why the product is
// built without them) says so in its version string; lib.load() refuses such a library unless explicitly allowed.
#ifdef AVI_BUILD_PACKED_FP32
extern "C" const char* avi_version(void) { return "avi_talking_hip 0.2.0 (gfx950, packed-fp32 DIAGNOSTIC build)"; }
#else
extern "C" const char* avi_version(void) { return "avi_talking_hip 0.2.0 (gfx950)"; }
#endif
