// Library-level entry points of libavi_talking_hip.so.
#include "common.h"

extern "C" const char* avi_version(void) { return "avi_talking_hip 0.1.0 (gfx950)"; }
