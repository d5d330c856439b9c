// Version / error-string entry points of libdispu_hip.so.
#include "common.h"

DISPU_EXPORT int dispu_version(void) { return 4; }   // history: include/dispu_hip.h
DISPU_EXPORT const char* dispu_error_string(int code) { return hipGetErrorString((hipError_t)code); }
