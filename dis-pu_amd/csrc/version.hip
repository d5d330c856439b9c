// Version / error-string entry points of libdispu_hip.so.
#include "common.h"

DISPU_EXPORT int dispu_version(void) { return 2; }   // 2: round-2 ABI (scratch arguments of match_cost(_grad), *_ws k-NN entries, bf16 GEMMs)
DISPU_EXPORT const char* dispu_error_string(int code) { return hipGetErrorString((hipError_t)code); }
