// Version / error-string entry points of libdispu_hip.so.
#include "common.h"

DISPU_EXPORT int dispu_version(void) { return 5; }   // history: include/dispu_hip.h
DISPU_EXPORT const char* dispu_error_string(int code) { return hipGetErrorString((hipError_t)code); }

// ---- launch-tape helpers (dis-pu_amd/_lib.py:Tape): the training step's eager launch sequence, recorded once and re-issued from a
// flat list, needs its stream / event / memset operations as plain C calls on raw handles (torch's wrappers cost 2 - 5 us each).
DISPU_EXPORT int dispu_event_record(void* event, void* stream) { return (int)hipEventRecord((hipEvent_t)event, (hipStream_t)stream); }
DISPU_EXPORT int dispu_stream_wait_event(void* stream, void* event) {
    return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
}
DISPU_EXPORT int dispu_memset_async(void* dst, int value, size_t bytes, void* stream) {
    if (!dst && bytes) return (int)hipErrorInvalidValue;
    return bytes ? (int)hipMemsetAsync(dst, value, bytes, (hipStream_t)stream) : 0;
}
