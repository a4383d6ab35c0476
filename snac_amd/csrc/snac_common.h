// snac_common.h -- what the translation units of libsnac_hip.so share besides include/snac_hip.h (internal, not installed):
// the thread-local error string behind snac_last_error() and the two helpers that fill it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>

#include "snac_hip.h"

namespace snac_detail {
extern thread_local char g_err[256];                                 // defined in snac_hip.hip
inline int fail(int code, const char* msg) {
    std::snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
inline int fail_hip(hipError_t e, const char* where) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return SNAC_ERR_HIP;
}
}  // namespace snac_detail
