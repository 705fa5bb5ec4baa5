// odam_err.h -- error plumbing shared by the C-ABI translation units (no exceptions cross the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>

inline thread_local char g_odam_err[512] = "";

inline int odam_fail(int code, const char* msg) {
    std::snprintf(g_odam_err, sizeof(g_odam_err), "%s", msg);
    return code;
}

#define ODAM_HIP(call)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            std::snprintf(g_odam_err, sizeof(g_odam_err), "%s failed: %s (%s:%d)", #call,         \
                          hipGetErrorString(e_), __FILE__, __LINE__);                            \
            return 2; /* ODAM_E_HIP */                                                           \
        }                                                                                        \
    } while (0)
