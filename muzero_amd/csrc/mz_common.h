// Shared host-side helpers of libmzhip.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/mzhip.h"

namespace mz {

void set_error(const char* fmt, ...);

inline int fail(mz_status code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(mz_status code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

}  // namespace mz

#define MZ_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return mz::fail(_e == hipErrorOutOfMemory ? MZ_ERR_OOM : MZ_ERR_HIP, "%s failed: %s (%s:%d)", \
                            #expr, hipGetErrorString(_e), __FILE__, __LINE__);                         \
    } while (0)

#define MZ_REQUIRE(cond, ...)                                      \
    do {                                                           \
        if (!(cond)) return mz::fail(MZ_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)
