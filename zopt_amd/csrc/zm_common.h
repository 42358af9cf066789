// Shared host-side helpers for the zopt_amd C ABI (error reporting, argument checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/zopt_amd.h"

namespace zm {

// Thread-local last-error buffer behind zm_last_error().
char* last_error_buf();
int set_error(int code, const char* fmt, ...);

#define ZM_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return zm::set_error((int)_e, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

}  // namespace zm
