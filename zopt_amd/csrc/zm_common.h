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


// Ordering of an LDS write against a later LDS read of the same bytes by ANOTHER LANE OF THE SAME WAVE (one-wave workgroups).
// The hardware runs a wave's DS instructions in issue order, so only the compiler must keep the program order of the two
// memory operations.  Default: a compiler-level memory barrier -- register-only instructions (MFMA, VALU) may be scheduled
// across it (K1 on resident data: 112 -> 106 us, outputs bit for bit the same; profiles/r02_k1_lab12.txt).
// -DZM_HEAVY_LDS_SYNC: wavefront-scope fences around a wave barrier, which is also a scheduling barrier for everything.
__device__ __forceinline__ void wave_lds_sync() {
#ifdef ZM_HEAVY_LDS_SYNC
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    asm volatile("" ::: "memory");
#endif
}

}  // namespace zm
