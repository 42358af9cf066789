// Shared host-side helpers for the zopt_amd C ABI (error reporting, argument checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/zopt_amd.h"

namespace zm {

// Run-time switches.  A PRODUCT build reads exactly four environment variables, each selecting a FALLBACK kernel -- the one that
// also serves the shapes / alignments its fast path refuses, so it is product code either way:
//     ZOPT_AMD_LQR_PATH=lds|reg   ZOPT_AMD_ILQR_PATH=reg   ZOPT_AMD_ROLLOUT_PATH=generic   ZOPT_AMD_MPC_PATH=lane
// Every other ZOPT_AMD_* variable is an A/B switch of the kernel lab (ring depth, packed / full operands, tail thresholds, ...):
// it exists only in a -DZM_LAB build (`make lab` -> libzopt_amd_lab.so, loaded through ZOPT_AMD_LIB by the A/B tests and the lab
// tools); in the product library lab_env() is the constant nullptr and the compiler removes the branch it guards.
inline const char* fallback_env(const char* name) { return getenv(name); }
inline const char* lab_env(const char* name) {
#ifdef ZM_LAB
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

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

// 32-bit halves of a double through one DPP move each (VALU speed; __shfl_xor is two ds_bpermute round trips through the LDS crossbar
// per level).  mov_dpp with bound_ctrl: every lane of these in-row permutations has a valid source, so the destination needs no
// initial value.
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(const double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// Sum / maximum over the 16 lanes of a row, the result in every lane of the row: butterflies by quad_perm (xor 1, xor 2),
// row_half_mirror and row_mirror.  The same bits as the __shfl_xor butterfly over offsets 1, 2, 4, 8: after the quad steps every lane of
// a quad holds the quad's value, so the mirrored partner of a step holds exactly what the xor partner holds, and + / max commute.
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_mov64<0xB1>(v);
    v += dpp_mov64<0x4E>(v);
    v += dpp_mov64<0x141>(v);
    v += dpp_mov64<0x140>(v);
    return v;
}
// The same sum in the order of a ladder that starts at offset 8: row_ror:8 is lane ^ 8 on 16 lanes; after it the values repeat with
// period 8, so row_ror:4 delivers what lane ^ 4 holds; then the quad steps.  Bits of `for (off = 8; off >= 1; off >>= 1) v += shfl_xor`.
__device__ __forceinline__ double row16_sum_from8(double v) {
    v += dpp_mov64<0x128>(v);
    v += dpp_mov64<0x124>(v);
    v += dpp_mov64<0x4E>(v);
    v += dpp_mov64<0xB1>(v);
    return v;
}
__device__ __forceinline__ double row16_max(double v) {
    v = __builtin_fmax(v, dpp_mov64<0xB1>(v));
    v = __builtin_fmax(v, dpp_mov64<0x4E>(v));
    v = __builtin_fmax(v, dpp_mov64<0x141>(v));
    v = __builtin_fmax(v, dpp_mov64<0x140>(v));
    return v;
}

// v + (the value of lane ^ 16) resp. lane ^ 32, by the gfx950 row / half-wave swaps (VALU; __shfl_xor across rows is a ds_bpermute
// round trip through LDS on the dependency chain).  permlane16_swap(a, b): odd rows of a <-> even rows of b; with a = b = v the two
// results hold the even-row resp. odd-row partner values of every row pair, whose sum is the butterfly sum in both rows (fp addition
// commutes: the same bits as v + __shfl_xor(v, 16)).  permlane32_swap likewise for the two half waves.
__device__ __forceinline__ double sum_xor16(const double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    const double x = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | (unsigned long long)lo[0]);
    const double y = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | (unsigned long long)lo[1]);
    return x + y;
}
__device__ __forceinline__ double sum_xor32(const double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    const double x = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | (unsigned long long)lo[0]);
    const double y = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | (unsigned long long)lo[1]);
    return x + y;
}

}  // namespace zm
