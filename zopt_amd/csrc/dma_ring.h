// LDS ring filled by DMA (global_load_lds): the pieces shared by the kernels that stage their per-step operands this way
// (lqr_backward_dma.hip, ilqr_backward.hip).  Every DMA wave-instruction copies 64 x 16 B, lane i -> LDS bytes [16 i, 16 i + 16).
#pragma once
#include <hip/hip_runtime.h>

namespace zm {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// 16 B of zeros: DMA source of the idle lanes, so that every ring slot's padding reads 0.0.
static __device__ __attribute__((aligned(16))) const double zm_zero_src[2] = {0.0, 0.0};
// 16 B of ones: a second constant chunk in a slot's padding for kernels whose operands are packed (entries that are not stored are 0 or 1)
static __device__ __attribute__((aligned(16))) const double zm_one_src[2] = {1.0, 1.0};

template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
    static_assert(N >= 0 && N <= 15, "vmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}

// DMA groups retire in issue order and are issued in decreasing step order, so when step j is consumed exactly
// min(D-1, j) younger DMA groups may still be in flight.  vmcnt counts stores too (gfx9: loads, stores and LDS-DMA share the
// counter, in issue order): a kernel that issues ONE store per step after the DMA of that step has, at the wait for step j,
// min(D-1, T-1-j) younger stores outstanding on top of the younger DMA groups.  STORES = false ignores them (over-waits: the
// wait then also drains most of the next step's DMA, i.e. gives away prefetch distance); STORES = true counts them.
template <int NI, int D, bool STORES = false>
__device__ __forceinline__ void wait_for_step(const int j, const int T = 0) {
    if constexpr (STORES) {
        // issue order (every iteration i: wait(i); operand reads; DMA(i-D); ...; store L(i)), so between DMA(j) -- issued in
        // iteration j+D, or in the prologue -- and the wait of iteration j lie:  store L(j+D), DMA(j-1), store L(j+D-1), ...,
        // DMA(j-D+1), store L(j+1).  Younger than DMA(j): min(D-1, j) DMA groups and min(D, T-1-j) stores.
        const int st = (T - 1 - j) < D ? (T - 1 - j) : D;
        if (j >= D - 1) {
            if (st >= D) wait_vmcnt_imm<NI*(D - 1) + D>();          // steady state
            else if (st == 0) wait_vmcnt_imm<NI*(D - 1)>();          // the first D steps of the sweep: fewer stores behind
            else if (st == 1) wait_vmcnt_imm<NI*(D - 1) + 1>();
            else if (st == 2) wait_vmcnt_imm<NI*(D - 1) + 2>();
            else wait_vmcnt_imm<NI*(D - 1) + 3>();
        } else {                           // the last D-1 steps: fewer DMA groups behind; stores not counted (over-waits, 2 steps)
            if (j == 0) wait_vmcnt_imm<0>();
            else if (j == 1) wait_vmcnt_imm<NI>();
            else wait_vmcnt_imm<2 * NI>();
        }
    } else {
        if (j >= D - 1) {
            wait_vmcnt_imm<NI*(D - 1)>();
        } else if (D >= 3 && j == 1) {
            wait_vmcnt_imm<NI>();
        } else if (D >= 4 && j == 2) {
            wait_vmcnt_imm<2 * NI>();
        } else {
            wait_vmcnt_imm<0>();
        }
    }
}

}  // namespace zm
