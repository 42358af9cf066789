// Core of the MFMA tile kernels for the LQR backward sweep with LARGE states, shared by the fp32 (n <= 64) and fp64 (n <= 48)
// instantiations.  See lqr_backward_tiled_f32.hip for the derivation; the only type-dependent facts are collected in the traits:
// scalar / tile types, the MFMA instruction, which matrix row register r of lane group g holds, and the LDS access idioms.
#pragma once
#include <hip/hip_runtime.h>

#include "zm_common.h"

namespace zm {

typedef float tf4 __attribute__((ext_vector_type(4)));
typedef double td4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void t_lds_sync() { wave_lds_sync(); }

// v_mfma_f32_16x16x4_f32: accumulator register r of lane (g, c) holds row 4g + r.
struct TileF32 {
    using S = float;
    using V4 = tf4;
    static constexpr bool FUSED_DPP = true;   // the solve's broadcasts ride on the FMA (v_fmac_f32_dpp ... row_newbcast)
    // One batch of the elimination at pivot KK, rows r0 .. r0+CNT-1 (lane (g, c) holds column c of Suu in u[], one column of Sux in
    // x[]): m_q = u_q * inv (the multipliers, meaningful in lane c == KK of every 16-lane row), then
    //     x_q += bcast_KK(m_q) * nxk,   u_q += bcast_KK(m_q) * nuk         (nxk = -x[KK], nuk = -u[KK])
    // with the broadcast as the DPP operand of the FMA itself: 3 instructions per row instead of 5 (multiplier, v_readlane, two
    // FMAs) and no scalar registers.  ONE asm statement: the CNT multiplies come first, so every multiplier is at least two
    // instructions old when a DPP operand reads it (the wait states hipcc does not insert inside asm).
    template <int KK, int CNT>
    static __device__ __forceinline__ void elim_batch(const float inv, const float nxk, const float nuk, float (&m)[4], float& x0,
                                                      float& x1, float& x2, float& x3, float& u0, float& u1, float& u2, float& u3) {
        if constexpr (CNT == 4)
            asm("v_mul_f32 %0, %8, %12\n\tv_mul_f32 %1, %9, %12\n\tv_mul_f32 %2, %10, %12\n\tv_mul_f32 %3, %11, %12\n\t"
                "v_fmac_f32_dpp %4, %0, %13 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %8, %0, %14 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %5, %1, %13 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %9, %1, %14 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %6, %2, %13 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %10, %2, %14 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %7, %3, %13 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %11, %3, %14 row_newbcast:%15 row_mask:0xf bank_mask:0xf"
                : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(u0), "+v"(u1),
                  "+v"(u2), "+v"(u3)
                : "v"(inv), "v"(nxk), "v"(nuk), "i"(KK));
        else if constexpr (CNT == 3)
            asm("v_mul_f32 %0, %6, %9\n\tv_mul_f32 %1, %7, %9\n\tv_mul_f32 %2, %8, %9\n\t"
                "v_fmac_f32_dpp %3, %0, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %6, %0, %11 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %4, %1, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %7, %1, %11 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %5, %2, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %8, %2, %11 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
                : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(u0), "+v"(u1), "+v"(u2)
                : "v"(inv), "v"(nxk), "v"(nuk), "i"(KK));
        else if constexpr (CNT == 2)
            asm("v_mul_f32 %0, %4, %6\n\tv_mul_f32 %1, %5, %6\n\ts_nop 0\n\t"
                "v_fmac_f32_dpp %2, %0, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %4, %0, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %3, %1, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %5, %1, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
                : "=&v"(m[0]), "=&v"(m[1]), "+v"(x0), "+v"(x1), "+v"(u0), "+v"(u1)
                : "v"(inv), "v"(nxk), "v"(nuk), "i"(KK));
        else
            asm("v_mul_f32 %0, %2, %3\n\ts_nop 1\n\t"
                "v_fmac_f32_dpp %1, %0, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %2, %0, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
                : "=&v"(m[0]), "+v"(x0), "+v"(u0)
                : "v"(inv), "v"(nxk), "v"(nuk), "i"(KK));
    }
    // acc += bcast_L(v) * c  (substitution: v = -u[kk] of the row, lane L holds column L's entry)
    template <int L>
    static __device__ __forceinline__ void fma_bc(float& acc, const float v, const float c) {
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(c), "i"(L));
    }
    // bcast_L(v) * c
    template <int L>
    static __device__ __forceinline__ float mul_bc(const float v, const float c) {
        float r;
        asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(c), "i"(L));
        return r;
    }
    static __device__ __forceinline__ float dpp_src(float v) {   // two wait states between a VALU write and a DPP read of v
        asm("s_nop 1" : "+v"(v));
        return v;
    }
    static constexpr int TLD = 20;   // row stride of the LDS buffers (floats): 80 B rows keep the b128 accesses 16 B-aligned
    static __device__ __forceinline__ V4 zero() { return V4{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ int row(const int g, const int r) { return 4 * g + r; }
    static __device__ __forceinline__ V4 mfma(const S a, const S b, const V4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ S fma(const S a, const S b, const S c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ S abs(const S a) { return __builtin_fabsf(a); }
    static __device__ __forceinline__ S huge() { return 3.0e38f; }
    static __device__ __forceinline__ void pin(S& v) { asm volatile("" : "+v"(v)); }
    static __device__ __forceinline__ S readlane(const S v, const int l) {   // wave-uniform copy of lane l's value
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
    }
    static __device__ __forceinline__ S rcp(const S a) {   // 1/a to fp32 rounding: hardware estimate (1 ulp) + one Newton step
        const float y = __builtin_amdgcn_rcpf(a);
        return __builtin_fmaf(__builtin_fmaf(-a, y, 1.0f), y, y);
    }
    // D-layout tile -> LDS, transposed: buf[col * TLD + row] (one b128 per lane: the 4 rows of a lane are consecutive)
    static __device__ __forceinline__ void tile_to_lds_T(S* buf, const V4 t, const int g, const int c) {
        *reinterpret_cast<V4*>(buf + c * TLD + 4 * g) = t;
    }
    // D-layout tile -> LDS row-major buf[row * TLD + col]; tile_from_lds_T then returns the transpose: out[r] = X[c][4g + r]
    static __device__ __forceinline__ void tile_to_lds(S* buf, const V4 t, const int g, const int c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(4 * g + r) * TLD + c] = t[r];
    }
    static __device__ __forceinline__ V4 tile_from_lds_T(const S* buf, const int g, const int c) {
        return *reinterpret_cast<const V4*>(buf + c * TLD + 4 * g);
    }
};

// v_mfma_f64_16x16x4_f64: accumulator register r of lane (g, c) holds row 4r + g.
struct TileF64 {
    using S = double;
    using V4 = td4;
    static constexpr bool FUSED_DPP = false;
    static constexpr int TLD = 18;   // 144 B rows
    static __device__ __forceinline__ V4 zero() { return V4{0.0, 0.0, 0.0, 0.0}; }
    static __device__ __forceinline__ int row(const int g, const int r) { return 4 * r + g; }
    static __device__ __forceinline__ V4 mfma(const S a, const S b, const V4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ S fma(const S a, const S b, const S c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ S abs(const S a) { return __builtin_fabs(a); }
    static __device__ __forceinline__ S huge() { return 1.0e300; }
    static __device__ __forceinline__ void pin(S& v) { asm volatile("" : "+v"(v)); }
    static __device__ __forceinline__ S readlane(const S v, const int l) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ S rcp(const S a) {   // hardware estimate + two Newton steps
        double r = __builtin_amdgcn_rcp(a);
        r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
        return r;
    }
    static __device__ __forceinline__ void tile_to_lds_T(S* buf, const V4 t, const int g, const int c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[c * TLD + 4 * r + g] = t[r];
    }
    static __device__ __forceinline__ void tile_to_lds(S* buf, const V4 t, const int g, const int c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(4 * r + g) * TLD + c] = t[r];
    }
    static __device__ __forceinline__ V4 tile_from_lds_T(const S* buf, const int g, const int c) {
        V4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = buf[c * TLD + 4 * r + g];
        return t;
    }
};

// acc + X^T Y for D-layout tiles: register s is K-step s (for both layouts the K index of lane group g in step s is the row
// the tile register s of that group holds, so a tile is directly a B operand and, read as the A operand, its transpose)
template <class TR>
__device__ __forceinline__ typename TR::V4 op(const typename TR::V4 x, const typename TR::V4 y, typename TR::V4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = TR::mfma(x[s], y[s], acc);
    return acc;
}

// tile (K, J) of a row-major (nrows x ncols) matrix in D layout; out-of-range elements read as `diag` on the diagonal, else 0
template <class TR, bool EXACT>
__device__ __forceinline__ typename TR::V4 load_tile(const typename TR::S* __restrict__ X, const int nrows, const int ncols,
                                                     const int K, const int J, const int g, const int c,
                                                     const typename TR::S diag = 0) {
    typename TR::V4 t;
    const int col = 16 * J + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * K + TR::row(g, r);
        if constexpr (EXACT) {
#ifdef ZM_TILED_NT_LOADS
            t[r] = __builtin_nontemporal_load(&X[row * ncols + col]);   // operands are streamed once (A/B build: -DZM_TILED_NT_LOADS)
#else
            t[r] = X[row * ncols + col];
#endif
        } else {
            const bool ok = row < nrows && col < ncols;
            const typename TR::S v = X[ok ? row * ncols + col : 0];
            t[r] = ok ? v : ((row == col) ? diag : typename TR::S(0));
        }
    }
    return t;
}

// Elimination / substitution of the 16 x 16 solve with the broadcasts fused into the FMAs (TR::FUSED_DPP): compile-time recursion over
// the pivot, because the DPP control (row_newbcast:KK) is an immediate.  x: the lane's column of Sux; u: its column of Suu;
// inv[kk]: 1 / pivot kk (meaningful in lane c == kk of every 16-lane row).
template <class TR, int KK>
__device__ __forceinline__ void solve_elim(typename TR::S (&x)[16], typename TR::S (&u)[16], typename TR::S (&inv)[16], bool& okm,
                                           unsigned long long& bad) {
    using S = typename TR::S;
    if constexpr (KK < 16) {
        inv[KK] = TR::rcp(u[KK]);
        const S nxk = -x[KK], nuk = -u[KK];
        // growth check of the unpivoted elimination: the squares of the column's multipliers must sum to <= 16, which implies
        // every |multiplier| <= 4 (the bound of the register-level fast path); one FMA per row instead of a compare and a
        // mask update, and a NaN or infinite multiplier fails it (NaN compares false)
        S sq = S(0);
        constexpr int R0 = KK + 1;
#define ZM_ELIM_BATCH(B)                                                                                                        \
        if constexpr (R0 + 4 * (B) < 16) {                                                                                      \
            constexpr int r0 = R0 + 4 * (B), cnt = (16 - r0) < 4 ? (16 - r0) : 4;                                               \
            S m[4] = {S(0), S(0), S(0), S(0)};                                                                                  \
            constexpr int i1 = r0 + 1 < 16 ? r0 + 1 : 15, i2 = r0 + 2 < 16 ? r0 + 2 : 15, i3 = r0 + 3 < 16 ? r0 + 3 : 15;       \
            TR::template elim_batch<KK, cnt>(inv[KK], nxk, nuk, m, x[r0], x[i1], x[i2], x[i3], u[r0], u[i1], u[i2], u[i3]);     \
            _Pragma("unroll") for (int q = 0; q < cnt; ++q) sq = TR::fma(m[q], m[q], sq);   /* NaN / inf propagate */               \
        }
        ZM_ELIM_BATCH(0)
        ZM_ELIM_BATCH(1)
        ZM_ELIM_BATCH(2)
        ZM_ELIM_BATCH(3)
#undef ZM_ELIM_BATCH
        bad |= __ballot(!(sq <= S(16))) & (0x0001000100010001ull << KK);   // only lane c == KK of each row holds the column's multipliers
        (void)okm;
        solve_elim<TR, KK + 1>(x, u, inv, okm, bad);
    }
}

template <class TR, int KK, int R>
__device__ __forceinline__ void subst_terms(typename TR::S& a0, typename TR::S& a1, const typename TR::S nuk, const typename TR::S (&x)[16]) {
    if constexpr (R < 16) {
        if constexpr ((R - KK) & 1)
            TR::template fma_bc<R>(a0, nuk, x[R]);     // a0 += bcast_R(-U[KK][R]) * x[R]
        else
            TR::template fma_bc<R>(a1, nuk, x[R]);
        subst_terms<TR, KK, R + 1>(a0, a1, nuk, x);
    }
}

template <class TR, int KK>
__device__ __forceinline__ void solve_subst(typename TR::S (&x)[16], const typename TR::S (&u)[16], const typename TR::S (&inv)[16]) {
    using S = typename TR::S;
    if constexpr (KK >= 0) {
        S a0 = x[KK], a1 = S(0);
        const S nuk = TR::dpp_src(-u[KK]);             // lane R of a row holds -U[KK][R]
        subst_terms<TR, KK, KK + 1>(a0, a1, nuk, x);
        x[KK] = TR::template mul_bc<KK>(TR::dpp_src(inv[KK]), a0 + a1);   // (a0 + a1) / pivot
        solve_subst<TR, KK - 1>(x, u, inv);
    }
}

// Diagnostic build (-DZM_TILED_LAB, tools/k1t_lab.hip): s_memtime stamps at the phase boundaries of a step, summed per wave into
// zm_tiled_stamps.  The stamps' waits forbid overlaps the product has: read SHARES, not lengths.
#ifdef ZM_TILED_LAB
__device__ unsigned long long zm_tiled_stamps[12];
#define ZT_STAMP(i)                                                                                   \
    {                                                                                                 \
        unsigned long long t_;                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        zt_acc[i] += t_ - zt_last;                                                                    \
        zt_last = t_;                                                                                 \
    }
#else
#define ZT_STAMP(i)
#endif

// PREFETCH: the operands of step k-1 are fetched into a second register set (Fn, Rn) while step k computes.  false (fp64 at four tile
// rows, where V + F + Y alone take 448 of the 512 registers): the step's operands are loaded at its head instead -- the HBM latency
// of one step's loads is exposed once per ~20 us step, which costs far less than spilling a second operand set.
// DARE (discreteInfiniteHorizonLqr for large states, zopt/lqrUtils.py:176-204): the same step iterated on TIME-INVARIANT operands -- A, B,
// Q, R are (batch, ., .) without a time axis, T is the iteration cap, L (batch, m, n) is overwritten every iteration -- from V = Q until
// the gain stops changing: max|L_k - L_{k-1}| <= tol max|L_k| (tested every 8th iteration, or a stall at the rounding floor); then
// P <- V (batch, n, n) and iters <- +k (converged) / -k (cap reached), as zm_dare_f64 documents.
template <class S>
struct TiledDareArgs {
    S* P;
    int* iters;
    S tol;
};

template <class TR, int NT, bool EXACT, bool PREFETCH = true, bool DARE = false>
__global__ __launch_bounds__(64) void lqr_backward_tiled(const typename TR::S* __restrict__ A, const typename TR::S* __restrict__ B,
                                                         const typename TR::S* __restrict__ Q, const typename TR::S* __restrict__ R,
                                                         typename TR::S* __restrict__ L, const long batch, const int T, const int n_,
                                                         const int m_, const TiledDareArgs<typename TR::S> dr = {nullptr, nullptr, 0}) {
    using S = typename TR::S;
    using f4 = typename TR::V4;
    constexpr int TLD = TR::TLD;
    constexpr int NP = 16 * NT;  // padded state dimension
    const int n = EXACT ? NP : n_, m = EXACT ? 16 : m_;
    // Solve buffer, column-major: element (row u, column j) of [Sux | Suu] at Sc[j * TLD + u]; columns NP..NP+15 are Suu.
    __shared__ __attribute__((aligned(16))) S Sc[(NP + 16) * TLD];
    __shared__ __attribute__((aligned(16))) S Tb[2 * NT + 1][16 * TLD];
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    // Ownership inside the LDS-resident (pivoted) solve: lane j < NP owns column j of Sux, lane c < 16 owns column c of Suu.
    // Every read-modify-write of an LDS word is done by its ONE owner: copies kept by several lanes are not safe, because
    // the compiler may sink the read into divergent branches, and lanes of different branches would apply the update twice.
    const int jl = (NT == 4) ? lane : (lane < NP ? lane : NP - 1);  // surplus lanes read the last column and write nothing
    const bool own_x = (NT == 4) || lane < NP;
    const bool own_u = lane < 16;
    const long traj = blockIdx.x;
    if (traj >= batch) return;
    const long nn = (long)n * n, nm = (long)n * m, mm = (long)m * m;
    const long TS = DARE ? 1 : T;              // operands per trajectory along the time axis
    const S* Ab = A + traj * TS * nn;
    const S* Bb = B + traj * TS * nm;
    const S* Qb = Q + traj * TS * nn;
    const S* Rb = R + traj * TS * mm;
    S* Lb = L + traj * TS * nm;
    const long k0 = DARE ? 0 : (long)(T - 1);  // time index of the first step

    f4 V[NT][NT], F[NT][NT + 1], Fn[PREFETCH ? NT : 1][NT + 1], Y[NT][NT + 1], Rt, Rn;
    // terminal value = last stage cost (lqrUtils.py:172); operands of the first step
#pragma unroll
    for (int K = 0; K < NT; ++K) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            V[K][J] = load_tile<TR, EXACT>(Qb + k0 * nn, n, n, K, J, g, c);
            if constexpr (PREFETCH) Fn[K][J] = load_tile<TR, EXACT>(Ab + k0 * nn, n, n, K, J, g, c);
        }
        if constexpr (PREFETCH) Fn[K][NT] = load_tile<TR, EXACT>(Bb + k0 * nm, n, m, K, 0, g, c);
    }
    if constexpr (PREFETCH) Rn = load_tile<TR, EXACT>(Rb + k0 * mm, m, m, 0, 0, g, c, S(1));
    // DARE: the previous iteration's gain (this lane's column), convergence state
    S xprev[DARE ? 16 : 1];
    S dprev = TR::huge();
    int stall = 0, iters_done = 0;
    bool stop = false, conv = false;
    if constexpr (DARE) {
#pragma unroll
        for (int u_ = 0; u_ < 16; ++u_) xprev[u_] = S(0);
    }

#ifdef ZM_TILED_LAB
    unsigned long long zt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, zt_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(zt_last)::"memory");
#endif
    for (int kk_ = T - 1; kk_ >= 0; --kk_) {
        const int k = DARE ? 0 : kk_;          // time index of this step's operands and of L_k
        if constexpr (PREFETCH) {
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J <= NT; ++J) F[K][J] = Fn[K][J];
            Rt = Rn;
        } else {
#pragma unroll
            for (int K = 0; K < NT; ++K) {
#pragma unroll
                for (int J = 0; J < NT; ++J) F[K][J] = load_tile<TR, EXACT>(Ab + k * nn, n, n, K, J, g, c);
                F[K][NT] = load_tile<TR, EXACT>(Bb + k * nm, n, m, K, 0, g, c);
            }
            Rt = load_tile<TR, EXACT>(Rb + k * mm, m, m, 0, 0, g, c, S(1));
        }
        ZT_STAMP(0)   // loop head: waits for the operands of this step (issued during the previous step)
        // Y_B = V^T B
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            f4 acc = TR::zero();
#pragma unroll
            for (int K = 0; K < NT; ++K) acc = op<TR>(V[K][I], F[K][NT], acc);
            Y[I][NT] = acc;
        }
        // S = Y_B^T F + [0 | R]  ->  LDS, one b128 per tile (rows 4g..4g+3 of column 16J+c)
#pragma unroll
        for (int J = 0; J <= NT; ++J) {
            f4 acc = (J == NT) ? Rt : TR::zero();
#pragma unroll
            for (int K = 0; K < NT; ++K) acc = op<TR>(Y[K][NT], F[K][J], acc);
            TR::tile_to_lds_T(Sc + 16 * J * TLD, acc, g, c);
        }
        // tiles that are needed transposed: B_K, Y_B,I, R
#pragma unroll
        for (int K = 0; K < NT; ++K) {
            TR::tile_to_lds(Tb[K], F[K][NT], g, c);
            TR::tile_to_lds(Tb[NT + K], Y[K][NT], g, c);
        }
        TR::tile_to_lds(Tb[2 * NT], Rt, g, c);
        ZT_STAMP(1)   // Y_B, S (144 MFMAs at NT = 4) and the LDS writes of S and the transposed tiles
        t_lds_sync();
        // column j of Sux and column c of Suu into registers
        S x[16], u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            x[i] = Sc[jl * TLD + i];
            u[i] = Sc[(NP + c) * TLD + i];
        }
        // Y_A = V^T A (4 NT^3 MFMAs, 32 cycles of matrix pipe each) is independent of the solve.  A wave issues in order, so
        // the solve's VALU work hides under these MFMAs only if the two are interleaved finely: ya(t) issues MFMA number t of
        // the Y_A sequence (row-tile I outermost) and closes the scheduling region, so the order written here is the order
        // executed: one MFMA per elimination / substitution unit.  When row-tile I is complete the V tiles it read (column I)
        // are dead and take Q_k[*][I], the accumulator init of V'.
        ZT_STAMP(2)   // solve operands from LDS
        int yq = 0;   // running MFMA number: a constant at every call once the loops below are unrolled
        auto ya = [&]() {
            const int t = yq++;
            if (t < 4 * NT * NT * NT) {
                // column tile J fastest: consecutive MFMAs accumulate into NT different tiles (a chain of 16 MFMAs into one tile runs
                // at the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32, not at its 32-cycle issue rate)
                const int J_ = t % NT, s_ = (t / NT) & 3, K_ = (t / (4 * NT)) % NT, I_ = t / (4 * NT * NT);
                const f4 a_ = (K_ == 0 && s_ == 0) ? TR::zero() : Y[I_][J_];
                Y[I_][J_] = TR::mfma(V[K_][I_][s_], F[K_][J_][s_], a_);
                if ((t + 1) % (4 * NT * NT) == 0) {
#pragma unroll
                    for (int K = 0; K < NT; ++K) V[K][I_] = load_tile<TR, EXACT>(Qb + k * nn, n, n, K, I_, g, c);
                }
            }
        };
        __builtin_amdgcn_sched_barrier(0);

        // ---- L = solve(Suu, Sux).  Fast path: LU WITHOUT row exchanges on registers (row operations are lane-local, the
        //      multipliers wave-uniform).  Accepted only if every multiplier stayed <= 4 in magnitude (partial pivoting keeps
        //      them <= 1; for the symmetric positive definite Suu of a regular LQR problem they are far below that), so the
        //      result differs from jnp.linalg.solve's pivoted LU by rounding only.  Otherwise: pivoted LU in LDS (below).
        unsigned long long bad = 0ull;
        S pinv[16];
      if constexpr (TR::FUSED_DPP) {
        // fp32 MFMA and fp32 VALU share the vector ALU (the fp32 "matrix peak" IS the vector rate): the solve's VALU work does not
        // hide under the Y_A MFMAs, it ADDS to them (stamps: 16.1 k cycles for this phase = 8.2 k of MFMA + the vector
        // instructions).  What counts is therefore the NUMBER of vector instructions: the broadcasts ride on the FMAs
        // (TR::elim_batch / fma_bc: v_fmac_f32_dpp row_newbcast), three instructions per eliminated row instead of five, one per
        // substitution term instead of two, no wave-uniform copies in scalar registers.
        bool okm = true;
        solve_elim<TR, 0>(x, u, pinv, okm, bad);
        bad |= __ballot(!(TR::abs(pinv[15]) < TR::huge())) & (0x0001000100010001ull << 15);
        solve_subst<TR, 15>(x, u, pinv);
#pragma unroll
        for (int t = 0; t < 4 * NT * NT * NT; ++t) ya();
      } else {
        // Every stage below is a run of mutually INDEPENDENT instructions (all multipliers of a column, then all their wave-uniform
        // copies, then all row updates), with one Y_A MFMA after every ~4 of them: issued in the order written (ya() closes the
        // scheduling region), a unit-by-unit order (multiplier -> readlane -> its two FMAs) stalls on each dependency and the
        // phase ran at twice the MFMA time (stamps: 16.1 k cycles against 8.2 k of MFMA issue; tools/k1t_lab.hip).
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const S inv = TR::rcp(u[kk]);
            pinv[kk] = TR::readlane(inv, kk);
            ya();
            bool okm = true;
            // rows in batches of four: 4 multipliers, their 4 wave-uniform copies, 8 updates -- independent instructions inside a
            // stage, few enough uniform values alive at a time for the scalar register file
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {      // (constant trip counts: the unroller must see them before kk is a constant)
                const int r0 = kk + 1 + 4 * bq;
                if (r0 >= 16) continue;
                S mv[4], ms[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r0 + q < 16) {
                        mv[q] = u[r0 + q] * inv;
                        okm &= (TR::abs(mv[q]) <= S(4));   // NaN compares false: a NaN multiplier fails the check
                    }
                ya();
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r0 + q < 16) ms[q] = TR::readlane(mv[q], kk);
                ya();
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r0 + q < 16) {
                        x[r0 + q] = TR::fma(-ms[q], x[kk], x[r0 + q]);
                        u[r0 + q] = TR::fma(-ms[q], u[kk], u[r0 + q]);
                        if (q == 1) ya();
                    }
                ya();
            }
            bad |= __ballot(!okm) & (0x0001000100010001ull << kk);   // only lane c == kk of each group holds the column's multipliers
            ya();
        }
        bad |= __ballot(!(TR::abs(pinv[15]) < TR::huge()));
#pragma unroll
        for (int kk = 15; kk >= 0; --kk) {
            S a0 = x[kk], a1 = S(0);                      // two partial chains over the row
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                const int r0 = kk + 1 + 4 * bq;
                if (r0 >= 16) continue;
                S ur[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r0 + q < 16) ur[q] = TR::readlane(u[kk], r0 + q);
                ya();
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r0 + q < 16) {
                        if (q & 1)
                            a1 = TR::fma(-ur[q], x[r0 + q], a1);
                        else
                            a0 = TR::fma(-ur[q], x[r0 + q], a0);
                    }
                ya();
            }
            x[kk] = (a0 + a1) * pinv[kk];
            ya();
        }
      }
        // (readlane path) ya() is called 260 times above (16 + 36 x 3 + 32 + 16 in the elimination, 36 x 2 + 16 in the substitution): the whole Y_A
        // sequence (256 MFMAs at NT = 4) is issued.  The count must stay a compile-time fact -- a loop that the compiler does not
        // unroll turns the tile indices inside ya() into run-time values and the register-resident tiles into scratch arrays.
        static_assert(4 * NT * NT * NT <= 256, "the Y_A sequence is at most 256 MFMAs");
        // The pivoted path below overwrites x, so the optimiser would sink the whole substitution past the branch -- away from
        // the MFMAs it is meant to hide under.  Pin the values here.
#pragma unroll
        for (int u_ = 0; u_ < 16; ++u_) TR::pin(x[u_]);
        ZT_STAMP(3)   // solve interleaved with Y_A (256 MFMAs)
        if (bad != 0ull) {   // wave-uniform, rare: LU with partial pivoting on the copy still in LDS (getrf / getrs order)
#define S_(r_, j_) Sc[(j_) * TLD + (r_)]
#pragma unroll 1
            for (int kk = 0; kk < 16; ++kk) {
                S pv = (c >= kk) ? TR::abs(S_(c, NP + kk)) : S(-1);
                int pi = c;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const S ov = __shfl_xor(pv, off, 16);
                    const int oi = __shfl_xor(pi, off, 16);
                    const bool take = (ov > pv) || (ov == pv && oi < pi);  // first largest entry, as isamax
                    pv = take ? ov : pv;
                    pi = take ? oi : pi;
                }
                const int p = __builtin_amdgcn_readfirstlane(pi);
                {  // swap rows kk and p (a no-op when p == kk)
                    const S a0 = S_(kk, jl), b0 = S_(p, jl);
                    const S a1 = S_(kk, NP + c), b1 = S_(p, NP + c);
                    t_lds_sync();
                    if (own_x) {
                        S_(kk, jl) = b0;
                        S_(p, jl) = a0;
                    }
                    if (own_u) {
                        S_(kk, NP + c) = b1;
                        S_(p, NP + c) = a1;
                    }
                    t_lds_sync();
                }
                const S inv = S(1) / S_(kk, NP + kk);
                const S pj = S_(kk, jl);
                const S pu = S_(kk, NP + c);
#pragma unroll 1
                for (int r = kk + 1; r < 16; ++r) {
                    const S mr = S_(r, NP + kk) * inv;
                    const S xj = S_(r, jl);
                    const S xu = S_(r, NP + c);
                    t_lds_sync();
                    if (own_x) S_(r, jl) = xj - mr * pj;
                    if (own_u && c > kk) S_(r, NP + c) = xu - mr * pu;
                }
                t_lds_sync();
            }
#pragma unroll
            for (int kk = 15; kk >= 0; --kk) {
                S acc = S_(kk, jl);
#pragma unroll
                for (int r = kk + 1; r < 16; ++r) acc -= S_(kk, NP + r) * x[r];
                x[kk] = acc / S_(kk, NP + kk);
            }
            t_lds_sync();
#undef S_
        }
        if constexpr (DARE) {   // has the gain stopped changing?  (wave-uniform decision; the step is finished either way)
            ++iters_done;
            if ((iters_done & 7) == 0 || kk_ == 0) {
                S dmax = S(0), smax = S(0);
#pragma unroll
                for (int u_ = 0; u_ < 16; ++u_) {
                    const S d_ = TR::abs(x[u_] - xprev[u_]), a_ = TR::abs(x[u_]);
                    dmax = (lane < n && d_ > dmax) || (lane < n && !(d_ == d_)) ? d_ : dmax;      // NaN sticks
                    smax = (lane < n && a_ > smax) ? a_ : smax;
                }
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const S od = __shfl_xor(dmax, off, 64), os = __shfl_xor(smax, off, 64);
                    dmax = (od > dmax || !(od == od)) ? od : dmax;
                    smax = os > smax ? os : smax;
                }
                if (!(dmax > dr.tol * smax)) {           // converged (or NaN: stop; the caller sees the non-finite gain)
                    conv = true;
                    stop = true;
                } else {
                    stall = (dmax >= dprev && dmax <= S(1e-9) * smax) ? stall + 1 : 0;   // rounding floor reached
                    if (stall >= 3) {
                        conv = true;
                        stop = true;
                    }
                    dprev = dmax;
                }
            }
#pragma unroll
            for (int u_ = 0; u_ < 16; ++u_) xprev[u_] = x[u_];
        }
        // L_k to HBM (row u: 64 consecutive floats across the wave), -L back to LDS (b128) for the tile reads
        if (lane < n) {
#pragma unroll
            for (int u_ = 0; u_ < 16; ++u_)
                if (EXACT || u_ < m) Lb[k * nm + (long)u_ * n + lane] = x[u_];
        }
        if (own_x) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Sc[jl * TLD + i] = -x[i];
        }
        t_lds_sync();
        f4 NL[NT], NRL[NT];
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) NL[J][r] = Sc[(16 * J + c) * TLD + TR::row(g, r)];
        ZT_STAMP(4)   // L store, -L through LDS back as tiles
        if constexpr (PREFETCH) {  // operands of step k-1, fetched under the ~15k cycles of MFMAs that follow (the last iteration
           // re-reads step 0: no branch around the loads); issued only now so that they do not hold 84 registers during the solve
            const int kn = (!DARE && k > 0) ? k - 1 : 0;
#pragma unroll
            for (int K = 0; K < NT; ++K) {
#pragma unroll
                for (int J = 0; J < NT; ++J) Fn[K][J] = load_tile<TR, EXACT>(Ab + kn * nn, n, n, K, J, g, c);
                Fn[K][NT] = load_tile<TR, EXACT>(Bb + kn * nm, n, m, K, 0, g, c);
            }
            Rn = load_tile<TR, EXACT>(Rb + kn * mm, m, m, 0, 0, g, c, S(1));
        }
        __builtin_amdgcn_sched_barrier(0);   // all 84 loads in flight before the MFMA stream starts
        ZT_STAMP(5)   // issue of the next step's operand loads
        // -RL = R (-L)
        {
            const f4 RT = TR::tile_from_lds_T(Tb[2 * NT], g, c);
#pragma unroll
            for (int J = 0; J < NT; ++J) NRL[J] = op<TR>(RT, NL[J], TR::zero());
        }
        // Acl = A + B(-L)  (in place),  W = Y_A + Y_B(-L)  (in place)
#pragma unroll
        for (int K = 0; K < NT; ++K) {
            const f4 BT = TR::tile_from_lds_T(Tb[K], g, c);
            const f4 YT = TR::tile_from_lds_T(Tb[NT + K], g, c);
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                F[K][J] = op<TR>(BT, NL[J], F[K][J]);
                Y[K][J] = op<TR>(YT, NL[J], Y[K][J]);
            }
        }
        // V' = Q + (-L)^T(-RL) + W^T Acl
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                f4 acc = op<TR>(NL[I], NRL[J], V[I][J]);
#pragma unroll
                for (int K = 0; K < NT; ++K) acc = op<TR>(Y[K][I], F[K][J], acc);
                V[I][J] = acc;
            }
        t_lds_sync();  // Sc / Tb are rewritten by the next step
        ZT_STAMP(6)   // -RL, Acl, W, V' (464 MFMAs)
        if constexpr (DARE) {
            if (stop) break;
        }
    }
    if constexpr (DARE) {
        if (dr.P) {
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * K + TR::row(g, r), j = 16 * J + c;
                        if (i < n && j < n) dr.P[traj * nn + (long)i * n + j] = V[K][J][r];
                    }
        }
        if (dr.iters && lane == 0) dr.iters[traj] = conv ? iters_done : -iters_done;
    }
#ifdef ZM_TILED_LAB
    if (lane == 0) {
        for (int q = 0; q < 7; ++q) atomicAdd(&zm_tiled_stamps[q], zt_acc[q]);
        atomicAdd(&zm_tiled_stamps[7], 1ull);
    }
#endif
}


template <class TR, int NT, bool PREFETCH = true>
static int launch_tiled(const typename TR::S* A, const typename TR::S* B, const typename TR::S* Q, const typename TR::S* R,
                        typename TR::S* L, int64_t batch, int T, int n, int m, hipStream_t st) {
    const bool exact = (n == 16 * NT) && (m == 16) && !zm::lab_env("ZOPT_AMD_TILED_GENERIC");
    if (exact)
        hipLaunchKernelGGL((lqr_backward_tiled<TR, NT, true, PREFETCH>), dim3((unsigned)batch), dim3(64), 0, st, A, B, Q, R, L,
                           (long)batch, T, n, m);
    else
        hipLaunchKernelGGL((lqr_backward_tiled<TR, NT, false, PREFETCH>), dim3((unsigned)batch), dim3(64), 0, st, A, B, Q, R, L,
                           (long)batch, T, n, m);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// DARE launcher (fp64): value iteration on time-invariant operands, see TiledDareArgs
template <class TR, int NT, bool PREFETCH = true>
static int launch_tiled_dare(const typename TR::S* A, const typename TR::S* B, const typename TR::S* Q, const typename TR::S* R,
                             typename TR::S* L, typename TR::S* P, int* iters, int64_t batch, int max_iter, int n, int m,
                             typename TR::S tol, hipStream_t st) {
    const TiledDareArgs<typename TR::S> dr{P, iters, tol};
    hipLaunchKernelGGL((lqr_backward_tiled<TR, NT, false, PREFETCH, true>), dim3((unsigned)batch), dim3(64), 0, st, A, B, Q, R, L,
                       (long)batch, max_iter, n, m, dr);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

}  // namespace zm
