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

template <class TR, int NT, bool EXACT>
__global__ __launch_bounds__(64) void lqr_backward_tiled(const typename TR::S* __restrict__ A, const typename TR::S* __restrict__ B,
                                                         const typename TR::S* __restrict__ Q, const typename TR::S* __restrict__ R,
                                                         typename TR::S* __restrict__ L, const long batch, const int T, const int n_,
                                                         const int m_) {
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
    const S* Ab = A + traj * T * nn;
    const S* Bb = B + traj * T * nm;
    const S* Qb = Q + traj * T * nn;
    const S* Rb = R + traj * T * mm;
    S* Lb = L + traj * T * nm;

    f4 V[NT][NT], F[NT][NT + 1], Fn[NT][NT + 1], Y[NT][NT + 1], Rt, Rn;
    // terminal value = last stage cost (lqrUtils.py:172); operands of the first step
#pragma unroll
    for (int K = 0; K < NT; ++K) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            V[K][J] = load_tile<TR, EXACT>(Qb + (long)(T - 1) * nn, n, n, K, J, g, c);
            Fn[K][J] = load_tile<TR, EXACT>(Ab + (long)(T - 1) * nn, n, n, K, J, g, c);
        }
        Fn[K][NT] = load_tile<TR, EXACT>(Bb + (long)(T - 1) * nm, n, m, K, 0, g, c);
    }
    Rn = load_tile<TR, EXACT>(Rb + (long)(T - 1) * mm, m, m, 0, 0, g, c, S(1));

    for (int k = T - 1; k >= 0; --k) {
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J <= NT; ++J) F[K][J] = Fn[K][J];
        Rt = Rn;
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
        int yq = 0;   // running MFMA number: a constant at every call once the loops below are unrolled
        auto ya = [&]() {
            const int t = yq++;
            if (t < 4 * NT * NT * NT) {
                const int s_ = t & 3, K_ = (t >> 2) % NT, J_ = ((t >> 2) / NT) % NT, I_ = (t >> 2) / (NT * NT);
                const f4 a_ = (K_ == 0 && s_ == 0) ? TR::zero() : Y[I_][J_];
                Y[I_][J_] = TR::mfma(V[K_][I_][s_], F[K_][J_][s_], a_);
                if ((t + 1) % (4 * NT * NT) == 0) {
#pragma unroll
                    for (int K = 0; K < NT; ++K) V[K][I_] = load_tile<TR, EXACT>(Qb + k * nn, n, n, K, I_, g, c);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        __builtin_amdgcn_sched_barrier(0);

        // ---- L = solve(Suu, Sux).  Fast path: LU WITHOUT row exchanges on registers (row operations are lane-local, the
        //      multipliers wave-uniform).  Accepted only if every multiplier stayed <= 4 in magnitude (partial pivoting keeps
        //      them <= 1; for the symmetric positive definite Suu of a regular LQR problem they are far below that), so the
        //      result differs from jnp.linalg.solve's pivoted LU by rounding only.  Otherwise: pivoted LU in LDS (below).
        unsigned long long bad = 0ull;
        S pinv[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const S inv = TR::rcp(u[kk]);
            pinv[kk] = TR::readlane(inv, kk);
            ya();
#pragma unroll
            for (int r = kk + 1; r < 16; ++r) {
                const S mv = u[r] * inv;
                bad |= __ballot(!(TR::abs(mv) <= S(4))) & (0x0001000100010001ull << kk);
                const S ms = TR::readlane(mv, kk);
                x[r] = TR::fma(-ms, x[kk], x[r]);
                u[r] = TR::fma(-ms, u[kk], u[r]);
                ya();
                if (((kk * 15 - kk * (kk - 1) / 2 + (r - kk - 1)) & 1) != 0) ya();   // an elimination unit is ~1.5 MFMAs long,
                                                                                     // a substitution unit ~0.5
            }
        }
        bad |= __ballot(!(TR::abs(pinv[15]) < TR::huge()));
#pragma unroll
        for (int kk = 15; kk >= 0; --kk) {
            S acc = x[kk];
#pragma unroll
            for (int r = kk + 1; r < 16; ++r) {
                acc = TR::fma(-TR::readlane(u[kk], r), x[r], acc);
                if (((kk * 15 - kk * (kk - 1) / 2 + (r - kk - 1)) & 1) == 0) ya();
            }
            x[kk] = acc * pinv[kk];
        }
        static_assert(4 * NT * NT * NT <= 256, "the 256 ya() calls above must cover the Y_A sequence");
        // The pivoted path below overwrites x, so the optimiser would sink the whole substitution past the branch -- away from
        // the MFMAs it is meant to hide under.  Pin the values here.
#pragma unroll
        for (int u_ = 0; u_ < 16; ++u_) TR::pin(x[u_]);
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
        {  // operands of step k-1, fetched under the ~15k cycles of MFMAs that follow (the last iteration re-reads step 0:
           // no branch around the loads); issued only now so that they do not hold 84 registers during the solve
            const int kn = k > 0 ? k - 1 : 0;
#pragma unroll
            for (int K = 0; K < NT; ++K) {
#pragma unroll
                for (int J = 0; J < NT; ++J) Fn[K][J] = load_tile<TR, EXACT>(Ab + kn * nn, n, n, K, J, g, c);
                Fn[K][NT] = load_tile<TR, EXACT>(Bb + kn * nm, n, m, K, 0, g, c);
            }
            Rn = load_tile<TR, EXACT>(Rb + kn * mm, m, m, 0, 0, g, c, S(1));
        }
        __builtin_amdgcn_sched_barrier(0);   // all 84 loads in flight before the MFMA stream starts
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
    }
}


template <class TR, int NT>
static int launch_tiled(const typename TR::S* A, const typename TR::S* B, const typename TR::S* Q, const typename TR::S* R,
                        typename TR::S* L, int64_t batch, int T, int n, int m, hipStream_t st) {
    const bool exact = (n == 16 * NT) && (m == 16) && !getenv("ZOPT_AMD_TILED_GENERIC");
    if (exact)
        hipLaunchKernelGGL((lqr_backward_tiled<TR, NT, true>), dim3((unsigned)batch), dim3(64), 0, st, A, B, Q, R, L, (long)batch, T, n,
                           m);
    else
        hipLaunchKernelGGL((lqr_backward_tiled<TR, NT, false>), dim3((unsigned)batch), dim3(64), 0, st, A, B, Q, R, L, (long)batch, T,
                           n, m);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

}  // namespace zm
