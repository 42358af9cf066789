// K1-T  lqr_backward_tiled_f32 -- finite-horizon LQR backward Riccati sweep for LARGE states (n <= 64, m <= 16), fp32, gfx950.
//
// Replaces zopt/lqrUtils.py:144-173 discreteFiniteHorizonLqr at the "large-state stress" shape of BASELINE configs[4]
// (n = 64, m = 16, T = 200, fp32): per trajectory
//     V <- Q[T-1];  for k = T-1..0:  L_k = solve(R_k + B_k^T V B_k, B_k^T V A_k)                     (:168)
//                                    V   = Q_k + L_k^T R_k L_k + (A_k - B_k L_k)^T V (A_k - B_k L_k)   (:169, Joseph form)
// At this shape the sweep is a chain of dense 64x64 contractions (1.7 Mflop per step against 42 kB of operands, 40 flop/B):
// it is bound by the fp32 matrix pipe, not by HBM, so the products run on v_mfma_f32_16x16x4_f32.
//
// One wave64 per trajectory (one wave per SIMD, all 512 registers): the value matrix V (16 tiles), the step's
// F = [A_k | B_k] (20 tiles) and Y = V^T F (20 tiles) live in registers as 16x16 tiles in the MFMA accumulator layout
//     lane l = (g = l >> 4, c = l & 15),  tile register r  <->  X[4g + r][c]                     ("D layout").
// With the K index of a product permuted consistently (K-step s of lane group g <-> row 4g+s) a D-layout tile is directly the
// B operand of the next product and, read as the A operand, its transpose:  op(X, Y) = X^T Y  costs 4 MFMAs and no lane
// movement.  Per step (NT = n/16 tile rows, u-index = one tile):
//     Y_B = V^T B                               4 NT^2   MFMA
//     S   = Y_B^T F + [0 | R] = [Sux | Suu]     4 NT(NT+1)
//     Y_A = V^T A                               4 NT^3
//     L   = solve(Suu, Sux)                     LU with partial pivoting, resident in LDS, one lane per column
//     Acl = A + B(-L),  -RL = R(-L)             4 NT^2 + 4 NT     (B^T, R^T tiles: transposed through LDS)
//     W   = Y_A + Y_B(-L) = V^T Acl             4 NT^2
//     V'  = Q + (-L)^T(-RL) + W^T Acl           4 NT^2 + 4 NT^3           (exact for nonsymmetric V, Q, R as well)
// = 864 MFMAs per step at NT = 4.  Operands of step k-1 are fetched into a second register set while step k computes.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "zm_common.h"

namespace zm {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void t_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc + X^T Y for D-layout tiles
__device__ __forceinline__ f4 op(const f4 x, const f4 y, f4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], y[s], acc, 0, 0, 0);
    return acc;
}

// tile (K, J) of a row-major (nrows x ncols) matrix in D layout; out-of-range elements read as `diag` on the diagonal, else 0
template <bool EXACT>
__device__ __forceinline__ f4 load_tile(const float* __restrict__ X, const int nrows, const int ncols, const int K, const int J,
                                        const int g, const int c, const float diag = 0.f) {
    f4 t;
    const int col = 16 * J + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * K + 4 * g + r;
        if constexpr (EXACT) {
            t[r] = X[row * ncols + col];
        } else {
            const bool ok = row < nrows && col < ncols;
            const float v = X[ok ? row * ncols + col : 0];
            t[r] = ok ? v : ((row == col) ? diag : 0.f);
        }
    }
    return t;
}

constexpr int TLD = 20;  // row stride of a transpose buffer (floats): 80 B rows keep the b128 accesses 16 B-aligned

// D-layout tile -> LDS, transposed: buf[col * TLD + row]  (one b128 per lane); reading it back with the roles of (g, c)
// swapped gives the transposed tile, reading TLD-strided columns gives one matrix column per lane.
__device__ __forceinline__ void tile_to_lds_T(float* buf, const f4 t, const int g, const int c) {
    *reinterpret_cast<f4*>(buf + c * TLD + 4 * g) = t;
}
// D-layout tile -> LDS row-major buf[row * TLD + col]; tile_from_lds_T then returns the transpose: out[r] = X[c][4g + r]
__device__ __forceinline__ void tile_to_lds(float* buf, const f4 t, const int g, const int c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(4 * g + r) * TLD + c] = t[r];
}
__device__ __forceinline__ f4 tile_from_lds_T(const float* buf, const int g, const int c) {
    return *reinterpret_cast<const f4*>(buf + c * TLD + 4 * g);
}

__device__ __forceinline__ float readlane_f(const float v, const int l) {   // wave-uniform copy of lane l's value
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ __forceinline__ float rcp_nr(const float a) {   // 1/a to fp32 rounding: hardware estimate (1 ulp) + one Newton step
    const float y = __builtin_amdgcn_rcpf(a);
    return __builtin_fmaf(__builtin_fmaf(-a, y, 1.0f), y, y);
}

template <int NT, bool EXACT>
__global__ __launch_bounds__(64) void lqr_backward_tiled_f32(const float* __restrict__ A, const float* __restrict__ B,
                                                             const float* __restrict__ Q, const float* __restrict__ R,
                                                             float* __restrict__ L, const long batch, const int T, const int n_,
                                                             const int m_) {
    constexpr int NP = 16 * NT;  // padded state dimension
    const int n = EXACT ? NP : n_, m = EXACT ? 16 : m_;
    // Solve buffer, column-major: element (row u, column j) of [Sux | Suu] at Sc[j * TLD + u]; columns NP..NP+15 are Suu.
    __shared__ __attribute__((aligned(16))) float Sc[(NP + 16) * TLD];
    __shared__ __attribute__((aligned(16))) float Tb[2 * NT + 1][16 * TLD];
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
    const float* Ab = A + traj * T * nn;
    const float* Bb = B + traj * T * nm;
    const float* Qb = Q + traj * T * nn;
    const float* Rb = R + traj * T * mm;
    float* Lb = L + traj * T * nm;

    f4 V[NT][NT], F[NT][NT + 1], Fn[NT][NT + 1], Y[NT][NT + 1], Rt, Rn;
    // terminal value = last stage cost (lqrUtils.py:172); operands of the first step
#pragma unroll
    for (int K = 0; K < NT; ++K) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            V[K][J] = load_tile<EXACT>(Qb + (long)(T - 1) * nn, n, n, K, J, g, c);
            Fn[K][J] = load_tile<EXACT>(Ab + (long)(T - 1) * nn, n, n, K, J, g, c);
        }
        Fn[K][NT] = load_tile<EXACT>(Bb + (long)(T - 1) * nm, n, m, K, 0, g, c);
    }
    Rn = load_tile<EXACT>(Rb + (long)(T - 1) * mm, m, m, 0, 0, g, c, 1.f);

    for (int k = T - 1; k >= 0; --k) {
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J <= NT; ++J) F[K][J] = Fn[K][J];
        Rt = Rn;
        // Y_B = V^T B
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int K = 0; K < NT; ++K) acc = op(V[K][I], F[K][NT], acc);
            Y[I][NT] = acc;
        }
        // S = Y_B^T F + [0 | R]  ->  LDS, one b128 per tile (rows 4g..4g+3 of column 16J+c)
#pragma unroll
        for (int J = 0; J <= NT; ++J) {
            f4 acc = (J == NT) ? Rt : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int K = 0; K < NT; ++K) acc = op(Y[K][NT], F[K][J], acc);
            tile_to_lds_T(Sc + 16 * J * TLD, acc, g, c);
        }
        // tiles that are needed transposed: B_K, Y_B,I, R
#pragma unroll
        for (int K = 0; K < NT; ++K) {
            tile_to_lds(Tb[K], F[K][NT], g, c);
            tile_to_lds(Tb[NT + K], Y[K][NT], g, c);
        }
        tile_to_lds(Tb[2 * NT], Rt, g, c);
        t_lds_sync();
        // column j of Sux and column c of Suu into registers
        float x[16], u[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 a = *reinterpret_cast<const f4*>(Sc + jl * TLD + 4 * q);
            const f4 b = *reinterpret_cast<const f4*>(Sc + (NP + c) * TLD + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[4 * q + r] = a[r];
                u[4 * q + r] = b[r];
            }
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
                const f4 a_ = (K_ == 0 && s_ == 0) ? f4{0.f, 0.f, 0.f, 0.f} : Y[I_][J_];
                Y[I_][J_] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[K_][I_][s_], F[K_][J_][s_], a_, 0, 0, 0);
                if ((t + 1) % (4 * NT * NT) == 0) {
#pragma unroll
                    for (int K = 0; K < NT; ++K) V[K][I_] = load_tile<EXACT>(Qb + k * nn, n, n, K, I_, g, c);
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
        float pinv[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float inv = rcp_nr(u[kk]);
            pinv[kk] = readlane_f(inv, kk);
            ya();
#pragma unroll
            for (int r = kk + 1; r < 16; ++r) {
                const float mv = u[r] * inv;
                bad |= __ballot(!(__builtin_fabsf(mv) <= 4.0f)) & (0x0001000100010001ull << kk);
                const float ms = readlane_f(mv, kk);
                x[r] = __builtin_fmaf(-ms, x[kk], x[r]);
                u[r] = __builtin_fmaf(-ms, u[kk], u[r]);
                ya();
                if (((kk * 15 - kk * (kk - 1) / 2 + (r - kk - 1)) & 1) != 0) ya();   // an elimination unit is ~1.5 MFMAs long,
                                                                                     // a substitution unit ~0.5
            }
        }
        bad |= __ballot(!(__builtin_fabsf(pinv[15]) < 3.0e38f));
#pragma unroll
        for (int kk = 15; kk >= 0; --kk) {
            float acc = x[kk];
#pragma unroll
            for (int r = kk + 1; r < 16; ++r) {
                acc = __builtin_fmaf(-readlane_f(u[kk], r), x[r], acc);
                if (((kk * 15 - kk * (kk - 1) / 2 + (r - kk - 1)) & 1) == 0) ya();
            }
            x[kk] = acc * pinv[kk];
        }
        static_assert(4 * NT * NT * NT <= 256, "the 256 ya() calls above must cover the Y_A sequence");
        // The pivoted path below overwrites x, so the optimiser would sink the whole substitution past the branch -- away from
        // the MFMAs it is meant to hide under.  Pin the values here.
#pragma unroll
        for (int u_ = 0; u_ < 16; ++u_) asm volatile("" : "+v"(x[u_]));
        if (bad != 0ull) {   // wave-uniform, rare: LU with partial pivoting on the copy still in LDS (getrf / getrs order)
#define S_(r_, j_) Sc[(j_) * TLD + (r_)]
#pragma unroll 1
            for (int kk = 0; kk < 16; ++kk) {
                float pv = (c >= kk) ? __builtin_fabsf(S_(c, NP + kk)) : -1.f;
                int pi = c;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const float ov = __shfl_xor(pv, off, 16);
                    const int oi = __shfl_xor(pi, off, 16);
                    const bool take = (ov > pv) || (ov == pv && oi < pi);  // first largest entry, as isamax
                    pv = take ? ov : pv;
                    pi = take ? oi : pi;
                }
                const int p = __builtin_amdgcn_readfirstlane(pi);
                {  // swap rows kk and p (a no-op when p == kk)
                    const float a0 = S_(kk, jl), b0 = S_(p, jl);
                    const float a1 = S_(kk, NP + c), b1 = S_(p, NP + c);
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
                const float inv = 1.0f / S_(kk, NP + kk);
                const float pj = S_(kk, jl);
                const float pu = S_(kk, NP + c);
#pragma unroll 1
                for (int r = kk + 1; r < 16; ++r) {
                    const float mr = S_(r, NP + kk) * inv;
                    const float xj = S_(r, jl);
                    const float xu = S_(r, NP + c);
                    t_lds_sync();
                    if (own_x) S_(r, jl) = xj - mr * pj;
                    if (own_u && c > kk) S_(r, NP + c) = xu - mr * pu;
                }
                t_lds_sync();
            }
#pragma unroll
            for (int kk = 15; kk >= 0; --kk) {
                float acc = S_(kk, jl);
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
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f4*>(Sc + jl * TLD + 4 * q) = f4{-x[4 * q], -x[4 * q + 1], -x[4 * q + 2], -x[4 * q + 3]};
        }
        t_lds_sync();
        f4 NL[NT], NRL[NT];
#pragma unroll
        for (int J = 0; J < NT; ++J) NL[J] = *reinterpret_cast<const f4*>(Sc + (16 * J + c) * TLD + 4 * g);
        {  // operands of step k-1, fetched under the ~15k cycles of MFMAs that follow (the last iteration re-reads step 0:
           // no branch around the loads); issued only now so that they do not hold 84 registers during the solve
            const int kn = k > 0 ? k - 1 : 0;
#pragma unroll
            for (int K = 0; K < NT; ++K) {
#pragma unroll
                for (int J = 0; J < NT; ++J) Fn[K][J] = load_tile<EXACT>(Ab + kn * nn, n, n, K, J, g, c);
                Fn[K][NT] = load_tile<EXACT>(Bb + kn * nm, n, m, K, 0, g, c);
            }
            Rn = load_tile<EXACT>(Rb + kn * mm, m, m, 0, 0, g, c, 1.f);
        }
        __builtin_amdgcn_sched_barrier(0);   // all 84 loads in flight before the MFMA stream starts
        // -RL = R (-L)
        {
            const f4 RT = tile_from_lds_T(Tb[2 * NT], g, c);
#pragma unroll
            for (int J = 0; J < NT; ++J) NRL[J] = op(RT, NL[J], f4{0.f, 0.f, 0.f, 0.f});
        }
        // Acl = A + B(-L)  (in place),  W = Y_A + Y_B(-L)  (in place)
#pragma unroll
        for (int K = 0; K < NT; ++K) {
            const f4 BT = tile_from_lds_T(Tb[K], g, c);
            const f4 YT = tile_from_lds_T(Tb[NT + K], g, c);
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                F[K][J] = op(BT, NL[J], F[K][J]);
                Y[K][J] = op(YT, NL[J], Y[K][J]);
            }
        }
        // V' = Q + (-L)^T(-RL) + W^T Acl
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                f4 acc = op(NL[I], NRL[J], V[I][J]);
#pragma unroll
                for (int K = 0; K < NT; ++K) acc = op(Y[K][I], F[K][J], acc);
                V[I][J] = acc;
            }
        t_lds_sync();  // Sc / Tb are rewritten by the next step
    }
}

template <int NT>
static int launch_tiled(const float* A, const float* B, const float* Q, const float* R, float* L, int64_t batch, int T, int n,
                        int m, hipStream_t st) {
    const bool exact = (n == 16 * NT) && (m == 16) && !getenv("ZOPT_AMD_TILED_GENERIC");
    if (exact)
        hipLaunchKernelGGL((lqr_backward_tiled_f32<NT, true>), dim3((unsigned)batch), dim3(64), 0, st, A, B, Q, R, L, (long)batch,
                           T, n, m);
    else
        hipLaunchKernelGGL((lqr_backward_tiled_f32<NT, false>), dim3((unsigned)batch), dim3(64), 0, st, A, B, Q, R, L,
                           (long)batch, T, n, m);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

}  // namespace zm

extern "C" int zm_lqr_backward_f32(const float* A, const float* B, const float* Q, const float* R, float* L, int64_t batch,
                                   int T, int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!A || !B || !Q || !R || !L) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_f32: null pointer");
    if (batch < 0 || T < 0 || n < 1 || m < 1) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_f32: bad size");
    if (n > 64 || m > 16)
        return zm::set_error(ZM_EUNSUPPORTED, "zm_lqr_backward_f32: n=%d, m=%d outside n <= 64, m <= 16", n, m);
    if (batch > 0x7fffffffLL) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_f32: batch too large for one launch");
    if (batch == 0 || T == 0) return ZM_OK;
    hipStream_t st = (hipStream_t)stream;
    switch ((n + 15) / 16) {
        case 1: return zm::launch_tiled<1>(A, B, Q, R, L, batch, T, n, m, st);
        case 2: return zm::launch_tiled<2>(A, B, Q, R, L, batch, T, n, m, st);
        case 3: return zm::launch_tiled<3>(A, B, Q, R, L, batch, T, n, m, st);
        default: return zm::launch_tiled<4>(A, B, Q, R, L, batch, T, n, m, st);
    }
}
