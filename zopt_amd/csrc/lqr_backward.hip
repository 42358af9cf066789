// K1  lqr_backward -- batched discrete finite-horizon LQR backward Riccati recursion, fp64, gfx950.
//
// Replaces the arithmetic of zopt/lqrUtils.py:144-173 (discreteFiniteHorizonLqr), per trajectory:
//     V <- Q[T-1]                                                              (:172)
//     for k = T-1 .. 0:
//         L_k = solve(R_k + B_k^T V B_k,  B_k^T V A_k)                         (:168)
//         V   = Q_k + L_k^T R_k L_k + (A_k - B_k L_k)^T V (A_k - B_k L_k)      (:169)  Joseph form
//
// Mapping (see tile16_f64.h): one wave64 per trajectory, stacked index [x(0..n-1) | pad | u at NP..NP+m-1],
// NP = 4*KS >= n, NP + m <= 16.  Per step, with F = [A_k | B_k] as a 16-column tile:
//     Y  = V^T F                3 MFMA   (V's D-layout registers read as an A operand are V^T)
//     G  = Y^T F + [0;R]        3 MFMA   = F^T V F: rows NP..NP+3 of G = [B^T V A | R + B^T V B].  Feeding Y back
//                                          as the A operand undoes the transpose, so the result is exact for a
//                                          NONSYMMETRIC V too (nonsymmetric Q/R inputs), as the reference's is.
//     L  = solve(Suu, Sux)      4x16 tile through LDS, lane-local pivoted LU (tile16_f64.h)
//     Acl= A - B L              1 MFMA   (negated A operand), C = F
//     RL = R L                  1 MFMA
//     W  = V^T Acl              3 MFMA
//     V' = Q + L^T RL + W^T Acl 4 MFMA   (= Q + L^T R L + Acl^T V Acl, the reference's Joseph form)
// A_k, B_k, Q_k, R_k are read ONCE from HBM straight into their MFMA register layouts (every step's
// matrices are whole 128-B lines: 1152/384/1152/128 B at n=12, m=4), two steps ahead of their use; V never
// leaves registers; L_k is written once.  Algorithmic HBM traffic: 8*(2n^2 + 2nm + m^2) B per horizon step.
#include "lqr_step_core.h"
#include "zm_common.h"

#include <cstdlib>

namespace zm {

template <int KS>
struct LqrAddr {
    const double* pF0;  // row g of the tile; K-step s adds s*dF
    const double* pQ0;  // row g of Q; K-step s adds s*4*n
    int dF;             // per-lane K-step stride of pF (4*n for A lanes, 4*m for B lanes)
    const double* pRm;
    const double* pBt;
    const double* pRt;
    double* pL;
    int sF;   // per-lane step stride of pF (n*n for A lanes, n*m for B lanes)
    int nn, nm, mm, q4n;
    bool rowok[KS];  // 4s+g < n
    bool vF[KS], vQ[KS], vRm, vBt, vRt, vL;
    double rm_pad;
};

// Loads the step the pointers currently address, then moves every pointer one step back in time.
// Loads are UNCONDITIONAL (lanes outside a matrix read a clamped in-bounds address of the same step and the
// value is then zeroed): exec-masked loads would put every load in its own branch region and make hipcc
// fall back to s_waitcnt vmcnt(0), which would serialise the two-steps-ahead prefetch.
template <int KS>
__device__ __forceinline__ void lqr_load_step(LqrStepRegs<KS>& d, LqrAddr<KS>& a) {
    double f[KS], q[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = a.pF0[a.rowok[s] ? s * a.dF : 0];
    a.pF0 -= a.sF;
#pragma unroll
    for (int s = 0; s < KS; ++s) q[s] = a.pQ0[a.rowok[s] ? s * a.q4n : 0];
    a.pQ0 -= a.nn;
    const double rm = *a.pRm;
    a.pRm -= a.mm;
    const double bt = *a.pBt;
    a.pBt -= a.nm;
    const double rt = *a.pRt;
    a.pRt -= a.mm;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        d.F[s] = a.vF[s] ? f[s] : 0.0;
        d.Qd[s] = a.vQ[s] ? q[s] : 0.0;
    }
    d.Rm = a.vRm ? rm : a.rm_pad;
    d.Bt = a.vBt ? bt : 0.0;
    d.Rt = a.vRt ? rt : 0.0;
}

template <int KS, bool PREFETCH>
__device__ __forceinline__ void lqr_step(double (&V)[KS], LqrStepRegs<KS>& d, LqrAddr<KS>& a, double* smw,
                                         const int g, const int c) {
    // Refill this step buffer with step k-2 only after the solve: the 4x4 system is the register-pressure peak.
    const double lv = lqr_step_core<KS>(V, d, smw, g, c, a.vL, [&]() {
        if constexpr (PREFETCH) lqr_load_step(d, a);
    });
    if (a.vL) *a.pL = lv;
    a.pL -= a.nm;
}

// KS = NP/4; NC/MC = compile-time n/m (0 = take the runtime arguments); WPB = waves (trajectories) per block.
template <int KS, int NC, int MC, int WPB>
__global__ __launch_bounds__(64 * WPB, 4) void lqr_backward_t16_f64(const double* __restrict__ A,
                                                                 const double* __restrict__ B,
                                                                 const double* __restrict__ Q,
                                                                 const double* __restrict__ R,
                                                                 double* __restrict__ L, const long batch,
                                                                 const int T, const int n_rt, const int m_rt) {
    constexpr int NP = 4 * KS;
    const int n = NC ? NC : n_rt;
    const int m = MC ? MC : m_rt;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long traj = (long)blockIdx.x * WPB + wave;
    if (traj >= batch) return;  // whole wave leaves; no block-level barrier is ever used
    const int g = lane >> 4, c = lane & 15;

    __shared__ double sm[WPB][64];
    double* smw = sm[wave];

    LqrAddr<KS> a;
    a.nn = n * n;
    a.nm = n * m;
    a.mm = m * m;
    const bool cA = c < n;
    const bool cB = (c >= NP) && (c < NP + m);
    const long last = (long)(T - 1);
    const double* At = A + (traj * T + last) * a.nn;
    const double* Bt = B + (traj * T + last) * a.nm;
    const double* Qt = Q + (traj * T + last) * a.nn;
    const double* Rt = R + (traj * T + last) * a.mm;
    // Lanes / rows outside the matrices read an in-bounds address of the same step (element 0 of A_k, or
    // row g when only the K-step row 4s+g is out of range) and are zeroed after the load.  Base pointer,
    // K-step stride and step stride are always chosen TOGETHER so a clamped lane never leaves its array.
    const bool row0 = g < n;
    const bool laneA = row0 && cA;   // reads A_k[4s+g][c]
    const bool laneB = row0 && cB;   // reads B_k[4s+g][c-NP]
    a.q4n = 4 * n;
    a.dF = laneA ? 4 * n : laneB ? 4 * m : 0;
    a.sF = laneB ? a.nm : a.nn;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int row = 4 * s + g;
        a.rowok[s] = row < n;
        a.vF[s] = (row < n) && (cA || cB);
        a.vQ[s] = (row < n) && cA;
    }
    a.pF0 = laneA ? (At + g * n + c) : laneB ? (Bt + g * m + (c - NP)) : At;
    a.pQ0 = laneA ? (Qt + g * n + c) : Qt;
    if (!laneA) a.q4n = 0;
    a.vRm = (g < m) && cB;
    a.pRm = a.vRm ? (Rt + g * m + (c - NP)) : Rt;
    a.rm_pad = (g >= m && c == NP + g) ? 1.0 : 0.0;
    a.vBt = cA && (g < m);
    a.pBt = a.vBt ? (Bt + c * m + g) : Bt;
    a.vRt = (c < m) && (g < m);
    a.pRt = a.vRt ? (Rt + c * m + g) : Rt;
    a.vL = (g < m) && cA;
    a.pL = L + (traj * T + last) * a.nm + g * n + c;

    // d0 <- step T-1, d1 <- step T-2; each step refills its own buffer with the step two back in time.
    LqrStepRegs<KS> d0, d1;
    lqr_load_step(d0, a);
    if (T >= 2) lqr_load_step(d1, a);
    double V[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) V[s] = d0.Qd[s];  // V <- Q[T-1]   (lqrUtils.py:172)

    int k = T - 1;  // invariant: d0 holds step k, d1 holds step k-1
    while (k >= 3) {
        lqr_step<KS, true>(V, d0, a, smw, g, c);
        lqr_step<KS, true>(V, d1, a, smw, g, c);
        k -= 2;
    }
    // peeled tail (k in {0,1,2}): prefetch only while a step k-2 >= 0 exists
    if (k == 2) {
        lqr_step<KS, true>(V, d0, a, smw, g, c);
        lqr_step<KS, false>(V, d1, a, smw, g, c);
        lqr_step<KS, false>(V, d0, a, smw, g, c);
    } else if (k == 1) {
        lqr_step<KS, false>(V, d0, a, smw, g, c);
        lqr_step<KS, false>(V, d1, a, smw, g, c);
    } else {
        lqr_step<KS, false>(V, d0, a, smw, g, c);
    }
}

template <int KS, int NC, int MC>
static int launch_t16(const double* A, const double* B, const double* Q, const double* R, double* L, int64_t batch,
                      int T, int n, int m, hipStream_t stream) {
    constexpr int WPB = 1;
    const long blocks = (batch + WPB - 1) / WPB;
    hipLaunchKernelGGL((lqr_backward_t16_f64<KS, NC, MC, WPB>), dim3((unsigned)blocks), dim3(64 * WPB), 0, stream, A,
                       B, Q, R, L, (long)batch, T, n, m);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// DARE by Riccati value iteration: the same step with time-invariant operands held in registers, repeated from V = Q
// until the value matrix stops changing.  Replaces zopt/lqrUtils.py:176-204 discreteInfiniteHorizonLqr
// (V = scipy.linalg.solve_discrete_are(A, B, Q, R); L = solve(R + B^T V B, B^T V A)): for a stabilisable / detectable
// problem the recursion from V = Q >= 0 converges to that stabilising solution; the reference's own test uses the long
// finite horizon as the cross-check of this function.  One wave per system, no HBM traffic inside the loop.
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = __builtin_fmax(v, __shfl_xor(v, off, 64));
    return v;
}

template <int KS>
__global__ __launch_bounds__(64, 4) void dare_t16_f64(const double* __restrict__ A, const double* __restrict__ B,
                                                      const double* __restrict__ Q, const double* __restrict__ R,
                                                      double* __restrict__ L, double* __restrict__ P, int* __restrict__ iters,
                                                      const long batch, const int n, const int m, const double tol,
                                                      const int max_iter) {
    constexpr int NP = 4 * KS;
    const int lane = threadIdx.x & 63;
    const long sys = blockIdx.x;
    if (sys >= batch) return;
    const int g = lane >> 4, c = lane & 15;
    __shared__ double smw[64];

    LqrAddr<KS> a;
    a.nn = n * n;
    a.nm = n * m;
    a.mm = m * m;
    const bool cA = c < n;
    const bool cB = (c >= NP) && (c < NP + m);
    const double* At = A + sys * a.nn;
    const double* Bt = B + sys * a.nm;
    const double* Qt = Q + sys * a.nn;
    const double* Rt = R + sys * a.mm;
    const bool row0 = g < n;
    const bool laneA = row0 && cA, laneB = row0 && cB;
    a.q4n = 4 * n;
    a.dF = laneA ? 4 * n : laneB ? 4 * m : 0;
    a.sF = 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int row = 4 * s + g;
        a.rowok[s] = row < n;
        a.vF[s] = (row < n) && (cA || cB);
        a.vQ[s] = (row < n) && cA;
    }
    a.pF0 = laneA ? (At + g * n + c) : laneB ? (Bt + g * m + (c - NP)) : At;
    a.pQ0 = laneA ? (Qt + g * n + c) : Qt;
    if (!laneA) a.q4n = 0;
    a.vRm = (g < m) && cB;
    a.pRm = a.vRm ? (Rt + g * m + (c - NP)) : Rt;
    a.rm_pad = (g >= m && c == NP + g) ? 1.0 : 0.0;
    a.vBt = cA && (g < m);
    a.pBt = a.vBt ? (Bt + c * m + g) : Bt;
    a.vRt = (c < m) && (g < m);
    a.pRt = a.vRt ? (Rt + c * m + g) : Rt;
    a.vL = (g < m) && cA;
    a.nn = a.nm = a.mm = 0;   // time-invariant: lqr_load_step must not move the pointers
    LqrStepRegs<KS> d;
    lqr_load_step(d, a);
    double V[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) V[s] = d.Qd[s];

    double lv = 0.0, prev = 1e300;
    int it = 0, stall = 0;
    bool conv = false;
    while (it < max_iter) {   // every quantity in the loop condition is wave-uniform
        double Vo[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) Vo[s] = V[s];
        lv = lqr_step_core<KS>(V, d, smw, g, c, a.vL, []() {});
        ++it;
        if ((it & 3) == 0 || it == max_iter) {
            double df = 0.0, sc = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                df = __builtin_fmax(df, __builtin_fabs(V[s] - Vo[s]));
                sc = __builtin_fmax(sc, __builtin_fabs(V[s]));
            }
            df = wave_max(df);
            sc = wave_max(sc);
            if (!(df > tol * sc)) {                                // converged (or NaN: stop; the caller sees the non-finite gain)
                conv = true;
                break;
            }
            stall = (df >= prev && df <= 1e-9 * sc) ? stall + 1 : 0;   // rounding floor reached
            if (stall >= 3) {
                conv = true;
                break;
            }
            prev = df;
        }
    }
    // gain of the converged value: one more solve with V fixed (lv above belongs to the previous iterate)
    {
        double Vc[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) Vc[s] = V[s];
        lv = lqr_step_core<KS>(Vc, d, smw, g, c, a.vL, []() {});
    }
    if (a.vL) L[sys * (long)(n * m) + g * n + c] = lv;
    if (P) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (4 * s + g < n && cA) P[sys * (long)(n * n) + (4 * s + g) * n + c] = V[s];
    }
    // explicit status: +iterations when the test above was met (also when that happened exactly on the last allowed iteration),
    // -iterations when the cap ended the loop first
    if (iters && lane == 0) iters[sys] = conv ? it : -it;
}

// lqr_backward_dma.hip: LDS-DMA staged fast path (even n, m; 16-B aligned pointers); ZM_EUNSUPPORTED otherwise.
int lqr_backward_dma_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L,
                              int64_t batch, int T, int n, int m, hipStream_t stream);

}  // namespace zm

namespace zm {
int lqr_backward_lds_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L, int64_t batch,
                              int T, int n, int m, hipStream_t st);   // lqr_backward_lds_f64.hip
int lqr_backward_tiled_f64_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L, int64_t batch,
                                    int T, int n, int m, hipStream_t st);   // lqr_backward_tiled_f64.hip (n <= 64)
int lqr_dare_tiled_f64_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L, double* P, int* iters,
                                int64_t batch, int n, int m, double tol, int max_iter, hipStream_t st);
}

extern "C" int zm_lqr_backward_supported(int n, int m, int elem_size) {
    if (n < 1 || m < 1 || n > 64 || m > 16) return 0;
    return (elem_size == 8 || elem_size == 4) ? 1 : 0;
}

extern "C" int zm_lqr_backward_f64(const double* A, const double* B, const double* Q, const double* R, double* L,
                                   int64_t batch, int T, int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!A || !B || !Q || !R || !L) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_f64: null pointer");
    if (batch < 0 || T < 1 || n < 1 || m < 1)
        return zm::set_error(ZM_EINVAL, "zm_lqr_backward_f64: bad size batch=%lld T=%d n=%d m=%d", (long long)batch, T,
                             n, m);
    if (!zm_lqr_backward_supported(n, m, 8))
        return zm::set_error(ZM_EUNSUPPORTED, "zm_lqr_backward_f64: (n=%d, m=%d) not covered (need n<=64, m<=16)", n, m);
    if ((int64_t)T * n * n >= (int64_t)1 << 31 || batch >= ((int64_t)1 << 31))
        return zm::set_error(ZM_EUNSUPPORTED, "zm_lqr_backward_f64: T*n*n or batch too large");
    if (batch == 0) return ZM_OK;
    hipStream_t st = (hipStream_t)stream;
    if (n > 12 || m > 4) {
        // medium sizes: fp64 MFMA tile kernel (n <= 48); beyond, or with ZOPT_AMD_LQR_PATH=lds: the LDS coverage kernel
        static const bool force_lds = [] {
            const char* e = zm::fallback_env("ZOPT_AMD_LQR_PATH");
            return e && e[0] == 'l';
        }();
        if (!force_lds) {
            const int rc = zm::lqr_backward_tiled_f64_dispatch(A, B, Q, R, L, batch, T, n, m, st);
            if (rc != ZM_EUNSUPPORTED) return rc;
        }
        return zm::lqr_backward_lds_dispatch(A, B, Q, R, L, batch, T, n, m, st);
    }
    // ZOPT_AMD_LQR_PATH=reg forces the register-prefetch kernel (A/B measurements); default: LDS-DMA when eligible.
    static const bool force_reg = [] {
        const char* e = zm::fallback_env("ZOPT_AMD_LQR_PATH");
        return e && e[0] == 'r';
    }();
    if (!force_reg) {
        const int rc = zm::lqr_backward_dma_dispatch(A, B, Q, R, L, batch, T, n, m, st);
        if (rc != ZM_EUNSUPPORTED) return rc;
    }
    if (n == 12 && m == 4) return zm::launch_t16<3, 12, 4>(A, B, Q, R, L, batch, T, n, m, st);
    if (n == 8 && m == 4) return zm::launch_t16<2, 8, 4>(A, B, Q, R, L, batch, T, n, m, st);
    if (n == 4 && m == 1) return zm::launch_t16<1, 4, 1>(A, B, Q, R, L, batch, T, n, m, st);
    if (n <= 4) return zm::launch_t16<1, 0, 0>(A, B, Q, R, L, batch, T, n, m, st);
    if (n <= 8) return zm::launch_t16<2, 0, 0>(A, B, Q, R, L, batch, T, n, m, st);
    return zm::launch_t16<3, 0, 0>(A, B, Q, R, L, batch, T, n, m, st);
}

extern "C" int zm_dare_f64(const double* A, const double* B, const double* Q, const double* R, double* L, double* P,
                           int32_t* iters, int64_t batch, int n, int m, double tol, int max_iter, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!A || !B || !Q || !R || !L) return zm::set_error(ZM_EINVAL, "zm_dare_f64: null pointer");
    if (batch < 0 || n < 1 || m < 1 || max_iter < 1 || !(tol >= 0.0)) return zm::set_error(ZM_EINVAL, "zm_dare_f64: bad argument");
    if (n > 64 || m > 16) return zm::set_error(ZM_EUNSUPPORTED, "zm_dare_f64: (n=%d, m=%d) not covered (need n<=64, m<=16)", n, m);
    if (batch >= ((int64_t)1 << 31)) return zm::set_error(ZM_EUNSUPPORTED, "zm_dare_f64: batch too large");
    if (n > 12 || m > 4)   // large states: the fp64 tile kernel's step on time-invariant operands
        return zm::lqr_dare_tiled_f64_dispatch(A, B, Q, R, L, P, (int*)iters, batch, n, m, tol, max_iter, (hipStream_t)stream);
    if (batch == 0) return ZM_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)batch), block(64);
    if (n <= 4)
        hipLaunchKernelGGL((zm::dare_t16_f64<1>), grid, block, 0, st, A, B, Q, R, L, P, (int*)iters, (long)batch, n, m, tol, max_iter);
    else if (n <= 8)
        hipLaunchKernelGGL((zm::dare_t16_f64<2>), grid, block, 0, st, A, B, Q, R, L, P, (int*)iters, (long)batch, n, m, tol, max_iter);
    else
        hipLaunchKernelGGL((zm::dare_t16_f64<3>), grid, block, 0, st, A, B, Q, R, L, P, (int*)iters, (long)batch, n, m, tol, max_iter);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
