// Per-step arithmetic of the LQR backward sweep shared by the tile-16 kernels (register-prefetch and LDS-DMA).
// See lqr_backward.hip for the derivation; reference: zopt/lqrUtils.py:167-170.
#pragma once
#include "tile16_f64.h"

namespace zm {

template <int KS>
struct LqrStepRegs {
    double F[KS];   // F[4s+g][c]   : A_k (c < n) | B_k (NP <= c < NP+m)      B-operand / A-operand(F^T)
    double Qd[KS];  // Q_k[4s+g][c] : D-layout accumulator init of V'
    double Rm;      // R_k[g][c-NP] : D-layout row NP+g accumulator init of G (identity padding for g >= m)
    double Bt;      // B_k[c][g]    : A-operand of B L
    double Rt;      // R_k[c][g]    : A-operand of R L
};

// One Riccati step on registers.  V (D-layout, KS regs) is updated in place; returns this lane's L_k[g][c]
// (zero outside g < m, c < n).  `after_solve()` runs once the step's operands have been consumed into
// accumulators and the m x m solve is done (the register-pressure peak) -- the register-prefetch kernel refills
// its step buffer there.  `exch` = 64 doubles of LDS private to this wave.
template <int KS, typename AfterSolve>
__device__ __forceinline__ double lqr_step_core(double (&V)[KS], const LqrStepRegs<KS>& d, double* exch, const int g,
                                                const int c, const bool vL, AfterSolve&& after_solve) {
    constexpr int NP = 4 * KS;
    // Y = V^T F
    d4 y = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) y = mfma(V[s], d.F[s], y);
    // G = Y^T F + [0 ; R] = F^T V F + [0 ; R]  -> row NP+g : [ B^T V A | R + B^T V B ]
    d4 gacc = zero4();
    gacc[KS] = d.Rm;
#pragma unroll
    for (int s = 0; s < KS; ++s) gacc = mfma(y[s], d.F[s], gacc);
    const double mrow = gacc[KS];

    // Accumulator inits consume the step operands.
    d4 aacc = zero4();
    d4 vacc = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        aacc[s] = d.F[s];
        vacc[s] = d.Qd[s];
    }
    const double bt = d.Bt, rt = d.Rt;

    // 4 x 16 tile [Sux | Suu] through LDS: every lane reads all of Suu (broadcast) and its own RHS column.
    exch[g * 16 + c] = mrow;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double S[4][4], b[4], x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i][j] = exch[i * 16 + NP + j];
        b[i] = exch[i * 16 + c];
    }
    __builtin_amdgcn_wave_barrier();
    lu_solve4(S, b, x);
    const double x01 = (g & 1) ? x[1] : x[0];
    const double x23 = (g & 1) ? x[3] : x[2];
    double lv = (g & 2) ? x23 : x01;
    lv = vL ? lv : 0.0;  // L_k[g][c], zero outside (g < m, c < n)
    after_solve();

    // Acl = A - B L   (columns >= NP keep B; they only ever feed padding rows/columns)
    aacc = mfma<true>(bt, lv, aacc);
    // RL = R L  (rows 0..m-1 -> D reg 0 == B operand, K-step 0)
    const d4 racc = mfma(rt, lv, zero4());
    // W = V^T Acl
    d4 w = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) w = mfma(V[s], aacc[s], w);
    // V' = Q + L^T (R L) + W^T Acl = Q + L^T R L + Acl^T V Acl
    vacc = mfma(lv, racc[0], vacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) vacc = mfma(w[s], aacc[s], vacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) V[s] = vacc[s];
    return lv;
}

}  // namespace zm
