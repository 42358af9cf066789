// Device building blocks for the "tile-16" fp64 Riccati kernels (gfx950 / CDNA4, wave64).
//
// One wavefront owns one trajectory.  The stacked [state ; control] index (n + m <= 16) is mapped onto one
// 16x16 v_mfma_f64_16x16x4_f64 tile, lane l = (g = l >> 4, c = l & 15):
//
//   "D-layout"  (MFMA C/D, 4 regs r):  X[4r + g][c]
//   "A-operand" (one reg per K-step s): Aop[i = c][k = 4s + g]
//   "B-operand" (one reg per K-step s): Bop[k = 4s + g][j = c]
//
// => a D-layout result can be fed back UNCHANGED as the B operand of the next product (reg r = K-step r), and
//    the same registers read as an A operand are the TRANSPOSE (X^T).  Every product of the Riccati step
//    (F^T V F, A - B L, Acl^T V Acl, L^T R L) is therefore chained MFMA-to-MFMA with no lane movement; the
//    only cross-lane traffic per step is one 4x16 tile through LDS for the m x m solve.
#pragma once
#include <hip/hip_runtime.h>

namespace zm {

typedef double d4 __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16) + C, fp64.  NEG_A uses the free operand-negate modifier (blgp bit 0 -> neg:[1,0,0]).
template <bool NEG_A = false>
__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
    if constexpr (NEG_A)
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
    else
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ d4 zero4() { return d4{0.0, 0.0, 0.0, 0.0}; }

// Lane-local solve of the 4x4 system  S x = b  by LU with partial pivoting (first maximum wins, as LAPACK
// idamax) followed by forward/back substitution -- the arithmetic of `solve` in lqrUtils.py:168 /
// ilqrUtils.py:167-168 (jnp.linalg.solve = getrf + getrs).  S is identical in every lane (read by
// broadcast from LDS), b is the lane's own right-hand-side column.  Rows i >= m are identity padding.
// Pivot search is a running compare-exchange: row k ends up holding the first max |S[i][k]|, i >= k; the rows
// below are a permutation of LAPACK's, which changes nothing downstream (elimination is row-wise).
// Pivots are inverted once (1/p, IEEE division) and applied by multiplication.
__device__ __forceinline__ void lu_solve4(double (&S)[4][4], double (&b)[4], double (&x)[4]) {
    double rinv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const bool sw = __builtin_fabs(S[i][k]) > __builtin_fabs(S[k][k]);
#pragma unroll
            for (int j = k; j < 4; ++j) {
                const double t = S[k][j];
                S[k][j] = sw ? S[i][j] : t;
                S[i][j] = sw ? t : S[i][j];
            }
            const double t = b[k];
            b[k] = sw ? b[i] : t;
            b[i] = sw ? t : b[i];
        }
        rinv[k] = 1.0 / S[k][k];
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const double f = S[i][k] * rinv[k];
#pragma unroll
            for (int j = k + 1; j < 4; ++j) S[i][j] = __builtin_fma(-f, S[k][j], S[i][j]);
            b[i] = __builtin_fma(-f, b[k], b[i]);
        }
    }
#pragma unroll
    for (int k = 3; k >= 0; --k) {
        double s = b[k];
#pragma unroll
        for (int j = k + 1; j < 4; ++j) s = __builtin_fma(-S[k][j], x[j], s);
        x[k] = s * rinv[k];
    }
}

// 1/a to fp64 accuracy (<= ~1 ulp) from v_rcp_f64 (measured max relative error 4.6e-8, tools/rcp_accuracy.hip) and ONE
// third-order correction r0 (1 + e + e^2), e = 1 - a r0 (error e^3 ~ 1e-22): three FMAs, one fewer than two Newton steps and
// as accurate.  a = 0, inf, NaN or denormal-range pivots give inf/NaN here -- callers test the result and take the IEEE /
// pivoted path in that case.
__device__ __forceinline__ double fast_rcp(const double a) {
    const double r = __builtin_amdgcn_rcp(a);
    const double e = __builtin_fma(-a, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

// Fast path of the m x m solve: Gaussian elimination WITHOUT row exchanges (no register shuffling at all),
// valid whenever partial pivoting is not needed for stability.  Returns false -- and the caller must redo the
// solve with lu_solve4 (pivoted, IEEE division) -- unless every multiplier satisfies |l_ik| <= 4 (element growth
// then stays <= 5^3) and every pivot reciprocal is finite.  NaN compares false, so NaN/zero pivots always take
// the pivoted path and inf/NaN propagate exactly as there.  Differs from the pivoted path only by rounding.
__device__ __forceinline__ bool lu_solve4_nopivot(const double (&S)[4][4], const double (&b)[4], double (&x)[4]) {
    const double lim = 4.0;
    const double r0 = fast_rcp(S[0][0]);
    const double f10 = S[1][0] * r0, f20 = S[2][0] * r0, f30 = S[3][0] * r0;
    bool ok = (__builtin_fabs(f10) <= lim) & (__builtin_fabs(f20) <= lim) & (__builtin_fabs(f30) <= lim);
    const double a11 = __builtin_fma(-f10, S[0][1], S[1][1]), a12 = __builtin_fma(-f10, S[0][2], S[1][2]);
    const double a13 = __builtin_fma(-f10, S[0][3], S[1][3]), b1 = __builtin_fma(-f10, b[0], b[1]);
    const double a21 = __builtin_fma(-f20, S[0][1], S[2][1]), a22 = __builtin_fma(-f20, S[0][2], S[2][2]);
    const double a23 = __builtin_fma(-f20, S[0][3], S[2][3]), b2 = __builtin_fma(-f20, b[0], b[2]);
    const double a31 = __builtin_fma(-f30, S[0][1], S[3][1]), a32 = __builtin_fma(-f30, S[0][2], S[3][2]);
    const double a33 = __builtin_fma(-f30, S[0][3], S[3][3]), b3 = __builtin_fma(-f30, b[0], b[3]);
    const double r1 = fast_rcp(a11);
    const double f21 = a21 * r1, f31 = a31 * r1;
    ok &= (__builtin_fabs(f21) <= lim) & (__builtin_fabs(f31) <= lim);
    const double c22 = __builtin_fma(-f21, a12, a22), c23 = __builtin_fma(-f21, a13, a23);
    const double d2 = __builtin_fma(-f21, b1, b2);
    const double c32 = __builtin_fma(-f31, a12, a32), c33 = __builtin_fma(-f31, a13, a33);
    const double d3 = __builtin_fma(-f31, b1, b3);
    const double r2 = fast_rcp(c22);
    const double f32 = c32 * r2;
    ok &= (__builtin_fabs(f32) <= lim);
    const double e33 = __builtin_fma(-f32, c23, c33), g3 = __builtin_fma(-f32, d2, d3);
    const double r3 = fast_rcp(e33);
    ok &= (__builtin_fabs(r3) <= 1.79769313486231570815e308);
    x[3] = g3 * r3;
    x[2] = __builtin_fma(-c23, x[3], d2) * r2;
    x[1] = __builtin_fma(-a13, x[3], __builtin_fma(-a12, x[2], b1)) * r1;
    x[0] = __builtin_fma(-S[0][3], x[3], __builtin_fma(-S[0][2], x[2], __builtin_fma(-S[0][1], x[1], b[0]))) * r0;
    return ok;
}

// The re-solve after a failed growth check, shared by the sweep kernels: LU with partial pivoting -- unless S holds a NaN.  Then
// every component of the solution is NaN whichever elimination order runs (a NaN row update f * S[k][j] poisons its whole row even
// through zeros, a NaN pivot poisons every row below it, and the back substitution multiplies every x_j into the rows above it), so
// the result is written directly.  This matters for speed, not for values: an iLQR start that has diverged to NaN (0.9 % of
// BASELINE configs[3]'s starts, which the reference keeps iterating to maxIter) would otherwise take the slow pivoted path in every
// step of every remaining sweep -- and a launch lasts as long as its slowest wave (tail sweeps: 109 instead of 77 us).
__device__ __forceinline__ void lu_solve4_fallback(double (&S)[4][4], double (&b)[4], double (&x)[4]) {
    bool has_nan = false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) has_nan |= (S[i][j] != S[i][j]);
    if (has_nan) {
        const double qnan = __builtin_nan("");
        x[0] = x[1] = x[2] = x[3] = qnan;
    } else {
        lu_solve4(S, b, x);
    }
}

}  // namespace zm
