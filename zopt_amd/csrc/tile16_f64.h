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

}  // namespace zm
