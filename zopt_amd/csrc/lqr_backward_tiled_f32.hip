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
#include "lqr_tiled_core.h"

#include <cstdlib>

namespace zm {
int lqr_backward_dma_dispatch_f32io(const float* A, const float* B, const float* Q, const float* R, float* L, int64_t batch, int T,
                                    int n, int m, hipStream_t stream);
}

extern "C" int zm_lqr_backward_f32(const float* A, const float* B, const float* Q, const float* R, float* L, int64_t batch,
                                   int T, int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!A || !B || !Q || !R || !L) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_f32: null pointer");
    if (batch < 0 || T < 0 || n < 1 || m < 1) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_f32: bad size");
    if (n > 64 || m > 16)
        return zm::set_error(ZM_EUNSUPPORTED, "zm_lqr_backward_f32: n=%d, m=%d outside n <= 64, m <= 16", n, m);
    if (batch > 0x7fffffffLL) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_f32: batch too large for one launch");
    if (T == 0) return ZM_OK;
    hipStream_t st = (hipStream_t)stream;
    {   // the small fast-path shapes: K1 (LDS-DMA ring, tile-16 fp64 MFMA) on fp32 arrays -- fp32 storage, fp64 arithmetic, half the
        // HBM bytes of the fp64 call and no conversion passes (lqr_backward_dma.hip).  ZOPT_AMD_LQR_F32=tile: the fp32 tile kernel.
        static const bool tile_only = [] {
            const char* e = zm::lab_env("ZOPT_AMD_LQR_F32");
            return e && e[0] == 't';
        }();
        if (!tile_only) {
            const int rc = zm::lqr_backward_dma_dispatch_f32io(A, B, Q, R, L, batch, T, n, m, st);
            if (rc != ZM_EUNSUPPORTED) return rc;
        }
    }
    switch ((n + 15) / 16) {
        case 1: return zm::launch_tiled<zm::TileF32, 1>(A, B, Q, R, L, batch, T, n, m, st);
        case 2: return zm::launch_tiled<zm::TileF32, 2>(A, B, Q, R, L, batch, T, n, m, st);
        case 3: return zm::launch_tiled<zm::TileF32, 3>(A, B, Q, R, L, batch, T, n, m, st);
        default: return zm::launch_tiled<zm::TileF32, 4>(A, B, Q, R, L, batch, T, n, m, st);
    }
}
