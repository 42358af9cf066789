// K1-T64  the MFMA tile kernel of lqr_backward_tiled_f32.hip instantiated for fp64 (v_mfma_f64_16x16x4_f64), n <= 64, m <= 16:
// zopt/lqrUtils.py:144-173 for medium-sized and large fp64 systems.  At 8 registers per tile, three tile rows (n <= 48) fit the
// register-resident design with its second operand set (V + F + Fn + Y = 45 tiles); at four tile rows (48 < n <= 64) V + F + Y alone
// are 448 of the 512 registers, so that instantiation drops the prefetched operand set (PREFETCH = false: a step's operands are
// loaded at its head) and lets the compiler spill what the solve needs on top (162 registers at the exact shape) -- still an order
// of magnitude faster than the LDS coverage kernel (lqr_backward_lds_f64.hip) it replaces as the default there.
#include "lqr_tiled_core.h"

#include <cstdlib>

namespace zm {
int lqr_backward_tiled_f64_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L, int64_t batch,
                                    int T, int n, int m, hipStream_t st) {
    if (n > 64 || m > 16 || n < 1 || m < 1) return ZM_EUNSUPPORTED;
    if (n <= 16) return launch_tiled<TileF64, 1>(A, B, Q, R, L, batch, T, n, m, st);
    if (n <= 32) return launch_tiled<TileF64, 2>(A, B, Q, R, L, batch, T, n, m, st);
    if (n <= 48) return launch_tiled<TileF64, 3>(A, B, Q, R, L, batch, T, n, m, st);
    return launch_tiled<TileF64, 4, false>(A, B, Q, R, L, batch, T, n, m, st);
}

// discreteInfiniteHorizonLqr beyond the tile-16 shapes (12 < n <= 64 or 4 < m <= 16): the tile kernel's Joseph-form step iterated on
// time-invariant operands (lqr_tiled_core.h: DARE)
int lqr_dare_tiled_f64_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L, double* P, int* iters,
                                int64_t batch, int n, int m, double tol, int max_iter, hipStream_t st) {
    if (n > 64 || m > 16 || n < 1 || m < 1) return ZM_EUNSUPPORTED;
    if (n <= 16) return launch_tiled_dare<TileF64, 1>(A, B, Q, R, L, P, iters, batch, max_iter, n, m, tol, st);
    if (n <= 32) return launch_tiled_dare<TileF64, 2>(A, B, Q, R, L, P, iters, batch, max_iter, n, m, tol, st);
    if (n <= 48) return launch_tiled_dare<TileF64, 3>(A, B, Q, R, L, P, iters, batch, max_iter, n, m, tol, st);
    return launch_tiled_dare<TileF64, 4, false>(A, B, Q, R, L, P, iters, batch, max_iter, n, m, tol, st);
}
}  // namespace zm
