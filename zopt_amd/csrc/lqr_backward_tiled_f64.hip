// K1-T64  the MFMA tile kernel of lqr_backward_tiled_f32.hip instantiated for fp64 (v_mfma_f64_16x16x4_f64) and n <= 48, m <= 16:
// zopt/lqrUtils.py:144-173 for medium-sized fp64 systems.  (At 8 registers per tile the register-resident design stops at three tile
// rows; 48 < n <= 64 in fp64 stays on the LDS coverage kernel, lqr_backward_lds_f64.hip.)
#include "lqr_tiled_core.h"

#include <cstdlib>

namespace zm {
int lqr_backward_tiled_f64_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L, int64_t batch,
                                    int T, int n, int m, hipStream_t st) {
    if (n > 48 || m > 16 || n < 1 || m < 1) return ZM_EUNSUPPORTED;
    if (n <= 16) return launch_tiled<TileF64, 1>(A, B, Q, R, L, batch, T, n, m, st);
    if (n <= 32) return launch_tiled<TileF64, 2>(A, B, Q, R, L, batch, T, n, m, st);
    return launch_tiled<TileF64, 3>(A, B, Q, R, L, batch, T, n, m, st);
}
}  // namespace zm
