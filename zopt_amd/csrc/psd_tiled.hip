// K5-T  psd_project for LARGE symmetric matrices (16 < k <= 64), fp64 -- the PD projection  a -> V max(w, eps) V^T  of
// zopt/ilqrUtils.py:217-219 (ensurePositiveDefinite) and its users conditionQuadraticCost (:222-234, the stacked (n+m)^2 cost Hessian)
// and conditionValueFunction (:254-257) beyond the one-tile kernel of psd.hip.
//
// Same method as ns16.h, on NT x NT tiles of 16 x 16 (one wave per matrix, every tile in MFMA registers): with X = sym(a) - eps I,
//     V max(w, eps) V^T = eps I + (X + |X|) / 2,   |X| = sign(X) X,
// sign(X) by the cubic Newton-Schulz step  Z <- Z (3 I - Z^2) / 2  behind quintic boosters  Z <- Z (a I + b Z^2 + c Z^4)  while
// F = |I - Z^2|_F^2 > 0.9 -- the same control flow, constants and iteration caps as ns16.h (tools/ns_psd_model.py is its NumPy model:
// <= 4e-12 relative against eigh on adversarial spectra), only the product is NT^3 tile products (4 NT^3 MFMAs) instead of one.
// Three tile matrices live in registers (Z, Z^2, the polynomial: 48 tiles = 384 registers at NT = 4); X itself is re-read from memory
// after the iteration.  Indices whose row and column are exactly zero stay out of the iteration and get eps on the diagonal.
// A coverage path: 0.3 ms per 64 x 64 matrix and wave -- the large shapes' cost Hessians are projected once per solve or per point.
#include "lqr_tiled_core.h"
#include "psd_mats.h"

namespace zm {

template <int NT>
struct TMat {
    td4 t[NT][NT];
};

// P = X^T Y
template <int NT>
__device__ __forceinline__ void tm_op(const TMat<NT>& X, const TMat<NT>& Y, TMat<NT>& P) {
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            td4 acc = TileF64::zero();
#pragma unroll
            for (int K = 0; K < NT; ++K) acc = op<TileF64>(X.t[K][I], Y.t[K][J], acc);
            P.t[I][J] = acc;
        }
}

template <int NT>
__device__ __forceinline__ double tm_wave_sum(double v) {
    return sum_xor32(sum_xor16(row16_sum(v)));
}

// M <- (M + M^T) / 2, tile pair by tile pair through two LDS buffers
template <int NT>
__device__ __forceinline__ void tm_symmetrise(TMat<NT>& M, double* buf, const int g, const int c) {
    constexpr int TLD = TileF64::TLD;
    double* b0 = buf;
    double* b1 = buf + 16 * TLD;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = I; J < NT; ++J) {
            TileF64::tile_to_lds(b0, M.t[I][J], g, c);
            if (J != I) TileF64::tile_to_lds(b1, M.t[J][I], g, c);
            t_lds_sync();
            const td4 tij = TileF64::tile_from_lds_T(J != I ? b1 : b0, g, c);   // (M[J][I])^T
            td4 tji = tij;
            if (J != I) tji = TileF64::tile_from_lds_T(b0, g, c);              // (M[I][J])^T
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double a = M.t[I][J][r], b = M.t[J][I][r];
                M.t[I][J][r] = 0.5 * (a + tij[r]);
                if (J != I) M.t[J][I][r] = 0.5 * (b + tji[r]);
            }
            t_lds_sync();   // the buffers are rewritten by the next pair
        }
}

// tile (K, J) of sym-less input: element (16K + 4r + g, 16J + c), zero beyond k
template <class Mat, int NT>
__device__ __forceinline__ void tm_load(const Mat& M, const long mat, const int k, TMat<NT>& A, const int g, const int c) {
#pragma unroll
    for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * K + 4 * r + g, j = 16 * J + c;
                const bool in = i < k && j < k;
                A.t[K][J][r] = in ? M.load(mat, in ? i : 0, in ? j : 0) : 0.0;
            }
}

template <class Mat, int NT>
__global__ __launch_bounds__(64) void psd_project_tiled_kernel(const Mat M, const int k, const double eps, const long count) {
    __shared__ __attribute__((aligned(16))) double buf[2 * 16 * TileF64::TLD];
    const long mat = blockIdx.x;
    if (mat >= count) return;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    TMat<NT> Z, Z2, W;
    // ---- X = sym(a) - eps I on the live, structurally nonzero indices; Z = X / |X|_F
    tm_load<Mat, NT>(M, mat, k, Z, g, c);
    tm_symmetrise<NT>(Z, buf, g, c);                       // jnp.linalg.eigh symmetrises its input
    double idr[NT][4], live[NT][4];                        // diagonal masks of tile (K, K)
    {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            bool nz = false;
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int r = 0; r < 4; ++r) nz |= (Z.t[K][J][r] != 0.0);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(nz);
            const unsigned colmask = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull);
            const bool colnz = (colmask >> c) & 1u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool diag = (4 * r + g == c) && (16 * J + c < k);
                live[J][r] = diag ? 1.0 : 0.0;
                idr[J][r] = (diag && colnz) ? 1.0 : 0.0;
            }
        }
    }
    double ss = 0.0;
#pragma unroll
    for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double x = Z.t[K][J][r] - (K == J ? eps * idr[K][r] : 0.0);
                Z.t[K][J][r] = x;
                ss = __builtin_fma(x, x, ss);
            }
    ss = tm_wave_sum<NT>(ss);
    const bool iterate = ss > 0.0;                         // X = 0: |X| = 0, nothing to iterate
    if (iterate) {
        const double inv = 1.0 / sqrt(ss);
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) Z.t[K][J][r] *= inv;
        constexpr double QA = 3.4445, QB = -4.7750, QC = 2.0315;
        constexpr int MAX_PAIRS = 18, MAX_CUBIC = 14;
        int pairs = 0, cubic = 0;
        // W <- QC W + QB Z2 + QA I   (W holds Z2^T Z2)
        auto quintic_poly = [&]() {
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        W.t[K][J][r] = __builtin_fma(QC, W.t[K][J][r], __builtin_fma(QB, Z2.t[K][J][r], K == J ? QA * idr[K][r] : 0.0));
        };
        // W <- 1.5 I - 0.5 Z2
        auto cubic_poly = [&]() {
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) W.t[K][J][r] = (K == J ? 1.5 * idr[K][r] : 0.0) - 0.5 * Z2.t[K][J][r];
        };
        for (;;) {
            tm_op<NT>(Z, Z, Z2);
            double f = 0.0;
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double e = (K == J ? idr[K][r] : 0.0) - Z2.t[K][J][r];
                        f = __builtin_fma(e, e, f);
                    }
            f = tm_wave_sum<NT>(f);
            if (f > 0.9 && pairs < MAX_PAIRS) {
                // booster group (quintic, quintic, cubic) as in ns16.h (NS_RIDER1 = 0.9: one riding quintic whenever a group runs)
                tm_op<NT>(Z2, Z2, W);
                quintic_poly();
                tm_op<NT>(W, Z, Z2);      // Z' = W Z (W is symmetric by construction)
                Z = Z2;
                tm_op<NT>(Z, Z, Z2);
                tm_op<NT>(Z2, Z2, W);
                quintic_poly();
                tm_op<NT>(W, Z, Z2);
                Z = Z2;
                tm_op<NT>(Z, Z, Z2);
                cubic_poly();
                tm_op<NT>(W, Z, Z2);
                Z = Z2;
                tm_symmetrise<NT>(Z, buf, g, c);
                ++pairs;
            } else {
                const bool last = (f < 1e-16) || (cubic + 1 >= MAX_CUBIC);
                cubic_poly();
                tm_op<NT>(Z, W, Z2);
                Z = Z2;
                if (last) {
                    tm_symmetrise<NT>(Z, buf, g, c);
                    ++cubic;
                    break;
                }
                tm_op<NT>(Z, Z, Z2);
                cubic_poly();
                tm_op<NT>(Z, W, Z2);
                Z = Z2;
                tm_symmetrise<NT>(Z, buf, g, c);
                cubic += 2;
                if (0.5625 * f * f < 1e-18 || cubic >= MAX_CUBIC) break;
            }
        }
    }
    // ---- X again (re-read; the same arithmetic as above), |X| = Z^T X, result = eps I + (X + |X|) / 2, symmetrised
    tm_load<Mat, NT>(M, mat, k, W, g, c);
    tm_symmetrise<NT>(W, buf, g, c);
#pragma unroll
    for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int r = 0; r < 4; ++r) W.t[K][K][r] -= eps * idr[K][r];
    if (iterate) {
        tm_op<NT>(Z, W, Z2);
    } else {
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) Z2.t[K][J] = TileF64::zero();
    }
#pragma unroll
    for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                W.t[K][J][r] = __builtin_fma(0.5, W.t[K][J][r] + Z2.t[K][J][r], K == J ? eps * live[K][r] : 0.0);
    tm_symmetrise<NT>(W, buf, g, c);
#pragma unroll
    for (int K = 0; K < NT; ++K)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * K + 4 * r + g, j = 16 * J + c;
                if (i < k && j < k) M.store(mat, i, j, W.t[K][J][r]);
            }
}

template <class Mat>
static int launch_psd_tiled(const Mat& M, int k, double eps, long count, hipStream_t st) {
    const dim3 grid((unsigned)count), block(64);
    if (k <= 32)
        hipLaunchKernelGGL((psd_project_tiled_kernel<Mat, 2>), grid, block, 0, st, M, k, eps, count);
    else if (k <= 48)
        hipLaunchKernelGGL((psd_project_tiled_kernel<Mat, 3>), grid, block, 0, st, M, k, eps, count);
    else
        hipLaunchKernelGGL((psd_project_tiled_kernel<Mat, 4>), grid, block, 0, st, M, k, eps, count);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// 16 < k <= 64 (psd.hip dispatches here)
int psd_project_tiled_plain(double* A, int64_t count, int k, double eps, hipStream_t st) {
    if (k <= 16 || k > 64) return ZM_EUNSUPPORTED;
    return launch_psd_tiled(PlainMat{A, k}, k, eps, (long)count, st);
}
int psd_project_tiled_cost(double* c_xx, double* c_ux, double* c_uu, int64_t count, int n, int m, double eps, hipStream_t st) {
    if (n + m <= 16 || n + m > 64) return ZM_EUNSUPPORTED;
    return launch_psd_tiled(StackedCost{c_xx, c_ux, c_uu, n, m}, n + m, eps, (long)count, st);
}

}  // namespace zm
