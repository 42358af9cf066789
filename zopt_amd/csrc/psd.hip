// K5  psd_project -- batched symmetric eigen-decomposition, eigenvalue clamp, reconstruction; fp64, gfx950.
//
// Replaces zopt/ilqrUtils.py:217-219 ensurePositiveDefinite(a, eps):  w, v = eigh(a);  (v * max(w, eps)) @ v.T
// (jnp.linalg.eigh symmetrises its input: (a + a^T)/2) and its users conditionQuadraticCost (:222-234) and
// conditionValueFunction (:254-257).  The result is a spectral function of the matrix, so it does not depend on
// the eigenvector signs / ordering an eigensolver happens to return.
//
// One wave64 per matrix (k <= 16), matrix and eigenvector accumulator in LDS.  Cyclic two-sided Jacobi with the
// round-robin parallel ordering: K-1 rounds per sweep, K/2 disjoint rotations per round; the rotations of a round are
// computed by K/2 lanes, then every lane applies them to its share of the column pairs (A, V) and of the row pairs (A).
// Sweeps stop when a whole sweep found every |a_pq| <= 2^-52 * sqrt(|a_pp a_qq|) (quadratic convergence: one extra
// sweep at most), or after 20 sweeps.
#include "jacobi16.h"
#include "ns16.h"
#include "psd_mats.h"
#include "zm_common.h"

namespace zm {

#ifndef ZM_PSD_JACOBI
// one wave per matrix: the k x k matrix sits zero-padded in one 16 x 16 MFMA tile (lane (g, c) holds rows 4r+g of column c) and
// is projected by the matrix-sign iteration of ns16.h -- ~100-200 fp64 MFMAs instead of ~10 Jacobi sweeps through LDS
template <class Mat, int KSZ>
__device__ __forceinline__ void psd_project_tile(const Mat& M, const long mat, const int k, const double eps, double* T) {
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    d4 a;
    bool live[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        const bool in = (i < k) && (c < k);
        a[r] = in ? M.load(mat, in ? i : 0, in ? c : 0) : 0.0;
        live[r] = in && (i == c);
    }
    psd_project_ns<KSZ>(a, live, eps, T, g, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        if (i < k && c < k) M.store(mat, i, c, a[r]);
    }
}

template <class Mat>
__global__ __launch_bounds__(64) void psd_project_kernel(const Mat M, const int k, const double eps, const long count) {
    __shared__ double T[NS_LDS_DOUBLES];
    const long mat = blockIdx.x;
    if (mat >= count) return;
    if (k <= 4) psd_project_tile<Mat, 1>(M, mat, k, eps, T);
    else if (k <= 8) psd_project_tile<Mat, 2>(M, mat, k, eps, T);
    else if (k <= 12) psd_project_tile<Mat, 3>(M, mat, k, eps, T);
    else psd_project_tile<Mat, 4>(M, mat, k, eps, T);
}
#else
template <class Mat>
__global__ __launch_bounds__(64) void psd_project_kernel(const Mat M, const int k, const double eps, const long count) {
    __shared__ double As[PK * PLD], Vs[PK * PLD], cs[PK];
    __shared__ int pq[PK];
    const int lane = threadIdx.x;
    const long mat = blockIdx.x;
    if (mat >= count) return;
    // load and symmetrise (jnp.linalg.eigh: symmetrize_input=True)
    for (int e = lane; e < k * k; e += 64) {
        const int i = e / k, j = e % k;
        As[i * PLD + j] = 0.5 * (M.load(mat, i, j) + M.load(mat, j, i));
    }
    wave_lds_sync();
    psd_project_lds(As, Vs, cs, pq, k, eps, lane);
    for (int e = lane; e < k * k; e += 64) M.store(mat, e / k, e % k, As[(e / k) * PLD + (e % k)]);
}

#endif

// psd_tiled.hip: the same projection on NT x NT tiles for 16 < k <= 64
int psd_project_tiled_plain(double* A, int64_t count, int k, double eps, hipStream_t st);
int psd_project_tiled_cost(double* c_xx, double* c_ux, double* c_uu, int64_t count, int n, int m, double eps, hipStream_t st);

}  // namespace zm

extern "C" int zm_psd_project_f64(double* A, int64_t count, int k, double eps, void* stream) {
    if (count == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!A) return zm::set_error(ZM_EINVAL, "zm_psd_project_f64: null pointer");
    if (count < 0 || k < 1) return zm::set_error(ZM_EINVAL, "zm_psd_project_f64: bad size");
    if (k > 64) return zm::set_error(ZM_EUNSUPPORTED, "zm_psd_project_f64: k=%d > 64", k);
    if (k > zm::PK) return zm::psd_project_tiled_plain(A, count, k, eps, (hipStream_t)stream);
    hipLaunchKernelGGL((zm::psd_project_kernel<zm::PlainMat>), dim3((unsigned)count), dim3(64), 0, (hipStream_t)stream,
                       zm::PlainMat{A, k}, k, eps, (long)count);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

extern "C" int zm_condition_cost_f64(double* c_xx, double* c_ux, double* c_uu, int64_t count, int n, int m, double eps,
                                     void* stream) {
    if (count == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!c_xx || !c_ux || !c_uu) return zm::set_error(ZM_EINVAL, "zm_condition_cost_f64: null pointer");
    if (count < 0 || n < 1 || m < 1) return zm::set_error(ZM_EINVAL, "zm_condition_cost_f64: bad size");
    if (n + m > 64) return zm::set_error(ZM_EUNSUPPORTED, "zm_condition_cost_f64: n+m=%d > 64", n + m);
    if (n + m > zm::PK) return zm::psd_project_tiled_cost(c_xx, c_ux, c_uu, count, n, m, eps, (hipStream_t)stream);
    hipLaunchKernelGGL((zm::psd_project_kernel<zm::StackedCost>), dim3((unsigned)count), dim3(64), 0, (hipStream_t)stream,
                       zm::StackedCost{c_xx, c_ux, c_uu, n, m}, n + m, eps, (long)count);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

extern "C" int zm_condition_dynamics_f64(const double* f_xx, const double* f_ux, const double* f_uu, const double* v_x,
                                         double* vf_xx, double* vf_ux, double* vf_uu, int64_t count, int n, int m, double eps,
                                         void* stream) {
    if (count == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!f_xx || !f_ux || !f_uu || !v_x || !vf_xx || !vf_ux || !vf_uu)
        return zm::set_error(ZM_EINVAL, "zm_condition_dynamics_f64: null pointer");
    if (count < 0 || n < 1 || m < 1) return zm::set_error(ZM_EINVAL, "zm_condition_dynamics_f64: bad size");
    if (n + m > zm::PK) return zm::set_error(ZM_EUNSUPPORTED, "zm_condition_dynamics_f64: n+m=%d > 16", n + m);
    hipLaunchKernelGGL((zm::psd_project_kernel<zm::ContractedDynamics>), dim3((unsigned)count), dim3(64), 0, (hipStream_t)stream,
                       zm::ContractedDynamics{f_xx, f_ux, f_uu, v_x, vf_xx, vf_ux, vf_uu, n, m}, n + m, eps, (long)count);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
