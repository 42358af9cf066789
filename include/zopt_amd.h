/* zopt_amd -- C ABI of the MI355X-native batched LQR / iLQR / MPC solve engine.
 *
 * The reference (zprihoda/zopt) is pure Python and has NO FFI layer: its boundary for this path is
 * the Python signatures of zopt/lqrUtils.py, zopt/ilqrUtils.py and zopt/mpcUtils.py.  This header is
 * the C-ABI a maintainer would bind underneath those signatures (ctypes stub: INTEGRATION.md).
 * Each entry point names the reference function (file:line) whose arithmetic it replaces.
 *
 * Conventions
 *   - All arrays are C-contiguous with the reference's time-first layout (lqrUtils.py:156-159) and ONE
 *     extra leading axis `batch` (new in the build; the reference is batch == 1).
 *   - `*_f64` entry points take DEVICE pointers (hipMalloc / torch ROCm `data_ptr()`), launch
 *     asynchronously on `stream` (a hipStream_t passed as void*, NULL = default stream) and return
 *     without synchronising.  `*_host_*` variants take HOST pointers, stage through device memory
 *     they allocate and free themselves, and synchronise before returning.
 *   - The caller owns every buffer; the library never retains a pointer after the call returns.
 *   - Return value: 0 = ok; <0 = ZM_E* below (bad argument / unsupported shape); >0 = hipError_t.
 *     `zm_last_error()` gives a thread-local message for the last non-zero return.
 *   - Numerics follow the reference: a singular `solve` propagates inf/NaN, nothing throws.
 */
#ifndef ZOPT_AMD_H
#define ZOPT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZM_OK 0
#define ZM_EINVAL (-1)       /* null pointer, non-positive size */
#define ZM_EUNSUPPORTED (-2) /* shape outside what the compiled kernels cover (see zm_lqr_backward_supported) */

/* Library version (major*10000 + minor*100 + patch). */
int zm_version(void);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* zm_last_error(void);

/* 1 if (n, m) is covered by the compiled kernels for the given element size (8 = fp64), else 0. */
int zm_lqr_backward_supported(int n, int m, int elem_size);

/* Batched discrete finite-horizon LQR backward Riccati recursion (Joseph-form value update).
 * Replaces: zopt/lqrUtils.py:144-173  discreteFiniteHorizonLqr(A, B, Q, R, N) -> L
 *     V <- Q[T-1];  for k = T-1..0:  L_k = solve(R_k + B_k^T V B_k, B_k^T V A_k)              (:168)
 *                                    V = Q_k + L_k^T R_k L_k + (A_k-B_k L_k)^T V (A_k-B_k L_k) (:169)
 * in : A (batch,T,n,n)  B (batch,T,n,m)  Q (batch,T,n,n)  R (batch,T,m,m)
 * out: L (batch,T,m,n)          control law u = -L x (lqrUtils.py:151)
 */
int zm_lqr_backward_f64(const double* A, const double* B, const double* Q, const double* R, double* L,
                        int64_t batch, int T, int n, int m, void* stream);

/* Same, HOST pointers (NumPy arrays): allocates device staging, copies in, solves, copies L back. */
int zm_lqr_backward_host_f64(const double* A, const double* B, const double* Q, const double* R, double* L,
                             int64_t batch, int T, int n, int m);

/* Batched iLQR backward pass: affine policy (l, L) from the quadratic model along a trajectory.
 * Replaces: zopt/ilqrUtils.py:153-181  riccatiStep_ilqr / backwardPass_ilqr(dynamics, cost, Vf) -> AffinePolicy
 *     Q_x = c_x + f_x^T v_x   Q_u = c_u + f_u^T v_x   Q_xx = c_xx + f_x^T v_xx f_x   Q_uu = c_uu + f_u^T v_xx f_u
 *     Q_ux = c_ux + f_u^T v_xx f_x   l = -solve(Q_uu, Q_u)   L = -solve(Q_uu, Q_ux)                     (:160-168)
 *     v_x' = Q_x - L^T Q_uu l        v_xx' = Q_xx - L^T Q_uu L                                          (:170)
 * (the scalar terms c, v and AffineDynamics.f do not influence the returned policy and are not taken)
 * in : f_x (batch,T,n,n) f_u (batch,T,n,m)                          AffineDynamics   (pytrees.py:129-136)
 *      c_x (batch,T,n) c_u (batch,T,m) c_xx (batch,T,n,n) c_ux (batch,T,m,n) c_uu (batch,T,m,m)   QuadraticCostFunction (:84-98)
 *      vf_x (batch,n) vf_xx (batch,n,n)                             QuadraticValueFunction Vf (:58-69)
 * out: l (batch,T,m)  L (batch,T,m,n)                               AffinePolicy (:207-213), u = alpha*l + L dx + uPrev
 */
int zm_ilqr_backward_f64(const double* f_x, const double* f_u, const double* c_x, const double* c_u,
                         const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                         const double* vf_xx, double* l, double* L, int64_t batch, int T, int n, int m, void* stream);

/* ---- registered device models (the reference differentiates / rolls out arbitrary Python callables with JAX;
 *      a kernel needs the model in device code) ------------------------------------------------------------------ */
#define ZM_MODEL_LINEAR 1     /* x+ = A x + B u, A (n,n), B (n,m) device pointers shared by the whole batch           */
#define ZM_MODEL_QUADCOPTER 2 /* x+ = x + dt * Quadcopter.inertialDynamics(x,u)  (zopt/quadcopter.py:116-144), n=12, m=4 */
typedef struct zm_model_t {
    int kind, n, m, reserved;
    double dt;              /* quadcopter: Euler step (demos/iterativeLqr.py:23,35) */
    const double* A;        /* linear: device pointers; else NULL */
    const double* B;
} zm_model_t;

/* c(x,u) = x^T Q x + u^T R u,  c_f(x) = x^T Qf x  (demos/iterativeLqr.py:12-13,37; no 1/2).  Device pointers. */
typedef struct zm_quadcost_t {
    const double* Q;        /* (n,n) */
    const double* R;        /* (m,m) */
    const double* Qf;       /* (n,n) */
} zm_quadcost_t;

/* Batched policy rollout with a parallel line search over step sizes.
 * Replaces: zopt/ilqrUtils.py:33-66 trajectoryRollout (n_alpha = 1, cost may be NULL) and :116-150 forwardPass2
 *           (n_alpha = 16, alphas = 0.5^j), with AffinePolicy.__call__ (pytrees.py:215-220) and CostFunction.__call__
 *           (pytrees.py:49-52):
 *     for each alpha:  x_0 = x0;  u_k = alpha*l_k + L_k (x_k - xPrev_k) + uPrev_k;  x_{k+1} = f(x_k, u_k)
 *                      J = sum_k c(x_k,u_k) + c_f(x_T);        result = the rollout with the smallest J (NaN wins, as argmin)
 * in : x0 (batch,n)  l (batch,T,m)  L (batch,T,m,n)  xPrev (batch,T+1,n)  uPrev (batch,T,m)  alphas (n_alpha) [device]
 *      active (batch) int32 or NULL: trajectories with active==0 are skipped (their outputs are left untouched)
 * out: xTraj (batch,T+1,n)  uTraj (batch,T,m)  J (batch) or NULL  alpha_idx (batch) int32 or NULL
 */
int zm_rollout_linesearch_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* l,
                              const double* L, const double* xPrev, const double* uPrev, const double* alphas,
                              int n_alpha, const int32_t* active, double* xTraj, double* uTraj, double* J,
                              int32_t* alpha_idx, int64_t batch, int T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ZOPT_AMD_H */
