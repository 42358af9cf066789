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

/* Releases the few host-side resources the library keeps between calls (one pinned int32 + one event per device and concurrent
 * zm_ilqr_solve_f64 caller, used for the drivers' host round trips); returns how many were released.  Call it before unloading the
 * library or destroying the HIP context; calling it at any other time is harmless (the next solve re-creates what it needs). */
int zm_shutdown(void);

/* 1 if (n, m) is covered by the compiled LQR sweep kernels for the given element size, else 0.
 *   8 = fp64: tile-16 MFMA kernels for n <= 12, m <= 4 (the LDS-DMA fast path at n in {8, 12}, m = 4: ~1.5e9 steps/s at (12, 4));
 *             register-tile fp64 MFMA kernel for n <= 64, m <= 16: three tile rows with a prefetched operand set up to n = 48
 *             (54 M steps/s at (48, 16)), four tile rows without it for 48 < n <= 64 (17 M steps/s at (64, 16); the LDS-resident
 *             coverage kernel it replaced there ran 1.4 M and stays selectable with ZOPT_AMD_LQR_PATH=lds);
 *   4 = fp32: zm_lqr_backward_f32, n <= 64, m <= 16 (n in {8, 12}, m = 4 with 16-B aligned arrays: fp32 storage, fp64 arithmetic on
 *             the LDS-DMA ring kernel -- 1.8e9 steps/s at (12, 4); the other shapes up to 16: the fp32 MFMA tile kernel). */
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

/* Finite-horizon LQR with bilinear cost (cross term H), affine dynamics (offset d) and linear costs (q, r).
 * Replaces: zopt/lqrUtils.py:207-262  bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, N) -> (L, l)
 *     carry (V, v) <- (Q[T-1], q[T-1]);  Su = r + v^T B + d^T V B   Suu = R + B^T V B   Sux = H + B^T V A   (:244-246)
 *     L = solve(Suu, Sux)  l = solve(Suu, Su)  V' = Q + A^T V A - L^T Suu L  v' = q + A^T (v + V d) - Sux^T l  (:248-252)
 * (q0 / v0 never influence L or l and are not taken.)          control law u = -L x - l
 * in : A (batch,T,n,n) B (batch,T,n,m) d (batch,T,n) Q (batch,T,n,n) R (batch,T,m,m) H (batch,T,m,n) q (batch,T,n) r (batch,T,m)
 * out: L (batch,T,m,n)  l (batch,T,m)
 */
int zm_lqr_backward_affine_f64(const double* A, const double* B, const double* d, const double* Q, const double* R,
                               const double* H, const double* q, const double* r, double* L, double* l,
                               int64_t batch, int T, int n, int m, void* stream);

/* Batched discrete-time infinite-horizon LQR (DARE) by Riccati value iteration from V = Q on registers.
 * Replaces: zopt/lqrUtils.py:176-204 discreteInfiniteHorizonLqr (SciPy solve_discrete_are + one solve).
 * in : A (batch,n,n)  B (batch,n,m)  Q (batch,n,n)  R (batch,m,m)   [device]; n <= 64, m <= 16 (n <= 12, m <= 4: K1's step with the
 *      operands in registers, stopping on max|V' - V| <= tol max|V'|; beyond: the fp64 tile kernel's Joseph-form step on time-invariant
 *      operands, stopping on max|L_k - L_{k-1}| <= tol max|L_k|, P = the value after that step)
 *      tol: stop when max|V' - V| <= tol * max|V'| (or at the rounding floor); max_iter: iteration cap
 * out: L (batch,m,n) with u = -L x;  P (batch,n,n) or NULL: the value matrix;
 *      iters (batch) or NULL: +k = converged after k iterations (also when k == max_iter), -k = the cap ended the loop after k
 *      iterations without the stopping test being met (the gain is then not a stationary one)
 */
int zm_dare_f64(const double* A, const double* B, const double* Q, const double* R, double* L, double* P, int32_t* iters,
                int64_t batch, int n, int m, double tol, int max_iter, void* stream);

/* Batched continuous-time infinite-horizon LQR (CARE  A^T P + P A - P B R^-1 B^T P + Q = 0) by the structure-preserving
 * doubling algorithm, one wave per design.
 * Replaces: zopt/lqrUtils.py:13-36 infiniteHorizonLqr (SciPy solve_continuous_are + one solve); with the integral-augmented
 *           system of lqrUtils.py:101-141 also infiniteHorizonIntegralLqr.
 * in : A (batch,n,n)  B (batch,n,m)  Q (batch,n,n)  R (batch,m,m)   [device]; n <= 16, m <= 16; Q, R symmetric
 *      tol: stop when max|P' - P| <= tol * max|P'|; max_iter: cap on doubling steps (each squares the contraction: ~10 suffice)
 * out: K (batch,m,n) = R^-1 B^T P with u = -K x;  P (batch,n,n) or NULL;
 *      info (batch) or NULL: doubling steps taken, -1 not converged, -2 singular pivot / no stabilising solution
 */
int zm_care_f64(const double* A, const double* B, const double* Q, const double* R, double* K, double* P, int32_t* info,
                int64_t batch, int n, int m, double tol, int max_iter, void* stream);

/* Batched continuous-time finite-horizon LQR value function: dV/dt = -Q + V S V - V A - A^T V, V(T) = Qf, S = B R^-1 B^T,
 * integrated backwards with an adaptive Dormand-Prince 5(4) pair, output at t_j = j T / (N - 1), j = 0..N-1.
 * Replaces: zopt/lqrUtils.py:39-98 finiteHorizonLqr (_lqrHjb + jax.experimental.ode.odeint; its gain interpolation,
 *           jaxUtils.py:7-24, stays on the host side of the boundary).
 * in : A_s, Q_s (batch,n_samples,n,n), B_s (batch,n_samples,n,m), Rinv_s (batch,n_samples,m,m) [device]: coefficients sampled
 *      at linspace(0, T, n_samples), linear in between (n_samples = 1: time-invariant);  Qf (batch,n,n);  n <= 16, m <= 16;
 *      rtol, atol: local error control (odeint: 1.4e-8 both)
 * out: V (batch,N,n,n);  info (batch) or NULL: integration steps taken, -1 step cap reached, -2 NaN (finite escape time)
 */
int zm_riccati_ode_f64(const double* A_s, const double* B_s, const double* Rinv_s, const double* Q_s, const double* Qf,
                       double* V, int32_t* info, int64_t batch, int n, int m, int n_samples, int N, double T, double rtol,
                       double atol, int max_steps, void* stream);

/* fp32 batched finite-horizon LQR backward sweep (n <= 64, m <= 16): the fp32 MFMA tile kernel; at the fast-path shapes (n in {8, 12},
 * m = 4, 16-B aligned arrays) the LDS-DMA ring kernel on fp32 storage with fp64 arithmetic (the fp64 result rounded once).
 * Replaces: zopt/lqrUtils.py:144-173 discreteFiniteHorizonLqr when JAX runs in its default fp32 mode (x64 disabled, quirk Q8);
 *           the "large-state stress" shape n=64, m=16, T=200 of BASELINE configs[4].
 * in : A (batch,T,n,n)  B (batch,T,n,m)  Q (batch,T,n,n)  R (batch,T,m,m)   [device, C-contiguous float]
 * out: L (batch,T,m,n)
 */
int zm_lqr_backward_f32(const float* A, const float* B, const float* Q, const float* R, float* L, int64_t batch, int T, int n,
                        int m, void* stream);

/* zm_lqr_backward_f64 with HOST pointers (NumPy arrays): allocates device staging, copies in, solves, copies L back. */
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
 * Shapes: n <= 12, m <= 4 on the tile-16 kernels (LDS-DMA ring at n in {8, 12}, m = 4); beyond, up to n <= 48, m <= 16, on the fp64
 * MFMA tile sweep (sweep_tiled_f64.hip: V, [f_x | f_u] and every product as 16 x 16 register tiles, one wave per trajectory);
 * ZM_EUNSUPPORTED above.  zm_lqr_backward_affine_f64 takes the same shapes.
 */
int zm_ilqr_backward_f64(const double* f_x, const double* f_u, const double* c_x, const double* c_u,
                         const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                         const double* vf_xx, double* l, double* L, int64_t batch, int T, int n, int m, void* stream);

/* ---- registered device models (the reference differentiates / rolls out arbitrary Python callables with JAX;
 *      a kernel needs the model in device code) ------------------------------------------------------------------ */
#define ZM_MODEL_LINEAR 1     /* x+ = A x + B u, A (n,n), B (n,m) device pointers shared by the whole batch           */
#define ZM_MODEL_QUADCOPTER 2 /* x+ = x + dt * Quadcopter.inertialDynamics(x,u)  (zopt/quadcopter.py:116-144), n=12, m=4; dt = 0: the derivative */
#define ZM_MODEL_QUADCOPTER_RB 3 /* same with Quadcopter.rigidBodyDynamics (:70-113), n=8, m=4; wind_ned holds the BODY-frame wind */
typedef struct zm_model_t {
    int kind, n, m, reserved;
    double dt;              /* quadcopter: Euler step (demos/iterativeLqr.py:23,35) */
    const double* A;        /* linear: device pointers; else NULL */
    const double* B;
    double wind_ned[3];     /* quadcopter: constant wind in the north-east-down frame (quadcopter.py:117,138; the closed-loop
                               simulation of demos/iterativeLqr.py:48 uses (3,1,0)); zeros for the solvers */
} zm_model_t;

/* c(x,u) = x^T Q x + u^T R u,  c_f(x) = x^T Qf x  (demos/iterativeLqr.py:12-13,37; no 1/2).  Device pointers. */
typedef struct zm_quadcost_t {
    const double* Q;        /* (n,n) */
    const double* R;        /* (m,m) */
    const double* Qf;       /* (n,n) */
    int32_t diagonal;       /* 1: the caller asserts that Q, R and Qf are diagonal (the demos' weights) -- off-diagonal entries are
                               then never read and the rollout runs a leaner kernel; 0: unknown, the kernels look for themselves */
    int32_t reserved;
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
 * Shapes: registered models with n <= 12, m <= 4 (one lane per rollout; the (12, 4) fast paths of rollout_fast.hip / rollout_quad.hip);
 *      ZM_MODEL_LINEAR also beyond, up to n <= 64, m <= 16 (rollout_wide.hip: one wave per rollout, lane i owns state i; four waves
 *      per trajectory share the 16 step sizes, the winner is rolled out once more with stores; no scratch).
 */
int zm_rollout_linesearch_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* l,
                              const double* L, const double* xPrev, const double* uPrev, const double* alphas,
                              int n_alpha, const int32_t* active, double* xTraj, double* uTraj, double* J,
                              int32_t* alpha_idx, int64_t batch, int T, void* stream);

/* Same, over a compacted list of trajectory ids: only list[0..count) are processed (densely packed into waves), all
 * other trajectories keep their outputs.  Arrays keep their full (batch, ...) shapes and are indexed by trajectory id.
 * Used by the iLQR / DDP drivers once most of the batch has converged (a mask would leave mostly idle waves).  `active`
 * (batch) int32 or NULL: listed trajectories with active == 0 are skipped as well, so a list may be a few iterations old. */
int zm_rollout_linesearch_list_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* l,
                                   const double* L, const double* xPrev, const double* uPrev, const double* alphas,
                                   int n_alpha, const int32_t* list, int64_t count, const int32_t* active, double* xTraj,
                                   double* uTraj, double* J, int32_t* alpha_idx, int64_t batch, int T, void* stream);

/* Acceptance step of the iLQR / DDP loop (zopt/ilqrUtils.py:316-320) for the trajectories in list[0..count):
 *     converged = |J - Jn| <= tol;  J <- Jn;  xTraj <- xTrajNew;  uTraj <- uTrajNew;  active <- !converged
 * Arrays keep their (batch, ...) shapes; rows of trajectories not listed -- or listed but with active == 0 -- are left untouched. */
int zm_ilqr_accept_f64(const int32_t* list, int64_t count, double* J, const double* Jn, double* xTraj, const double* xTrajNew,
                       double* uTraj, const double* uTrajNew, int32_t* converged, int32_t* active, double tol, int64_t batch,
                       int T, int n, int m, void* stream);

/* First-order expansion of a registered model along a trajectory.
 * Replaces: zopt/pytrees.py:139-153 AffineDynamics.from_function / from_trajectory (jax.jacobian of dynFun at
 *           (xTraj[:-1], uTraj)):  f = dynFun(x_k,u_k), f_x = d dynFun/dx, f_u = d dynFun/du.
 * in : xTraj (batch,T+1,n)  uTraj (batch,T,m)   active (batch) int32 or NULL (inactive trajectories are skipped)
 * out: f (batch,T,n) or NULL   f_x (batch,T,n,n)   f_u (batch,T,n,m)
 */
int zm_linearize_dynamics_f64(const zm_model_t* model, const double* xTraj, const double* uTraj, const int32_t* active,
                              double* f, double* f_x, double* f_u, int64_t batch, int T, void* stream);

/* Second-order expansion of the registered quadratic cost along a trajectory, and of the terminal cost at x_T.
 * Replaces: zopt/pytrees.py:100-115 QuadraticCostFunction.from_function / from_trajectory and :72-81
 *           QuadraticValueFunction.fromTerminalCostFunction for c = x'Qx + u'Ru, c_f = x'Qf x:
 *     c_x = (Q+Q')x  c_u = (R+R')u  c_xx = Q+Q'  c_ux = 0  c_uu = R+R'    v = x'Qf x  v_x = (Qf+Qf')x  v_xx = Qf+Qf'
 * The Hessians do not depend on the trajectory: they are written ONCE (c_xx (n,n), c_ux (m,n), c_uu (m,m),
 * v_xx (n,n)) and consumed with `shared_hessian = 1` by zm_ilqr_backward_ex_f64.
 * in : xTraj (batch,T+1,n)  uTraj (batch,T,m)  active or NULL
 * out: c (batch,T)  c_x (batch,T,n)  c_u (batch,T,m)  v (batch)  v_x (batch,n)   (each may be NULL)
 *      c_xx (n,n)  c_ux (m,n)  c_uu (m,m)  v_xx (n,n)                              (each may be NULL)
 */
int zm_quadratize_cost_f64(const zm_quadcost_t* cost, int n, int m, const double* xTraj, const double* uTraj,
                           const int32_t* active, double* c, double* c_x, double* c_u, double* v, double* v_x,
                           double* c_xx, double* c_ux, double* c_uu, double* v_xx, int64_t batch, int T, void* stream);

/* zm_ilqr_backward_f64 with (a) an `active` mask (inactive trajectories keep their previous l, L) and
 * (b) `shared_hessian` != 0: c_xx (n,n), c_ux (m,n), c_uu (m,m) and vf_xx (n,n) are single matrices shared by every
 * trajectory and step (time-invariant quadratic cost) instead of (batch,T,.,.) / (batch,.,.) arrays. */
int zm_ilqr_backward_ex_f64(const double* f_x, const double* f_u, const double* c_x, const double* c_u,
                            const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                            const double* vf_xx, const int32_t* active, int shared_hessian, double* l, double* L,
                            int64_t batch, int T, int n, int m, void* stream);

/* Batched DDP backward pass: zm_ilqr_backward_ex_f64 plus the second-order dynamics terms.
 * Replaces: zopt/ilqrUtils.py:184-214 riccatiStep_ddp / backwardPass_ddp and :237-251 conditionQuadraticDynamics:
 *     vf_.. = einsum('i,ijk', v_x, f_..);  [[vf_xx, vf_ux^T],[vf_ux, vf_uu]] <- ensurePositiveDefinite(.)  per step,
 *     Q_xx += vf_xx, Q_uu += vf_uu, Q_ux += vf_ux, then the iLQR step.
 * in : as zm_ilqr_backward_ex_f64, plus QuadraticDynamics (pytrees.py:165-177)
 *      f_xx (batch,T,n,n,n)  f_ux (batch,T,n,m,n)  f_uu (batch,T,n,m,m)     [f_..[i,j,k] = d2 f_i / d._j d._k]
 *      (f_ux and f_uu both NULL: identically zero, i.e. dynamics affine in the controls)
 * out: l (batch,T,m)  L (batch,T,m,n)
 */
int zm_ddp_backward_f64(const double* f_x, const double* f_u, const double* f_xx, const double* f_ux, const double* f_uu,
                        const double* c_x, const double* c_u, const double* c_xx, const double* c_ux, const double* c_uu,
                        const double* vf_x, const double* vf_xx, const int32_t* active, int shared_hessian, double* l,
                        double* L, int64_t batch, int T, int n, int m, void* stream);

/* The same sweeps, also returning the quadratic value function they end with -- what the reference's single-step helpers
 * riccatiStep_ilqr (ilqrUtils.py:153-173) and riccatiStep_ddp (:184-206) return next to the policy (T = 1), and the value at the
 * trajectory start for T > 1:   v' = (c + v) - 1/2 l^T Q_uu l,  v_x' = Q_x - L^T Q_uu l,  v_xx' = Q_xx - L^T Q_uu L   (:170).
 * in : as zm_ilqr_backward_f64, plus c (batch,T) or NULL and vf (batch) or NULL (the scalar terms);
 *      f_xx, f_ux, f_uu all NULL: iLQR step; all given: DDP step (PD-projected second-order dynamics terms)
 * out: l, L as before;  v_out (batch), vx_out (batch,n), vxx_out (batch,n,n), each may be NULL  * Shapes: n <= 12, m <= 4; the iLQR form (f_xx == f_ux == f_uu == NULL) also up to n <= 48, m <= 16 on the tile sweep (sweep_tiled_f64.hip;
 * the three value outputs are then required).
 */
int zm_riccati_value_f64(const double* f_x, const double* f_u, const double* f_xx, const double* f_ux, const double* f_uu,
                         const double* c, const double* c_x, const double* c_u, const double* c_xx, const double* c_ux,
                         const double* c_uu, const double* vf, const double* vf_x, const double* vf_xx, double* l, double* L,
                         double* v_out, double* vx_out, double* vxx_out, int64_t batch, int T, int n, int m, void* stream);

/* zm_ilqr_backward_ex_f64 / zm_ddp_backward_f64 over a compacted list of trajectory ids: one wave per LISTED trajectory (grid =
 * count), so that the few hundred stragglers of a solve spread over the whole chip instead of sharing SIMDs among 8192 mostly idle
 * blocks.  Arrays keep their (batch, ...) shapes; listed trajectories with active == 0 are skipped; list == NULL: all of them. */
int zm_ilqr_backward_list_f64(const double* f_x, const double* f_u, const double* c_x, const double* c_u, const double* c_xx,
                              const double* c_ux, const double* c_uu, const double* vf_x, const double* vf_xx, const int32_t* list,
                              int64_t count, const int32_t* active, int shared_hessian, double* l, double* L, int64_t batch, int T,
                              int n, int m, void* stream);
int zm_ddp_backward_list_f64(const double* f_x, const double* f_u, const double* f_xx, const double* f_ux, const double* f_uu,
                             const double* c_x, const double* c_u, const double* c_xx, const double* c_ux, const double* c_uu,
                             const double* vf_x, const double* vf_xx, const int32_t* list, int64_t count, const int32_t* active,
                             int shared_hessian, double* l, double* L, int64_t batch, int T, int n, int m, void* stream);

/* The three expansions over a compacted list of trajectory ids (as zm_rollout_linesearch_list_f64): only list[0..count) are expanded
 * -- the grids shrink with the list -- all arrays keep their full (batch, ...) shapes and are indexed by trajectory id; listed
 * trajectories with active == 0 are skipped.  list == NULL: every trajectory, i.e. the plain entry points below / above. */
int zm_linearize_dynamics_list_f64(const zm_model_t* model, const double* xTraj, const double* uTraj, const int32_t* list,
                                   int64_t count, const int32_t* active, double* f, double* f_x, double* f_u, int64_t batch, int T,
                                   void* stream);
int zm_quadratize_cost_list_f64(const zm_quadcost_t* cost, int n, int m, const double* xTraj, const double* uTraj,
                                const int32_t* list, int64_t count, const int32_t* active, double* c, double* c_x, double* c_u,
                                double* v, double* v_x, double* c_xx, double* c_ux, double* c_uu, double* v_xx, int64_t batch,
                                int T, void* stream);
int zm_quadratic_dynamics_list_f64(const zm_model_t* model, const double* xTraj, const double* uTraj, const int32_t* list,
                                   int64_t count, const int32_t* active, double* f_xx, double* f_ux, double* f_uu, int64_t batch,
                                   int T, void* stream);

/* Variables in which a registered model is NOT affine: bit i = state i, bit n + j = control j.  Only pairs of these have a nonzero
 * second derivative -- the second-order expansion evaluates only those pairs, and a driver need not materialise the zero blocks. */
int zm_model_nonlinear_mask(const zm_model_t* model, uint32_t* mask);

/* Second-order expansion of a registered model along a trajectory (forward-mode hyper-dual numbers).
 * Replaces: zopt/pytrees.py:180-194 QuadraticDynamics.from_function / from_trajectory (jax.hessian of dynFun):
 * in : xTraj (batch,T+1,n)  uTraj (batch,T,m)  active or NULL
 * out: f_xx (batch,T,n,n,n)  f_ux (batch,T,n,m,n)  f_uu (batch,T,n,m,m)
 *      f_ux and f_uu may both be NULL for a model that is affine in its controls (no control bit in zm_model_nonlinear_mask):
 *      they are identically zero and zm_ddp_backward_f64 / zm_riccati_value_f64 take NULL for them as well.
 */
int zm_quadratic_dynamics_f64(const zm_model_t* model, const double* xTraj, const double* uTraj, const int32_t* active,
                              double* f_xx, double* f_ux, double* f_uu, int64_t batch, int T, void* stream);

/* Batched trim of the quadcopter (the producer side of the path).
 * Replaces: zopt/quadcopter.py:146-177 Quadcopter.trim(uvwTrim): find x = [uvw, p,q,r,phi,theta], u = [thrust,mx,my,mz] with
 *           rigidBodyDynamics(x, u) = 0 (the reference minimises the squared residual with SciPy BFGS; its test accepts
 *           |residual| <= 1e-3).  Levenberg-Marquardt from the reference's start point, one lane per instance.
 * in : uvw (batch,3) [device]; wind_body (3) HOST pointer or NULL (body-frame wind, zeros in the reference's trim)
 * out: xTrim (batch,8)  uTrim (batch,4)  resid (batch) = |rigidBodyDynamics|_2 or NULL  ok (batch) int32 (resid <= tol) or NULL */
int zm_quadcopter_trim_f64(const double* uvw, const double* wind_body, double* xTrim, double* uTrim, double* resid, int32_t* ok,
                           int64_t batch, double tol, void* stream);

/* Batched projection onto the positive definite cone: A <- V max(w, eps) V^T with (w, V) = eigh((A + A^T)/2).
 * Replaces: zopt/ilqrUtils.py:217-219 ensurePositiveDefinite (jnp.linalg.eigh symmetrises its input) and its users
 *           :254-257 conditionValueFunction (k = n).       in/out: A (count,k,k) in place, k <= 64: one wave per matrix, matrix-sign
 *           iterations on fp64 MFMA tiles instead of an eigen-decomposition -- one 16 x 16 tile up to k = 16 (psd.hip, ns16.h), NT x NT
 *           tiles beyond (psd_tiled.hip; the same iteration, constants and caps).  zm_condition_cost_f64 likewise for n + m <= 64;
 *           zm_condition_dynamics_f64 (the DDP path) stays at n + m <= 16.
 */
int zm_psd_project_f64(double* A, int64_t count, int k, double eps, void* stream);

/* Same projection applied to the stacked cost Hessian [[c_xx, c_ux^T],[c_ux, c_uu]] of every step, written back
 * into its blocks.  Replaces: zopt/ilqrUtils.py:222-234 conditionQuadraticCost.
 * in/out: c_xx (count,n,n)  c_ux (count,m,n)  c_uu (count,m,m) in place (count = batch*T, or 1 for a shared Hessian)
 */
int zm_condition_cost_f64(double* c_xx, double* c_ux, double* c_uu, int64_t count, int n, int m, double eps,
                          void* stream);

/* Second-order dynamics terms of the DDP step, PD-conditioned.
 * Replaces: zopt/ilqrUtils.py:237-251 conditionQuadraticDynamics(quadratic_dynamics, v_x):
 *     vf_.. = einsum('i,ijk', v_x, f_..);  [[vf_xx, vf_ux^T],[vf_ux, vf_uu]] <- ensurePositiveDefinite(.);  blocks sliced back
 * in : f_xx (count,n,n,n)  f_ux (count,n,m,n)  f_uu (count,n,m,m)  v_x (count,n)
 * out: vf_xx (count,n,n)  vf_ux (count,m,n)  vf_uu (count,m,m) */
int zm_condition_dynamics_f64(const double* f_xx, const double* f_ux, const double* f_uu, const double* v_x, double* vf_xx,
                              double* vf_ux, double* vf_uu, int64_t count, int n, int m, double eps, void* stream);

/* ---- box-constrained LQ-MPC (reference: zopt/mpcUtils.py:12-81, class lqrMpc; the reference hands this QP to
 *      cvxpy -> OSQP, whose arithmetic is not in the reference tree: numeric parity is "unpinned", acceptance is by KKT
 *      residuals -- see DESIGN.md) -------------------------------------------------------------------------------
 *   min  sum_{k<N} (x_k'Q x_k + u_k'R u_k) + x_N'Qf x_N        (mpcUtils.py:52-54, no 1/2)
 *   s.t. x_{k+1} = A x_k + B u_k,  x_lb <= x_k <= x_ub (k = 0..N),  u_lb <= u_k <= u_ub,  x_0 = x0      (:55-58)
 * Method: ADMM on the splitting {dynamics-feasible trajectory w} / {box copy y}; the w-update is an LQ tracking
 * problem solved exactly by a Riccati factorisation that does not depend on the iterates (zm_mpc_setup_f64, once per
 * problem), so one ADMM iteration is an affine backward sweep + a forward rollout + a clip.
 */
#define ZM_MPC_OPTIMAL 1            /* cvxpy status "optimal"            */
#define ZM_MPC_INFEASIBLE 2         /* "infeasible"                      */
#define ZM_MPC_USER_LIMIT 3         /* "user_limit" (max_iter reached)   */
#define ZM_MPC_OPTIMAL_INACCURATE 4 /* "optimal_inaccurate": max_iter reached with both residuals within 10x their tolerances -- OSQP's
                                       "solved inaccurate", which cvxpy reports under this name (mpcUtils.py:74,78).  The rest of cvxpy's
                                       vocabulary cannot arise here: "unbounded" needs a direction of unbounded descent, which a cost
                                       with Q, Qf >= 0 and R > 0 over dynamics-feasible trajectories does not have, and
                                       "infeasible_inaccurate" is reported as "user_limit" (an uncertified instance at the cap) */

/* Riccati tables of the ADMM w-update for penalty rho:  K (N,m,n), Minv (N,m,m)   [all device pointers]
 * in : A (n,n) B (n,m) Q (n,n) R (m,m) Qf (n,n); n <= 24, m <= 8.
 * Shapes of the solve entry points: (n, m) in {(12,4), (8,4), (4,2), (4,1), (2,2), (2,1), (1,1)} (16 lanes per instance, adaptive penalty
 * levels; the Python mirror embeds any smaller shape with inert padding) and (24, 8) (one lane per instance, the penalty of `level0`
 * only; registers + scratch: a coverage path for problems beyond the 16-index tile). */
int zm_mpc_setup_f64(const double* A, const double* B, const double* Q, const double* R, const double* Qf, double rho,
                     int N, int n, int m, double* K, double* Minv, void* stream);

/* Solve `batch` instances (one per initial state) of the QP above.
 * in : A, B (shared), K, Minv from zm_mpc_setup_f64 (same rho), bounds x_lb, x_ub (n), u_lb, u_ub (m) (+-inf allowed),
 *      x0 (batch,n); workspace: 4 * batch * N * (n+m) doubles
 * out: xTraj (batch,N+1,n)  uTraj (batch,N,m)  status (batch) ZM_MPC_*  iters (batch) or NULL
 *      resid (batch,2) = final (primal, dual) residual inf-norms, or NULL
 */
int zm_mpc_solve_f64(const double* A, const double* B, const double* K, const double* Minv, const double* x_lb,
                     const double* x_ub, const double* u_lb, const double* u_ub, const double* x0, double rho,
                     double eps_abs, double eps_rel, double eps_prim_inf, int max_iter, double* workspace, double* xTraj,
                     double* uTraj,
                     int32_t* status, int32_t* iters, double* resid, int64_t batch, int N, int n, int m, void* stream);

/* Same, with OSQP-style warm starting (cvxpy's default `warm_start=True`, mpcUtils.py:77): with warm_start != 0 the
 * workspace must be the one a previous call for the same (batch, N, n, m, rho) left behind; its iterates (box copy y,
 * scaled dual lam) start the ADMM instead of zeros; warm_start == 2 additionally advances them by one horizon step
 * (iterate k <- iterate k+1, the last one repeated), the natural guess when x0 is the previous plan's x_1 as in the
 * receding-horizon loop (demos/lqrMpc.py:41-48).  Only instances whose previous solve ended "optimal" are warm-started. */
int zm_mpc_solve_warm_f64(const double* A, const double* B, const double* K, const double* Minv, const double* x_lb,
                          const double* x_ub, const double* u_lb, const double* u_ub, const double* x0, double rho,
                          double eps_abs, double eps_rel, double eps_prim_inf, int max_iter, int warm_start,
                          double* workspace, double* xTraj, double* uTraj, int32_t* status, int32_t* iters, double* resid,
                          int64_t batch, int N, int n, int m, void* stream);

/* Same, with OSQP's adaptive penalty (`adaptive_rho`, on by default in OSQP / cvxpy): K and Minv hold the tables of
 * zm_mpc_setup_f64 for n_levels penalties rho * rho_step^(l - level0), level-major (n_levels, N, m, n) / (n_levels, N, m, m).
 * Every 8 iterations an instance moves to the level nearest (on the log scale) rho * sqrt(primal / dual residual ratio),
 * rescaling its scaled dual.  n_levels = 1 is the fixed-penalty solve above. */
int zm_mpc_solve_adaptive_f64(const double* A, const double* B, const double* K, const double* Minv, int n_levels, int level0,
                              double rho_step, const double* x_lb, const double* x_ub, const double* u_lb, const double* u_ub,
                              const double* x0, double rho, double eps_abs, double eps_rel, double eps_prim_inf, int max_iter,
                              int warm_start, double* workspace, double* xTraj, double* uTraj, int32_t* status, int32_t* iters,
                              double* resid, int64_t batch, int N, int n, int m, void* stream);

/* Second derivatives in PACKED form, for models that declare which variable pairs can have a nonzero second derivative.
 * zm_model_hessian_pairs: pairs (2 * npairs int32, host, may be NULL) <- (a, b), a <= b, in the stacked variable index (states
 *     0..n-1, controls n..n+m-1); npairs (host) <- their number, 0 if the model declares none.  The quadcopter declares 28.
 * zm_quadratic_dynamics_pairs_list_f64: H (batch,T,npairs,n) <- d2 f_i / dz_a dz_b: the nonzero entries of
 *     QuadraticDynamics.from_trajectory's f_xx / f_ux / f_uu (zopt/pytrees.py:180-194), 2 688 B per point for the quadcopter
 *     instead of 19 968 B of mostly zeros; `list` / `active` as in zm_quadratic_dynamics_list_f64 (list may be NULL).
 * zm_ddp_backward_pairs_list_f64: zm_ddp_backward_list_f64 (zopt/ilqrUtils.py:184-214, 237-251) reading H instead of the three
 *     tensors -- the contraction sum_i v_x[i] H[p][i] runs in the same order, so the policy is bitwise the same.  Used by
 *     zm_ilqr_solve_f64; the array-level API (full tensors) is zm_ddp_backward_f64. */
int zm_model_hessian_pairs(const zm_model_t* model, int32_t* pairs, int32_t* npairs);
int zm_quadratic_dynamics_pairs_list_f64(const zm_model_t* model, const double* xTraj, const double* uTraj, const int32_t* list,
                                         int64_t count, const int32_t* active, double* H, int64_t batch, int T, void* stream);
int zm_ddp_backward_pairs_list_f64(const zm_model_t* model, const double* f_x, const double* f_u, const double* H,
                                   const double* c_x, const double* c_u, const double* c_xx, const double* c_ux,
                                   const double* c_uu, const double* vf_x, const double* vf_xx, const int32_t* list, int64_t count,
                                   const int32_t* active, int shared_hessian, double* l, double* L, int64_t batch, int T,
                                   void* stream);

/* The whole iLQR / DDP solve of a batch as one call: the outer loop of zopt/ilqrUtils.py:290-327 (iterativeLqr, ddp = 0) and
 * :360-397 (differentialDynamicProgramming, ddp != 0) for a registered model and quadratic cost -- initial rollout of
 * (uGuess, L = 0), then per iteration the expansions along the trajectory, the PD-conditioned Hessians, the backward pass, the
 * 16-way line search and `converged = |J - J_new| <= tol`, each over the compacted list of the trajectories that have not
 * converged yet (rebuilt on the device every `sync_every` >= 1 iterations; results do not depend on it).
 * HOST-SYNCHRONOUS, unlike the other `*_f64` entry points: the host side of the loop runs inside this call and BLOCKS the calling
 * thread on an event every `sync_every` iterations (it must learn how many trajectories are left to size the next launches), so the
 * call returns only when the loop has ended -- after `max_iter` iterations or once every trajectory has converged -- with the last
 * iteration's kernels and the final collect possibly still queued on `stream` (synchronise it before reading the outputs).  Work
 * queued on OTHER streams overlaps as usual.
 * in : x0 (batch,n)  uGuess (batch,T,m)  [device]
 *      workspace: at least zm_ilqr_solve_workspace_f64(model, batch, T, ddp) doubles, 16-B aligned [device]
 *      iwork: 2 * batch + 2 int32 [device]
 * out: xTraj (batch,T+1,n)  uTraj (batch,T,m)  L (batch,T,m,n)  J (batch)  converged (batch) int32  [device]
 *      iterations: HOST int32 or NULL -- iterations the loop ran (<= max_iter) */
int64_t zm_ilqr_solve_workspace_f64(const zm_model_t* model, int64_t batch, int T, int ddp);
int zm_ilqr_solve_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* uGuess, int ddp,
                      int max_iter, double tol, int sync_every, double* workspace, int64_t workspace_doubles, int32_t* iwork,
                      double* xTraj, double* uTraj, double* L, double* J, int32_t* converged, int32_t* iterations, int64_t batch,
                      int T, void* stream);
/* The same solve with a per-iteration record (diagnostics; the parity tests compare it with the oracle loop iteration by
 * iteration, zopt/ilqrUtils.py:305-322): J_trace (max_iter, batch) [device, may be NULL]: row i <- every trajectory's cost after
 * iteration i's acceptance step (`J_new` of :316-320; the cost it retired with once it has converged); alpha_trace (max_iter, batch)
 * int32 [device, may be NULL]: row i <- index into 0.5**arange(16) of the step size forwardPass2's argmin picked (:145-149),
 * meaningful for the trajectories that were still active in iteration i.  Rows >= *iterations are not written. */
int zm_ilqr_solve_trace_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* uGuess, int ddp,
                            int max_iter, double tol, int sync_every, double* workspace, int64_t workspace_doubles, int32_t* iwork,
                            double* xTraj, double* uTraj, double* L, double* J, int32_t* converged, int32_t* iterations,
                            int64_t batch, int T, void* stream, double* J_trace, int32_t* alpha_trace);

/* Same, with OSQP's over-relaxation `alpha` in (0, 2) (OSQP / cvxpy default 1.6, which is what the reference's
 * `prob.solve(**kwargs)` runs with, mpcUtils.py:77): the relaxed iterate alpha w + (1 - alpha) y_prev enters the projection and
 * the dual update; residuals are those of the unrelaxed iterate.  alpha = 1 is zm_mpc_solve_adaptive_f64. */
int zm_mpc_solve_relaxed_f64(const double* A, const double* B, const double* K, const double* Minv, int n_levels, int level0,
                             double rho_step, double alpha, const double* x_lb, const double* x_ub, const double* u_lb,
                             const double* u_ub, const double* x0, double rho, double eps_abs, double eps_rel, double eps_prim_inf,
                             int max_iter, int warm_start, double* workspace, double* xTraj, double* uTraj, int32_t* status,
                             int32_t* iters, double* resid, int64_t batch, int N, int n, int m, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ZOPT_AMD_H */
