// K7  linearize_dynamics / quadratize_cost -- model expansions along a trajectory, fp64, gfx950.
//
// Replaces the JAX autodiff constructors of zopt/pytrees.py: AffineDynamics.from_trajectory (:139-153),
// QuadraticCostFunction.from_trajectory (:100-115) and QuadraticValueFunction.fromTerminalCostFunction (:72-81)
// for registered device models (models.h).
//
// linearize_dynamics: embarrassingly parallel over (trajectory, step): 16 lanes per point, lane j evaluates the model
// once on forward-mode dual numbers seeded in direction e_j of z = [x ; u] and owns COLUMN j of [f_x | f_u]
// (exactly what jax.jacobian builds column by column).  4 points per wave64.
// quadratize_cost: c_x = (Q+Q^T)x, c_u = (R+R^T)u, c = x^TQx + u^TRu per point; the Hessians Q+Q^T, R+R^T, Qf+Qf^T
// are trajectory-independent and written once.
#include "models.h"
#include "quad_derivs_gen.h"
#include "zm_common.h"

namespace zm {

// what the generated closed forms of the quadcopter's derivatives read (quad_derivs_gen.h): state, thrust, wind, one sincos per angle
__device__ __forceinline__ QuadAtoms quad_atoms(const zm_model_t& md, const double* xk, const double* uk) {
    QuadAtoms a;
#pragma unroll
    for (int i = 0; i < 12; ++i) a.x[i] = xk[i];
    a.u0 = uk[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.w[i] = md.wind_ned[i];
    zm_sincos(a.x[6], &a.s6, &a.c6);
    zm_sincos(a.x[7], &a.s7, &a.c7);
    zm_sincos(a.x[8], &a.s8, &a.c8);
    a.ic7 = 1.0 / a.c7;
    return a;
}

// Optional fused cost expansion (the solvers' per-iteration call, zm::expand_list): the lane of variable j also writes
// c_x[j] = ((Q + Q^T) x)[j] resp. c_u[j - n] = ((R + R^T) u)[j - n] of its point, and the lanes of a trajectory's last point the
// terminal gradient v_x = (Qf + Qf^T) x_T -- the same expressions, in the same order, as quadratize_cost_kernel.
struct ExpandCost {
    zm_quadcost_t cs;
    double* c_x;
    double* c_u;
    double* v_x;
};

// Launch shape of the expansion kernels: 16 lanes per trajectory point, LIN_WAVES waves per workgroup.  Measured on the stand-alone
// copy of this kernel (tools/lin_lab.hip, profiles/r02_lin_lab*.txt): with the closed forms in their first, fully expanded shape the
// launch was bound by their ~700 fp64 operations per wave (as the dual-number evaluation before them), not by memory -- without
// loads AND stores it took as long; one-wave or four-wave workgroups, one point or several per group, whole-line or column-strided
// stores moved it by less than 10 %.  With the factored forms it writes its 1 536 B per point at the rate HBM takes them.
constexpr int LIN_WAVES = 4;
// WIND (quadcopter): constant NED wind -- the still-air closed forms are less than half as long; carrying both behind a run-time test
// cost the kernel 40 registers (130 against 92) and with them two of its five waves per SIMD
// PACKED (quadcopter, dt != 0): `f_x` receives the packed image of [f_x | f_u] -- the structurally nonzero entries of d xd / d z only
// (quad_derivs_gen.h: 56 per point in still air, 59 (+1 pad) with wind; everything else is the identity's 0 or 1) -- 448 B per point
// instead of 1 536 B; `f_u` is not written.  The solvers' sweeps read this form (ilqr_backward_dma_f64<..., NJP>).
template <bool COST, bool QUAD, bool WIND = false, bool PACKED = false>
__global__ __launch_bounds__(64 * LIN_WAVES, 3) void linearize_dynamics_kernel(const zm_model_t md, const double* __restrict__ xTraj,
                                                                const double* __restrict__ uTraj,
                                                                const int* __restrict__ active, double* __restrict__ f,
                                                                double* __restrict__ f_x, double* __restrict__ f_u,
                                                                const long batch, const int T,
                                                                const int* __restrict__ list, const long count, const ExpandCost ec) {
    // one point's [f_x (n, n) | f_u (n, m)] image per 16-lane group: the columns the lanes computed are transposed here so that the
    // group stores whole 128-byte lines instead of column-strided 8-byte pieces
    __shared__ double tile[4 * LIN_WAVES][MAXN * (MAXN + MAXM)];
    const int lane = threadIdx.x;
    const int j = lane & 15;                                   // seed direction / Jacobian column
    const int q = lane >> 4;                                   // group of this workgroup
    const long gi = (long)blockIdx.x * (4 * LIN_WAVES) + q;    // slot * T + k; slots are the listed trajectories, or all of them
    const long nslot = list ? count : batch;
    if (gi >= nslot * T) return;
    const long slot = gi / T;
    const int k = (int)(gi - slot * T);
    const long traj = list ? (long)list[slot] : slot;
    const int n = QUAD ? 12 : md.n, m = QUAD ? 4 : md.m;   // (compile-time in the quadcopter variant)
    const long pt = traj * T + k;                              // point index
    double xv[MAXN], uv[MAXM];
    const double* xk = xTraj + (traj * (T + 1) + k) * n;
    {   // the point's loads go out together with the mask's
        const double* uk = uTraj + pt * m;
#pragma unroll
        for (int i = 0; i < MAXN; ++i) xv[i] = (i < n) ? xk[i] : 0.0;
#pragma unroll
        for (int i = 0; i < MAXM; ++i) uv[i] = (i < m) ? uk[i] : 0.0;
    }
    if (active && active[traj] == 0) return;
    {
        // (the cost part first: its weight loads leave with the state's, ahead of the LDS exchange below -- a compiler barrier)
        if constexpr (COST) {
            const zm_quadcost_t& cs = ec.cs;
            if (j < n) {
                double g;
                if (cs.diagonal == 1) {
                    double xj = 0.0;
#pragma unroll
                    for (int i = 0; i < MAXN; ++i) xj = (i == j) ? xv[i] : xj;
                    g = (cs.Q[j * n + j] + cs.Q[j * n + j]) * xj;
                } else {
                    g = 0.0;
#pragma unroll
                    for (int i = 0; i < MAXN; ++i)
                        if (i < n) g = __builtin_fma(cs.Q[j * n + i] + cs.Q[i * n + j], xv[i], g);
                }
                ec.c_x[pt * n + j] = g;
                if (k == T - 1) {   // terminal gradient from x_T
                    const double* xT = xk + n;
                    double gv;
                    if (cs.diagonal == 1) {
                        gv = (cs.Qf[j * n + j] + cs.Qf[j * n + j]) * xT[j];
                    } else {
                        gv = 0.0;
                        for (int i = 0; i < n; ++i) gv = __builtin_fma(cs.Qf[j * n + i] + cs.Qf[i * n + j], xT[i], gv);
                    }
                    ec.v_x[traj * n + j] = gv;
                }
            } else if (j < n + m) {
                const int ju = j - n;
                double g;
                if (cs.diagonal == 1) {
                    double uj = 0.0;
#pragma unroll
                    for (int i = 0; i < MAXM; ++i) uj = (i == ju) ? uv[i] : uj;
                    g = (cs.R[ju * m + ju] + cs.R[ju * m + ju]) * uj;
                } else {
                    g = 0.0;
#pragma unroll
                    for (int i = 0; i < MAXM; ++i)
                        if (i < m) g = __builtin_fma(cs.R[ju * m + i] + cs.R[i * m + ju], uv[i], g);
                }
                ec.c_u[pt * m + ju] = g;
            }
        }
        if constexpr (PACKED) {
            static_assert(QUAD, "packed Jacobians: the quadcopter's closed forms");
            constexpr int NJ = WIND ? QUAD_NJ_WIND : QUAD_NJ_STILL, NJP = (NJ + 1) & ~1;
            const QuadAtoms a = quad_atoms(md, xv, uv);
            if (NJP != NJ && j == 0) tile[q][NJP - 1] = 0.0;
            quad_jac_column_packed<WIND>(j, a, md.dt, tile[q]);
            wave_lds_sync();
            double* op = f_x + pt * NJP;
            for (int e = j; e < NJP; e += 16) op[e] = tile[q][e];
            return;
        }
        double col[MAXN];                                      // column j of [f_x | f_u] (lanes j >= n + m: unused)
        if constexpr (QUAD) {
            // closed-form column j of d xd / d z (generated, quad_derivs_gen.h): the trigonometric values once, a few operations
            // per entry -- instead of the whole model on dual numbers per column.  x+ = x + dt xd  =>  column of [f_x | f_u] =
            // e_j + dt d xd  (dt = 0: the derivative of xd itself, as model_step defines that case).
            const QuadAtoms a = quad_atoms(md, xv, uv);
            double o[12];
            quad_jac_column<WIND>(j, a, o);
#pragma unroll
            for (int i = 0; i < 12; ++i) col[i] = (md.dt == 0.0) ? o[i] : __builtin_fma(md.dt, o[i], (i == j) ? 1.0 : 0.0);
            if (f && j == 0) {   // model_step's quadcopter branch, spelled out (the generic call drags the linear model's A, B loads
                                 // into this loop as invariants)
                double xd[12];
                quad_inertial_dynamics<double>(xv, uv, md.wind_ned, xd);
#pragma unroll
                for (int i = 0; i < 12; ++i) f[pt * 12 + i] = (md.dt == 0.0) ? xd[i] : xv[i] + md.dt * xd[i];
            }
        } else {
            Dual x[MAXN], u[MAXM], xn[MAXN];
#pragma unroll
            for (int i = 0; i < MAXN; ++i) x[i] = Dual{xv[i], (i == j) ? 1.0 : 0.0};
#pragma unroll
            for (int i = 0; i < MAXM; ++i) u[i] = Dual{uv[i], (n + i == j) ? 1.0 : 0.0};
            model_step<Dual>(md, x, u, xn);
#pragma unroll
            for (int i = 0; i < MAXN; ++i) col[i] = xn[i].d;
            if (f && j == 0) {
#pragma unroll
                for (int i = 0; i < MAXN; ++i)
                    if (i < n) f[pt * n + i] = xn[i].v;
            }
        }
        if (j < n + m) {
            double* t = tile[q] + ((j < n) ? j : n * n + (j - n));
            const int st = (j < n) ? n : m;
#pragma unroll
            for (int i = 0; i < MAXN; ++i)
                if (i < n) t[i * st] = col[i];
        }
        wave_lds_sync();
        {
            double* ox = f_x + pt * n * n;
            double* ou = f_u + pt * n * m;
            for (int e = j; e < n * n; e += 16) ox[e] = tile[q][e];
            for (int e = j; e < n * m; e += 16) ou[e] = tile[q][n * n + e];
        }
    }
}

// quadratic_dynamics: the unordered derivative pairs (a, b) that can be nonzero are spread over the lanes, each evaluated once on
// hyper-dual numbers seeded e_a, e_b; every other entry is zero.  PPW points per wave: 2 when the model declares at most 32 pairs
// (model_hessian_pairs), else 1 with the pairs of the variables the model is not affine in (model_nonlinear_mask).
template <int PPW>
__global__ __launch_bounds__(64) void quadratic_dynamics_kernel(const zm_model_t md, const double* __restrict__ xTraj,
                                                                const double* __restrict__ uTraj,
                                                                const int* __restrict__ active, double* __restrict__ f_xx,
                                                                double* __restrict__ f_ux, double* __restrict__ f_uu,
                                                                const long batch, const int T, const int* __restrict__ list,
                                                                const long count) {
    constexpr int LPP = 64 / PPW;                       // lanes per point
    const int sub = threadIdx.x / LPP, lp = threadIdx.x % LPP;
    const long nslot = list ? count : batch;
    const long sp = (long)blockIdx.x * PPW + sub;       // slot * T + k
    if (sp >= nslot * T) return;
    const long slot = sp / T;
    const int k = (int)(sp - slot * T);
    const long traj = list ? (long)list[slot] : slot;
    if (active && active[traj] == 0) return;
    const long pt = traj * T + k;
    const int n = md.n, m = md.m, K = n + m;
    const double* xk = xTraj + (traj * (T + 1) + k) * n;
    const double* uk = uTraj + pt * m;
    double* oxx = f_xx + pt * n * n * n;
    double* oux = f_ux + pt * n * m * n;
    double* ouu = f_uu + pt * n * m * m;
    // everything is zero-filled with coalesced stores (full lines: leaving the structural zeros unwritten in a buffer zeroed once
    // is slower, the entries then reach HBM as partial-line writes), then the pairs are evaluated
    {
        const int nxx = n * n * n, nux = n * m * n, nuu = n * m * m;
        for (int e = lp; e < nxx; e += LPP) oxx[e] = 0.0;
        if (f_ux) {   // NULL (with f_uu): the model is affine in its controls, the caller does not materialise these zeros
            for (int e = lp; e < nux; e += LPP) oux[e] = 0.0;
            for (int e = lp; e < nuu; e += LPP) ouu[e] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the zeros land before the entries written below
        __builtin_amdgcn_s_waitcnt(0);
    }
    int ta = 0, tb = 0;
    const int declared = model_hessian_pairs(md, lp, ta, tb);
    const unsigned mask = model_nonlinear_mask(md);
    int nl[MAXN + MAXM], V = 0;
    for (int i = 0; i < K; ++i)
        if (mask >> i & 1u) nl[V++] = i;
    const int npairs = declared ? declared : V * (V + 1) / 2;
    for (int p = lp; p < npairs; p += LPP) {
        int a = ta, b = tb;
        if (!declared) {
            int ia = 0, rem = p;
            while (rem >= V - ia) {  // row ia of the upper triangle holds V - ia pairs (ia, ia..V-1)
                rem -= V - ia;
                ++ia;
            }
            a = 0;               // nl[] by a select chain: a dynamic index would put the array into scratch
            b = 0;
#pragma unroll
            for (int q = 0; q < MAXN + MAXM; ++q) {
                a = (q == ia) ? nl[q] : a;
                b = (q == ia + rem) ? nl[q] : b;
            }
        }
        Hyper x[MAXN], u[MAXM], xn[MAXN];
#pragma unroll
        for (int i = 0; i < MAXN; ++i) x[i] = Hyper{(i < n) ? xk[i] : 0.0, (i == a) ? 1.0 : 0.0, (i == b) ? 1.0 : 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < MAXM; ++i)
            u[i] = Hyper{(i < m) ? uk[i] : 0.0, (n + i == a) ? 1.0 : 0.0, (n + i == b) ? 1.0 : 0.0, 0.0};
        model_step<Hyper>(md, x, u, xn);
#pragma unroll
        for (int i = 0; i < MAXN; ++i) {
            if (i < n) {
                const double h = xn[i].d12;  // d2 f_i / dz_a dz_b,  a <= b
                if (b < n) {                 // both states: f_xx[i][a][b] = f_xx[i][b][a]
                    oxx[(i * n + a) * n + b] = h;
                    oxx[(i * n + b) * n + a] = h;
                } else if (a < n) {          // a state, b control: f_ux[i][b-n][a]
                    oux[(i * m + (b - n)) * n + a] = h;
                } else {                     // both controls
                    ouu[(i * m + (a - n)) * m + (b - n)] = h;
                    ouu[(i * m + (b - n)) * m + (a - n)] = h;
                }
            }
        }
    }
}

// The same second derivatives PACKED: H[pt][p][i] = d2 f_i / dz_a dz_b for the model's declared pairs p = (a <= b) only -- every
// other entry of f_xx / f_ux / f_uu is structurally zero.  The quadcopter: 28 x 12 doubles = 2 688 B per trajectory point instead of
// 19 968 B (13 824 B without the control blocks), all of it written by coalesced stores, nothing zero-filled.  Consumed by the
// DDP sweep (zm_ddp_backward_pairs_list_f64), whose contraction sum_i v_x[i] H[p][i] runs in the same order as over the full
// tensors: bitwise the same result.
template <int PPW>
__global__ __launch_bounds__(64 * LIN_WAVES) void quadratic_dynamics_pairs_kernel(const zm_model_t md, const double* __restrict__ xTraj,
                                                                      const double* __restrict__ uTraj,
                                                                      const int* __restrict__ active, double* __restrict__ H,
                                                                      const long batch, const int T,
                                                                      const int* __restrict__ list, const long count) {
    constexpr int LPP = 64 / PPW;                       // lanes per point
    const int sub = threadIdx.x / LPP, lp = threadIdx.x % LPP;
    const long nslot = list ? count : batch;
    const long sp = (long)blockIdx.x * (PPW * LIN_WAVES) + sub;       // slot * T + k  (LIN_WAVES waves per workgroup, as linearize)
    if (sp >= nslot * T) return;
    const long slot = sp / T;
    const int k = (int)(sp - slot * T);
    const long traj = list ? (long)list[slot] : slot;
    if (active && active[traj] == 0) return;
    const long pt = traj * T + k;
    const int n = md.n, m = md.m;
    const double* xk = xTraj + (traj * (T + 1) + k) * n;
    const double* uk = uTraj + pt * m;
    int a = 0, b = 0;
    const int npairs = model_hessian_pairs(md, lp, a, b);
    // the pairs' rows are gathered in LDS and leave as whole lines (a row per lane -- 96 B apart -- reached HBM as partial-line writes)
    __shared__ double tile[PPW * LIN_WAVES][ZM_MAX_PAIRS * MAXN];
    double h[MAXN];
    if (md.kind == ZM_MODEL_QUADCOPTER) {
        // closed-form second derivatives of pair lp (generated, quad_derivs_gen.h) instead of the model on hyper-dual numbers;
        // x+ = x + dt xd  =>  d2 f = dt d2 xd  (dt = 0: d2 xd itself)
        const QuadAtoms at = quad_atoms(md, xk, uk);
        double o[12];
        if (md.wind_ned[0] == 0.0 && md.wind_ned[1] == 0.0 && md.wind_ned[2] == 0.0) quad_hess_pair<false>(lp, at, o);   // still air
        else quad_hess_pair<true>(lp, at, o);
#pragma unroll
        for (int i = 0; i < 12; ++i) h[i] = (md.dt == 0.0) ? o[i] : md.dt * o[i];
    } else {
        Hyper x[MAXN], u[MAXM], xn[MAXN];
#pragma unroll
        for (int i = 0; i < MAXN; ++i) x[i] = Hyper{(i < n) ? xk[i] : 0.0, (i == a) ? 1.0 : 0.0, (i == b) ? 1.0 : 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < MAXM; ++i) u[i] = Hyper{(i < m) ? uk[i] : 0.0, (n + i == a) ? 1.0 : 0.0, (n + i == b) ? 1.0 : 0.0, 0.0};
        model_step<Hyper>(md, x, u, xn);
#pragma unroll
        for (int i = 0; i < MAXN; ++i) h[i] = xn[i].d12;
    }
    if (lp < npairs) {
#pragma unroll
        for (int i = 0; i < MAXN; ++i)
            if (i < n) tile[sub][lp * n + i] = h[i];
    }
    wave_lds_sync();
    double* o = H + pt * npairs * n;
    for (int e = lp; e < npairs * n; e += LPP) o[e] = tile[sub][e];
}

// The quadcopter's packed second derivatives with 16 lanes per point (4 points per wave instead of 2): lane j < 14 evaluates the
// closed forms of pairs 2j and 2j+1 (quad_hess_pair2), the 28 x 12 image of the point is gathered in LDS and leaves as whole lines.
// Halves the instructions per point of quadratic_dynamics_pairs_kernel (whose 32-lane groups leave every second SIMD lane group idle
// through the shared part).
// SPARSE: only the structurally nonzero entries leave (QUAD_NH_*: 69 in still air, 85 with wind, padded to an even count; positions
// QUAD_HDENSE_*) -- 560 B per point instead of 2 688 B; the DDP ring sweep scatters them into its dense LDS image.
template <bool WIND, bool SPARSE = false>
__global__ __launch_bounds__(64 * LIN_WAVES) void quad_hessian_pairs16_kernel(const zm_model_t md, const double* __restrict__ xTraj,
                                                                              const double* __restrict__ uTraj,
                                                                              const int* __restrict__ active, double* __restrict__ H,
                                                                              const long batch, const int T,
                                                                              const int* __restrict__ list, const long count) {
    constexpr int NPR = 28, ROW = NPR * 12;
    __shared__ double tile[4 * LIN_WAVES][ROW];
    const int j = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long nslot = list ? count : batch;
    const long sp = (long)blockIdx.x * (4 * LIN_WAVES) + q;   // slot * T + k
    if (sp >= nslot * T) return;
    const long slot = sp / T;
    const int k = (int)(sp - slot * T);
    const long traj = list ? (long)list[slot] : slot;
    const long pt = traj * T + k;
    const double* xk = xTraj + (traj * (T + 1) + k) * 12;
    const double* uk = uTraj + pt * 4;
    const QuadAtoms at = quad_atoms(md, xk, uk);
    if (active && active[traj] == 0) return;
    if constexpr (SPARSE) {
        constexpr int NH = WIND ? QUAD_NH_WIND : QUAD_NH_STILL, NHP = (NH + 1) & ~1;
        if (NHP != NH && j == 15) tile[q][NHP - 1] = 0.0;
        quad_hess_pair2_packed<WIND>(j, at, md.dt, tile[q]);
        wave_lds_sync();
        double* out = H + pt * NHP;
        for (int e = j; e < NHP; e += 16) out[e] = tile[q][e];
        return;
    }
    double o[24];
    quad_hess_pair2<WIND>(j, at, o);
    if (j < NPR / 2) {
#pragma unroll
        for (int i = 0; i < 24; ++i) tile[q][j * 24 + i] = (md.dt == 0.0) ? o[i] : md.dt * o[i];   // x+ = x + dt xd  =>  d2 f = dt d2 xd
    }
    wave_lds_sync();
    double* out = H + pt * ROW;
#pragma unroll
    for (int e = 0; e < ROW / 16; ++e) out[j + 16 * e] = tile[q][j + 16 * e];
}

// one thread per (trajectory, step) point plus one per trajectory for the terminal expansion
__global__ __launch_bounds__(256) void quadratize_cost_kernel(const zm_quadcost_t cs, const int n, const int m,
                                                              const double* __restrict__ xTraj,
                                                              const double* __restrict__ uTraj,
                                                              const int* __restrict__ active, double* __restrict__ c,
                                                              double* __restrict__ c_x, double* __restrict__ c_u,
                                                              double* __restrict__ v, double* __restrict__ v_x,
                                                              const long batch, const int T, const int* __restrict__ list,
                                                              const long count) {
    const long sid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nslot = list ? count : batch;
    const long npts = nslot * T;
    if (sid >= npts + nslot) return;
    const bool terminal = sid >= npts;
    const long slot = terminal ? (sid - npts) : (sid / T);
    const long traj = list ? (long)list[slot] : slot;
    if (active && active[traj] == 0) return;
    const int k = terminal ? T : (int)(sid - slot * T);
    const long id = traj * T + k;                              // point index (unused for the terminal expansion)
    const double* xk = xTraj + (traj * (T + 1) + k) * n;
    double x[MAXN], u[MAXM];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) x[i] = (i < n) ? xk[i] : 0.0;
    const double* Qm = terminal ? cs.Qf : cs.Q;
    double* gx = terminal ? (v_x ? v_x + traj * n : nullptr) : (c_x ? c_x + id * n : nullptr);
    if (gx) {
        if (cs.diagonal == 1) {   // asserted diagonal weights: the n^2 - n vanishing terms of (Q + Q^T) x are not formed
#pragma unroll
            for (int jj = 0; jj < MAXN; ++jj)
                if (jj < n) gx[jj] = (Qm[jj * n + jj] + Qm[jj * n + jj]) * x[jj];
        } else {
#pragma unroll
            for (int jj = 0; jj < MAXN; ++jj) {
                if (jj < n) {
                    double s = 0.0;   // ((Q + Q^T) x)[jj]
#pragma unroll
                    for (int i = 0; i < MAXN; ++i)
                        if (i < n) s = __builtin_fma(Qm[jj * n + i] + Qm[i * n + jj], x[i], s);
                    gx[jj] = s;
                }
            }
        }
    }
    if (terminal) {
        if (v) v[traj] = terminal_cost(cs, n, x);
        return;
    }
    const double* uk = uTraj + id * m;
#pragma unroll
    for (int i = 0; i < MAXM; ++i) u[i] = (i < m) ? uk[i] : 0.0;
    if (c_u) {
        if (cs.diagonal == 1) {
#pragma unroll
            for (int jj = 0; jj < MAXM; ++jj)
                if (jj < m) c_u[id * m + jj] = (cs.R[jj * m + jj] + cs.R[jj * m + jj]) * u[jj];
        } else {
#pragma unroll
            for (int jj = 0; jj < MAXM; ++jj) {
                if (jj < m) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < MAXM; ++i)
                        if (i < m) s = __builtin_fma(cs.R[jj * m + i] + cs.R[i * m + jj], u[i], s);
                    c_u[id * m + jj] = s;
                }
            }
        }
    }
    if (c) c[id] = running_cost(cs, n, m, x, u);
}

__global__ __launch_bounds__(256) void cost_hessians_kernel(const zm_quadcost_t cs, const int n, const int m,
                                                            double* __restrict__ c_xx, double* __restrict__ c_ux,
                                                            double* __restrict__ c_uu, double* __restrict__ v_xx) {
    const int t = threadIdx.x;
    if (t < n * n) {
        const int i = t / n, jj = t % n;
        if (c_xx) c_xx[t] = cs.Q[i * n + jj] + cs.Q[jj * n + i];
        if (v_xx) v_xx[t] = cs.Qf[i * n + jj] + cs.Qf[jj * n + i];
    }
    if (t < m * m && c_uu) {
        const int i = t / m, jj = t % m;
        c_uu[t] = cs.R[i * m + jj] + cs.R[jj * m + i];
    }
    if (t < m * n && c_ux) c_ux[t] = 0.0;
}

}  // namespace zm

static int zm_check_model(const zm_model_t* model, zm_model_t& md, const char* who) {
    if (!model) return zm::set_error(ZM_EINVAL, "%s: null model", who);
    md = *model;
    if (md.kind == ZM_MODEL_QUADCOPTER) {
        md.n = 12;
        md.m = 4;
    } else if (md.kind == ZM_MODEL_QUADCOPTER_RB) {
        md.n = 8;
        md.m = 4;
    } else if (md.kind == ZM_MODEL_LINEAR) {
        if (!md.A || !md.B) return zm::set_error(ZM_EINVAL, "%s: linear model needs A, B", who);
    } else {
        return zm::set_error(ZM_EUNSUPPORTED, "%s: unknown model kind %d", who, md.kind);
    }
    if (md.n < 1 || md.n > zm::MAXN || md.m < 1 || md.m > zm::MAXM)
        return zm::set_error(ZM_EUNSUPPORTED, "%s: (n=%d, m=%d) not covered (n<=12, m<=4)", who, md.n, md.m);
    return ZM_OK;
}

// list == NULL: every trajectory (subject to `active`); else the `count` listed ones -- the grid shrinks with the list
static int zm_check_list(const char* who, const int32_t* list, int64_t count, int64_t batch) {
    if (list && (count < 0 || count > batch)) return zm::set_error(ZM_EINVAL, "%s: bad list length", who);
    return ZM_OK;
}

extern "C" int zm_linearize_dynamics_list_f64(const zm_model_t* model, const double* xTraj, const double* uTraj,
                                              const int32_t* list, int64_t count, const int32_t* active, double* f,
                                              double* f_x, double* f_u, int64_t batch, int T, void* stream) {
    if (batch == 0 || (list && count == 0)) return ZM_OK;   /* nothing to do (pointers of empty arrays may be NULL) */
    zm_model_t md;
    int rc = zm_check_model(model, md, "zm_linearize_dynamics_f64");
    if (rc) return rc;
    if (!xTraj || !uTraj || !f_x || !f_u) return zm::set_error(ZM_EINVAL, "zm_linearize_dynamics_f64: null pointer");
    if (batch < 0 || T < 1) return zm::set_error(ZM_EINVAL, "zm_linearize_dynamics_f64: bad size");
    if ((rc = zm_check_list("zm_linearize_dynamics_list_f64", list, count, batch))) return rc;
    const long npts = (long)(list ? count : batch) * T;
    constexpr int GPB = 4 * zm::LIN_WAVES;   // 16-lane groups per workgroup
    const bool quad = md.kind == ZM_MODEL_QUADCOPTER;
    const long ngrp = (long)(list ? count : batch) * T;
    (void)npts;
    const dim3 grid((unsigned)((ngrp + GPB - 1) / GPB)), block(64 * zm::LIN_WAVES);
    const bool windy = md.wind_ned[0] != 0.0 || md.wind_ned[1] != 0.0 || md.wind_ned[2] != 0.0;
    if (quad && windy)
        hipLaunchKernelGGL((zm::linearize_dynamics_kernel<false, true, true>), grid, block, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, f, f_x, f_u, (long)batch, T, (const int*)list, (long)count, zm::ExpandCost{});
    else if (quad)
        hipLaunchKernelGGL((zm::linearize_dynamics_kernel<false, true, false>), grid, block, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, f, f_x, f_u, (long)batch, T, (const int*)list, (long)count, zm::ExpandCost{});
    else
        hipLaunchKernelGGL((zm::linearize_dynamics_kernel<false, false>), grid, block, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, f, f_x, f_u, (long)batch, T, (const int*)list, (long)count, zm::ExpandCost{});
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

namespace zm {
// The solvers' expansion for the quadcopter, ONE LANE PER TRAJECTORY POINT (packed Jacobians + cost gradients).  With 16 lanes per
// point (linearize_dynamics_kernel) a wave serves four points and pays the trigonometric values, the shared temporaries and -- its
// lanes diverging over the columns -- every column's code for them: ~425 fp64 instructions per four points, which bound that launch
// (0.30 ms per 8192 x 100 points where its stores need 0.1).  Here a lane evaluates the whole straight-line form of its point
// (quad_jac_all_packed: the same temporaries and expressions, 119 operations in still air) into its row of a wave-private LDS tile,
// and the wave then copies the tile out as the contiguous block it is in memory (a chunk = up to QP_R consecutive points of one
// trajectory: NJP doubles per point for the Jacobians, 12 and 4 for the gradients).  Four waves per workgroup, no barrier.
// copy-out of a wave's LDS tile: `tot` consecutive doubles of the output block, element e from row e / ROW, column COL0 + e % ROW of
// the tile; eight independent LDS reads in flight per lane before their stores
template <int ROW, int COL0, int LD>
__device__ __forceinline__ void tile_copy_out(double* __restrict__ op, const double* tile, const int tot, const int lane) {
    for (int e0 = lane; e0 < tot; e0 += 64 * 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + 64 * q;
            const int p = e / ROW;
            v[q] = (e < tot) ? tile[p * LD + COL0 + (e - p * ROW)] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + 64 * q;
            if (e < tot) op[e] = v[q];
        }
    }
}

constexpr int QP_WAVES = 4;
#ifndef QP_ROWS
#define QP_ROWS 32
#endif
// Points per chunk = rows of a wave's LDS tile.  64 (every lane a point) leaves one wave per SIMD (37 KB of LDS per wave) and the launch
// waits on its loads and stores half of its time; with 32 or 16 the idle lanes cost little (the arithmetic is ~25 % of a wave's life at 50
// points) and two to four times as many waves overlap: expansion at the full batch 133-142 / 128-132 / 122-127 us for 64 / 32 / 16 rows.
// 32: the wind forms carry twice the arithmetic.
constexpr int QP_R = QP_ROWS;
template <bool WIND>
__global__ __launch_bounds__(64 * QP_WAVES) void expand_quad_points_kernel(const zm_model_t md, const double* __restrict__ xTraj,
                                                                           const double* __restrict__ uTraj,
                                                                           const int* __restrict__ active, double* __restrict__ f_x,
                                                                           const int T, const int* __restrict__ list, const long nslot,
                                                                           const ExpandCost ec, const int per, const int nchunk) {
    constexpr int NJ = WIND ? QUAD_NJ_WIND : QUAD_NJ_STILL, NJP = (NJ + 1) & ~1;
    constexpr int LD = NJP + 16 + 1;   // a point's row: packed Jacobians, c_x, c_u; odd, so that the lanes' rows start in different banks
    extern __shared__ __attribute__((aligned(16))) double qp_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* tile = qp_lds + (long)wv * QP_R * LD;
    const long chunk = (long)blockIdx.x * QP_WAVES + wv;
    if (chunk >= nslot * nchunk) return;
    const long slot = chunk / nchunk;
    const int ch = (int)(chunk - slot * nchunk);
    const long traj = list ? (long)list[slot] : slot;
    if (active && active[traj] == 0) return;   // (whole wave)
    const int k0 = ch * per;
    const int np = (T - k0 < per) ? T - k0 : per;
    const zm_quadcost_t& cs = ec.cs;
    if (lane < np) {
        const int k = k0 + lane;
        const double* xk = xTraj + (traj * (T + 1) + k) * 12;
        const double* uk = uTraj + (traj * T + k) * 4;
        double xv[12], uv[4];
#pragma unroll
        for (int i = 0; i < 12; ++i) xv[i] = xk[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) uv[i] = uk[i];
        double* t = tile + lane * LD;
        // cost gradients: the expressions of quadratize_cost_kernel / linearize_dynamics_kernel, in their order
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            double g;
            if (cs.diagonal == 1) {
                g = (cs.Q[j * 12 + j] + cs.Q[j * 12 + j]) * xv[j];
            } else {
                g = 0.0;
#pragma unroll
                for (int i = 0; i < 12; ++i) g = __builtin_fma(cs.Q[j * 12 + i] + cs.Q[i * 12 + j], xv[i], g);
            }
            t[NJP + j] = g;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double g;
            if (cs.diagonal == 1) {
                g = (cs.R[j * 4 + j] + cs.R[j * 4 + j]) * uv[j];
            } else {
                g = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) g = __builtin_fma(cs.R[j * 4 + i] + cs.R[i * 4 + j], uv[i], g);
            }
            t[NJP + 12 + j] = g;
        }
        const QuadAtoms a = quad_atoms(md, xv, uv);
        if (NJP != NJ) t[NJP - 1] = 0.0;
        quad_jac_all_packed<WIND>(a, md.dt, t);
    }
    if (k0 + np == T && lane < 12) {   // terminal gradient from x_T, one lane per component
        const double* xT = xTraj + (traj * (T + 1) + T) * 12;
        const int j = lane;
        double gv;
        if (cs.diagonal == 1) {
            gv = (cs.Qf[j * 12 + j] + cs.Qf[j * 12 + j]) * xT[j];
        } else {
            gv = 0.0;
#pragma unroll
            for (int i = 0; i < 12; ++i) gv = __builtin_fma(cs.Qf[j * 12 + i] + cs.Qf[i * 12 + j], xT[i], gv);
        }
        ec.v_x[traj * 12 + j] = gv;
    }
    wave_lds_sync();
    const long p0 = traj * T + k0;
    tile_copy_out<NJP, 0, LD>(f_x + p0 * NJP, tile, np * NJP, lane);
    tile_copy_out<12, NJP, LD>(ec.c_x + p0 * 12, tile, np * 12, lane);
    tile_copy_out<4, NJP + 12, LD>(ec.c_u + p0 * 4, tile, np * 4, lane);
}

template <bool WIND>
static int launch_expand_quad_points(const zm_model_t& md, const double* xTraj, const double* uTraj, const int* active, double* f_x,
                                     const int T, const int* list, const long nslot, const ExpandCost& ec, hipStream_t st) {
    constexpr int NJ = WIND ? QUAD_NJ_WIND : QUAD_NJ_STILL, NJP = (NJ + 1) & ~1, LD = NJP + 16 + 1;
    const size_t bytes = (size_t)QP_WAVES * QP_R * LD * sizeof(double);
    static_assert((size_t)QP_WAVES * QP_R * (((QUAD_NJ_WIND + 1) & ~1) + 17) * sizeof(double) <= 160 * 1024, "LDS tile of four waves");
    const int nchunk = (T + QP_R - 1) / QP_R, per = (T + nchunk - 1) / nchunk;
    const long chunks = nslot * nchunk;
    // per launch (cheap): the attribute is per device, and several devices may be driven from one process
    ZM_HIP_CHECK(hipFuncSetAttribute((const void*)expand_quad_points_kernel<WIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((expand_quad_points_kernel<WIND>), dim3((unsigned)((chunks + QP_WAVES - 1) / QP_WAVES)), dim3(64 * QP_WAVES), bytes,
                       st, md, xTraj, uTraj, active, f_x, T, list, nslot, ec, per, nchunk);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
}  // namespace zm

// Solver-internal (ilqr_solve.hip): the per-iteration expansions of the listed trajectories in ONE launch -- [f_x | f_u] as
// zm_linearize_dynamics_list_f64, c_x / c_u / v_x as zm_quadratize_cost_list_f64 (same expressions; tests/test_ilqr_solve_gpu.py).
namespace zm {
int expand_list(const zm_model_t* model, const zm_quadcost_t* cost, const double* xTraj, const double* uTraj, const int32_t* list,
                int64_t count, const int32_t* active, double* f_x, double* f_u, double* c_x, double* c_u, double* v_x, int64_t batch,
                int T, void* stream, int packed) {
    if (batch == 0 || count == 0) return ZM_OK;
    zm_model_t md;
    int rc = zm_check_model(model, md, "expand_list");
    if (rc) return rc;
    if (!cost || !cost->Q || !cost->R || !cost->Qf || !xTraj || !uTraj || !list || !f_x || !f_u || !c_x || !c_u || !v_x)
        return set_error(ZM_EINVAL, "expand_list: null pointer");
    if (batch < 0 || T < 1 || count < 0 || count > batch) return set_error(ZM_EINVAL, "expand_list: bad size");
    const long npts = (long)count * T;
    constexpr int GPB = 4 * LIN_WAVES;
    const bool quad = md.kind == ZM_MODEL_QUADCOPTER;
    const long ngrp = (long)count * T;
    (void)npts;
    const dim3 grid((unsigned)((ngrp + GPB - 1) / GPB)), block(64 * LIN_WAVES);
    const ExpandCost ec{*cost, c_x, c_u, v_x};
    const bool windy = md.wind_ned[0] != 0.0 || md.wind_ned[1] != 0.0 || md.wind_ned[2] != 0.0;
    if (packed && (!quad || md.dt == 0.0)) return set_error(ZM_EUNSUPPORTED, "expand_list: packed Jacobians need the quadcopter with dt != 0");
    // ZOPT_AMD_EXPAND=group: the packed expansion with 16 lanes per point (A/B; same results)
    static const bool by_points = [] {
        const char* e = zm::lab_env("ZOPT_AMD_EXPAND");
        return !(e && e[0] == 'g');
    }();
    if (packed && by_points) {
        if (windy) return launch_expand_quad_points<true>(md, xTraj, uTraj, (const int*)active, f_x, T, (const int*)list, (long)count, ec, (hipStream_t)stream);
        return launch_expand_quad_points<false>(md, xTraj, uTraj, (const int*)active, f_x, T, (const int*)list, (long)count, ec, (hipStream_t)stream);
    }
    if (packed && windy)
        hipLaunchKernelGGL((linearize_dynamics_kernel<true, true, true, true>), grid, block, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, (double*)nullptr, f_x, f_u, (long)batch, T, (const int*)list, (long)count, ec);
    else if (packed)
        hipLaunchKernelGGL((linearize_dynamics_kernel<true, true, false, true>), grid, block, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, (double*)nullptr, f_x, f_u, (long)batch, T, (const int*)list, (long)count, ec);
    else if (quad && windy)
        hipLaunchKernelGGL((linearize_dynamics_kernel<true, true, true>), grid, block, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, (double*)nullptr, f_x, f_u, (long)batch, T, (const int*)list, (long)count, ec);
    else if (quad)
        hipLaunchKernelGGL((linearize_dynamics_kernel<true, true, false>), grid, block, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, (double*)nullptr, f_x, f_u, (long)batch, T, (const int*)list, (long)count, ec);
    else
        hipLaunchKernelGGL((linearize_dynamics_kernel<true, false>), grid, block, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, (double*)nullptr, f_x, f_u, (long)batch, T, (const int*)list, (long)count, ec);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
}  // namespace zm

extern "C" int zm_linearize_dynamics_f64(const zm_model_t* model, const double* xTraj, const double* uTraj,
                                         const int32_t* active, double* f, double* f_x, double* f_u, int64_t batch, int T,
                                         void* stream) {
    return zm_linearize_dynamics_list_f64(model, xTraj, uTraj, nullptr, 0, active, f, f_x, f_u, batch, T, stream);
}

extern "C" int zm_quadratize_cost_list_f64(const zm_quadcost_t* cost, int n, int m, const double* xTraj, const double* uTraj,
                                           const int32_t* list, int64_t count, const int32_t* active, double* c, double* c_x,
                                           double* c_u, double* v, double* v_x, double* c_xx, double* c_ux, double* c_uu,
                                           double* v_xx, int64_t batch, int T, void* stream) {
    /* batch == 0 is a valid call: it still writes the trajectory-independent Hessians (xTraj, uTraj may then be NULL) */
    if (!cost || !cost->Q || !cost->R || !cost->Qf || (batch != 0 && (!xTraj || !uTraj)))
        return zm::set_error(ZM_EINVAL, "zm_quadratize_cost_f64: null pointer");
    if (n < 1 || n > zm::MAXN || m < 1 || m > zm::MAXM)
        return zm::set_error(ZM_EUNSUPPORTED, "zm_quadratize_cost_f64: (n=%d, m=%d) not covered", n, m);
    if (batch < 0 || T < 1) return zm::set_error(ZM_EINVAL, "zm_quadratize_cost_f64: bad size");
    if (const int rcl = zm_check_list("zm_quadratize_cost_list_f64", list, count, batch)) return rcl;
    hipStream_t st = (hipStream_t)stream;
    if (c_xx || c_ux || c_uu || v_xx)
        hipLaunchKernelGGL(zm::cost_hessians_kernel, dim3(1), dim3(256), 0, st, *cost, n, m, c_xx, c_ux, c_uu, v_xx);
    const long nslot = list ? (long)count : (long)batch;
    if (nslot > 0 && (c || c_x || c_u || v || v_x)) {
        const long work = nslot * T + nslot;
        hipLaunchKernelGGL(zm::quadratize_cost_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, *cost, n, m,
                           xTraj, uTraj, (const int*)active, c, c_x, c_u, v, v_x, (long)batch, T, (const int*)list, (long)count);
    }
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

extern "C" int zm_quadratize_cost_f64(const zm_quadcost_t* cost, int n, int m, const double* xTraj, const double* uTraj,
                                      const int32_t* active, double* c, double* c_x, double* c_u, double* v, double* v_x,
                                      double* c_xx, double* c_ux, double* c_uu, double* v_xx, int64_t batch, int T,
                                      void* stream) {
    return zm_quadratize_cost_list_f64(cost, n, m, xTraj, uTraj, nullptr, 0, active, c, c_x, c_u, v, v_x, c_xx, c_ux, c_uu, v_xx,
                                       batch, T, stream);
}

extern "C" int zm_quadratic_dynamics_list_f64(const zm_model_t* model, const double* xTraj, const double* uTraj,
                                              const int32_t* list, int64_t count, const int32_t* active, double* f_xx,
                                              double* f_ux, double* f_uu, int64_t batch, int T, void* stream) {
    if (batch == 0 || (list && count == 0)) return ZM_OK;   /* nothing to do (pointers of empty arrays may be NULL) */
    zm_model_t md;
    int rc = zm_check_model(model, md, "zm_quadratic_dynamics_f64");
    if (rc) return rc;
    if (!xTraj || !uTraj || !f_xx || (!f_ux != !f_uu)) return zm::set_error(ZM_EINVAL, "zm_quadratic_dynamics_f64: null pointer");
    if (!f_ux && (zm::model_nonlinear_mask(md) >> md.n) != 0u)
        return zm::set_error(ZM_EINVAL, "zm_quadratic_dynamics_f64: f_ux / f_uu may only be omitted for a model that is affine in its "
                                        "controls (zm_model_nonlinear_mask)");
    if (batch < 0 || T < 1) return zm::set_error(ZM_EINVAL, "zm_quadratic_dynamics_f64: bad size");
    if ((rc = zm_check_list("zm_quadratic_dynamics_list_f64", list, count, batch))) return rc;
    const long npts = (list ? (long)count : (long)batch) * T;
    if (md.kind == ZM_MODEL_QUADCOPTER)   // declares 28 pairs (model_hessian_pairs): two points per wave
        hipLaunchKernelGGL(zm::quadratic_dynamics_kernel<2>, dim3((unsigned)((npts + 1) / 2)), dim3(64), 0, (hipStream_t)stream, md,
                           xTraj, uTraj, (const int*)active, f_xx, f_ux, f_uu, (long)batch, T, (const int*)list, (long)count);
    else
        hipLaunchKernelGGL(zm::quadratic_dynamics_kernel<1>, dim3((unsigned)npts), dim3(64), 0, (hipStream_t)stream, md, xTraj,
                           uTraj, (const int*)active, f_xx, f_ux, f_uu, (long)batch, T, (const int*)list, (long)count);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

extern "C" int zm_model_hessian_pairs(const zm_model_t* model, int32_t* pairs, int32_t* npairs) {
    zm_model_t md;
    const int rc = zm_check_model(model, md, "zm_model_hessian_pairs");
    if (rc) return rc;
    if (!npairs) return zm::set_error(ZM_EINVAL, "zm_model_hessian_pairs: null pointer");
    const zm::PairTab t = zm::model_pair_table(md.kind);
    *npairs = t.n;
    if (pairs)
        for (int q = 0; q < t.n; ++q) {
            pairs[2 * q] = t.ab[q] >> 4;
            pairs[2 * q + 1] = t.ab[q] & 15;
        }
    return ZM_OK;
}

extern "C" int zm_quadratic_dynamics_pairs_list_f64(const zm_model_t* model, const double* xTraj, const double* uTraj,
                                                    const int32_t* list, int64_t count, const int32_t* active, double* H,
                                                    int64_t batch, int T, void* stream) {
    if (batch == 0 || (list && count == 0)) return ZM_OK;
    zm_model_t md;
    int rc = zm_check_model(model, md, "zm_quadratic_dynamics_pairs_list_f64");
    if (rc) return rc;
    if (!xTraj || !uTraj || !H) return zm::set_error(ZM_EINVAL, "zm_quadratic_dynamics_pairs_list_f64: null pointer");
    if (batch < 0 || T < 1) return zm::set_error(ZM_EINVAL, "zm_quadratic_dynamics_pairs_list_f64: bad size");
    if ((rc = zm_check_list("zm_quadratic_dynamics_pairs_list_f64", list, count, batch))) return rc;
    const int np = zm::model_pair_table(md.kind).n;
    if (np < 1 || np > 32)
        return zm::set_error(ZM_EUNSUPPORTED, "zm_quadratic_dynamics_pairs_list_f64: the model declares no Hessian pairs "
                                              "(zm_model_hessian_pairs); use zm_quadratic_dynamics_list_f64");
    const long npts = (list ? (long)count : (long)batch) * T;
    if (md.kind == ZM_MODEL_QUADCOPTER) {
        constexpr int PPB16 = 4 * zm::LIN_WAVES;
        const dim3 grid16((unsigned)((npts + PPB16 - 1) / PPB16)), block16(64 * zm::LIN_WAVES);
        if (md.wind_ned[0] != 0.0 || md.wind_ned[1] != 0.0 || md.wind_ned[2] != 0.0)
            hipLaunchKernelGGL(zm::quad_hessian_pairs16_kernel<true>, grid16, block16, 0, (hipStream_t)stream, md, xTraj, uTraj,
                               (const int*)active, H, (long)batch, T, (const int*)list, (long)count);
        else
            hipLaunchKernelGGL(zm::quad_hessian_pairs16_kernel<false>, grid16, block16, 0, (hipStream_t)stream, md, xTraj, uTraj,
                               (const int*)active, H, (long)batch, T, (const int*)list, (long)count);
        ZM_HIP_CHECK(hipGetLastError());
        return ZM_OK;
    }
    constexpr int PPB = 2 * zm::LIN_WAVES;   // points per workgroup
    hipLaunchKernelGGL(zm::quadratic_dynamics_pairs_kernel<2>, dim3((unsigned)((npts + PPB - 1) / PPB)), dim3(64 * zm::LIN_WAVES), 0, (hipStream_t)stream, md,
                       xTraj, uTraj, (const int*)active, H, (long)batch, T, (const int*)list, (long)count);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// Solver-internal (ilqr_solve.hip): the quadcopter's second derivatives in SPARSE form for the listed trajectories
namespace zm {
// The sparse second derivatives with ONE LANE PER TRAJECTORY POINT, as expand_quad_points_kernel does the Jacobians: the lane runs the
// straight-line form quad_hess_all_packed (143 operations in still air, 377 with wind -- quad_hessian_pairs16_kernel pays the
// trigonometric values, the shared part and all 14 cases for every four points) into its LDS row; the wave copies the chunk's
// contiguous block out.  W waves per workgroup (the wind form's rows leave room for three).
template <bool WIND, int W>
__global__ __launch_bounds__(64 * W) void quad_hessian_points_kernel(const zm_model_t md, const double* __restrict__ xTraj,
                                                                     const double* __restrict__ uTraj, const int* __restrict__ active,
                                                                     double* __restrict__ Hs, const int T, const int* __restrict__ list,
                                                                     const long nslot, const int per, const int nchunk) {
    constexpr int NH = WIND ? QUAD_NH_WIND : QUAD_NH_STILL, NHP = (NH + 1) & ~1, LD = NHP + 1;
    extern __shared__ __attribute__((aligned(16))) double qp_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* tile = qp_lds + (long)wv * QP_R * LD;
    const long chunk = (long)blockIdx.x * W + wv;
    if (chunk >= nslot * nchunk) return;
    const long slot = chunk / nchunk;
    const int ch = (int)(chunk - slot * nchunk);
    const long traj = list ? (long)list[slot] : slot;
    if (active && active[traj] == 0) return;   // (whole wave)
    const int k0 = ch * per;
    const int np = (T - k0 < per) ? T - k0 : per;
    if (lane < np) {
        const int k = k0 + lane;
        const QuadAtoms at = quad_atoms(md, xTraj + (traj * (T + 1) + k) * 12, uTraj + (traj * T + k) * 4);
        double* t = tile + lane * LD;
        if (NHP != NH) t[NHP - 1] = 0.0;
        quad_hess_all_packed<WIND>(at, md.dt, t);
    }
    wave_lds_sync();
    tile_copy_out<NHP, 0, LD>(Hs + (traj * T + k0) * NHP, tile, np * NHP, lane);
}

template <bool WIND>
static int launch_quad_hessian_points(const zm_model_t& md, const double* xTraj, const double* uTraj, const int* active, double* Hs,
                                      const int T, const int* list, const long nslot, hipStream_t st) {
    constexpr int NH = WIND ? QUAD_NH_WIND : QUAD_NH_STILL, NHP = (NH + 1) & ~1, LD = NHP + 1;
    constexpr int W = (4 * QP_R * LD * 8 <= 160 * 1024) ? 4 : 3;
    static_assert(W * QP_R * LD * 8 <= 160 * 1024, "LDS tile of the workgroup");
    const size_t bytes = (size_t)W * QP_R * LD * sizeof(double);
    const int nchunk = (T + QP_R - 1) / QP_R, per = (T + nchunk - 1) / nchunk;
    const long chunks = nslot * nchunk;
    ZM_HIP_CHECK(hipFuncSetAttribute((const void*)quad_hessian_points_kernel<WIND, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((quad_hessian_points_kernel<WIND, W>), dim3((unsigned)((chunks + W - 1) / W)), dim3(64 * W), bytes, st, md, xTraj,
                       uTraj, active, Hs, T, list, nslot, per, nchunk);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
int quad_hessian_sparse_list(const zm_model_t* model, const double* xTraj, const double* uTraj, const int32_t* list, int64_t count,
                             const int32_t* active, double* Hs, int64_t batch, int T, void* stream) {
    if (batch == 0 || count == 0) return ZM_OK;
    zm_model_t md;
    int rc = zm_check_model(model, md, "quad_hessian_sparse_list");
    if (rc) return rc;
    if (md.kind != ZM_MODEL_QUADCOPTER || !xTraj || !uTraj || !list || !Hs || T < 1 || count < 0 || count > batch)
        return set_error(ZM_EINVAL, "quad_hessian_sparse_list: bad argument");
    constexpr int PPB16 = 4 * LIN_WAVES;
    const long npts = (long)count * T;
    const dim3 grid16((unsigned)((npts + PPB16 - 1) / PPB16)), block16(64 * LIN_WAVES);
    const bool windy = md.wind_ned[0] != 0.0 || md.wind_ned[1] != 0.0 || md.wind_ned[2] != 0.0;
    static const bool by_points = [] {   // ZOPT_AMD_EXPAND=group: 16 lanes per point (A/B; same results)
        const char* e = zm::lab_env("ZOPT_AMD_EXPAND");
        return !(e && e[0] == 'g');
    }();
    if (by_points) {
        if (windy) return launch_quad_hessian_points<true>(md, xTraj, uTraj, (const int*)active, Hs, T, (const int*)list, (long)count, (hipStream_t)stream);
        return launch_quad_hessian_points<false>(md, xTraj, uTraj, (const int*)active, Hs, T, (const int*)list, (long)count, (hipStream_t)stream);
    }
    if (md.wind_ned[0] != 0.0 || md.wind_ned[1] != 0.0 || md.wind_ned[2] != 0.0)
        hipLaunchKernelGGL((quad_hessian_pairs16_kernel<true, true>), grid16, block16, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, Hs, (long)batch, T, (const int*)list, (long)count);
    else
        hipLaunchKernelGGL((quad_hessian_pairs16_kernel<false, true>), grid16, block16, 0, (hipStream_t)stream, md, xTraj, uTraj,
                           (const int*)active, Hs, (long)batch, T, (const int*)list, (long)count);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
}  // namespace zm

extern "C" int zm_quadratic_dynamics_f64(const zm_model_t* model, const double* xTraj, const double* uTraj,
                                         const int32_t* active, double* f_xx, double* f_ux, double* f_uu, int64_t batch,
                                         int T, void* stream) {
    return zm_quadratic_dynamics_list_f64(model, xTraj, uTraj, nullptr, 0, active, f_xx, f_ux, f_uu, batch, T, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched trim of the quadcopter: for given body velocities uvw find (p, q, r, phi, theta) and (thrust, mx, my, mz) with
// rigidBodyDynamics(x, u) = 0.  Replaces zopt/quadcopter.py:146-177 Quadcopter.trim, which minimises the squared residual with
// SciPy BFGS from z0 = (0,0,0,0,0, g,0,0,0); the reference's test accepts any point with |residual| <= 1e-3
// (tests/test_quadcopter.py:89-99).  Here: Levenberg-Marquardt on the 8 residuals from the same z0, Jacobian by dual numbers,
// one lane per instance (8 equations, 9 unknowns: the damping selects the small-step solution).
namespace zm {

__global__ __launch_bounds__(64) void quad_trim_kernel(const double* __restrict__ uvw, const double wb0, const double wb1,
                                                       const double wb2, double* __restrict__ xTrim,
                                                       double* __restrict__ uTrim, double* __restrict__ resid,
                                                       int* __restrict__ ok, const long batch, const double tol) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const double v0 = uvw[i * 3], v1 = uvw[i * 3 + 1], v2 = uvw[i * 3 + 2];
    double z[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 9.807, 0.0, 0.0, 0.0};
    auto residual = [&](const double (&zz)[9], double (&r)[8]) {
        const double x8[8] = {v0, v1, v2, zz[0], zz[1], zz[2], zz[3], zz[4]};
        const double u4[4] = {zz[5], zz[6], zz[7], zz[8]};
        const double wb[3] = {wb0, wb1, wb2};
        quad_rigid_body<double>(x8, u4, wb, r);
    };
    double r[8];
    residual(z, r);
    double cost = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) cost += r[e] * r[e];
    double lam = 1e-3;
    for (int it = 0; it < 200 && cost > tol * tol * 1e-6; ++it) {
        // Jacobian column by column on dual numbers; normal equations A = J^T J, b = J^T r
        double J[8][9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            Dual x8[8] = {{v0, 0}, {v1, 0}, {v2, 0}, {z[0], 0}, {z[1], 0}, {z[2], 0}, {z[3], 0}, {z[4], 0}};
            Dual u4[4] = {{z[5], 0}, {z[6], 0}, {z[7], 0}, {z[8], 0}};
            if (j < 5) x8[3 + j].d = 1.0; else u4[j - 5].d = 1.0;
            const Dual wb[3] = {{wb0, 0}, {wb1, 0}, {wb2, 0}};
            Dual rd[8];
            quad_rigid_body<Dual>(x8, u4, wb, rd);
#pragma unroll
            for (int e = 0; e < 8; ++e) J[e][j] = rd[e].d;
        }
        double A[9][9], b[9];
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += J[e][p] * r[e];
            b[p] = s;
#pragma unroll
            for (int q = 0; q <= p; ++q) {
                double t = 0.0;
#pragma unroll
                for (int e = 0; e < 8; ++e) t += J[e][p] * J[e][q];
                A[p][q] = t;
                A[q][p] = t;
            }
        }
        bool accepted = false;
        for (int tr = 0; tr < 12 && !accepted; ++tr) {
            // Cholesky of A + lam (diag(A) + 1e-9 I), solve for the step
            double Lc[9][9], y[9], d[9];
            bool pd = true;
#pragma unroll
            for (int p = 0; p < 9; ++p) {
#pragma unroll
                for (int q = 0; q <= p; ++q) {
                    double s = A[p][q] + ((p == q) ? lam * (A[p][p] + 1e-9) : 0.0);
#pragma unroll
                    for (int k = 0; k < q; ++k) s -= Lc[p][k] * Lc[q][k];
                    if (p == q) {
                        pd = pd && (s > 0.0);
                        Lc[p][p] = sqrt(s > 0.0 ? s : 1.0);
                    } else {
                        Lc[p][q] = s / Lc[q][q];
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                double s = -b[p];
#pragma unroll
                for (int k = 0; k < p; ++k) s -= Lc[p][k] * y[k];
                y[p] = s / Lc[p][p];
            }
#pragma unroll
            for (int p = 8; p >= 0; --p) {
                double s = y[p];
#pragma unroll
                for (int k = p + 1; k < 9; ++k) s -= Lc[k][p] * d[k];
                d[p] = s / Lc[p][p];
            }
            double zn[9], rn[8];
#pragma unroll
            for (int p = 0; p < 9; ++p) zn[p] = z[p] + d[p];
            residual(zn, rn);
            double cn = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) cn += rn[e] * rn[e];
            if (pd && cn < cost) {
#pragma unroll
                for (int p = 0; p < 9; ++p) z[p] = zn[p];
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = rn[e];
                cost = cn;
                lam = lam > 1e-10 ? lam * 0.1 : lam;
                accepted = true;
            } else {
                lam *= 10.0;
            }
        }
        if (!accepted) break;
    }
    xTrim[i * 8 + 0] = v0;
    xTrim[i * 8 + 1] = v1;
    xTrim[i * 8 + 2] = v2;
#pragma unroll
    for (int p = 0; p < 5; ++p) xTrim[i * 8 + 3 + p] = z[p];
#pragma unroll
    for (int p = 0; p < 4; ++p) uTrim[i * 4 + p] = z[5 + p];
    const double rn = sqrt(cost);
    if (resid) resid[i] = rn;
    if (ok) ok[i] = (rn <= tol) ? 1 : 0;
}

}  // namespace zm

extern "C" int zm_quadcopter_trim_f64(const double* uvw, const double* wind_body, double* xTrim, double* uTrim, double* resid,
                                      int32_t* ok, int64_t batch, double tol, void* stream) {
    if (batch == 0) return ZM_OK;
    if (!uvw || !xTrim || !uTrim) return zm::set_error(ZM_EINVAL, "zm_quadcopter_trim_f64: null pointer");
    if (batch < 0 || !(tol > 0.0)) return zm::set_error(ZM_EINVAL, "zm_quadcopter_trim_f64: bad argument");
    const double w0 = wind_body ? wind_body[0] : 0.0, w1 = wind_body ? wind_body[1] : 0.0, w2 = wind_body ? wind_body[2] : 0.0;
    hipLaunchKernelGGL(zm::quad_trim_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, (hipStream_t)stream, uvw, w0, w1, w2,
                       xTrim, uTrim, resid, (int*)ok, (long)batch, tol);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

extern "C" int zm_model_nonlinear_mask(const zm_model_t* model, uint32_t* mask) {
    zm_model_t md;
    const int rc = zm_check_model(model, md, "zm_model_nonlinear_mask");
    if (rc) return rc;
    if (!mask) return zm::set_error(ZM_EINVAL, "zm_model_nonlinear_mask: null pointer");
    *mask = zm::model_nonlinear_mask(md);
    return ZM_OK;
}
