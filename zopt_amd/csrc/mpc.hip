// K9  mpc_box_qp -- batched box-constrained LQ-MPC, fp64, gfx950.
//
// Problem statement: zopt/mpcUtils.py:48-59 (class lqrMpc).  The reference solves it with cvxpy -> OSQP (ADMM on a
// sparse KKT system), code that is not part of the reference tree; this is an MI355X-first ADMM for the same QP:
//
//   split   w = (x_1..x_N, u_0..u_{N-1})  consistent with x_{k+1} = A x_k + B u_k, x_0 = x0      (dynamics, exact)
//           y = copy of w inside the box [lb, ub]                                                  (bounds, exact)
//   iterate w <- argmin cost(w) + rho/2 |w - (y - lam)|^2  s.t. dynamics     (LQ tracking problem)
//           y <- clip(w + lam, lb, ub);   lam <- lam + w - y
//
// The w-update's Riccati matrices depend on (A, B, Q, R, Qf, rho) only -- not on x0, y, lam -- so they are factored
// ONCE (mpc_setup_kernel: K_k, Suu_k^-1), shared by every instance and every iteration.  One ADMM iteration is then
//   backward:  p <- -rho z_N;  k = N-1..0:  Qu = -rho zu_k + B^T p;  kf_k = Suu_k^-1 Qu;  p <- hx_k + A^T p - K_k^T Qu
//   forward :  x <- x0;        k = 0..N-1:  u = -K_k x - kf_k;  x <- A x + B u;  y, lam update, residuals
// = ~500 FMAs per stage, no factorisation, no branching on data.
//
// Mapping: ONE LANE per MPC instance (instances are independent: mpcUtils.py:76-81); the shared tables are read at
// wave-uniform addresses (scalar loads), x and p live in registers, the iterates y, lam, kf in a batch-minor
// workspace (lane i <-> consecutive addresses: coalesced).  Each lane runs its own ADMM to convergence
// (OSQP-style criteria); a wave retires when all its lanes have.
//
// Termination (as OSQP): r_prim = |w - y|_inf <= eps_abs + eps_rel max(|w|,|y|),  r_dual = rho |y - y_prev|_inf <=
// eps_abs + eps_rel rho |lam|.  Infeasibility: x0 outside its bounds; or, every ZM_MPC_CHK (8) iterations, OSQP's primal
// infeasibility certificate on the dual step v = w - y:  |G^T v|_inf <= eps_pinf |v|_inf  (G = the linear map u -> w;
// computed by an adjoint sweep) and  v^T w(u=0) - support_box(v) > eps_pinf |v|_inf,  i.e. v separates the dynamics
// subspace from the box.  The certificate is sound but can need thousands of iterations; an infeasible instance that
// is not certified within max_iter reports "user_limit".
#include "mpc_common.h"

#include <cstdlib>
#include <type_traits>

namespace zm {

// ----------------------------------------------------------------------------------------------------------------
// setup: single workgroup, matrices in LDS, threads spread over matrix elements
// ----------------------------------------------------------------------------------------------------------------
constexpr int SN = 12, SM = 4;      // the shapes of the 16-lanes-per-instance kernel (mpc_wave.hip) and of the small lane kernels
constexpr int SNL = 24, SML = 8;    // larger problems: the lane-per-instance kernel only (fixed penalty), see zm_mpc_solve_relaxed_f64

__device__ __forceinline__ void mm_nn(double* C, const double* A, const double* B, int p, int q, int r) {  // C = A(p,q) B(q,r)
    for (int e = threadIdx.x; e < p * r; e += blockDim.x) {
        const int i = e / r, j = e % r;
        double s = 0.0;
        for (int k = 0; k < q; ++k) s = __builtin_fma(A[i * q + k], B[k * r + j], s);
        C[e] = s;
    }
    __syncthreads();
}
__device__ __forceinline__ void mm_tn(double* C, const double* A, const double* B, int q, int p, int r) {  // C = A(q,p)^T B(q,r)
    for (int e = threadIdx.x; e < p * r; e += blockDim.x) {
        const int i = e / r, j = e % r;
        double s = 0.0;
        for (int k = 0; k < q; ++k) s = __builtin_fma(A[k * p + i], B[k * r + j], s);
        C[e] = s;
    }
    __syncthreads();
}

template <int SN, int SM>
__global__ __launch_bounds__(256) void mpc_setup_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                        const double* __restrict__ Q, const double* __restrict__ R,
                                                        const double* __restrict__ Qf, const double rho, const int N,
                                                        const int n, const int m, double* __restrict__ Kout,
                                                        double* __restrict__ Minvout) {
    __shared__ double As[SN * SN], Bs[SN * SM], P[SN * SN], PA[SN * SN], PB[SN * SM], Sux[SM * SN], Suu[SM * SM],
        Mi[SM * SM], K[SM * SN], T1[SN * SN], T2[SN * SN];
    const int t = threadIdx.x;
    for (int e = t; e < n * n; e += blockDim.x) {
        As[e] = A[e];
        P[e] = 2.0 * Qf[e] + ((e / n == e % n) ? rho : 0.0);  // P_N = 2 Qf + rho I   (1/2-form Hessian of x'Qf x + rho/2 |x-z|^2)
    }
    for (int e = t; e < n * m; e += blockDim.x) Bs[e] = B[e];
    __syncthreads();
    for (int k = N - 1; k >= 0; --k) {
        mm_nn(PA, P, As, n, n, n);
        mm_nn(PB, P, Bs, n, n, m);
        mm_tn(Sux, Bs, PA, n, m, n);  // B^T P A
        mm_tn(Suu, Bs, PB, n, m, m);  // B^T P B
        if (t < m * m) Suu[t] += 2.0 * R[t] + ((t / m == t % m) ? rho : 0.0);
        __syncthreads();
        if (t == 0) {  // m x m inverse by Gauss-Jordan with partial pivoting (m <= 4)
            double a[SM][2 * SM];
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < m; ++j) {
                    a[i][j] = Suu[i * m + j];
                    a[i][m + j] = (i == j) ? 1.0 : 0.0;
                }
            for (int c = 0; c < m; ++c) {
                int pv = c;
                for (int i = c + 1; i < m; ++i)
                    if (__builtin_fabs(a[i][c]) > __builtin_fabs(a[pv][c])) pv = i;
                for (int j = 0; j < 2 * m; ++j) {
                    const double tmp = a[c][j];
                    a[c][j] = a[pv][j];
                    a[pv][j] = tmp;
                }
                const double inv = 1.0 / a[c][c];
                for (int j = 0; j < 2 * m; ++j) a[c][j] *= inv;
                for (int i = 0; i < m; ++i)
                    if (i != c) {
                        const double f = a[i][c];
                        for (int j = 0; j < 2 * m; ++j) a[i][j] = __builtin_fma(-f, a[c][j], a[i][j]);
                    }
            }
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < m; ++j) Mi[i * m + j] = a[i][m + j];
        }
        __syncthreads();
        mm_nn(K, Mi, Sux, m, m, n);     // K_k = Suu^-1 B^T P A
        mm_tn(T1, As, PA, n, n, n);     // A^T P A
        mm_tn(T2, Sux, K, m, n, n);     // Sux^T K
        for (int e = t; e < n * n; e += blockDim.x)
            P[e] = (2.0 * Q[e] + ((e / n == e % n) ? rho : 0.0)) + T1[e] - T2[e];
        for (int e = t; e < m * n; e += blockDim.x) Kout[(long)k * m * n + e] = K[e];
        for (int e = t; e < m * m; e += blockDim.x) Minvout[(long)k * m * m + e] = Mi[e];
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------------------------
// solve: one lane per instance
// ----------------------------------------------------------------------------------------------------------------

// The shared tables are separate `const __restrict__` kernel arguments so that hipcc can prove them read-only and
// fetch them with scalar loads (wave-uniform addresses) instead of per-lane vector loads held in hundreds of VGPRs.
template <int NS, int MC>
__global__ __launch_bounds__(64) void mpc_solve_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                       const double* __restrict__ Ktab, const double* __restrict__ Mtab,
                                                       const double* __restrict__ x_lb, const double* __restrict__ x_ub,
                                                       const double* __restrict__ u_lb, const double* __restrict__ u_ub,
                                                       const MpcArgs g) {
    constexpr int W = NS + MC;
    const long inst = (long)blockIdx.x * 64 + threadIdx.x;
    // Lanes beyond the batch leave at once: the sweeps below store unconditionally (no branch per store), so no lane may
    // alias another instance's slots; the wave-level votes (__all / __any) only count the lanes that are still here.
    if (inst >= g.batch) return;
    constexpr bool live = true;
    const long ii = inst;
    const long bt = g.batch;
    const int N = g.N;
    const double rho = g.rho;
    // workspace, batch-minor: y[k][i][inst], lam[k][i][inst], kf[k][j][inst] (+ spare), rv[k][i][inst]
    double* y = g.ws;
    double* lam = g.ws + (long)N * W * bt;
    double* kf = g.ws + 2L * N * W * bt;
    double* rv = g.ws + 3L * N * W * bt;

    double x0[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) x0[i] = g.x0[ii * NS + i];
    bool x0_ok = true;
#pragma unroll
    for (int i = 0; i < NS; ++i) x0_ok &= (x0[i] >= x_lb[i]) && (x0[i] <= x_ub[i]);  // x_0 = x0 is box-constrained too (:56,:58)

    // warm start is per instance: only iterates of a solve that ended "optimal" are reused (the flag lives in the spare
    // part of the kf block); an instance that was infeasible / hit the limit last time starts cold
    double* okflag = g.ws + 2L * N * W * bt + (long)N * MC * bt;
    const bool lane_warm = g.warm && okflag[ii] == 1.0;
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i < W; ++i) {
            if (!lane_warm) {
                y[((long)k * W + i) * bt + ii] = 0.0;
                lam[((long)k * W + i) * bt + ii] = 0.0;
            } else if (g.warm == 2 && k + 1 < N) {   // receding horizon: the old plan advanced by one step (tail repeated)
                y[((long)k * W + i) * bt + ii] = y[((long)(k + 1) * W + i) * bt + ii];
                lam[((long)k * W + i) * bt + ii] = lam[((long)(k + 1) * W + i) * bt + ii];
            }
            rv[((long)k * W + i) * bt + ii] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < MC; ++j) kf[((long)k * MC + j) * bt + ii] = 0.0;
    }

    int status = x0_ok ? 0 : ZM_MPC_INFEASIBLE;
    int it = 0;  // this lane's ADMM iterations
    double rp = 0.0, rd = 0.0;
    bool near_ok = false;   // the last iterate's residuals are within 10x the tolerances (OSQP's "solved inaccurate" test at the cap)
    bool done = !live || status != 0;
    for (int gi = 0; gi < g.max_iter; ++gi) {  // gi is wave-uniform
        if (__all(done)) break;
        const bool chk = ((gi + 1) % ZM_MPC_CHK) == 0;  // infeasibility certificate on this iteration
        // ---- backward affine sweep.  Costate of x_{k+1}: p = -rho z(x_{k+1}) + (A^T p - K^T Qu)_{k+1};
        //      Qu = -rho z(u_k) + B^T p;  kf_k = Suu_k^-1 Qu.  Stage k touches only block k of (y, lam) (its states are the copy
        //      of x_{k+1}), and block k-1 is fetched while stage k computes: no load is predicated, no store is conditional
        //      (a finished lane rewrites the values it read), so the loop has no branch and one memory latency
        //      per stage is hidden behind ~500 FMAs.
        double p[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) p[i] = 0.0;
        double yb[W], lb[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const long e = ((long)(N - 1) * W + i) * bt + ii;
            yb[i] = y[e];
            lb[i] = lam[e];
        }
#pragma unroll 1
        for (int k = N - 1; k >= 0; --k) {
            const double* Kk = Ktab + (long)k * MC * NS;
            const double* Mk = Mtab + (long)k * MC * MC;
            double yq[W], lq[W], kfo[MC];
            {
                const int kp = k > 0 ? k - 1 : 0;
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const long e = ((long)kp * W + i) * bt + ii;
                    yq[i] = y[e];
                    lq[i] = lam[e];
                }
#pragma unroll
                for (int j = 0; j < MC; ++j) kfo[j] = kf[((long)k * MC + j) * bt + ii];
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) p[i] = __builtin_fma(-rho, yb[i] - lb[i], p[i]);
            double qu[MC];
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                double sacc = -rho * (yb[NS + j] - lb[NS + j]);
#pragma unroll
                for (int i = 0; i < NS; ++i) sacc = __builtin_fma(B[i * MC + j], p[i], sacc);
                qu[j] = sacc;
            }
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                double sacc = 0.0;
#pragma unroll
                for (int l = 0; l < MC; ++l) sacc = __builtin_fma(Mk[j * MC + l], qu[l], sacc);
                kf[((long)k * MC + j) * bt + ii] = done ? kfo[j] : sacc;   // a finished lane keeps the kf of its last iterate
            }
            double pn[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                double sacc = 0.0;
#pragma unroll
                for (int l = 0; l < NS; ++l) sacc = __builtin_fma(A[l * NS + i], p[l], sacc);
#pragma unroll
                for (int j = 0; j < MC; ++j) sacc = __builtin_fma(-Kk[j * NS + i], qu[j], sacc);
                pn[i] = sacc;
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) p[i] = pn[i];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                yb[i] = yq[i];
                lb[i] = lq[i];
            }
        }
        // ---- forward rollout w, projection y, dual update lam, residual norms (and r = w - y, support function on chk)
        double x[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = x0[i];
        double nrp = 0.0, nrd = 0.0, nw = 0.0, ny = 0.0, nl = 0.0, sup = 0.0, ndl = 0.0;
        auto forward = [&](auto chk_c) {
            constexpr bool CHK = decltype(chk_c)::value;
            double kfc[MC];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const long e = (long)i * bt + ii;
                yb[i] = y[e];
                lb[i] = lam[e];
            }
#pragma unroll
            for (int j = 0; j < MC; ++j) kfc[j] = kf[(long)j * bt + ii];
#pragma unroll 1
            for (int k = 0; k < N; ++k) {
                const double* Kk = Ktab + (long)k * MC * NS;
                double yq[W], lq[W], kfq[MC];
                {
                    const int kn = k + 1 < N ? k + 1 : N - 1;
#pragma unroll
                    for (int i = 0; i < W; ++i) {
                        const long e = ((long)kn * W + i) * bt + ii;
                        yq[i] = y[e];
                        lq[i] = lam[e];
                    }
#pragma unroll
                    for (int j = 0; j < MC; ++j) kfq[j] = kf[((long)kn * MC + j) * bt + ii];
                }
                double u[MC], xn[NS];
#pragma unroll
                for (int j = 0; j < MC; ++j) {
                    double sacc = -kfc[j];
#pragma unroll
                    for (int i = 0; i < NS; ++i) sacc = __builtin_fma(-Kk[j * NS + i], x[i], sacc);
                    u[j] = sacc;
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double sacc = 0.0;
#pragma unroll
                    for (int l = 0; l < NS; ++l) sacc = __builtin_fma(A[i * NS + l], x[l], sacc);
#pragma unroll
                    for (int j = 0; j < MC; ++j) sacc = __builtin_fma(B[i * MC + j], u[j], sacc);
                    xn[i] = sacc;
                }
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const double wv = (i < NS) ? xn[i < NS ? i : 0] : u[i >= NS ? i - NS : 0];
                    const double lo = (i < NS) ? x_lb[i < NS ? i : 0] : u_lb[i >= NS ? i - NS : 0];
                    const double hi = (i < NS) ? x_ub[i < NS ? i : 0] : u_ub[i >= NS ? i - NS : 0];
                    const long e = ((long)k * W + i) * bt + ii;
                    const double lold = lb[i], yold = yb[i];
                    const double wh = __builtin_fma(g.alpha, wv, (1.0 - g.alpha) * yold);   // relaxed iterate (alpha = 1: wv exactly)
                    double yn = wh + lold;
                    yn = yn < lo ? lo : (yn > hi ? hi : yn);
                    const double r = wv - yn, dl = wh - yn;   // primal residual; dual step
                    const double ln = lold + dl;
                    y[e] = done ? yold : yn;       // a finished lane keeps its iterate
                    lam[e] = done ? lold : ln;
                    if constexpr (CHK) {
                        rv[e] = dl;                // only read back by lanes that are not finished
                        sup += (dl > 0.0) ? dl * hi : ((dl < 0.0) ? dl * lo : 0.0);  // support function of the box at v = dl
                        ndl = __builtin_fmax(ndl, __builtin_fabs(dl));
                    }
                    nrp = __builtin_fmax(nrp, __builtin_fabs(r));
                    nrd = __builtin_fmax(nrd, __builtin_fabs(yn - yold));
                    nw = __builtin_fmax(nw, __builtin_fabs(wv));
                    ny = __builtin_fmax(ny, __builtin_fabs(yn));
                    nl = __builtin_fmax(nl, __builtin_fabs(ln));
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) x[i] = xn[i];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    yb[i] = yq[i];
                    lb[i] = lq[i];
                }
#pragma unroll
                for (int j = 0; j < MC; ++j) kfc[j] = kfq[j];
            }
        };
        if (chk)
            forward(std::true_type{});
        else
            forward(std::false_type{});
        bool need_cert = false;
        if (!done) {
            ++it;
            rp = nrp;
            rd = rho * nrd;
            const double ep = g.eps_abs + g.eps_rel * __builtin_fmax(nw, ny);
            const double ed = g.eps_abs + g.eps_rel * rho * nl;
            near_ok = (rp <= 10.0 * ep) && (rd <= 10.0 * ed);
            if (rp <= ep && rd <= ed) {
                status = ZM_MPC_OPTIMAL;
                done = true;
            } else if (!(rp == rp)) {
                done = true;  // NaN iterates (non-finite data): stop with the limit status
            } else {
                need_cert = chk;
            }
        }
        // ---- primal infeasibility certificate (see file header)
        if (chk && __any(need_cert)) {
            double sv[NS];
            {
                const long o = (long)(N - 1) * W;
#pragma unroll
                for (int i = 0; i < NS; ++i) sv[i] = rv[(o + i) * bt + ii];
            }
            double gmax = 0.0;
            for (int k = N - 1; k >= 0; --k) {
#pragma unroll
                for (int j = 0; j < MC; ++j) {
                    double sacc = rv[((long)k * W + NS + j) * bt + ii];
#pragma unroll
                    for (int i = 0; i < NS; ++i) sacc = __builtin_fma(B[i * MC + j], sv[i], sacc);
                    gmax = __builtin_fmax(gmax, __builtin_fabs(sacc));  // (G^T r)_k = ru_k + B^T s
                }
                double sn[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double sacc = (k >= 1) ? rv[((long)(k - 1) * W + i) * bt + ii] : 0.0;
#pragma unroll
                    for (int l = 0; l < NS; ++l) sacc = __builtin_fma(A[l * NS + i], sv[l], sacc);
                    sn[i] = sacc;
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) sv[i] = sn[i];
            }
            double vw0 = 0.0;  // v^T w(u = 0) = s^T x0   (s = sum_j (A^j)^T vx_j after the sweep)
#pragma unroll
            for (int i = 0; i < NS; ++i) vw0 = __builtin_fma(sv[i], x0[i], vw0);
            if (need_cert && gmax <= g.eps_pinf * ndl && (vw0 - sup) > g.eps_pinf * ndl) {
                status = ZM_MPC_INFEASIBLE;
                done = true;
            }
        }
    }
    // final trajectory: the dynamics-exact rollout w of the last iterate
    if (live) {
        double x[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            x[i] = x0[i];
            g.xTraj[(ii * (N + 1)) * NS + i] = x[i];
        }
        for (int k = 0; k < N; ++k) {
            const double* Kk = Ktab + (long)k * MC * NS;
            double u[MC], xn[NS];
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                double sacc = -kf[((long)k * MC + j) * bt + ii];
#pragma unroll
                for (int i = 0; i < NS; ++i) sacc = __builtin_fma(-Kk[j * NS + i], x[i], sacc);
                u[j] = sacc;
                g.uTraj[(ii * N + k) * MC + j] = sacc;
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                double sacc = 0.0;
#pragma unroll
                for (int l = 0; l < NS; ++l) sacc = __builtin_fma(A[i * NS + l], x[l], sacc);
#pragma unroll
                for (int j = 0; j < MC; ++j) sacc = __builtin_fma(B[i * MC + j], u[j], sacc);
                xn[i] = sacc;
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                x[i] = xn[i];
                g.xTraj[(ii * (N + 1) + k + 1) * NS + i] = x[i];
            }
        }
        g.status[ii] = status ? status : (near_ok ? ZM_MPC_OPTIMAL_INACCURATE : ZM_MPC_USER_LIMIT);
        okflag[ii] = (status == ZM_MPC_OPTIMAL) ? 1.0 : 0.0;
        if (g.iters) g.iters[ii] = it;
        if (g.resid) {
            g.resid[ii * 2] = rp;
            g.resid[ii * 2 + 1] = rd;
        }
    }
}


template <int NS, int MC>
static int launch_mpc(const MpcTabs& t, const MpcArgs& g, hipStream_t st) {
    hipLaunchKernelGGL((mpc_solve_kernel<NS, MC>), dim3((unsigned)((g.batch + 63) / 64)), dim3(64), 0, st, t.A, t.B, t.K,
                       t.Minv, t.x_lb, t.x_ub, t.u_lb, t.u_ub, g);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

}  // namespace zm

extern "C" int zm_mpc_setup_f64(const double* A, const double* B, const double* Q, const double* R, const double* Qf,
                                double rho, int N, int n, int m, double* K, double* Minv, void* stream) {
    if (!A || !B || !Q || !R || !Qf || !K || !Minv) return zm::set_error(ZM_EINVAL, "zm_mpc_setup_f64: null pointer");
    if (N < 1 || n < 1 || m < 1 || !(rho > 0.0)) return zm::set_error(ZM_EINVAL, "zm_mpc_setup_f64: bad size / rho");
    if (n > zm::SNL || m > zm::SML) return zm::set_error(ZM_EUNSUPPORTED, "zm_mpc_setup_f64: (n=%d, m=%d) not covered (n <= 24, m <= 8)", n, m);
    if (n <= zm::SN && m <= zm::SM)
        hipLaunchKernelGGL((zm::mpc_setup_kernel<zm::SN, zm::SM>), dim3(1), dim3(256), 0, (hipStream_t)stream, A, B, Q, R, Qf, rho, N, n,
                           m, K, Minv);
    else
        hipLaunchKernelGGL((zm::mpc_setup_kernel<zm::SNL, zm::SML>), dim3(1), dim3(256), 0, (hipStream_t)stream, A, B, Q, R, Qf, rho, N, n,
                           m, K, Minv);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

extern "C" int zm_mpc_solve_f64(const double* A, const double* B, const double* K, const double* Minv,
                                const double* x_lb, const double* x_ub, const double* u_lb, const double* u_ub,
                                const double* x0, double rho, double eps_abs, double eps_rel, double eps_prim_inf,
                                int max_iter, double* workspace, double* xTraj, double* uTraj, int32_t* status, int32_t* iters,
                                double* resid, int64_t batch, int N, int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    return zm_mpc_solve_warm_f64(A, B, K, Minv, x_lb, x_ub, u_lb, u_ub, x0, rho, eps_abs, eps_rel, eps_prim_inf, max_iter, 0,
                                 workspace, xTraj, uTraj, status, iters, resid, batch, N, n, m, stream);
}

extern "C" int zm_mpc_solve_warm_f64(const double* A, const double* B, const double* K, const double* Minv,
                                     const double* x_lb, const double* x_ub, const double* u_lb, const double* u_ub,
                                     const double* x0, double rho, double eps_abs, double eps_rel, double eps_prim_inf,
                                     int max_iter, int warm_start, double* workspace, double* xTraj, double* uTraj,
                                     int32_t* status, int32_t* iters, double* resid, int64_t batch, int N, int n, int m,
                                     void* stream) {
    return zm_mpc_solve_adaptive_f64(A, B, K, Minv, 1, 0, 1.0, x_lb, x_ub, u_lb, u_ub, x0, rho, eps_abs, eps_rel, eps_prim_inf,
                                     max_iter, warm_start, workspace, xTraj, uTraj, status, iters, resid, batch, N, n, m, stream);
}

extern "C" int zm_mpc_solve_adaptive_f64(const double* A, const double* B, const double* K, const double* Minv, int n_levels,
                                         int level0, double rho_step, const double* x_lb, const double* x_ub,
                                         const double* u_lb, const double* u_ub, const double* x0, double rho, double eps_abs,
                                         double eps_rel, double eps_prim_inf, int max_iter, int warm_start, double* workspace,
                                         double* xTraj, double* uTraj, int32_t* status, int32_t* iters, double* resid,
                                         int64_t batch, int N, int n, int m, void* stream) {
    return zm_mpc_solve_relaxed_f64(A, B, K, Minv, n_levels, level0, rho_step, 1.0, x_lb, x_ub, u_lb, u_ub, x0, rho, eps_abs, eps_rel,
                                    eps_prim_inf, max_iter, warm_start, workspace, xTraj, uTraj, status, iters, resid, batch, N, n, m,
                                    stream);
}

extern "C" int zm_mpc_solve_relaxed_f64(const double* A, const double* B, const double* K, const double* Minv, int n_levels,
                                        int level0, double rho_step, double alpha, const double* x_lb, const double* x_ub,
                                        const double* u_lb, const double* u_ub, const double* x0, double rho, double eps_abs,
                                        double eps_rel, double eps_prim_inf, int max_iter, int warm_start, double* workspace,
                                        double* xTraj, double* uTraj, int32_t* status, int32_t* iters, double* resid,
                                        int64_t batch, int N, int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;
    if (!(alpha > 0.0 && alpha < 2.0)) return zm::set_error(ZM_EINVAL, "zm_mpc_solve_relaxed_f64: alpha must lie in (0, 2)");   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!A || !B || !K || !Minv || !x_lb || !x_ub || !u_lb || !u_ub || !x0 || !workspace || !xTraj || !uTraj || !status)
        return zm::set_error(ZM_EINVAL, "zm_mpc_solve_f64: null pointer");
    if (batch < 0 || N < 1 || max_iter < 0 || !(rho > 0.0)) return zm::set_error(ZM_EINVAL, "zm_mpc_solve_f64: bad size");
    if (n_levels < 1 || level0 < 0 || level0 >= n_levels || (n_levels > 1 && !(rho_step > 1.0)))
        return zm::set_error(ZM_EINVAL, "zm_mpc_solve_adaptive_f64: bad penalty levels");
    zm::MpcTabs t{A, B, K, Minv, x_lb, x_ub, u_lb, u_ub};
    zm::MpcArgs g{x0, rho, eps_abs, eps_rel, eps_prim_inf, max_iter, warm_start == 2 ? 2 : (warm_start ? 1 : 0), workspace, xTraj, uTraj, (int*)status, (int*)iters, resid,
                  (long)batch, N, n_levels, level0, rho_step, alpha};
    hipStream_t st = (hipStream_t)stream;
    // default: 16 lanes per instance with the iterates in LDS (mpc_wave.hip); ZOPT_AMD_MPC_PATH=lane forces the
    // lane-per-instance kernel below, which also takes the horizons that do not fit LDS (fixed penalty: level0 only)
    static const bool force_lane = [] {
        const char* e = zm::fallback_env("ZOPT_AMD_MPC_PATH");
        return e && e[0] == 'l';
    }();
    if (!force_lane) {
        const int rc = zm::mpc_wave_dispatch(t, g, n, m, st);
        if (rc != ZM_EUNSUPPORTED) return rc;
    }
    t.K = K + (long)level0 * N * m * n;
    t.Minv = Minv + (long)level0 * N * m * m;
    if (n == 24 && m == 8) return zm::launch_mpc<24, 8>(t, g, st);   // beyond the 16-index tile: lane per instance, registers + scratch
    if (n == 12 && m == 4) return zm::launch_mpc<12, 4>(t, g, st);
    if (n == 8 && m == 4) return zm::launch_mpc<8, 4>(t, g, st);
    if (n == 4 && m == 2) return zm::launch_mpc<4, 2>(t, g, st);
    if (n == 4 && m == 1) return zm::launch_mpc<4, 1>(t, g, st);
    if (n == 2 && m == 2) return zm::launch_mpc<2, 2>(t, g, st);
    if (n == 2 && m == 1) return zm::launch_mpc<2, 1>(t, g, st);
    if (n == 1 && m == 1) return zm::launch_mpc<1, 1>(t, g, st);
    return zm::set_error(ZM_EUNSUPPORTED, "zm_mpc_solve_f64: (n=%d, m=%d) not among the compiled shapes", n, m);
}
