// Shared argument blocks of the MPC solve kernels (mpc.hip: one lane per instance; mpc_wave.hip: 16 lanes per instance).
#pragma once
#include "zm_common.h"

// Every ZM_MPC_CHK-th ADMM iteration checks the primal-infeasibility certificate and lets the adaptive penalty move (OSQP's
// `check_termination` / `adaptive_rho_interval`, both tunables of the solver, not of the problem).  8 with moves to the NEAREST
// tabulated level: measured on BASELINE configs[2] (profiles/r03_mpc_check_interval_ab.txt) 106 -> 60 worst-case iterations.
#ifndef ZM_MPC_CHK
#define ZM_MPC_CHK 8
#endif

namespace zm {

struct MpcArgs {
    const double* x0;
    double rho, eps_abs, eps_rel, eps_pinf;
    int max_iter;
    int warm;   // 1: the workspace holds the iterates (y, lam) of a previous solve of the same problem family; 2: same, shifted by one step
    double *ws, *xTraj, *uTraj;
    int *status, *iters;
    double* resid;
    long batch;
    int N;
    // adaptive penalty (OSQP's adaptive_rho): tables for n_levels penalties rho * rho_step^(l - level0), level-major in K / Minv
    int n_levels, level0;
    double rho_step;
    // over-relaxation (OSQP's alpha; 1 = plain ADMM): w_hat = alpha w + (1 - alpha) y_prev enters the projection and the dual update
    double alpha;
};

struct MpcTabs {
    const double *A, *B, *K, *Minv, *x_lb, *x_ub, *u_lb, *u_ub;
};

// mpc_wave.hip: 16 lanes per instance, iterates in LDS.  ZM_EUNSUPPORTED if the shape / horizon does not fit.
int mpc_wave_dispatch(const MpcTabs& t, const MpcArgs& g, int n, int m, hipStream_t st);

}  // namespace zm
