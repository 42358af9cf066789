// K6-W  rollout + 16-way line search for LARGE linear models (ZM_MODEL_LINEAR with 12 < n <= 64 or 4 < m <= 16), fp64.
//
// Replaces zopt/ilqrUtils.py:33-66 (trajectoryRollout), :116-150 (forwardPass2), pytrees.py:215-220 (AffinePolicy call) and
// pytrees.py:49-52 (CostFunction call) where the lane-per-rollout kernels of rollout.hip (register arrays of MAXN = 12 states) stop:
//     for each alpha:  x_0 = x0;  u_k = alpha l_k + L_k (x_k - xPrev_k) + uPrev_k;  x_{k+1} = A x_k + B u_k
//                      J = sum_k (x_k^T Q x_k + u_k^T R u_k) + x_T^T Qf x_T;     result = the rollout with the smallest J
// Mapping: one WAVE per rollout, lane i owns state component i: row i of A, Q (and B, R for the control lanes) sits in the lane's
// registers for the whole horizon, the state / deviation / control vectors of the step go through a wave-private LDS strip and
// are read back as broadcasts.  The policy product L_k dx (m x n) runs on all 64 lanes: lane (q, r) sums the 16 columns
// 16q .. 16q+15 of row r, the four partial sums meet by the row swaps of zm_common.h.  A workgroup = four waves = one
// trajectory: wave w rolls the step sizes w, w+4, w+8, w+12 out one after the other (costs only), the 16 costs meet in LDS,
// `jnp.argmin` semantics pick the winner (NaN beats everything, first index wins), and wave 0 rolls the winner out once more
// with stores -- one launch, no scratch memory, the other 15 rollouts never touch HBM.
#include "models.h"
#include "zm_common.h"

namespace zm {

struct WideArgs {
    const double *A, *B, *Q, *R, *Qf;
    const double *x0, *l, *L, *xPrev, *uPrev, *alphas;
    const int *active, *list;
    long count;
    double *xTraj, *uTraj, *J;
    int* idx;
    long batch;
    int T, n, m, n_alpha, has_cost;
};

template <int NP>
struct WideRows {   // what a lane keeps for the whole horizon
    double a[NP], q[NP], b[16], r[16];
};

// one rollout of the wave's trajectory at step size `alpha`; returns the wave-wide cost (every lane); STORE: writes xTraj / uTraj
template <int NP, bool STORE>
__device__ __forceinline__ double wide_rollout(const WideArgs& g, const WideRows<NP>& w, const long traj, const double alpha,
                                               double* __restrict__ xs, double* __restrict__ dxs, double* __restrict__ us,
                                               const int lane) {
    const int n = g.n, m = g.m, T = g.T;
    const int q4 = lane >> 4, r = lane & 15;
    const double* lb = g.l + traj * (long)T * m;
    const double* Lb = g.L + traj * (long)T * m * n;
    const double* xpb = g.xPrev + traj * (long)(T + 1) * n;
    const double* upb = g.uPrev + traj * (long)T * m;
    double* xT = g.xTraj + traj * (long)(T + 1) * n;
    double* uT = g.uTraj + traj * (long)T * m;
    double xi = lane < n ? g.x0[traj * n + lane] : 0.0;
    xs[lane] = xi;
    if (STORE && lane < n) xT[lane] = xi;
    double Jl = 0.0;
    wave_lds_sync();
    for (int k = 0; k < T; ++k) {
        // deviation from the previous trajectory, by state index
        dxs[lane] = lane < n ? xi - xpb[(long)k * n + lane] : 0.0;
        wave_lds_sync();
        // u_r = (alpha l_r + sum_j L[r][j] dx_j) + uPrev_r     (pytrees.py:220, ilqrUtils.py:60)
        double s = 0.0;
        if (r < m) {
            const double* Lr = Lb + ((long)k * m + r) * n + 16 * q4;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj)
                if (16 * q4 + jj < n) s = __builtin_fma(Lr[jj], dxs[16 * q4 + jj], s);
        }
        s = sum_xor32(sum_xor16(s));
        const double ur = r < m ? (alpha * lb[(long)k * m + r] + s) + upb[(long)k * m + r] : 0.0;
        if (q4 == 0) us[r] = ur;
        wave_lds_sync();
        // running cost x^T Q x + u^T R u and the step x+ = A x + B u: one pass over the broadcast state / control
        double qx = 0.0, xn = 0.0;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const double xj = xs[j];
            qx = __builtin_fma(w.q[j], xj, qx);
            xn = __builtin_fma(w.a[j], xj, xn);
        }
        double ru = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double uj = us[j];
            ru = __builtin_fma(w.r[j], uj, ru);
            xn = __builtin_fma(w.b[j], uj, xn);
        }
        Jl = __builtin_fma(xi, qx, Jl);
        if (lane < m) Jl = __builtin_fma(ur, ru, Jl);
        wave_lds_sync();                // every lane has read xs: it may be rewritten
        xi = lane < n ? xn : 0.0;
        xs[lane] = xi;
        if (STORE) {
            if (lane < m) uT[(long)k * m + lane] = ur;
            if (lane < n) xT[(long)(k + 1) * n + lane] = xi;
        }
        wave_lds_sync();
    }
    if (g.has_cost) {   // terminal cost x_T^T Qf x_T
        double qf = 0.0;
        if (lane < n) {
            for (int j = 0; j < n; ++j) qf = __builtin_fma(g.Qf[lane * n + j], xs[j], qf);
        }
        Jl = __builtin_fma(xi, qf, Jl);
    }
    return sum_xor32(sum_xor16(row16_sum(Jl)));
}

template <int NP>
__global__ __launch_bounds__(256) void rollout_wide_kernel(const WideArgs g) {
    __shared__ double strips[4][64 + 64 + 16];
    __shared__ double Js[16];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const long slot = blockIdx.x;
    const long traj = g.list ? (long)g.list[slot] : slot;
    if (traj >= g.batch || (g.active && !g.active[traj])) return;   // whole workgroups only: the barrier below is safe
    const int n = g.n, m = g.m;
    WideRows<NP> w;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const bool ok = lane < n && j < n;
        w.a[j] = ok ? g.A[lane * n + j] : 0.0;
        w.q[j] = (ok && g.has_cost) ? g.Q[lane * n + j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        w.b[j] = (lane < n && j < m) ? g.B[lane * m + j] : 0.0;
        w.r[j] = (lane < m && j < m && g.has_cost) ? g.R[lane * m + j] : 0.0;
    }
    double* xs = strips[wave];
    double* dxs = xs + 64;
    double* us = dxs + 64;
    if (g.n_alpha == 1) {   // trajectoryRollout: one rollout, stored as it runs
        if (wave == 0) {
            const double J = wide_rollout<NP, true>(g, w, traj, g.alphas[0], xs, dxs, us, lane);
            if (lane == 0) {
                if (g.J) g.J[traj] = J;
                if (g.idx) g.idx[traj] = 0;
            }
        }
        return;
    }
    for (int a = wave; a < g.n_alpha; a += 4) {
        const double J = wide_rollout<NP, false>(g, w, traj, g.alphas[a], xs, dxs, us, lane);
        if (lane == 0) Js[a] = J;
    }
    __syncthreads();
    if (wave == 0) {
        int best = 0;   // jnp.argmin: NaN is the minimum, the first index wins ties (ilqrUtils.py:147)
        double jb = Js[0];
        for (int a = 1; a < g.n_alpha; ++a) {
            const double ja = Js[a];
            const bool take = (ja != ja) ? !(jb != jb) : (!(jb != jb) && ja < jb);
            if (take) {
                best = a;
                jb = ja;
            }
        }
        const double J = wide_rollout<NP, true>(g, w, traj, g.alphas[best], xs, dxs, us, lane);   // same arithmetic, now stored
        if (lane == 0) {
            if (g.J) g.J[traj] = J;
            if (g.idx) g.idx[traj] = best;
        }
    }
}

// ZM_EUNSUPPORTED unless the model is linear with n <= 64, m <= 16.
int rollout_wide_dispatch(const zm_model_t& md, const zm_quadcost_t* cost, const double* x0, const double* l, const double* L,
                          const double* xPrev, const double* uPrev, const double* alphas, int n_alpha, const int* active,
                          const int* list, long count, double* xTraj, double* uTraj, double* J, int* idx, long batch, int T,
                          hipStream_t st) {
    if (md.kind != ZM_MODEL_LINEAR || md.n < 1 || md.n > 64 || md.m < 1 || md.m > 16) return ZM_EUNSUPPORTED;
    const WideArgs g{md.A, md.B, cost ? cost->Q : nullptr, cost ? cost->R : nullptr, cost ? cost->Qf : nullptr, x0, l, L, xPrev, uPrev,
                     alphas, active, list, count, xTraj, uTraj, J, idx, batch, T, md.n, md.m, n_alpha, cost ? 1 : 0};
    const long nslot = list ? count : batch;
    if (nslot == 0) return ZM_OK;
    const dim3 grid((unsigned)nslot), block(256);
    if (md.n <= 16)
        hipLaunchKernelGGL((rollout_wide_kernel<16>), grid, block, 0, st, g);
    else if (md.n <= 32)
        hipLaunchKernelGGL((rollout_wide_kernel<32>), grid, block, 0, st, g);
    else if (md.n <= 48)
        hipLaunchKernelGGL((rollout_wide_kernel<48>), grid, block, 0, st, g);
    else
        hipLaunchKernelGGL((rollout_wide_kernel<64>), grid, block, 0, st, g);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

}  // namespace zm
