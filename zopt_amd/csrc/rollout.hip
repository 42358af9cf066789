// K6  rollout_linesearch -- batched policy rollout + parallel line search over step sizes, fp64, gfx950.
//
// Replaces zopt/ilqrUtils.py:33-66 (trajectoryRollout), :116-150 (forwardPass2), pytrees.py:215-220 (AffinePolicy call)
// and pytrees.py:49-52 (CostFunction call) for registered device models (models.h).
//
// Mapping: ONE LANE per (trajectory, step size).  With 16 step sizes a wave64 carries 4 trajectories; the 16 lanes of a
// trajectory read the same l_k, L_k, xPrev_k, uPrev_k addresses (one memory transaction each) and keep x, u in
// registers.  Pass 1 accumulates J for every step size; a 16-lane argmin (NaN wins, first index on ties, as
// jnp.argmin) picks the step; pass 2 re-rolls that step size (bit-identical arithmetic) and its first lane stores the
// trajectory -- the other 15 rollouts never touch HBM.  This chain is latency/ALU-bound, not HBM-bound: it reads
// 8(mn + 2m + n) = 544 B per trajectory step.
#include "models.h"
#include "zm_common.h"

#include <cstdlib>

namespace zm {

// one rollout; STORE: write xTraj/uTraj; returns J (0 when cost == nullptr)
template <bool STORE>
__device__ __forceinline__ double rollout_one(const zm_model_t& md, const zm_quadcost_t* cs, const double alpha,
                                              const double* __restrict__ x0, const double* __restrict__ l,
                                              const double* __restrict__ L, const double* __restrict__ xPrev,
                                              const double* __restrict__ uPrev, double* __restrict__ xTraj,
                                              double* __restrict__ uTraj, const int T) {
    const int n = md.n, m = md.m;
    double x[MAXN], u[MAXM], xn[MAXN];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) x[i] = (i < n) ? x0[i] : 0.0;
    if (STORE) {
#pragma unroll
        for (int i = 0; i < MAXN; ++i)
            if (i < n) xTraj[i] = x[i];
    }
    double J = 0.0;
    for (int k = 0; k < T; ++k) {
        const double* Lk = L + (long)k * m * n;
        const double* xp = xPrev + (long)k * n;
        double dx[MAXN];
#pragma unroll
        for (int j = 0; j < MAXN; ++j) dx[j] = (j < n) ? (x[j] - xp[j]) : 0.0;
#pragma unroll
        for (int i = 0; i < MAXM; ++i) {
            if (i < m) {
                double s = 0.0;  // L_k @ dx
#pragma unroll
                for (int j = 0; j < MAXN; ++j)
                    if (j < n) s = __builtin_fma(Lk[i * n + j], dx[j], s);
                u[i] = (alpha * l[(long)k * m + i] + s) + uPrev[(long)k * m + i];   // pytrees.py:220, ilqrUtils.py:60
            } else {
                u[i] = 0.0;
            }
        }
        if (cs) J += running_cost(*cs, n, m, x, u);
        model_step<double>(md, x, u, xn);
#pragma unroll
        for (int i = 0; i < MAXN; ++i) x[i] = xn[i];
        if (STORE) {
#pragma unroll
            for (int i = 0; i < MAXM; ++i)
                if (i < m) uTraj[(long)k * m + i] = u[i];
#pragma unroll
            for (int i = 0; i < MAXN; ++i)
                if (i < n) xTraj[(long)(k + 1) * n + i] = x[i];
        }
    }
    if (cs) J += terminal_cost(*cs, n, x);
    return J;
}

// NA = lanes per trajectory (power of two <= 16 that holds n_alpha)
template <int NA>
__global__ __launch_bounds__(64) void rollout_linesearch_kernel(const zm_model_t md, const zm_quadcost_t cost,
                                                                const bool has_cost, const double* __restrict__ x0,
                                                                const double* __restrict__ l, const double* __restrict__ L,
                                                                const double* __restrict__ xPrev,
                                                                const double* __restrict__ uPrev,
                                                                const double* __restrict__ alphas, const int n_alpha,
                                                                const int* __restrict__ active, const int* __restrict__ list,
                                                                const long count, double* __restrict__ xTraj,
                                                                double* __restrict__ uTraj, double* __restrict__ Jout,
                                                                int* __restrict__ idx_out, const long batch, const int T) {
    constexpr int TPW = 64 / NA;  // trajectories per wave
    const int lane = threadIdx.x;
    const int a = lane % NA;
    const long slot = (long)blockIdx.x * TPW + lane / NA;   // slot -> trajectory id (through `list` when given)
    const long nslot = list ? count : batch;
    const long traj = (slot < nslot) ? (list ? (long)list[slot] : slot) : 0;
    const bool live = (slot < nslot) && (a < n_alpha) && (active == nullptr || active[traj] != 0);
    const long t = live ? traj : 0;
    const int n = md.n, m = md.m;
    const zm_quadcost_t* cs = has_cost ? &cost : nullptr;
    const double alpha = alphas[a < n_alpha ? a : 0];
    const double* x0t = x0 + t * n;
    const double* lt = l + t * T * m;
    const double* Lt = L + t * T * m * n;
    const double* xpt = xPrev + t * (T + 1) * n;
    const double* upt = uPrev + t * T * m;
    double* xo = xTraj + t * (T + 1) * n;
    double* uo = uTraj + t * T * m;

    double J = 0.0;
    int best = 0;
    if (NA > 1) {
        if (live) J = rollout_one<false>(md, cs, alpha, x0t, lt, Lt, xpt, upt, nullptr, nullptr, T);
        // argmin over the NA lanes of this trajectory with jnp.argmin semantics: a NaN cost beats every number
        // (also -inf), the first index wins among equals.  Dead lanes carry (+inf, index NA).
        double key = live ? J : __builtin_inf();
        int isn = (live && (J != J)) ? 1 : 0;
        int who = live ? a : NA;
#pragma unroll
        for (int off = NA / 2; off >= 1; off >>= 1) {
            const double ok = __shfl_xor(key, off);
            const int on = __shfl_xor(isn, off);
            const int ow = __shfl_xor(who, off);
            const bool better = (on > isn) || (on == isn && ((on == 0 && ok < key) || ((on == 1 || ok == key) && ow < who)));
            key = better ? ok : key;
            isn = better ? on : isn;
            who = better ? ow : who;
        }
        best = who < NA ? who : 0;
    }
    // every lane of the trajectory re-rolls the winning step size; lane 0 of the group stores
    const double abest = alphas[best];
    if (live) {
        if (a == 0) {
            const double Jb = rollout_one<true>(md, cs, abest, x0t, lt, Lt, xpt, upt, xo, uo, T);
            if (Jout) Jout[t] = Jb;
            if (idx_out) idx_out[t] = best;
        }
    }
}

// rollout_fast.hip: compile-time (n, m) = (12, 4), 16 step sizes
int rollout_fast_dispatch(const zm_model_t& md, const double* Q, const double* R, const double* Qf, int diagonal, const double* x0,
                          const double* l, const double* L, const double* xPrev, const double* uPrev,
                          const double* alphas, int n_alpha, const int* active, const int* list, int64_t count, double* xTraj,
                          double* uTraj, double* J, int* idx, int64_t batch, int T, hipStream_t st, double* scratch);

}  // namespace zm

namespace zm {
int rollout_wide_dispatch(const zm_model_t& md, const zm_quadcost_t* cost, const double* x0, const double* l, const double* L,
                          const double* xPrev, const double* uPrev, const double* alphas, int n_alpha, const int* active,
                          const int* list, long count, double* xTraj, double* uTraj, double* J, int* idx, long batch, int T,
                          hipStream_t st);   // rollout_wide.hip
}
static int rollout_impl(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0,
                                         const double* l, const double* L, const double* xPrev, const double* uPrev,
                                         const double* alphas, int n_alpha, const int32_t* active, const int32_t* list,
                                         int64_t count, double* xTraj, double* uTraj, double* J, int32_t* alpha_idx,
                                         int64_t batch, int T, void* stream, double* scratch = nullptr) {
    if (!model || !x0 || !l || !L || !xPrev || !uPrev || !alphas || !xTraj || !uTraj)
        return zm::set_error(ZM_EINVAL, "zm_rollout_linesearch_f64: null pointer");
    if (batch < 0 || T < 0 || n_alpha < 1 || n_alpha > 16)
        return zm::set_error(ZM_EINVAL, "zm_rollout_linesearch_f64: bad size batch=%lld T=%d n_alpha=%d", (long long)batch,
                             T, n_alpha);
    if (n_alpha > 1 && !cost) return zm::set_error(ZM_EINVAL, "zm_rollout_linesearch_f64: a line search needs a cost");
    zm_model_t md = *model;
    if (md.kind == ZM_MODEL_QUADCOPTER) {
        md.n = 12;
        md.m = 4;
    } else if (md.kind == ZM_MODEL_QUADCOPTER_RB) {
        md.n = 8;
        md.m = 4;
    } else if (md.kind == ZM_MODEL_LINEAR) {
        if (!md.A || !md.B) return zm::set_error(ZM_EINVAL, "zm_rollout_linesearch_f64: linear model needs A, B");
    } else {
        return zm::set_error(ZM_EUNSUPPORTED, "zm_rollout_linesearch_f64: unknown model kind %d", md.kind);
    }
    const bool wide = md.kind == ZM_MODEL_LINEAR && (md.n > zm::MAXN || md.m > zm::MAXM) && md.n >= 1 && md.m >= 1 && md.n <= 64 &&
                      md.m <= 16;   // large linear models: one wave per rollout (rollout_wide.hip)
    if (!wide && (md.n < 1 || md.n > zm::MAXN || md.m < 1 || md.m > zm::MAXM))
        return zm::set_error(ZM_EUNSUPPORTED, "zm_rollout_linesearch_f64: (n=%d, m=%d) not covered (n<=12, m<=4; linear models n<=64, m<=16)",
                             md.n, md.m);
    if (cost && (!cost->Q || !cost->R || !cost->Qf))
        return zm::set_error(ZM_EINVAL, "zm_rollout_linesearch_f64: cost needs Q, R, Qf");
    if (batch == 0 || (list && count == 0)) return ZM_OK;
    if (count < 0 || count > batch) return zm::set_error(ZM_EINVAL, "zm_rollout_linesearch: bad list length");
    if (wide) {
        if (scratch) return zm::set_error(ZM_EUNSUPPORTED, "rollout: all-store mode needs the fast path");
        return zm::rollout_wide_dispatch(md, cost, x0, l, L, xPrev, uPrev, alphas, n_alpha, (const int*)active, (const int*)list,
                                         (long)count, xTraj, uTraj, J, (int*)alpha_idx, (long)batch, T, (hipStream_t)stream);
    }
    if (scratch && !(n_alpha == 16 && list && alpha_idx && J)) return zm::set_error(ZM_EINVAL, "rollout: all-store mode needs 16 step sizes, a list, J and alpha_idx");
    const int64_t nslot = list ? count : batch;
    hipStream_t st = (hipStream_t)stream;
    zm_quadcost_t cs = cost ? *cost : zm_quadcost_t{nullptr, nullptr, nullptr, 0, 0};
    const bool hc = cost != nullptr;
    const int* act = (const int*)active;
    static const bool force_generic = [] {
        const char* e = zm::fallback_env("ZOPT_AMD_ROLLOUT_PATH");
        return e && e[0] == 'g';
    }();
    const bool windy = md.wind_ned[0] != 0.0 || md.wind_ned[1] != 0.0 || md.wind_ned[2] != 0.0;   // fast path: still air only
    // (one step size -- the solvers' initial rollout -- runs the same kernel with all 16 lanes of a group on that step size: 16x
    //  redundant, and still several times faster than the generic lane-per-trajectory kernel with its uncoalesced policy reads)
    if (!force_generic && !windy && (n_alpha == 16 || n_alpha == 1) && md.n == 12 && md.m == 4 && cost && T >= 1)
        return zm::rollout_fast_dispatch(md, cs.Q, cs.R, cs.Qf, cs.diagonal, x0, l, L, xPrev, uPrev, alphas, n_alpha, act, (const int*)list, count, xTraj,
                                         uTraj, J, (int*)alpha_idx, batch, T, st, scratch);
    if (scratch) return zm::set_error(ZM_EUNSUPPORTED, "rollout: all-store mode needs the fast path");
    if (n_alpha == 1) {
        const unsigned blocks = (unsigned)((nslot + 63) / 64);
        hipLaunchKernelGGL((zm::rollout_linesearch_kernel<1>), dim3(blocks), dim3(64), 0, st, md, cs, hc, x0, l, L, xPrev,
                           uPrev, alphas, n_alpha, act, (const int*)list, (long)count, xTraj, uTraj, J, (int*)alpha_idx, (long)batch, T);
    } else {
        const unsigned blocks = (unsigned)((nslot + 3) / 4);
        hipLaunchKernelGGL((zm::rollout_linesearch_kernel<16>), dim3(blocks), dim3(64), 0, st, md, cs, hc, x0, l, L, xPrev,
                           uPrev, alphas, n_alpha, act, (const int*)list, (long)count, xTraj, uTraj, J, (int*)alpha_idx, (long)batch, T);
    }
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

extern "C" int zm_rollout_linesearch_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* l,
                                         const double* L, const double* xPrev, const double* uPrev, const double* alphas,
                                         int n_alpha, const int32_t* active, double* xTraj, double* uTraj, double* J,
                                         int32_t* alpha_idx, int64_t batch, int T, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    return rollout_impl(model, cost, x0, l, L, xPrev, uPrev, alphas, n_alpha, active, nullptr, 0, xTraj, uTraj, J, alpha_idx,
                        batch, T, stream);
}

extern "C" int zm_rollout_linesearch_list_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0,
                                              const double* l, const double* L, const double* xPrev, const double* uPrev,
                                              const double* alphas, int n_alpha, const int32_t* list, int64_t count,
                                              const int32_t* active, double* xTraj, double* uTraj, double* J, int32_t* alpha_idx,
                                              int64_t batch, int T, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!list) return zm::set_error(ZM_EINVAL, "zm_rollout_linesearch_list_f64: null list");
    return rollout_impl(model, cost, x0, l, L, xPrev, uPrev, alphas, n_alpha, active, list, count, xTraj, uTraj, J, alpha_idx,
                        batch, T, stream);
}

// Solver-internal (ilqr_solve.hip): the 16-way line search in all-store mode -- every step size's rollout goes to the scratch blocks
// of its slot ((T+1) * 256 doubles per slot, layout in rollout_fast.hip), alpha_idx[t] names the winner, nothing is written to
// xTraj / uTraj.
namespace zm {
// the part of the predicate below that is known when the workspace is sized (model kind, dimensions, wind, the fallback switch)
bool rollout_all_store_model_ok(const zm_model_t* model) {
    if (!model) return false;
    static const bool force_generic = [] {
        const char* e = zm::fallback_env("ZOPT_AMD_ROLLOUT_PATH");
        return e && e[0] == 'g';
    }();
    const bool windy = model->wind_ned[0] != 0.0 || model->wind_ned[1] != 0.0 || model->wind_ned[2] != 0.0;
    const bool dims = model->kind == ZM_MODEL_QUADCOPTER || (model->kind == ZM_MODEL_LINEAR && model->n == 12 && model->m == 4);
    return !force_generic && !windy && dims;
}
bool rollout_all_store_supported(const zm_model_t* model, const zm_quadcost_t* cost, int T) {
    return model && cost && T >= 1 && rollout_all_store_model_ok(model);
}
int rollout_linesearch_all_store(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* l, const double* L,
                                 const double* xPrev, const double* uPrev, const double* alphas, const int32_t* list, int64_t count,
                                 const int32_t* active, double* scratch, double* J, int32_t* alpha_idx, int64_t batch, int T,
                                 void* stream) {
    // (xTraj / uTraj are not written in this mode; the non-null placeholders only pass the argument check)
    return rollout_impl(model, cost, x0, l, L, xPrev, uPrev, alphas, 16, active, list, count, scratch, scratch, J, alpha_idx, batch, T,
                        stream, scratch);
}
}  // namespace zm

// ---------------------------------------------------------------------------------------------------------------------
// Acceptance step of the iLQR / DDP loop for the trajectories of a compacted id list: take the line search's result
// (trajectory, cost), test convergence and retire converged trajectories.  Replaces zopt/ilqrUtils.py:316-320
//     converged = abs(J - J_new) <= tol;  traj, J = traj_new, J_new
// One block per listed trajectory; only those rows are touched (a masked whole-array update would move the entire batch
// every iteration although most of it has converged).
namespace zm {
__global__ __launch_bounds__(256) void ilqr_accept_kernel(const int* __restrict__ list, const long count,
                                                          double* __restrict__ J, const double* __restrict__ Jn,
                                                          double* __restrict__ xT, const double* __restrict__ xT2,
                                                          double* __restrict__ uT, const double* __restrict__ uT2,
                                                          int* __restrict__ converged, int* __restrict__ active,
                                                          const double tol, const long xrow, const long urow,
                                                          const double* __restrict__ scratch, const int* __restrict__ idx,
                                                          int* __restrict__ where, const int where_val) {
    const long slot = blockIdx.x;
    if (slot >= count) return;
    const long t = list[slot];
    // The mask is read ONCE per block and shared: thread 0 clears active[t] at the end of this same block, and a wave that
    // is scheduled late must not see that store and skip its stripe of the copy (a torn trajectory).
    __shared__ int act;
    // (everything that depends on t alone leaves together with the mask's load: one memory round trip instead of three)
    const int win = scratch ? idx[t] : 0;
    double jn = 0.0, jo = 0.0;
    if (threadIdx.x == 0) {
        act = active[t];   // the list may be older than the mask (it is rebuilt only every few iterations)
        jn = Jn[t];
        jo = J[t];
    }
    __syncthreads();
    if (act == 0) return;
    // the new trajectory: row t of (xT2, uT2), or -- after an all-store line search (n = 12, m = 4) -- the winner's 16-byte pieces
    // of this slot's scratch blocks (layout: rollout_fast.hip, allstore)
    if (scratch) {
        // 16-byte pieces: (T + 1) * 6 of x, T * 2 of u
        constexpr int n = 12, m = 4, PB = (n + m) / 2;
        typedef double d2 __attribute__((ext_vector_type(2)));
        const double* sb = scratch + slot * (xrow / n) * (PB * 32) + win * 2;
        const long nxp = xrow / 2, nup = urow / 2;
        for (long e = threadIdx.x; e < nxp; e += blockDim.x) {
            const long k = e / (n / 2), p = e % (n / 2);
            *(d2*)(xT + t * xrow + 2 * e) = *(const d2*)(sb + (k * PB + p) * 32);
        }
        for (long e = threadIdx.x; e < nup; e += blockDim.x) {
            const long k = e / (m / 2), p = e % (m / 2);
            *(d2*)(uT + t * urow + 2 * e) = *(const d2*)(sb + ((k + 1) * PB + n / 2 + p) * 32);
        }
    } else if (xT2) {
        for (long e = threadIdx.x; e < xrow; e += blockDim.x) xT[t * xrow + e] = xT2[t * xrow + e];
        for (long e = threadIdx.x; e < urow; e += blockDim.x) uT[t * urow + e] = uT2[t * urow + e];
    }   // (else: the line search wrote the new rows where the caller wants them -- the solvers' alternating buffers)
    if (threadIdx.x == 0) {
        if (where) where[t] = where_val;   // which of the solver's two buffers holds this trajectory's newest rows
        const int cv = (__builtin_fabs(jo - jn) <= tol) ? 1 : 0;   // NaN compares false: never "converged"
        J[t] = jn;
        converged[t] = cv;
        active[t] = cv ? 0 : 1;
    }
}
}  // namespace zm

namespace zm {
int ilqr_accept(const int32_t* list, int64_t count, double* J, const double* Jn, double* xTraj, const double* xTrajNew, double* uTraj,
                const double* uTrajNew, int32_t* converged, int32_t* active, double tol, int64_t batch, int T, int n, int m,
                void* stream, const double* scratch, const int32_t* idx, int32_t* where, int where_val) {
    if (batch == 0 || count == 0) return ZM_OK;
    // (where != nullptr and no scratch: xTrajNew / uTrajNew may be nullptr -- nothing to copy, the new rows are in place)
    if (!list || !J || !Jn || !xTraj || !uTraj || !converged || !active || (scratch ? !idx : (!where && (!xTrajNew || !uTrajNew))))
        return set_error(ZM_EINVAL, "zm_ilqr_accept_f64: null pointer");
    if (count < 0 || count > batch || T < 1 || n < 1 || m < 1) return set_error(ZM_EINVAL, "zm_ilqr_accept_f64: bad size");
    if (scratch && (n != 12 || m != 4)) return set_error(ZM_EUNSUPPORTED, "ilqr_accept: all-store scratch is laid out for n = 12, m = 4");
    hipLaunchKernelGGL(ilqr_accept_kernel, dim3((unsigned)count), dim3(256), 0, (hipStream_t)stream, (const int*)list,
                       (long)count, J, Jn, xTraj, xTrajNew, uTraj, uTrajNew, (int*)converged, (int*)active, tol,
                       (long)(T + 1) * n, (long)T * m, scratch, (const int*)idx, (int*)where, where_val);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// End of a solve on alternating buffers: the rows of every trajectory whose newest state sits in the workspace buffer (where[t] != 0)
// move to the caller's arrays.  One block per trajectory.
__global__ __launch_bounds__(256) void ilqr_collect_kernel(const int* __restrict__ where, double* __restrict__ xT,
                                                           const double* __restrict__ xAlt, double* __restrict__ uT,
                                                           const double* __restrict__ uAlt, const long xrow, const long urow) {
    const long t = blockIdx.x;
    if (where[t] == 0) return;
    for (long e = threadIdx.x; e < xrow; e += blockDim.x) xT[t * xrow + e] = xAlt[t * xrow + e];
    for (long e = threadIdx.x; e < urow; e += blockDim.x) uT[t * urow + e] = uAlt[t * urow + e];
}
int ilqr_collect(const int32_t* where, double* xTraj, const double* xAlt, double* uTraj, const double* uAlt, int64_t batch, int T, int n,
                 int m, void* stream) {
    if (batch == 0) return ZM_OK;
    hipLaunchKernelGGL(ilqr_collect_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const int*)where, xTraj, xAlt,
                       uTraj, uAlt, (long)(T + 1) * n, (long)T * m);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
}  // namespace zm

extern "C" int zm_ilqr_accept_f64(const int32_t* list, int64_t count, double* J, const double* Jn, double* xTraj,
                                  const double* xTrajNew, double* uTraj, const double* uTrajNew, int32_t* converged,
                                  int32_t* active, double tol, int64_t batch, int T, int n, int m, void* stream) {
    return zm::ilqr_accept(list, count, J, Jn, xTraj, xTrajNew, uTraj, uTrajNew, converged, active, tol, batch, T, n, m, stream,
                           nullptr, nullptr, nullptr, 0);
}
