// K6 (fast path)  rollout_linesearch for compile-time (model, n = 12, m = 4), 16 step sizes -- gfx950.
//
// Same arithmetic as rollout.hip (reference ilqrUtils.py:33-66, 116-150; pytrees.py:49-52, 215-220); what changes:
//  * compile-time model / dimensions: no predicated loops, no branches on the model kind;
//  * a step's policy data [l_k | L_k | xPrev_k | uPrev_k] (68 doubles) is fetched ONCE per trajectory by the 16 lanes of
//    its group with coalesced loads, three steps ahead, and stays in the loading lanes' registers: every use is an FMA whose DPP
//    operand reads the owning lane (PolRegs below; the generic kernel issued 68 redundant global loads per lane and step, rounds
//    1-2 staged the data in LDS);
//  * cost matrices (and A, B of a linear model) live in LDS; diagonal Q, R, Qf (the demos' weights) use 12 + 4 FMAs;
//  * the alpha = 1 lane stores its rollout speculatively during pass 1: pass 2 (re-roll of the winner) only runs for
//    trajectories whose argmin is another step size;
//  * "all-store" mode (FastArgs::scratch, used by the solvers once few trajectories are left and a launch lasts as long as its
//    slowest wave's 2 x T dependent steps): EVERY lane stores its rollout into a scratch row [slot][step size], no second pass at
//    all -- the accept step copies the winner's row.  16x the stores, half the chain.
#include "models.h"
#include "quad_step.h"
#include "zm_common.h"

#include <cstdlib>

namespace zm {

constexpr int RN = 12, RM = 4;
constexpr int PSZ = RM + RM * RN + RN + RM;  // 68 doubles of policy data per step

struct FastArgs {
    const double* A;      // linear model (device) or nullptr
    const double* B;
    double dt;
    const double* Q;      // full (n,n) / (m,m) / (n,n) matrices
    const double* R;
    const double* Qf;
    const double* x0;
    const double* l;
    const double* L;
    const double* xPrev;
    const double* uPrev;
    const double* alphas;
    const int* active;
    const int* list;   // optional: trajectory ids to process (slot -> id), `count` of them; else slots are ids 0..batch-1
    long count;
    double* xTraj;
    double* uTraj;
    double* J;
    int* idx;
    long batch;
    int T;
    int n_alpha;       // 16, or 1: every lane of a trajectory's group rolls the same step size (lane 0 stores; no second pass)
    double* scratch;   // all-store mode (else nullptr): per slot (T+1) blocks of ALLSTORE_BLOCK doubles, see allstore_offset()
    int no_pass2;      // the winners are re-rolled by rollout_quad_reroll_kernel afterwards (needs idx)
};

// All-store scratch layout: the 16 lanes of a trajectory write NEIGHBOURING 16-byte pieces, so that one store instruction of a group
// covers 256 contiguous bytes (a row per step size instead -- 12.9 KB apart -- makes every store touch 64 different lines per wave
// and lengthens the chain by 40 %).  Block kb of a slot holds x_kb (pairs 0..5) and u_{kb-1} (pairs 6, 7; unused in block 0):
//     scratch[((slot * (T+1) + kb) * 8 + pair) * 32 + a * 2 + {0, 1}]          (8 pairs x 16 step sizes x 2 doubles per block)
constexpr int ALLSTORE_PAIRS = (RN + RM) / 2, ALLSTORE_BLOCK = ALLSTORE_PAIRS * 32;


template <int KIND>
__device__ __forceinline__ void fast_step(const double* As, const double* Bs, const double dt,
                                          const double (&x)[RN], const double (&u)[RM], double (&xn)[RN]) {
    if constexpr (KIND == ZM_MODEL_QUADCOPTER) {
        double sphi, cphi, sth, cth, spsi, cpsi;
        zm_sincos(x[6], &sphi, &cphi);
        zm_sincos(x[7], &sth, &cth);
        zm_sincos(x[8], &spsi, &cpsi);
        quad_euler_step_trig(x, u, dt, sphi, cphi, sth, cth, spsi, cpsi, xn);
    } else {
#pragma unroll
        for (int i = 0; i < RN; ++i) {
            double ax = x[0] * As[i * RN];
#pragma unroll
            for (int j = 1; j < RN; ++j) ax = ax + x[j] * As[i * RN + j];
            double bu = u[0] * Bs[i * RM];
#pragma unroll
            for (int j = 1; j < RM; ++j) bu = bu + u[j] * Bs[i * RM + j];
            xn[i] = ax + bu;
            __builtin_amdgcn_sched_barrier(0);  // one row of A, B in flight at a time (see quad_form)
        }
    }
}

template <bool DIAG, int K>
__device__ __forceinline__ double quad_form(const double* W, const double (&z)[K]) {
    double acc = 0.0;
    if constexpr (DIAG) {
#pragma unroll
        for (int j = 0; j < K; ++j) acc = __builtin_fma(z[j] * W[j * K + j], z[j], acc);   // (z^T W)_j = z_j W_jj
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) s = __builtin_fma(z[i], W[i * K + j], s);
            acc = __builtin_fma(s, z[j], acc);
            // keep the LDS reads of column j+1 behind this column's arithmetic: without the fence the scheduler
            // clusters all K*K table reads up front (hundreds of live VGPRs, spills)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

// cooperative fetch of one step's policy data of this lane's trajectory: element e = a + 16 i of [l|L|xPrev|uPrev]
struct PolPtrs {
    const double* p[5];
    int st[5];
};

// One step's policy data of a trajectory, spread over the 16 lanes that roll it out: element e of [l|L|xPrev|uPrev] sits in register
// v[e >> 4] of lane e & 15 -- exactly where the cooperative load puts it.  Nobody gathers it: every use is an FMA whose DPP source
// operand reads the owning lane (v_fmac_f64_dpp ... row_newbcast, the one DPP control 64-bit ALU instructions accept), so the data
// never goes through LDS (round 2 staged it there: 45 LDS instructions and a wave barrier per step, 38 % of the wave's cycles in
// s_waitcnt at two waves per SIMD -- profiles/r03_rollfast_pmc.txt).
struct PolRegs {
    double v[5];
};
// acc += c * (policy element E)
template <int E>
__device__ __forceinline__ void pol_fma(double& acc, const double c, const PolRegs& r) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(r.v[E >> 4]), "v"(c), "i"(E & 15));
}
// the two wait states a DPP read needs after a VALU write of its source (hipcc pads nothing inside asm): every step passes its five
// registers through this once, the broadcasts depend on the statement and hence follow it
__device__ __forceinline__ void pol_settle(PolRegs& r) {
#pragma unroll
    for (int i = 0; i < 5; ++i) asm("s_nop 1" : "+v"(r.v[i]));
}
// dx[j] = x[j] - xPrev_k[j]   as fma(-1, xPrev, x): one rounding of the same difference
template <int J = 0>
__device__ __forceinline__ void pol_dx(double (&dx)[RN], const double (&x)[RN], const double neg1, const PolRegs& r) {
    if constexpr (J < RN) {
        dx[J] = x[J];
        pol_fma<RM + RM * RN + J>(dx[J], neg1, r);
        pol_dx<J + 1>(dx, x, neg1, r);
    }
}
// s += sum_j L_k[I][j] dx[j], j ascending
template <int I, int J = 0>
__device__ __forceinline__ void pol_row(double& s, const double (&dx)[RN], const PolRegs& r) {
    if constexpr (J < RN) {
        pol_fma<RM + I * RN + J>(s, dx[J], r);
        pol_row<I, J + 1>(s, dx, r);
    }
}
// u = (alpha * l_k + L_k (x - xPrev_k)) + uPrev_k            (pytrees.py:220, ilqrUtils.py:59-60)
template <int I = 0>
__device__ __forceinline__ void pol_controls(double (&u)[RM], const double (&dx)[RN], const double al, const double one, const PolRegs& r) {
    if constexpr (I < RM) {
        double s = 0.0;
        pol_row<I>(s, dx, r);
        pol_fma<I>(s, al, r);                           // fma(l_k[I], alpha, s)
        pol_fma<RM + RM * RN + RN + I>(s, one, r);      // + uPrev_k[I]
        u[I] = s;
        pol_controls<I + 1>(u, dx, al, one, r);
    }
}

template <int KIND, bool DIAG, bool ALL>
// (no __restrict__ on the LDS tables: with it the loop-invariant LDS reads of Q, A, B ... are hoisted into registers --
//  several hundred VGPRs -- instead of being re-read by broadcast every step)
__device__ __forceinline__ void rollout_ls_body(const FastArgs& g, const double* Qs, const double* Rs, const double* Qfs,
                                                const double* As, const double* Bs) {
    const int lane = threadIdx.x;
    const int grp = lane >> 4, a = lane & 15;
    const long slot = (long)blockIdx.x * 4 + grp;
    const long nslot = g.list ? g.count : g.batch;
    const long traj = (slot < nslot) ? (g.list ? (long)g.list[slot] : slot) : 0;
    const bool live = (slot < nslot) && (g.active == nullptr || g.active[traj] != 0);
    const long t = live ? traj : 0;
    const int T = g.T;
    const double alpha = g.alphas[g.n_alpha == 1 ? 0 : a];

    PolPtrs pp;
    {
        const double* lt = g.l + t * T * RM;
        const double* Lt = g.L + t * T * RM * RN;
        const double* xpt = g.xPrev + t * (T + 1) * RN;
        const double* upt = g.uPrev + t * T * RM;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int e = a + 16 * i;
            if (e < RM) {
                pp.p[i] = lt + e;
                pp.st[i] = RM;
            } else if (e < RM + RM * RN) {
                pp.p[i] = Lt + (e - RM);
                pp.st[i] = RM * RN;
            } else if (e < RM + RM * RN + RN) {
                pp.p[i] = xpt + (e - RM - RM * RN);
                pp.st[i] = RN;
            } else if (e < PSZ) {
                pp.p[i] = upt + (e - RM - RM * RN - RN);
                pp.st[i] = RM;
            } else {  // lanes a >= 4 of the 5th load: re-read element 0 of uPrev_k into the buffer's padding
                pp.p[i] = upt;
                pp.st[i] = RM;
            }
        }
    }
    const double* x0t = g.x0 + t * RN;
    double* xo = g.xTraj + t * (T + 1) * RN;
    double* uo = g.uTraj + t * T * RM;
    double* so = ALL ? g.scratch + (live ? slot : 0) * (long)(T + 1) * ALLSTORE_BLOCK + a * 2 : nullptr;

    // one rollout of step size `al`; STORE: this lane writes xTraj / uTraj
    auto rollout = [&](const double al, const bool store) -> double {
        double x[RN], u[RM], xn[RN];
#pragma unroll
        for (int i = 0; i < RN; ++i) x[i] = x0t[i];
        if (store) {
            if constexpr (ALL) {
#pragma unroll
                for (int i = 0; i < RN; ++i) so[(i >> 1) * 32 + (i & 1)] = x[i];
            } else {
#pragma unroll
                for (int i = 0; i < RN; ++i) xo[i] = x[i];
            }
        }
        double J = 0.0;
        double qd[(ALL && DIAG) ? RN : 1], rd[(ALL && DIAG) ? RM : 1];
        if constexpr (ALL && DIAG) {
#pragma unroll
            for (int j = 0; j < RN; ++j) qd[j] = Qs[j * RN + j];
#pragma unroll
            for (int j = 0; j < RM; ++j) rd[j] = Rs[j * RM + j];
        }
        double one = 1.0, neg1 = -1.0;
        asm("" : "+v"(one), "+v"(neg1));   // in registers: VOP2 DPP takes no literal
        // the policy data of a step arrives three steps ahead, in four rotating register sets (five registers each)
        auto fetch = [&](PolRegs& r, const int k) {
#pragma unroll
            for (int i = 0; i < 5; ++i) r.v[i] = pp.p[i][(long)k * pp.st[i]];
        };
        auto body = [&](PolRegs& cur, PolRegs& ahead, const int k) {
            if (k + 3 < T) fetch(ahead, k + 3);
            pol_settle(cur);
            double dx[RN];
            pol_dx(dx, x, neg1, cur);
            pol_controls(u, dx, al, one, cur);
            if constexpr (ALL && DIAG) {   // the diagonal weights from registers (loaded before the loop); quad_form's arithmetic
                double jx = 0.0, ju = 0.0;
#pragma unroll
                for (int j = 0; j < RN; ++j) jx = __builtin_fma(x[j] * qd[j], x[j], jx);
#pragma unroll
                for (int j = 0; j < RM; ++j) ju = __builtin_fma(u[j] * rd[j], u[j], ju);
                J += jx + ju;
            } else {
                J += quad_form<DIAG, RN>(Qs, x) + quad_form<DIAG, RM>(Rs, u);
            }
            fast_step<KIND>(As, Bs, g.dt, x, u, xn);
#pragma unroll
            for (int i = 0; i < RN; ++i) x[i] = xn[i];
            if (store) {
                if constexpr (ALL) {
                    double* sb = so + (long)(k + 1) * ALLSTORE_BLOCK;
#pragma unroll
                    for (int i = 0; i < RN; ++i) sb[(i >> 1) * 32 + (i & 1)] = x[i];
#pragma unroll
                    for (int i = 0; i < RM; ++i) sb[(RN / 2 + (i >> 1)) * 32 + (i & 1)] = u[i];
                } else {
#pragma unroll
                    for (int i = 0; i < RM; ++i) uo[(long)k * RM + i] = u[i];
#pragma unroll
                    for (int i = 0; i < RN; ++i) xo[(long)(k + 1) * RN + i] = x[i];
                }
            }
        };
        PolRegs pa, pb, pc, pd;
        fetch(pa, 0);
        if (T > 1) fetch(pb, 1);
        if (T > 2) fetch(pc, 2);
        for (int k = 0; k < T; k += 4) {   // rotating sets: no copies
            body(pa, pd, k);
            if (k + 1 < T) body(pb, pa, k + 1);
            if (k + 2 < T) body(pc, pb, k + 2);
            if (k + 3 < T) body(pd, pc, k + 3);
        }
        J += quad_form<DIAG, RN>(Qfs, x);
        return J;
    };

    // pass 0: every lane its own step size; lane a == 0 (alpha_0) stores speculatively.
    // pass 1 (only where another step size than alpha_0 won): every lane of the group re-rolls the winner, lane 0 stores.
    double al = alpha;
    bool store = live && (a == 0 || ALL);
    for (int pass = 0; pass < 2; ++pass) {
        const double J = rollout(al, store);
        if (pass == 1) break;
        double key = live ? J : __builtin_inf();
        int isn = (live && (J != J)) ? 1 : 0;
        int who = a;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const double ok = __shfl_xor(key, off);
            const int on = __shfl_xor(isn, off);
            const int ow = __shfl_xor(who, off);
            const bool better = (on > isn) || (on == isn && ((on == 0 && ok < key) || ((on == 1 || ok == key) && ow < who)));
            key = better ? ok : key;
            isn = better ? on : isn;
            who = better ? ow : who;
        }
        const int best = who;
        const double Jbest = __shfl(J, (grp << 4) + best);
        if (live && a == 0) {
            if (g.J) g.J[t] = Jbest;
            if (g.idx) g.idx[t] = best;
        }
        const bool need2 = live && best != 0 && !ALL && !g.no_pass2;
        if (__ballot(need2) == 0ull) break;
        al = g.alphas[best];
        store = need2 && a == 0;
    }
}

// DIAG_ONLY: the caller asserted diagonal weights (zm_quadcost_t.diagonal): only the diagonal path is compiled in -- 158 VGPRs, three
// waves per SIMD.  Otherwise the kernel decides per launch; it then also carries the general path, whose hoisted weight matrices
// cost it 458 registers (one wave per SIMD) on either branch.
template <int KIND, bool DIAG_ONLY, bool ALL>
__global__ __launch_bounds__(64) void rollout_ls_fast_kernel(const FastArgs g) {
    __shared__ double Qs[RN * RN], Rs[RM * RM], Qfs[RN * RN];
    __shared__ double As[KIND == ZM_MODEL_LINEAR ? RN * RN : 1], Bs[KIND == ZM_MODEL_LINEAR ? RN * RM : 1];
    const int lane = threadIdx.x;
    bool offdiag = false;
    for (int e = lane; e < RN * RN; e += 64) {
        const double q = g.Q[e], qf = g.Qf[e];
        Qs[e] = q;
        Qfs[e] = qf;
        offdiag |= (e / RN != e % RN) && (q != 0.0 || qf != 0.0);
    }
    for (int e = lane; e < RM * RM; e += 64) {
        const double r = g.R[e];
        Rs[e] = r;
        offdiag |= (e / RM != e % RM) && (r != 0.0);
    }
    if constexpr (KIND == ZM_MODEL_LINEAR) {
        for (int e = lane; e < RN * RN; e += 64) As[e] = g.A[e];
        for (int e = lane; e < RN * RM; e += 64) Bs[e] = g.B[e];
    }
    __syncthreads();
    if constexpr (DIAG_ONLY) {
        rollout_ls_body<KIND, true, ALL>(g, Qs, Rs, Qfs, As, Bs);
    } else {
        // diagonal weights (the demos' Q = I, R = I, Qf = 10 I): x^T W x costs n FMAs instead of n^2; wave-uniform
        if (__ballot(offdiag) == 0ull)
            rollout_ls_body<KIND, true, ALL>(g, Qs, Rs, Qfs, As, Bs);
        else
            rollout_ls_body<KIND, false, ALL>(g, Qs, Rs, Qfs, As, Bs);
    }
}

template <int KIND>
static int launch_fast(const FastArgs& g, const bool diag_only, hipStream_t st) {
    const long nslot = g.list ? g.count : g.batch;
    const dim3 grid((unsigned)((nslot + 3) / 4)), block(64);
    if (g.scratch) {
        if (diag_only)
            hipLaunchKernelGGL((rollout_ls_fast_kernel<KIND, true, true>), grid, block, 0, st, g);
        else
            hipLaunchKernelGGL((rollout_ls_fast_kernel<KIND, false, true>), grid, block, 0, st, g);
    } else if (diag_only) {
        hipLaunchKernelGGL((rollout_ls_fast_kernel<KIND, true, false>), grid, block, 0, st, g);
    } else {
        hipLaunchKernelGGL((rollout_ls_fast_kernel<KIND, false, false>), grid, block, 0, st, g);
    }
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// rollout_quad.hip: four lanes per rollout (quadcopter, still air, diagonal weights)
struct QuadArgs {
    double dt;
    const double *Q, *R, *Qf, *x0, *l, *L, *xPrev, *uPrev, *alphas;
    const int *active, *list;
    long count;
    double *xTraj, *uTraj, *J;
    int* idx;
    long batch;
    int T;
    double* scratch;
};
int rollout_quad_all(const QuadArgs& g, hipStream_t st);
int rollout_quad_reroll(const QuadArgs& g, hipStream_t st);

// Fast path dispatch: (n, m) = (12, 4), 16 step sizes.
int rollout_fast_dispatch(const zm_model_t& md, const double* Q, const double* R, const double* Qf, int diagonal,
                          const double* x0, const double* l, const double* L, const double* xPrev, const double* uPrev,
                          const double* alphas, int n_alpha, const int* active, const int* list, int64_t count, double* xTraj,
                          double* uTraj, double* J, int* idx, int64_t batch, int T, hipStream_t st, double* scratch) {
    FastArgs g{md.A, md.B, md.dt, Q, R, Qf, x0, l, L, xPrev, uPrev, alphas, active, list, (long)count, xTraj, uTraj, J, idx,
               (long)batch, T, n_alpha, scratch, 0};
    // ZOPT_AMD_ROLLOUT_QUAD=0: everything in rollout_ls_fast_kernel (A/B; same results)
    static const bool quad_on = [] {
        const char* e = zm::lab_env("ZOPT_AMD_ROLLOUT_QUAD");
        return !(e && e[0] == '0');
    }();
    const bool quad = quad_on && md.kind == ZM_MODEL_QUADCOPTER && diagonal == 1 && n_alpha == 16 && J && idx;
    const QuadArgs qa{md.dt, Q, R, Qf, x0, l, L, xPrev, uPrev, alphas, active, list, (long)count, xTraj, uTraj, J, idx, (long)batch, T, scratch};
    // all-store line search with at most one wave per SIMD: one trajectory per wave, 4 lanes per step size (78 against 104 us per
    // launch; four-wave workgroups so that up to 1024 waves each get a SIMD).  With more trajectories its 4x as many waves no longer
    // run alone and the 4-trajectories-per-wave kernel is faster (measured: 1131 trajectories 145 against ~125 us; iLQR solve with the
    // switch-over at 640 / 800 / 1024 / 1280 trajectories: 42.65 / 42.4 / 42.3 / 43.0 ms).
    static const long quad_all_max = [] {   // ZOPT_AMD_QUAD_ALL_MAX: A/B of the switch-over (same results either side)
        const char* e = zm::lab_env("ZOPT_AMD_QUAD_ALL_MAX");
        return e ? atol(e) : 1024L;
    }();
    if (quad && scratch && count <= quad_all_max) return rollout_quad_all(qa, st);
    if (quad && !scratch) g.no_pass2 = 1;                              // two-pass line search: the second pass as its own, densely packed launch
    int rc;
    if (md.kind == ZM_MODEL_QUADCOPTER) rc = launch_fast<ZM_MODEL_QUADCOPTER>(g, diagonal == 1, st);
    else rc = launch_fast<ZM_MODEL_LINEAR>(g, diagonal == 1, st);
    if (rc || !quad || scratch) return rc;
    return rollout_quad_reroll(qa, st);
}

}  // namespace zm
