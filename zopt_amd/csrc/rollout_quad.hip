// K6-Q  the quadcopter's rollouts on FOUR lanes per (trajectory, step size) -- gfx950.
//
// Same arithmetic, bit for bit, as rollout_fast.hip (reference ilqrUtils.py:33-66, 116-150; pytrees.py:49-52, 215-220); what changes
// is who computes it.  A rollout is a chain of T dependent steps of ~310 VALU instructions per lane, and a lone wave issues dependent fp64
// instructions only every ~9.5 cycles (independent ones every 4.7): when few waves are in flight, the launch lasts as long as that chain.  Here the lanes of a quad
// share one rollout: lane q computes row q of the policy product L_k (x - xPrev_k) and the sine / cosine of one Euler angle, and
// loads only what it needs (row q of L_k, xPrev_k, l_k[q], uPrev_k[q]: 26 doubles instead of 68) straight from global memory, two
// steps ahead, in registers; controls and trigonometric values are exchanged by DPP quad broadcasts.  Dynamics, cost and state update
// stay replicated (their formulas differ per component: four lanes would diverge, not share).  ~200 instructions per step.
//
//  rollout_quad_all_kernel     the solvers' all-store line search (few trajectories left): one trajectory per wave, 16 step sizes x
//                              4 lanes, every rollout stored to the scratch blocks of rollout_fast.hip, argmin at the end
//  rollout_quad_reroll_kernel  the second pass of the two-pass line search as its own launch: 16 trajectories per wave, each
//                              re-rolled with its winning step size unless that is alpha_0 (whose rollout pass 1 stored already)
#include "models.h"
#include "quad_step.h"
#include "zm_common.h"

namespace zm {

constexpr int QN = 12, QM = 4;
constexpr int Q_ALLSTORE_BLOCK = ((QN + QM) / 2) * 32;   // rollout_fast.hip: ALLSTORE_BLOCK

struct QuadArgs {   // (declared identically in rollout_fast.hip, which decides when these kernels apply)
    double dt;
    const double *Q, *R, *Qf, *x0, *l, *L, *xPrev, *uPrev, *alphas;   // Q, R, Qf: full matrices, asserted diagonal by the caller
    const int *active, *list;   // list: trajectory ids (slot -> id), `count` of them; nullptr: slots are ids 0..batch-1
    long count;
    double *xTraj, *uTraj, *J;  // reroll: destination rows
    int* idx;                   // all: winner index out; reroll: winner index in
    long batch;
    int T;
    double* scratch;            // all: the all-store blocks
};

// value of lane I of this lane's quad
template <int I>
__device__ __forceinline__ double quad_bcast(const double v) {
    constexpr int CTRL = I | (I << 2) | (I << 4) | (I << 6);   // quad_perm:[I,I,I,I]
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// a step's operands of one lane
struct QuadOps {
    double Lq[QN], xp[QN], lq, upq;
};

__device__ __forceinline__ void quad_load(QuadOps& o, const double* Lr, const double* xp, const double* lq, const double* upq, const long k) {
#pragma unroll
    for (int j = 0; j < QN; ++j) o.Lq[j] = Lr[k * (QM * QN) + j];
#pragma unroll
    for (int j = 0; j < QN; ++j) o.xp[j] = xp[k * QN + j];
    o.lq = lq[k * QM];
    o.upq = upq[k * QM];
}

// one step in two halves.  First half: u = (alpha l_k + L_k (x - xPrev_k)) + uPrev_k  (pytrees.py:220, ilqrUtils.py:59-60) -- the only
// part that reads the step's operand registers, which the caller then refills for the step after next while the second half runs.
__device__ __forceinline__ void quad_policy(const QuadOps& o, const double al, const double (&x)[QN], double (&u)[QM]) {
    double dx[QN];
#pragma unroll
    for (int j = 0; j < QN; ++j) dx[j] = x[j] - o.xp[j];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < QN; ++j) s = __builtin_fma(o.Lq[j], dx[j], s);
    const double uq = __builtin_fma(al, o.lq, s) + o.upq;   // (explicit: the same contraction as rollout_fast.hip's DPP form)
    u[0] = quad_bcast<0>(uq);
    u[1] = quad_bcast<1>(uq);
    u[2] = quad_bcast<2>(uq);
    u[3] = quad_bcast<3>(uq);
}
// second half: cost (rollout_fast.hip's diagonal quad_form, in its order) ...
__device__ __forceinline__ void quad_cost(const double (&qd)[QN], const double (&rd)[QM], const double (&x)[QN], const double (&u)[QM],
                                          double& J) {
    double jx = 0.0, ju = 0.0;
#pragma unroll
    for (int j = 0; j < QN; ++j) jx = __builtin_fma(x[j] * qd[j], x[j], jx);
#pragma unroll
    for (int j = 0; j < QM; ++j) ju = __builtin_fma(u[j] * rd[j], u[j], ju);
    J += jx + ju;
}
// ... and x <- f(x, u)
__device__ __forceinline__ void quad_advance(const int q, const double dt, double (&x)[QN], const double (&u)[QM]) {
    // lane q < 3: sine and cosine of Euler angle q (lane 3 repeats psi); everyone receives all three pairs
    // (two selects kept apart: written as one expression the compiler indexes x[6 + min(q, 2)] dynamically and moves x to scratch)
    double ang = (q == 1) ? x[7] : x[8];
    asm("" : "+v"(ang));
    ang = (q == 0) ? x[6] : ang;
    double sq, cq;
    zm_sincos(ang, &sq, &cq);
    const double sphi = quad_bcast<0>(sq), cphi = quad_bcast<0>(cq);
    const double sth = quad_bcast<1>(sq), cth = quad_bcast<1>(cq);
    const double spsi = quad_bcast<2>(sq), cpsi = quad_bcast<2>(cq);
    double xn[QN];
    quad_euler_step_trig(x, u, dt, sphi, cphi, sth, cth, spsi, cpsi, xn);
#pragma unroll
    for (int i = 0; i < QN; ++i) x[i] = xn[i];
}

// Four-wave workgroups, one trajectory per wave and no barrier: a workgroup's waves go to the four SIMDs of a CU, whereas single-wave
// workgroups beyond three per CU start sharing a SIMD while another is free.
__global__ __launch_bounds__(256) void rollout_quad_all_kernel(const QuadArgs g) {
    const int lane = threadIdx.x & 63, q = lane & 3, a = lane >> 2;
    const long slot = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= g.count) return;
    const long t = g.list ? (long)g.list[slot] : slot;
    if (g.active && g.active[t] == 0) return;   // whole wave: one trajectory
    const int T = g.T;
    const double al = g.alphas[a];
    const double* Lr = g.L + (t * T * QM + q) * QN;
    const double* xp = g.xPrev + t * (T + 1) * QN;
    const double* lq = g.l + t * T * QM + q;
    const double* upq = g.uPrev + t * T * QM + q;
    double qd[QN], rd[QM], qf[QN];
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        qd[j] = g.Q[j * QN + j];
        qf[j] = g.Qf[j * QN + j];
    }
#pragma unroll
    for (int j = 0; j < QM; ++j) rd[j] = g.R[j * QM + j];
    double x[QN], u[QM];
#pragma unroll
    for (int i = 0; i < QN; ++i) x[i] = g.x0[t * QN + i];
    double* so = g.scratch + slot * (long)(T + 1) * Q_ALLSTORE_BLOCK + a * 2;
    const bool st = (q == 0);
    if (st) {
#pragma unroll
        for (int i = 0; i < QN; ++i) so[(i >> 1) * 32 + (i & 1)] = x[i];
    }
    double J = 0.0;
    // Two operand sets, each refilled for the step after next as soon as its step's policy half has read it -- i.e. while the rest of
    // that step (cost, sincos, dynamics: three quarters of it) and the whole next step run: ~1.75 steps of prefetch distance (with ~1000
    // trajectories in flight the operands come from HBM, and one step of this chain, ~0.8 us, is shorter than that round trip under
    // load) at the register cost of one step ahead (three rotating sets spilt 58 registers to AGPRs: 34 moves per step).
    QuadOps oa, ob;
    quad_load(oa, Lr, xp, lq, upq, 0);
    if (T > 1) quad_load(ob, Lr, xp, lq, upq, 1);
    auto body = [&](QuadOps& cur, const int k) {
        quad_policy(cur, al, x, u);
        if (k + 2 < T) quad_load(cur, Lr, xp, lq, upq, k + 2);
        quad_cost(qd, rd, x, u, J);
        quad_advance(q, g.dt, x, u);
        if (st) {
            double* sb = so + (long)(k + 1) * Q_ALLSTORE_BLOCK;
#pragma unroll
            for (int i = 0; i < QN; ++i) sb[(i >> 1) * 32 + (i & 1)] = x[i];
#pragma unroll
            for (int i = 0; i < QM; ++i) sb[(QN / 2 + (i >> 1)) * 32 + (i & 1)] = u[i];
        }
    };
    int k = 0;
    for (; k + 1 < T; k += 2) {   // ping-pong operand registers: no copies
        body(oa, k);
        body(ob, k + 1);
    }
    if (k < T) body(oa, k);
    {   // terminal cost, rollout_fast.hip's diagonal quad_form
        double jf = 0.0;
#pragma unroll
        for (int j = 0; j < QN; ++j) jf = __builtin_fma(x[j] * qf[j], x[j], jf);
        J += jf;
    }
    // argmin over the 16 step sizes with jnp.argmin semantics (NaN wins, first index on ties); the 4 lanes of a quad agree on J
    double key = J;
    int isn = (J != J) ? 1 : 0;
    int who = a;
#pragma unroll
    for (int off = 32; off >= 4; off >>= 1) {
        const double ok = __shfl_xor(key, off);
        const int on = __shfl_xor(isn, off);
        const int ow = __shfl_xor(who, off);
        const bool better = (on > isn) || (on == isn && ((on == 0 && ok < key) || ((on == 1 || ok == key) && ow < who)));
        key = better ? ok : key;
        isn = better ? on : isn;
        who = better ? ow : who;
    }
    const double Jbest = __shfl(J, who * 4);
    if (lane == 0) {
        g.J[t] = Jbest;
        g.idx[t] = who;
    }
}

__global__ __launch_bounds__(64) void rollout_quad_reroll_kernel(const QuadArgs g) {
    const int lane = threadIdx.x, q = lane & 3, s = lane >> 2;
    const long slot = (long)blockIdx.x * 16 + s;
    const long nslot = g.list ? g.count : g.batch;
    const long traj = (slot < nslot) ? (g.list ? (long)g.list[slot] : slot) : 0;
    const bool live = (slot < nslot) && (g.active == nullptr || g.active[traj] != 0);
    const int best = live ? g.idx[traj] : 0;
    const bool need = live && best != 0;       // alpha_0 won: pass 1 has stored that rollout
    if (__ballot(need) == 0ull) return;
    const long t = need ? traj : 0;
    const int T = g.T;
    const double al = g.alphas[best];
    const double* Lr = g.L + (t * T * QM + q) * QN;
    const double* xp = g.xPrev + t * (T + 1) * QN;
    const double* lq = g.l + t * T * QM + q;
    const double* upq = g.uPrev + t * T * QM + q;
    double x[QN], u[QM];
#pragma unroll
    for (int i = 0; i < QN; ++i) x[i] = g.x0[t * QN + i];
    double* xo = g.xTraj + t * (T + 1) * QN;
    double* uo = g.uTraj + t * T * QM;
    const bool st = need && (q == 0);
    if (st) {
#pragma unroll
        for (int i = 0; i < QN; ++i) xo[i] = x[i];
    }
    QuadOps oa, ob;   // refilled right after the policy half of their step, as in rollout_quad_all_kernel
    quad_load(oa, Lr, xp, lq, upq, 0);
    if (T > 1) quad_load(ob, Lr, xp, lq, upq, 1);
    auto body = [&](QuadOps& cur, const int k) {
        quad_policy(cur, al, x, u);
        if (k + 2 < T) quad_load(cur, Lr, xp, lq, upq, k + 2);
        quad_advance(q, g.dt, x, u);
        if (st) {
#pragma unroll
            for (int i = 0; i < QM; ++i) uo[(long)k * QM + i] = u[i];
#pragma unroll
            for (int i = 0; i < QN; ++i) xo[(long)(k + 1) * QN + i] = x[i];
        }
    };
    int k = 0;
    for (; k + 1 < T; k += 2) {
        body(oa, k);
        body(ob, k + 1);
    }
    if (k < T) body(oa, k);
}

// rollout_fast.hip decides when these apply (quadcopter in still air, diagonal weights asserted, 16 step sizes)
int rollout_quad_all(const QuadArgs& g, hipStream_t st) {
    hipLaunchKernelGGL(rollout_quad_all_kernel, dim3((unsigned)((g.count + 3) / 4)), dim3(256), 0, st, g);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
int rollout_quad_reroll(const QuadArgs& g, hipStream_t st) {
    const long nslot = g.list ? g.count : g.batch;
    hipLaunchKernelGGL(rollout_quad_reroll_kernel, dim3((unsigned)((nslot + 15) / 16)), dim3(64), 0, st, g);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

}  // namespace zm
