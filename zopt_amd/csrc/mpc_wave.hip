// K9-W  mpc_solve_wave -- the ADMM of mpc.hip with 16 LANES PER INSTANCE (4 instances per wave64), iterates in LDS.
//
// Same algorithm, same termination and certificate as mpc_solve_kernel (mpc.hip; problem: zopt/mpcUtils.py:48-59): only
// the mapping differs.  One lane per instance leaves an MI355X nearly idle at the batch sizes MPC is used with (1024
// instances = 16 waves) and makes every ADMM iteration a chain of 60 x ~500 dependent FMAs in one lane.  Here the 16 lanes of a
// group carry the STACKED index [x ; u]: lane i < n holds state component i, lane n + j control component j, each with its row / column
// of A, B, K_k, Suu_k^-1; a mat-vec is NL broadcasts (DPP row_newbcast, no LDS) + NL FMAs per lane, and it serves both blocks at once:
//     backward stage:  p = p' - rho z_x;  [A^T p ; Qu = -rho z_u + B^T p] in one pass over p;  [K^T Qu ; kf] in one pass over Qu
//     forward  stage:  [A x ; K x] in one pass over x;  + B u;  clip / dual update once, on the stacked iterate
// ~75 vector instructions per stage pair.  The iterates y, lam, kf (and the certificate's r) live in LDS, one private
// slot per lane and stage: no cross-lane LDS traffic, hence no barrier anywhere in the loop; HBM is touched only for the
// tables K_k, Suu_k^-1 (L2-resident, fetched three stages ahead) and at entry / exit (warm start, results).
#include "mpc_common.h"

#include <hip/hip_runtime.h>

namespace zm {

// acc += c * (value of lane L of this lane's 16-lane row of v): ONE instruction.  gfx90a+ accept a DPP source operand on
// 64-bit ALU instructions for exactly one control, row_newbcast -- the one a mat-vec needs -- so the broadcast costs no
// instruction of its own (two v_mov_b32_dpp + the FMA before: a third of the kernel's vector instructions were broadcasts).
// hipcc pads nothing inside asm: the two wait states a DPP read needs after a VALU write of its source come from dpp_src(),
// which every mat-vec passes its vector through once (the broadcasts then depend on that statement, hence follow it).
template <int L>
__device__ __forceinline__ void fma_bc(double& acc, const double c, const double v) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(c), "i"(L));
}
__device__ __forceinline__ double dpp_src(double v) {
    asm("s_nop 1" : "+v"(v));
    return v;
}
// a1 += sum_l c1[l] v_l,  a2 += sum_l c2[l] v_l   over the first NL lanes of the row.  The sums run as PS interleaved partial
// chains (term l goes to chain l % PS): a single chain of 12 dependent FMAs is the longest dependency of a stage, and the
// solve is the latency of 60 such stages per ADMM iteration.
// acc += sum_l c[l] v_(OFF + l): the vector's components sit in lanes OFF .. OFF + NL - 1 of the row.  The sums run as PS interleaved
// partial chains (term l goes to chain l % PS; chain 0 starts from acc): a single chain of 12 dependent FMAs would be the longest
// dependency of a stage.
template <int NL, int PS, int OFF, int L = 0>
__device__ __forceinline__ void mv_acc(const double (&c)[NL], const double v, double (&s)[PS]) {
    if constexpr (L < NL) {
        fma_bc<OFF + L>(s[L % PS], c[L], v);
        mv_acc<NL, PS, OFF, L + 1>(c, v, s);
    }
}
template <int NL, int OFF = 0>
__device__ __forceinline__ void mv(const double (&c)[NL], const double v, double& a) {
    constexpr int PS = NL >= 9 ? 3 : (NL >= 4 ? 2 : 1);
    double s[PS];
    s[0] = a;
#pragma unroll
    for (int i = 1; i < PS; ++i) s[i] = 0.0;
    mv_acc<NL, PS, OFF>(c, dpp_src(v), s);
    if constexpr (PS == 3)
        a = (s[0] + s[1]) + s[2];
    else if constexpr (PS == 2)
        a = s[0] + s[1];
    else
        a = s[0];
}
// the same as ONE chain (the short products onto an accumulator that is itself the end of a chain)
template <int NL, int OFF, int L = 0>
__device__ __forceinline__ void mv_seq_acc(const double (&c)[NL], const double v, double& a) {
    if constexpr (L < NL) {
        fma_bc<OFF + L>(a, c[L], v);
        mv_seq_acc<NL, OFF, L + 1>(c, v, a);
    }
}
template <int NL, int OFF>
__device__ __forceinline__ void mv_seq(const double (&c)[NL], const double v, double& a) {
    mv_seq_acc<NL, OFF>(c, dpp_src(v), a);
}
// acc = max(acc, |v|) in ONE instruction.  __builtin_fmax(acc, __builtin_fabs(v)) compiles to three (both operands are first passed
// through a canonicalising v_max with themselves, which only matters for signalling NaNs); five norms per forward stage made that a sixth
// of the stage.  Same result for every input the solve can produce (a quiet NaN is ignored by both forms).
__device__ __forceinline__ void amax(double& acc, const double v) {
    asm("v_max_f64 %0, %0, |%1|" : "+v"(acc) : "v"(v));
}
// (DPP butterflies: the __shfl_xor ladders they replace were eight ds_bpermute round trips per reduction, five reductions per iteration)
__device__ __forceinline__ double row_max(double v) { return row16_max(v); }
__device__ __forceinline__ double row_sum(double v) { return row16_sum_from8(v); }

// LDS per group and stage (doubles): y[16] lam[16] r[16] kf[16], one slot per lane
constexpr int WS_STAGE = 64;

// Lane roles (round 3): the STACKED index [x ; u] on the 16 lanes of a group -- lane i < NS owns state component i, lane NS + j owns
// control component j (NS + MC <= 16 for every compiled shape).  A mat-vec over the state lanes then serves both blocks of its result in
// ONE FMA per broadcast, each lane with its own coefficients ([A^T p ; B^T p], [A x ; K x]), and the projection / dual update runs once
// for the stacked iterate -- rounds 1-2 kept the control components in lanes 0 .. MC-1 next to the states, which cost two FMAs per
// broadcast and a second, four-lane copy of the update (~120 vector instructions per stage pair instead of ~75).  Same sums in the same
// order: the iterates are bit for bit those of the older mapping.
template <int NS, int MC>
__global__ __launch_bounds__(64) void mpc_solve_wave_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                            const double* __restrict__ Ktab, const double* __restrict__ Mtab,
                                                            const double* __restrict__ x_lb, const double* __restrict__ x_ub,
                                                            const double* __restrict__ u_lb, const double* __restrict__ u_ub,
                                                            const MpcArgs g) {
    static_assert(NS + MC <= 16, "the stacked index must fit the 16 lanes of a group");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int W = NS + MC;
    const int lane = threadIdx.x, grp = lane >> 4, li = lane & 15;
    const long inst_raw = (long)blockIdx.x * 4 + grp;
    const bool live = inst_raw < g.batch;          // uniform over the 16-lane group
    const long inst = live ? inst_raw : g.batch - 1;   // idle groups shadow the last instance and never store
    const int N = g.N;
    double rho = g.rho;        // penalty of this group (changes with adaptive levels)
    int lvl = g.level0;
    const bool sx = li < NS, su = (li >= NS) && (li < W), sw = li < W;   // this lane owns a state / a control / any component
    const int ix = sx ? li : 0, iu = su ? li - NS : 0, iw = sw ? li : 0;
    double* base = lds + (long)grp * N * WS_STAGE;
    double* yw = base + li;                        // + k * WS_STAGE
    double* lw = base + 16 + li;
    double* rw = base + 32 + li;
    double* kf = base + 48 + li;                   // (control lanes)

    // the lane's column of [A | B] (the adjoint products over the state lanes), its row of A (state lanes; the control lanes' row of
    // [A ; K_k] comes from the stage's table) and its row of B (zero outside the state lanes)
    double ABcol[NS], Arow[NS], Brow[MC];
#pragma unroll
    for (int l = 0; l < NS; ++l) {
        const double ac = A[l * NS + ix], bc = B[l * MC + iu], ar = A[ix * NS + l];
        ABcol[l] = sx ? ac : (su ? bc : 0.0);
        Arow[l] = sx ? ar : 0.0;
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
        const double br = B[ix * MC + j];
        Brow[j] = sx ? br : 0.0;
    }
    const double inf = __builtin_inf();
    const double lo = sx ? x_lb[ix] : (su ? u_lb[iu] : -inf), hi = sx ? x_ub[ix] : (su ? u_ub[iu] : inf);
    const double x0 = sx ? g.x0[inst * NS + ix] : 0.0;
    // x_0 = x0 is box-constrained too (mpcUtils.py:56,58): group-wide AND over the state lanes
    const double viol = (sx && !(x0 >= lo && x0 <= hi)) ? 1.0 : 0.0;
    const bool x0_in = row_max(viol) == 0.0;

    // per-instance block of the caller's workspace: [y (N,W) | lam (N,W) | kf (N,MC), ok flag, spare | unused]
    double* wsi = g.ws + inst * (4L * N * W);
    double* okflag = wsi + 2L * N * W + (long)N * MC;
    const bool warm = g.warm && (*okflag == 1.0);
    if (warm && g.n_levels > 1) {   // the stored lam is scaled by the penalty the previous solve ended with
        const int l = (int)okflag[1];
        if (l >= 0 && l < g.n_levels) {
            lvl = l;
            rho = g.rho * pow(g.rho_step, (double)(lvl - g.level0));
        }
    }
    for (int k = 0; k < N; ++k) {
        const int ks = (g.warm == 2 && k + 1 < N) ? k + 1 : k;    // shifted warm start: iterate k <- iterate k+1
        const double wy = warm ? wsi[(long)ks * W + iw] : 0.0, wl = warm ? wsi[(long)N * W + (long)ks * W + iw] : 0.0;
        yw[k * WS_STAGE] = sw ? wy : 0.0;
        lw[k * WS_STAGE] = sw ? wl : 0.0;
        rw[k * WS_STAGE] = 0.0;
        kf[k * WS_STAGE] = 0.0;
    }

    // Table slices of one stage, by lane role, and the lane's own iterates of that stage from LDS (y, lam, kf: stage-local, so reading
    // them stages ahead of their use is safe in both sweeps: no LDS round trip at the head of a stage's dependency chain):
    //     fwd[l]  the lane's row of [A ; K_k]          (state lanes: row of A, kept; control lanes: row of K_k, loaded under their mask)
    //     adj[j]  the lane's row of [K_k^T ; Suu_k^-1]  (state lanes: column of K_k; control lanes: row of Suu_k^-1)
    // No masking: a lane outside every role (shapes with NS + MC < 16) loads the finite entries of row / column 0 and computes finite
    // values nobody reads -- only lanes < NS of p / x and lanes NS .. W-1 of qu / u are ever broadcast, only lanes < W are stored or
    // enter a norm.  Stage indices are clamped into [0, N).
    struct Tab {
        double fwd[NS], adj[MC];
        double y, lam, kf;
    };
    // Table addresses: a per-lane base (role and penalty level: set at the top of every ADMM iteration) plus stage index x a per-lane
    // stage stride -- one v_mad per table and stage; the elements of a slice sit at compile-time offsets on either side of the role mask.
    // (Computed from lvl and k inside the stage, the addresses were a third of a backward stage's vector instructions.)
    const char *adj_base = nullptr, *fwd_base = nullptr;
    const unsigned adj_stride = (su ? MC * MC : MC * NS) * (unsigned)sizeof(double);
    auto set_level_bases = [&]() {
        adj_base = su ? (const char*)(Mtab + ((long)lvl * N * MC + iu) * MC) : (const char*)(Ktab + (long)lvl * N * MC * NS + ix);
        fwd_base = (const char*)(Ktab + (long)lvl * N * MC * NS + iu * NS);
    };
    auto load_tab = [&](int k, Tab& t) {
        k = k < 0 ? 0 : (k >= N ? N - 1 : k);
        t.y = yw[k * WS_STAGE];
        t.lam = lw[k * WS_STAGE];
        t.kf = kf[k * WS_STAGE];
        const double* pa = (const double*)(adj_base + (unsigned long long)(unsigned)k * adj_stride);
        if (su) {   // (only the control lanes' rows change with the stage: the state lanes keep their row of A, set once per sweep)
            const double* pf = (const double*)(fwd_base + (unsigned long long)(unsigned)k * (unsigned)(MC * NS * sizeof(double)));
#pragma unroll
            for (int i = 0; i < NS; ++i) t.fwd[i] = pf[i];
#pragma unroll
            for (int j = 0; j < MC; ++j) t.adj[j] = pa[j];           // row of Suu_k^-1
        } else {
#pragma unroll
            for (int j = 0; j < MC; ++j) t.adj[j] = pa[j * NS];      // column of K_k
        }
    };
    auto init_tab = [&](Tab& t) {
#pragma unroll
        for (int i = 0; i < NS; ++i) t.fwd[i] = Arow[i];
    };

    const double alpha = g.alpha, om_alpha = 1.0 - g.alpha;
    int status = x0_in ? 0 : ZM_MPC_INFEASIBLE;
    int it = 0;
    double rp = 0.0, rd = 0.0;
    bool near_ok = false;   // the last iterate's residuals are within 10x the tolerances (OSQP's "solved inaccurate" test at the cap)
    bool done = !live || status != 0;              // group-uniform
    for (int gi = 0; gi < g.max_iter; ++gi) {
        if (__all(done)) break;
        const bool chk = ((gi + 1) % ZM_MPC_CHK) == 0;
        set_level_bases();
        // ---- backward affine sweep.  The table slices come from L2 (~500+ cycles) and a stage is shorter than that, so they are
        //      fetched THREE stages ahead into a rotating set of registers (the loop is unrolled by three: no copies).
        double pp = 0.0;   // (A^T p - K^T Qu) of the stage above (state lanes)
        {
            auto bstage = [&](const int k, const Tab& t) {
                const double z = -rho * (t.y - t.lam);    // -rho z_x (state lanes), -rho z_u (control lanes)
                const double kfo = t.kf;
                const double p = pp + z;                  // costate of x_{k+1} (state lanes)
                double q = sx ? 0.0 : z;
                mv<NS>(ABcol, p, q);                      // A^T p (state lanes);  Qu = -rho z_u + B^T p (control lanes)
                double r = 0.0;
                mv<MC, NS>(t.adj, q, r);                  // K^T Qu (state lanes);  kf = Suu^-1 Qu (control lanes)
                if (su) kf[k * WS_STAGE] = done ? kfo : r;
                pp = q - r;
            };
            Tab t0, t1, t2;
            load_tab(N - 1, t0);
            load_tab(N - 2, t1);
            load_tab(N - 3, t2);
            int k = N - 1;
#pragma unroll 1
            for (; k >= 2; k -= 3) {
                bstage(k, t0);
                load_tab(k - 3, t0);
                bstage(k - 1, t1);
                load_tab(k - 4, t1);
                bstage(k - 2, t2);
                load_tab(k - 5, t2);
            }
            if (k >= 0) bstage(k, t0);
            if (k >= 1) bstage(k - 1, t1);
        }
        // ---- forward rollout, projection, dual update, residuals
        double x = x0;
        double nrp = 0.0, nrd = 0.0, nw = 0.0, ny = 0.0, nl = 0.0, sup = 0.0, ndl = 0.0;
        {
            auto fstage = [&](const int k, const Tab& t) {
                double ax = 0.0;
                mv<NS>(t.fwd, x, ax);                 // A x (state lanes), K x (control lanes)
                const double u = -t.kf - ax;          // (control lanes; only their u is ever broadcast, stored or projected)
                double xn = ax;
                mv_seq<MC, NS>(Brow, u, xn);          // + B u (state lanes)
                const double w = sx ? xn : u;         // the stacked iterate [x_{k+1} ; u_k]
                const double lold = t.lam, yold = t.y;
                const double wh = __builtin_fma(alpha, w, om_alpha * yold);   // relaxed iterate (alpha = 1: w exactly)
                double yn = wh + lold;
                yn = yn < lo ? lo : (yn > hi ? hi : yn);
                const double r = w - yn, dl = wh - yn, ln = lold + dl;        // primal residual; dual step
                yw[k * WS_STAGE] = (done || !sw) ? yold : yn;
                lw[k * WS_STAGE] = (done || !sw) ? lold : ln;
                if (chk) rw[k * WS_STAGE] = sw ? dl : 0.0;
                if (sw) {
                    if (chk) {
                        sup += (dl > 0.0) ? dl * hi : ((dl < 0.0) ? dl * lo : 0.0);
                        amax(ndl, dl);
                    }
                    amax(nrp, r);
                    amax(nrd, yn - yold);
                    amax(nw, w);
                    amax(ny, yn);
                    amax(nl, ln);
                }
                x = xn;                               // (only the state lanes' x is ever broadcast)
            };
            Tab t0, t1, t2;
            init_tab(t0);
            init_tab(t1);
            init_tab(t2);
            load_tab(0, t0);
            load_tab(1, t1);
            load_tab(2, t2);
            int k = 0;
#pragma unroll 1
            for (; k + 2 < N; k += 3) {
                fstage(k, t0);
                load_tab(k + 3, t0);
                fstage(k + 1, t1);
                load_tab(k + 4, t1);
                fstage(k + 2, t2);
                load_tab(k + 5, t2);
            }
            if (k < N) fstage(k, t0);
            if (k + 1 < N) fstage(k + 1, t1);
        }
        nrp = row_max(nrp);
        nrd = row_max(nrd);
        nw = row_max(nw);
        ny = row_max(ny);
        nl = row_max(nl);
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
                done = true;   // NaN iterates (non-finite data): stop with the limit status
            } else {
                need_cert = chk;
            }
        }
        // ---- adaptive penalty (OSQP adaptive_rho): rho <- rho sqrt(normalised primal / normalised dual residual), taken in
        //      whole steps of the tabulated levels: to the level nearest the wanted penalty on the log scale (so a move happens
        //      when the penalty is off by at least sqrt(rho_step)); the scaled dual lam = mu / rho is rescaled so that the
        //      unscaled multiplier mu is unchanged
        if (g.n_levels > 1 && chk && !done) {
            const double tiny = 1e-300;
            const double rpn = rp / __builtin_fmax(__builtin_fmax(nw, ny), tiny);
            const double rdn = rd / __builtin_fmax(rho * nl, tiny);
            const double want = __builtin_sqrt(rpn / __builtin_fmax(rdn, tiny));
            int dl = 0;
            if (want == want && want > 0.0) dl = (int)lrint(log(want) / log(g.rho_step));   // the NEAREST tabulated level
            int nl_ = lvl + dl;
            nl_ = nl_ < 0 ? 0 : (nl_ >= g.n_levels ? g.n_levels - 1 : nl_);
            if (nl_ != lvl) {
                const double rnew = g.rho * pow(g.rho_step, (double)(nl_ - g.level0));
                const double sc = rho / rnew;
                for (int k = 0; k < N; ++k) lw[k * WS_STAGE] *= sc;
                rho = rnew;
                lvl = nl_;
            }
        }
        // ---- primal infeasibility certificate (mpc.hip header): adjoint sweep over r = w - y
        if (chk && __any(need_cert)) {
            sup = row_sum(sup);
            double sv = sx ? rw[(N - 1) * WS_STAGE] : 0.0;
            double gmax = 0.0;
#pragma unroll 1
            for (int k = N - 1; k >= 0; --k) {
                const double rk = rw[k * WS_STAGE], rkm = (k >= 1) ? rw[(k - 1) * WS_STAGE] : 0.0;
                double gs = su ? rk : (sx ? rkm : 0.0);
                mv<NS>(ABcol, sv, gs);                // (G^T r)_k = r_u,k + B^T s (control lanes);   s <- r_x,k-1 + A^T s (state lanes)
                if (su) gmax = __builtin_fmax(gmax, __builtin_fabs(gs));
                sv = sx ? gs : 0.0;
            }
            gmax = row_max(gmax);
            const double vw0 = row_sum(sv * x0);
            const double dn = row_max(ndl);          // |dual step|: the certificate's scale (= rp without relaxation)
            if (need_cert && gmax <= g.eps_pinf * dn && (vw0 - sup) > g.eps_pinf * dn) {
                status = ZM_MPC_INFEASIBLE;
                done = true;
            }
        }
    }
    // ---- final trajectory (the dynamics-exact rollout of the last iterate) and the iterates for a later warm start
    if (live) {
        set_level_bases();
        double x = x0;
        if (sx) g.xTraj[(inst * (N + 1)) * NS + ix] = x;
        Tab tf;
        init_tab(tf);
#pragma unroll 1
        for (int k = 0; k < N; ++k) {
            load_tab(k, tf);
            double ax = 0.0;
            mv<NS>(tf.fwd, x, ax);
            const double u = su ? -kf[k * WS_STAGE] - ax : 0.0;
            double xn = ax;
            mv_seq<MC, NS>(Brow, u, xn);
            if (su) g.uTraj[(inst * N + k) * MC + iu] = u;
            x = sx ? xn : 0.0;
            if (sx) g.xTraj[(inst * (N + 1) + k + 1) * NS + ix] = x;
            if (sw) {
                wsi[(long)k * W + iw] = yw[k * WS_STAGE];
                wsi[(long)N * W + (long)k * W + iw] = lw[k * WS_STAGE];
            }
        }
        if (li == 0) {
            g.status[inst] = status ? status : (near_ok ? ZM_MPC_OPTIMAL_INACCURATE : ZM_MPC_USER_LIMIT);
            *okflag = (status == ZM_MPC_OPTIMAL) ? 1.0 : 0.0;
            okflag[1] = (double)lvl;
            if (g.iters) g.iters[inst] = it;
            if (g.resid) {
                g.resid[inst * 2] = rp;
                g.resid[inst * 2 + 1] = rd;
            }
        }
    }
}

template <int NS, int MC>
static int launch_wave(const MpcTabs& t, const MpcArgs& g, hipStream_t st) {
    const size_t bytes = (size_t)4 * g.N * WS_STAGE * sizeof(double);
    if (bytes > 150 * 1024) return ZM_EUNSUPPORTED;   // horizon too long for LDS: the lane-per-instance kernel takes it
    // per launch (cheap): the attribute is per device, and several devices may be driven from one process
    ZM_HIP_CHECK(hipFuncSetAttribute((const void*)mpc_solve_wave_kernel<NS, MC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     150 * 1024));
    hipLaunchKernelGGL((mpc_solve_wave_kernel<NS, MC>), dim3((unsigned)((g.batch + 3) / 4)), dim3(64), bytes, st, t.A, t.B, t.K,
                       t.Minv, t.x_lb, t.x_ub, t.u_lb, t.u_ub, g);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

int mpc_wave_dispatch(const MpcTabs& t, const MpcArgs& g, int n, int m, hipStream_t st) {
    if (n == 12 && m == 4) return launch_wave<12, 4>(t, g, st);
    if (n == 8 && m == 4) return launch_wave<8, 4>(t, g, st);
    if (n == 4 && m == 2) return launch_wave<4, 2>(t, g, st);
    if (n == 4 && m == 1) return launch_wave<4, 1>(t, g, st);
    if (n == 2 && m == 2) return launch_wave<2, 2>(t, g, st);
    if (n == 2 && m == 1) return launch_wave<2, 1>(t, g, st);
    if (n == 1 && m == 1) return launch_wave<1, 1>(t, g, st);
    return ZM_EUNSUPPORTED;
}

}  // namespace zm
