// K3-T / K2-T: the iLQR backward pass and the bilinear-affine LQR sweep for LARGE states (12 < n <= 48 or 4 < m <= 16, fp64) on the
// fp64 MFMA tile algebra of lqr_tiled_core.h -- one wave per trajectory, the value Hessian V, the step's [f_x | f_u] and every
// product as 16 x 16 D-layout tiles in registers, op(X, Y) = X^T Y with no lane movement.
//
// Replaces (per trajectory, k = T-1 .. 0):
//   MODE 0  zopt/ilqrUtils.py:153-181 riccatiStep_ilqr / backwardPass_ilqr
//       Q_x = c_x + f_x^T v_x          Q_u = c_u + f_u^T v_x
//       Q_xx = c_xx + f_x^T v_xx f_x   Q_uu = c_uu + f_u^T v_xx f_u   Q_ux = c_ux + f_u^T v_xx f_x                 (:160-165)
//       l = -solve(Q_uu, Q_u)          L = -solve(Q_uu, Q_ux)                                                        (:167-168)
//       v_x' = Q_x - L^T Q_uu l        v_xx' = Q_xx - L^T Q_uu L                                                     (:170)
//   MODE 1  zopt/lqrUtils.py:242-261 bilinearAffineLqr (carry (V, v) <- (Q[-1], q[-1]), :261)
//       Su = r + B^T (v + V^T d)       Suu = R + B^T V B              Sux = H + B^T V A                              (:244-247)
//       L = solve(Suu, Sux)            l = solve(Suu, Su)                                                            (:248-249)
//       V' = Q + A^T V A - L^T Suu L   v' = q + A^T (v + V d) - Sux^T l                                              (:250-251)
// (`d^T V B` of :244 is B^T (V^T d), `A^T (v + V d)` of :251 uses V d: the two are kept apart, so a nonsymmetric V is treated as
// the reference treats it; with F = [f_x | f_u] the matrix part is G = F^T V F + C computed as (V^T F)^T F, which is exact for a
// nonsymmetric V too.)
//
// The m x m solve is the 16 x 16 register solve of the LQR tile kernel: lane j < NP owns column j of Q_ux, lane NP the vector
// right-hand side (Q_u resp. Su), every lane column c of Q_uu; LU WITHOUT row exchanges on registers, accepted only while every
// multiplier stays <= 4 in magnitude (differs from jnp.linalg.solve's pivoted LU by rounding only), otherwise LU with partial
// pivoting on the copy in LDS.  The vector terms are VALU work on the tile registers: column sums reduced over the four lane
// groups, row sums over the 16 lanes of a group.
//
// Not tuned like K1-T (no operand prefetch, no MFMA / VALU interleave): it lifts the shape limit of the tile-16 kernels
// (ilqr_backward.hip: n <= 12, m <= 4) for the array-level sweeps; measured rates in DESIGN.md.
#include "lqr_tiled_core.h"
#include "tile16_f64.h"

namespace zm {

struct TrajListT {
    const int* list;
    long count;
};

struct SweepTiledArgs {
    const double *f_x, *f_u, *c_x, *c_u, *c_xx, *c_ux, *c_uu, *vf_x, *vf_xx, *dvec;
    long svx, svxx;        // trajectory strides (doubles) of vf_x / vf_xx
    const int* active;
    int shared_hessian;    // c_xx / c_ux / c_uu are one (n,n) / (m,n) / (m,m) set for every trajectory and step
    double *l, *L;
    long batch;
    int T, n, m;
    TrajListT tl;
    // MODE 0, optional: the value function riccatiStep_ilqr returns (ilqrUtils.py:170): scalar stage costs c (batch,T) and terminal v
    // (batch) in; v, v_x (batch,n), v_xx (batch,n,n) at the start of the horizon out (all five NULL: policy only)
    const double *cs, *vf;
    double *v_out, *vx_out, *vxx_out;
};

// M4: m <= 4 -- the m x m solve is then the lane-local 4 x 4 elimination of the tile-16 kernels (tile16_f64.h: every lane reads the
// 4 x 4 block of Q_uu by broadcast and eliminates its own right-hand-side column, ~60 instructions) instead of the 16 x 16 register
// solve (~1 300): at (16, 4) the solve was two thirds of the step.
template <int NT, int MODE, bool M4>
__global__ __launch_bounds__(64) void sweep_tiled_f64(const SweepTiledArgs a) {
    using TR = TileF64;
    using f4 = td4;
    constexpr int TLD = TR::TLD;
    constexpr int NP = 16 * NT;
    constexpr int UC = NP + 16;   // first column of Q_uu in the solve buffer (column NP: the vector right-hand side)
    __shared__ __attribute__((aligned(16))) double Sc[(NP + 32) * TLD];   // column-major: element (row u, column j) at Sc[j * TLD + u]
    __shared__ __attribute__((aligned(16))) double Tq[16 * TLD];          // Q_uu row-major
    __shared__ double vecs[2][NP + 16];                                   // [0]: value gradient by state index, [1]: V^T d
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const long slot = blockIdx.x;
    const long traj = a.tl.list ? (long)a.tl.list[slot] : slot;
    if (traj >= a.batch || (a.active && !a.active[traj])) return;
    const int n = a.n, m = a.m, T = a.T;
    const long nn = (long)n * n, nm = (long)n * m, mm = (long)m * m;
    const double* fxb = a.f_x + traj * T * nn;
    const double* fub = a.f_u + traj * T * nm;
    const double* cxb = a.c_x + traj * T * n;
    const double* cub = a.c_u + traj * T * m;
    const long hs = a.shared_hessian ? 0 : 1;
    const double* cxxb = a.c_xx + hs * traj * T * nn;
    const double* cuxb = a.c_ux + hs * traj * T * nm;
    const double* cuub = a.c_uu + hs * traj * T * mm;
    const double* db = MODE == 1 ? a.dvec + traj * T * n : nullptr;
    double* Lb = a.L + traj * T * nm;
    double* lb = a.l + traj * T * m;

    const int jl = lane < NP ? lane : NP;      // lanes >= NP all read the vector column; only lane NP's copy is used
    const bool own_x = lane <= NP;
    const bool own_u = lane < 16;

    f4 V[NT][NT];
    {
        const double* vxx = a.vf_xx + traj * a.svxx;
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int J = 0; J < NT; ++J) V[K][J] = load_tile<TR, false>(vxx, n, n, K, J, g, c);
        const double* vx = a.vf_x + traj * a.svx;
        if (lane < NP) vecs[0][lane] = lane < n ? vx[lane] : 0.0;
    }
    t_lds_sync();
    double vs = (MODE == 0 && a.v_out && a.vf) ? a.vf[traj] : 0.0;   // scalar part of the value function (wave-uniform)

    // operands of a step: F = [f_x | f_u], the stacked cost Hessian (accumulator init of G), the cost gradients at this lane's
    // column / control.  They do not depend on the recursion: step k-1's are fetched into a second register set while step k computes.
    // (three tile rows: the second set would be 200 more registers and the compiler spills them -- there the operands are loaded at
    //  the head of their own step, as the LQR tile kernel does at four tile rows)
    constexpr bool PF = NT <= 2;
    f4 Fn[PF ? NT : 1][NT + 1], Gn[PF ? NT : 1][NT], Gun[PF ? NT + 1 : 1];
    double gxn = 0.0, gun = 0.0;
    auto fetch = [&](const int k, f4 (&Fd)[PF ? NT : 1][NT + 1], f4 (&Gd)[PF ? NT : 1][NT], f4 (&Gud)[PF ? NT + 1 : 1], double& gxd, double& gud) {
        if constexpr (PF) {
#pragma unroll
            for (int K = 0; K < NT; ++K) {
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    Fd[K][J] = load_tile<TR, false>(fxb + k * nn, n, n, K, J, g, c);
                    Gd[K][J] = load_tile<TR, false>(cxxb + hs * k * nn, n, n, K, J, g, c);
                }
                Fd[K][NT] = load_tile<TR, false>(fub + k * nm, n, m, K, 0, g, c);
                Gud[K] = load_tile<TR, false>(cuxb + hs * k * nm, m, n, 0, K, g, c);
            }
            Gud[NT] = load_tile<TR, false>(cuub + hs * k * mm, m, m, 0, 0, g, c, 1.0);   // padded controls: identity pivots
            gxd = lane < n ? cxb[k * n + lane] : 0.0;                                    // c_x resp. q at this lane's column
            gud = c < m ? cub[k * m + c] : 0.0;                                          // c_u resp. r at this lane's control
        }
    };
    if constexpr (PF) fetch(T - 1, Fn, Gn, Gun, gxn, gun);
    for (int k = T - 1; k >= 0; --k) {
        f4 F[NT][NT + 1], G[NT][NT], Gu[NT + 1];
        double gx, gu;
        if constexpr (PF) {
#pragma unroll
            for (int K = 0; K < NT; ++K) {
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    F[K][J] = Fn[K][J];
                    G[K][J] = Gn[K][J];
                }
                F[K][NT] = Fn[K][NT];
                Gu[K] = Gun[K];
            }
            Gu[NT] = Gun[NT];
            gx = gxn;
            gu = gun;
            fetch(k > 0 ? k - 1 : 0, Fn, Gn, Gun, gxn, gun);   // (the last iteration re-reads step 0: no branch around the loads)
        } else {
#pragma unroll
            for (int K = 0; K < NT; ++K) {
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    F[K][J] = load_tile<TR, false>(fxb + k * nn, n, n, K, J, g, c);
                    G[K][J] = load_tile<TR, false>(cxxb + hs * k * nn, n, n, K, J, g, c);
                }
                F[K][NT] = load_tile<TR, false>(fub + k * nm, n, m, K, 0, g, c);
                Gu[K] = load_tile<TR, false>(cuxb + hs * k * nm, m, n, 0, K, g, c);
            }
            Gu[NT] = load_tile<TR, false>(cuub + hs * k * mm, m, m, 0, 0, g, c, 1.0);
            gx = lane < n ? cxb[k * n + lane] : 0.0;
            gu = c < m ? cub[k * m + c] : 0.0;
        }

        // ---- vector terms (they need V BEFORE its update): the vector(s) at this lane's tile rows 16 K + 4 r + g
        double tr_[NT][4], sr_[NT][4];   // multiplied into the f_x columns / into the f_u columns
#pragma unroll
        for (int K = 0; K < NT; ++K)
#pragma unroll
            for (int r = 0; r < 4; ++r) tr_[K][r] = sr_[K][r] = vecs[0][16 * K + 4 * r + g];
        if constexpr (MODE == 1) {
            double dr[NT][4], dc[NT];
#pragma unroll
            for (int K = 0; K < NT; ++K) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * K + 4 * r + g;
                    dr[K][r] = i < n ? db[k * n + i] : 0.0;
                }
                const int j = 16 * K + c;
                dc[K] = j < n ? db[k * n + j] : 0.0;
            }
            // V^T d: column sums over this lane's rows, reduced over the four lane groups; parked in LDS by state index
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                double p = 0.0;
#pragma unroll
                for (int K = 0; K < NT; ++K)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p = __builtin_fma(V[K][J][r], dr[K][r], p);
                p = sum_xor32(sum_xor16(p));
                if (g == J) vecs[1][16 * J + c] = p;
            }
            // V d: row sums over this lane's columns, reduced over the 16 lanes of the group
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double p = 0.0;
#pragma unroll
                    for (int J = 0; J < NT; ++J) p = __builtin_fma(V[K][J][r], dc[J], p);
                    p = row16_sum(p);   // (DPP butterflies, the bits of the __shfl_xor ladder over 1, 2, 4, 8)
                    tr_[K][r] += p;                                    // v + V d      (lqrUtils.py:251)
                }
            t_lds_sync();
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int r = 0; r < 4; ++r) sr_[K][r] += vecs[1][16 * K + 4 * r + g];   // v + V^T d    (:244)
        }
        double qx = gx, qu = gu;   // Q_x at column `lane` (lanes < NP), Q_u at control c
#pragma unroll
        for (int J = 0; J <= NT; ++J) {
            double p = 0.0;
#pragma unroll
            for (int K = 0; K < NT; ++K)
#pragma unroll
                for (int r = 0; r < 4; ++r) p = __builtin_fma(F[K][J][r], (J == NT ? sr_[K][r] : tr_[K][r]), p);
            p = sum_xor32(sum_xor16(p));
            if (J == NT)
                qu += p;
            else if (g == J)
                qx += p;
        }

        // ---- Y = V^T F;  G = C + Y^T F  (Q_xx, Q_ux, Q_uu)
        {
            f4 Y[NT][NT + 1];
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = 0; J <= NT; ++J) {
                    f4 acc = TR::zero();
#pragma unroll
                    for (int K = 0; K < NT; ++K) acc = op<TR>(V[K][I], F[K][J], acc);
                    Y[I][J] = acc;
                }
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int K = 0; K < NT; ++K) G[I][J] = op<TR>(Y[K][I], F[K][J], G[I][J]);
#pragma unroll
            for (int J = 0; J <= NT; ++J)
#pragma unroll
                for (int K = 0; K < NT; ++K) Gu[J] = op<TR>(Y[K][NT], F[K][J], Gu[J]);
        }

        // ---- [Q_ux | Q_u | Q_uu] to the solve buffer
#pragma unroll
        for (int J = 0; J < NT; ++J) TR::tile_to_lds_T(Sc + 16 * J * TLD, Gu[J], g, c);
        TR::tile_to_lds_T(Sc + UC * TLD, Gu[NT], g, c);
        TR::tile_to_lds(Tq, Gu[NT], g, c);
        if (g == 0) Sc[NP * TLD + c] = qu;
        t_lds_sync();
        double x[16], u[16], x0[MODE == 1 ? 16 : 1];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            x[i] = Sc[jl * TLD + i];
            u[i] = Sc[(UC + c) * TLD + i];
            if constexpr (MODE == 1) x0[i] = x[i];
        }
      if constexpr (M4) {
        // m <= 4: the 4 x 4 block of Q_uu (rows / columns >= m are identity padding) by broadcast reads, this lane's right-hand side
        double S4[4][4], b4[4], x4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) S4[i][jj] = Sc[(UC + jj) * TLD + i];
            b4[i] = x[i];
        }
        if (__builtin_amdgcn_ballot_w64(!lu_solve4_nopivot(S4, b4, x4)) != 0ull) lu_solve4_fallback(S4, b4, x4);
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = i < 4 ? x4[i] : 0.0;
      } else {
        // ---- LU without row exchanges on registers (row operations are lane-local, the multipliers wave-uniform)
        unsigned long long bad = 0ull;
        double pinv[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const double inv = TR::rcp(u[kk]);
            pinv[kk] = TR::readlane(inv, kk);
            bool okm = true;
#pragma unroll
            for (int r = kk + 1; r < 16; ++r) {
                const double mv = u[r] * inv;
                okm &= (__builtin_fabs(mv) <= 4.0);          // NaN compares false: a NaN multiplier fails the check
                const double ms = TR::readlane(mv, kk);
                x[r] = __builtin_fma(-ms, x[kk], x[r]);
                u[r] = __builtin_fma(-ms, u[kk], u[r]);
            }
            bad |= __ballot(!okm) & (0x0001000100010001ull << kk);   // only lane c == kk of each group holds the column's multipliers
        }
        bad |= __ballot(!(__builtin_fabs(pinv[15]) < TR::huge()));
#pragma unroll
        for (int kk = 15; kk >= 0; --kk) {
            double a0 = x[kk], a1 = 0.0;
#pragma unroll
            for (int r = kk + 1; r < 16; ++r) {
                const double ur = TR::readlane(u[kk], r);
                if (r & 1)
                    a1 = __builtin_fma(-ur, x[r], a1);
                else
                    a0 = __builtin_fma(-ur, x[r], a0);
            }
            x[kk] = (a0 + a1) * pinv[kk];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) TR::pin(x[i]);
        if (bad != 0ull) {   // wave-uniform, rare: LU with partial pivoting on the copy still in LDS (getrf / getrs order)
#define S_(r_, j_) Sc[(j_) * TLD + (r_)]
#pragma unroll 1
            for (int kk = 0; kk < 16; ++kk) {
                double pv = (c >= kk) ? __builtin_fabs(S_(c, UC + kk)) : -1.0;
                int pi = c;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const double ov = __shfl_xor(pv, off, 16);
                    const int oi = __shfl_xor(pi, off, 16);
                    const bool take = (ov > pv) || (ov == pv && oi < pi);  // first largest entry, as idamax
                    pv = take ? ov : pv;
                    pi = take ? oi : pi;
                }
                const int p = __builtin_amdgcn_readfirstlane(pi);
                {  // swap rows kk and p (a no-op when p == kk); every LDS word has ONE owner lane
                    const double a0 = S_(kk, jl), b0 = S_(p, jl);
                    const double a1 = S_(kk, UC + c), b1 = S_(p, UC + c);
                    t_lds_sync();
                    if (own_x) {
                        S_(kk, jl) = b0;
                        S_(p, jl) = a0;
                    }
                    if (own_u) {
                        S_(kk, UC + c) = b1;
                        S_(p, UC + c) = a1;
                    }
                    t_lds_sync();
                }
                const double inv = 1.0 / S_(kk, UC + kk);
                const double pj = S_(kk, jl);
                const double pu = S_(kk, UC + c);
#pragma unroll 1
                for (int r = kk + 1; r < 16; ++r) {
                    const double mr = S_(r, UC + kk) * inv;
                    const double xj = S_(r, jl);
                    const double xu = S_(r, UC + c);
                    t_lds_sync();
                    if (own_x) S_(r, jl) = xj - mr * pj;
                    if (own_u && c > kk) S_(r, UC + c) = xu - mr * pu;
                }
                t_lds_sync();
            }
#pragma unroll
            for (int kk = 15; kk >= 0; --kk) {
                double acc = S_(kk, jl);
#pragma unroll
                for (int r = kk + 1; r < 16; ++r) acc -= S_(kk, UC + r) * x[r];
                x[kk] = acc / S_(kk, UC + kk);
            }
            t_lds_sync();
#undef S_
        }
      }
        // x = column `lane` of Xs = Q_uu^-1 Q_ux (lanes < NP); lane NP: xs = Q_uu^-1 Q_u.  The policy of this step:
        constexpr double sgn = MODE == 0 ? -1.0 : 1.0;   // ilqrUtils.py:167-168 negates, lqrUtils.py:248-249 does not
        if (lane < n) {
#pragma unroll
            for (int u_ = 0; u_ < 16; ++u_)
                if (u_ < m) Lb[k * nm + (long)u_ * n + lane] = sgn * x[u_];
        }
        if (lane == NP) {
#pragma unroll
            for (int u_ = 0; u_ < 16; ++u_)
                if (u_ < m) lb[k * m + u_] = sgn * x[u_];
        }
        // ---- value update.  Matrix part: V' = G - Xs^T (Q_uu Xs)
        if (lane < NP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Sc[lane * TLD + i] = x[i];
        }
        t_lds_sync();
        {
            const f4 QT = TR::tile_from_lds_T(Tq, g, c);   // Q_uu^T as a tile: op(QT, X) = Q_uu X
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                f4 Xs;
#pragma unroll
                for (int r = 0; r < 4; ++r) Xs[r] = Sc[(16 * J + c) * TLD + TR::row(g, r)];
                Gu[J] = op<TR>(QT, Xs, TR::zero());        // M_J = Q_uu Xs_J   (Gu is free now)
            }
#pragma unroll
            for (int I = 0; I < NT; ++I) {
                f4 NXs;
#pragma unroll
                for (int r = 0; r < 4; ++r) NXs[r] = -Sc[(16 * I + c) * TLD + TR::row(g, r)];
#pragma unroll
                for (int J = 0; J < NT; ++J) V[I][J] = op<TR>(NXs, Gu[J], G[I][J]);
            }
        }
        // Vector part.  xl[k]: the vector solve, wave-uniform
        double vnew;
        if constexpr (MODE == 0) {
            double w = 0.0;   // w_c = (Q_uu xs)[c]
#pragma unroll
            for (int kq = 0; kq < 16; ++kq) w = __builtin_fma(Tq[c * TLD + kq], TR::readlane(x[kq], NP), w);
            vnew = qx;        // v_x' = Q_x - Xs^T (Q_uu xs)  (= Q_x - L^T Q_uu l)
#pragma unroll
            for (int u_ = 0; u_ < 16; ++u_) vnew = __builtin_fma(-x[u_], TR::readlane(w, u_), vnew);
            if (a.v_out) {    // v' = (c + v) - l^T Q_uu l / 2,  l^T Q_uu l = xs^T (Q_uu xs)                       (ilqrUtils.py:170)
                double lql = 0.0;
#pragma unroll
                for (int u_ = 0; u_ < 16; ++u_) lql = __builtin_fma(TR::readlane(x[u_], NP), TR::readlane(w, u_), lql);
                vs = ((a.cs ? a.cs[traj * T + k] : 0.0) + vs) - 0.5 * lql;
            }
        } else {
            vnew = qx;        // v' = q + A^T (v + V d) - Sux^T l
#pragma unroll
            for (int u_ = 0; u_ < 16; ++u_) vnew = __builtin_fma(-x0[u_], TR::readlane(x[u_], NP), vnew);
        }
        if (lane < NP) vecs[0][lane] = vnew;
        t_lds_sync();
    }
    if constexpr (MODE == 0) {
        if (a.v_out) {   // value function at the start of the horizon
            if (lane == 0) a.v_out[traj] = vs;
            if (a.vx_out && lane < n) a.vx_out[traj * n + lane] = vecs[0][lane];
            if (a.vxx_out) {
#pragma unroll
                for (int K = 0; K < NT; ++K)
#pragma unroll
                    for (int J = 0; J < NT; ++J)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * K + TR::row(g, r), j = 16 * J + c;
                            if (i < n && j < n) a.vxx_out[traj * nn + (long)i * n + j] = V[K][J][r];
                        }
            }
        }
    }
}

template <int NT>
static int launch_sweep_tiled(int mode, const SweepTiledArgs& a, long nslot, hipStream_t st) {
    const dim3 grid((unsigned)nslot), block(64);
    const bool m4 = a.m <= 4;
    if (mode == 0 && m4)
        hipLaunchKernelGGL((sweep_tiled_f64<NT, 0, true>), grid, block, 0, st, a);
    else if (mode == 0)
        hipLaunchKernelGGL((sweep_tiled_f64<NT, 0, false>), grid, block, 0, st, a);
    else if (m4)
        hipLaunchKernelGGL((sweep_tiled_f64<NT, 1, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((sweep_tiled_f64<NT, 1, false>), grid, block, 0, st, a);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// mode 0: iLQR backward pass; mode 1: bilinear-affine LQR (operands as launch_ilqr<MODE> in ilqr_backward.hip takes them).
// ZM_EUNSUPPORTED beyond n <= 48, m <= 16.
int sweep_tiled_f64_dispatch(int mode, const double* f_x, const double* f_u, const double* c_x, const double* c_u, const double* c_xx,
                             const double* c_ux, const double* c_uu, const double* vf_x, const double* vf_xx, const double* dvec,
                             long svx, long svxx, const int* active, int shared_hessian, double* l, double* L, int64_t batch, int T,
                             int n, int m, hipStream_t st, const int* list, long count, const double* cs, const double* vf,
                             double* v_out, double* vx_out, double* vxx_out) {
    if (n < 1 || m < 1 || n > 48 || m > 16 || (mode != 0 && mode != 1)) return ZM_EUNSUPPORTED;
    if (mode == 1 && !dvec) return set_error(ZM_EINVAL, "sweep_tiled_f64: the affine sweep needs d");
    const SweepTiledArgs a{f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, dvec, svx, svxx, active, shared_hessian, l, L,
                           (long)batch, T, n, m, TrajListT{list, count}, cs, vf, v_out, vx_out, vxx_out};
    const long nslot = list ? count : (long)batch;
    if (nslot == 0) return ZM_OK;
    if (n <= 16) return launch_sweep_tiled<1>(mode, a, nslot, st);
    if (n <= 32) return launch_sweep_tiled<2>(mode, a, nslot, st);
    return launch_sweep_tiled<3>(mode, a, nslot, st);
}

}  // namespace zm
