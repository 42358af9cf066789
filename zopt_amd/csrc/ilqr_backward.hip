// K3  ilqr_backward -- batched iLQR backward pass (affine policy from a quadratic value sweep), fp64, gfx950.
//
// Replaces the arithmetic of zopt/ilqrUtils.py:153-181 (riccatiStep_ilqr / backwardPass_ilqr), per trajectory,
// for k = T-1 .. 0 with carry (v_x, v_xx) <- Vf:
//     Q_x  = c_x  + f_x^T v_x                 Q_u  = c_u  + f_u^T v_x                              (:161-162)
//     Q_xx = c_xx + f_x^T v_xx f_x   Q_uu = c_uu + f_u^T v_xx f_u   Q_ux = c_ux + f_u^T v_xx f_x    (:163-165)
//     l = -solve(Q_uu, Q_u)          L = -solve(Q_uu, Q_ux)                                        (:167-168)
//     v_x' = Q_x - L^T Q_uu l        v_xx' = Q_xx - L^T Q_uu L                                     (:170)
// The scalar part (c, v) never influences l or L and backwardPass_ilqr returns only the policy, so it is not
// carried.  `Q_uu l` is taken as -Q_u (its value by construction of l); the difference is the residual of the
// m x m solve (rounding level).
//
// Tile-16 mapping (tile16_f64.h): one wave64 per trajectory, F = [f_x | f_u] as a 16-column tile,
//     Y = v_xx^T F (KS MFMA),  G = Y^T F + [[c_xx, .],[c_ux, c_uu]] (KS MFMA)  ->  Q_xx (rows < NP), [Q_ux | Q_uu] (row NP+g)
//     q = [c_x ; c_u] + F^T v_x : 3 FMAs per lane + a 4-group xor reduction (column-indexed, replicated over g)
//     solve: the 4 x 16 tile [Q_ux | Q_uu] and the row q through LDS; lane c < n solves column c of Q_ux,
//            lane c == NP solves Q_u.  -x[g] is L_k[g][c] resp. l_k[g].
//     T1 = Q_uu L (1 MFMA),  v_xx' = Q_xx - L^T T1 (1 MFMA, negated A operand)
// 2*KS + 2 MFMAs per step.  Inputs are read once with 8-byte loads straight into their register layouts, two steps
// ahead (register double buffer).  Shapes: n <= 12, m <= 4.
//
// MODE 2 = K4 ddp_backward: zopt/ilqrUtils.py:184-214 (riccatiStep_ddp / backwardPass_ddp) = MODE 0 plus, per step,
//     vf_zz = [[vf_xx, vf_ux^T],[vf_ux, vf_uu]],  vf_.. = einsum('i,ijk', v_x, f_..)                  (:240-247)
//     vf_zz <- ensurePositiveDefinite(vf_zz)   (eigenvalue clamp 1e-3, INSIDE the sequential sweep)    (:248)
//     Q_xx += vf_xx,  Q_uu += vf_uu,  Q_ux += vf_ux                                                    (:196-198)
// The contraction reads f_xx (n,n,n), f_ux (n,m,n), f_uu (n,m,m) of the step straight from HBM (n terms per tile
// element), the compact (n+m) x (n+m) matrix goes through LDS into the wave-level Jacobi projection of jacobi16.h and
// comes back as an accumulator init of G.
//
// MODE 1 = K2 lqr_backward_affine: the same sweep for zopt/lqrUtils.py:207-262 (bilinearAffineLqr), i.e. with
//     f_x,f_u <- A,B   c_xx,c_ux,c_uu <- Q,H,R   c_x,c_u <- q,r   and the affine-dynamics offset d:
//     Su  = r + B^T (v + V^T d)      Suu = R + B^T V B      Sux = H + B^T V A                       (:244-246)
//     L = solve(Suu, Sux)   l = solve(Suu, Su)       (no sign flip; law u = -L x - l)                (:248-249)
//     V' = Q + A^T V A - L^T Suu L      v' = q + A^T (v + V d) - Sux^T l                             (:251-252)
// (v0 / q0 never influence L or l and are not carried.)  V d is a row reduction of the V registers (4 xor-shuffles per
// K-step), V^T d a column reduction (2 shuffles) re-laid out through LDS; both are kept apart so that a
// nonsymmetric V is treated exactly as the reference does.
#include "jacobi16.h"
#include "tile16_f64.h"
#include "zm_common.h"

namespace zm {

template <int KS>
struct IlqrStepRegs {
    double F[KS];   // [f_x | f_u][4s+g][c]
    double C[KS];   // c_xx[4s+g][c]
    double Cu;      // row NP+g of the stacked cost Hessian: c_ux[g][c] (c < n) | c_uu[g][c-NP] | identity padding
    double cv;      // [c_x ; c_u][c]  (column-indexed)
    double dc;      // MODE 1: d[c]       (column-indexed)
    double dr[KS];  // MODE 1: d[4s+g]    (row-indexed)
};

template <int KS>
struct IlqrAddr {
    const double* pF0;   // K-step s adds (rowok ? s*dF : 0)
    const double* pC0;   // K-step s adds (rowok ? s*4n : 0)
    const double* pCu;
    const double* pcv;
    const double* pdc;   // MODE 1
    const double* pdr0;  // MODE 1: K-step s adds (rowok ? 4 s : 0)
    const double* pz[4]; // MODE 2: element (4r+g, c) of the stacked second-derivative tensor slice i = 0; +i*sz[r]
    int sz[4], stz[4];   // MODE 2: stride over i, stride over the time step
    int zc[4];           // MODE 2: compact LDS index a*PLD+b of tile element (4r+g, c), or -1
    double* pOut;        // L_k[g][c] (c < n) or l_k[g] (c == NP)
    int dF, dC, sF, sC, sCu, scv, sOut, sd;
    bool rowok[KS], vF[KS], vC[KS], vCu, vcv, vOut, vL, cA;
    bool warm_v = false;   // MODE 2: the Jacobi eigenvector buffer holds the previous step's result
    double vsum = 0.0;     // MODE 0/2, lanes c == NP: sum over the steps of -1/2 l^T Q_uu l  (scalar part of the value function)
    double cu_pad;
};

template <int KS, int MODE>
__device__ __forceinline__ void ilqr_load_step(IlqrStepRegs<KS>& d, IlqrAddr<KS>& a) {
    double f[KS], cc[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = a.pF0[a.rowok[s] ? s * a.dF : 0];
    a.pF0 -= a.sF;
#pragma unroll
    for (int s = 0; s < KS; ++s) cc[s] = a.pC0[a.rowok[s] ? s * a.dC : 0];
    a.pC0 -= a.sC;
    const double cu = *a.pCu;
    a.pCu -= a.sCu;
    const double cv = *a.pcv;
    a.pcv -= a.scv;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        d.F[s] = a.vF[s] ? f[s] : 0.0;
        d.C[s] = a.vC[s] ? cc[s] : 0.0;
    }
    d.Cu = a.vCu ? cu : a.cu_pad;
    d.cv = a.vcv ? cv : 0.0;
    if constexpr (MODE == 1) {
        const double dc = *a.pdc;
        double dr[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) dr[s] = a.pdr0[a.rowok[s] ? 4 * s : 0];
        a.pdc -= a.sd;
        a.pdr0 -= a.sd;
        d.dc = a.cA ? dc : 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) d.dr[s] = a.rowok[s] ? dr[s] : 0.0;
    }
}

// LDS per wave (doubles): [0,64) tile rows 0..3 = [Q_ux | Q_uu]; [64,80) row 4 = q; [80,96) v_x' (column-indexed)
//                       MODE 1: [96,112) V^T d (column-indexed); [112,116) l
constexpr int ILQR_LDS_DOUBLES = 116;

__device__ __forceinline__ void ilqr_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KS, int MODE, bool PREFETCH>
__device__ __forceinline__ void ilqr_step(double (&Vxx)[KS], double (&vxr)[KS], IlqrStepRegs<KS>& d, IlqrAddr<KS>& a,
                                          double* sm, const int g, const int c, const int ob0, const int ob1,
                                          const int ob2, const int ob3, const int oqa, double* jA, double* jV,
                                          double* jcs, int* jpq, const int n, const int m) {
    constexpr int NP = 4 * KS;
    // MODE 2: vf_zz = sum_i v_x[i] * d2f_i/dz2, PD-projected, as extra accumulator init
    d4 pz = zero4();
    if constexpr (MODE == 2) {
        const int lane = g * 16 + c;
        double z[4] = {0.0, 0.0, 0.0, 0.0};
        // the loads of 4 slices (16 per lane) are in flight before their FMAs (unrolled, unconditional: slices i >= n
        // re-read slice 0 with weight 0); more in flight would cost the second wave per SIMD its registers
#pragma unroll
        for (int i0 = 0; i0 < NP; i0 += 4) {
            double fz[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ic = (i0 + i) < n ? (i0 + i) : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) fz[i][r] = a.pz[r][(long)ic * a.sz[r]];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double vxi = (i0 + i) < n ? sm[80 + i0 + i] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = __builtin_fma(vxi, fz[i][r], z[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a.pz[r] -= a.stz[r];
            if (a.zc[r] >= 0) jA[a.zc[r]] = z[r];
        }
        ilqr_lds_sync();
        // symmetrise (jnp.linalg.eigh does) -- in place through registers
        const int k = n + m;
        double sy[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = lane + 64 * r;
            const int i = e / k, j = e % k;
            sy[r] = (e < k * k) ? 0.5 * (jA[i * PLD + j] + jA[j * PLD + i]) : 0.0;
        }
        ilqr_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = lane + 64 * r;
            if (e < k * k) jA[(e / k) * PLD + (e % k)] = sy[r];
        }
        ilqr_lds_sync();
        psd_project_lds(jA, jV, jcs, jpq, k, 1e-3, lane, a.warm_v);
        a.warm_v = true;   // jV now holds eigenvectors of a neighbouring step's matrix
#pragma unroll
        for (int r = 0; r < 4; ++r) pz[r] = (a.zc[r] >= 0) ? jA[a.zc[r]] : 0.0;
    }
    // Y = v_xx^T F
    d4 y = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) y = mfma(Vxx[s], d.F[s], y);
    // G = Y^T F + C0
    d4 gacc = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) gacc[s] = d.C[s] + pz[s];
    gacc[KS] = d.Cu + pz[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) gacc = mfma(y[s], d.F[s], gacc);
    // q[c] = cv[c] + sum_k F[k][c] w[k]   (partial over this lane's rows, then over the 4 lane groups);
    // MODE 0: w = v_x.  MODE 1: w = v + V d under the state columns (A^T (v + V d), :252) and v + V^T d under the
    // control columns (v^T B + d^T V B, :244).
    double qp = 0.0;
    if constexpr (MODE == 1) {
        double vtd = 0.0;  // (V^T d)[c]: column reduction
#pragma unroll
        for (int s = 0; s < KS; ++s) vtd = __builtin_fma(Vxx[s], d.dr[s], vtd);
        vtd += __shfl_xor(vtd, 16);
        vtd += __shfl_xor(vtd, 32);
        if (g == 0) sm[96 + c] = a.cA ? vtd : 0.0;
        ilqr_lds_sync();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            double vd = Vxx[s] * d.dc;  // (V d)[4s+g]: row reduction over the 16 lanes of the group
            vd += __shfl_xor(vd, 1);
            vd += __shfl_xor(vd, 2);
            vd += __shfl_xor(vd, 4);
            vd += __shfl_xor(vd, 8);
            const double w = vxr[s] + (a.cA ? vd : sm[96 + 4 * s + g]);
            qp = __builtin_fma(d.F[s], w, qp);
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) qp = __builtin_fma(d.F[s], vxr[s], qp);
    }
    qp += __shfl_xor(qp, 16);
    qp += __shfl_xor(qp, 32);
    const double qv = d.cv + qp;
    if constexpr (PREFETCH) ilqr_load_step<KS, MODE>(d, a);

    // solve: tile + q row through LDS
    sm[g * 16 + c] = gacc[KS];
    if (g == 0) sm[64 + c] = qv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double S[4][4], b[4], x[4], qu[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) S[i][jj] = sm[i * 16 + NP + jj];
        qu[i] = sm[64 + NP + i];  // Q_u (zero beyond m)
    }
    b[0] = sm[ob0];
    b[1] = sm[ob1];
    b[2] = sm[ob2];
    b[3] = sm[ob3];
    const double quu_a = sm[oqa];  // Q_uu[c][g]: A operand of Q_uu L (0-padded through the address choice below)
    __builtin_amdgcn_wave_barrier();
    const double b0[4] = {b[0], b[1], b[2], b[3]};  // original right-hand side (Sux column) for MODE 1's v' update
    if (__builtin_amdgcn_ballot_w64(!lu_solve4_nopivot(S, b, x)) != 0ull) lu_solve4(S, b, x);
    const double x01 = (g & 1) ? x[1] : x[0];
    const double x23 = (g & 1) ? x[3] : x[2];
    const double xg = (g & 2) ? x23 : x01;
    const double out = (MODE == 1) ? xg : -xg;  // L_k[g][c] for c < n, l_k[g] for c == NP
    if (a.vOut) *a.pOut = out;
    a.pOut -= a.sOut;
    const double lv = a.vL ? out : 0.0;
    if constexpr (MODE != 1) {
        // -1/2 l^T Q_uu l with l = -x, Q_uu x = Q_u  (ilqrUtils.py:170 / :203), on the lanes that solved for l
        if (c == NP) a.vsum -= 0.5 * ((x[0] * qu[0] + x[1] * qu[1]) + (x[2] * qu[2] + x[3] * qu[3]));
    }

    double vxn = qv;
    if constexpr (MODE == 1) {
        // v'[c] = (q + A^T (v + V d))[c] - sum_i Sux[i][c] l[i]                                   (lqrUtils.py:252)
        if (g == 0 && c == NP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sm[112 + i] = x[i];
        }
        ilqr_lds_sync();
#pragma unroll
        for (int i = 0; i < 4; ++i) vxn = __builtin_fma(-b0[i], sm[112 + i], vxn);
    } else {
        // v_x'[c] = Q_x[c] + sum_i L[i][c] Q_u[i]      (= Q_x - L^T Q_uu l with Q_uu l = -Q_u)
#pragma unroll
        for (int i = 0; i < 4; ++i) vxn = __builtin_fma(-x[i], qu[i], vxn);
    }
    if (g == 0) sm[80 + c] = a.cA ? vxn : 0.0;
    // T1 = Q_uu L ;  v_xx' = Q_xx - L^T T1
    const d4 t1 = mfma(quu_a, lv, zero4());
    d4 vacc = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) vacc[s] = gacc[s];
    vacc = mfma<true>(lv, t1[0], vacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) Vxx[s] = vacc[s];
    ilqr_lds_sync();
#pragma unroll
    for (int s = 0; s < KS; ++s) vxr[s] = sm[80 + 4 * s + g];
    __builtin_amdgcn_wave_barrier();
}

template <int KS, int MODE>
__global__ __launch_bounds__(64, 2) void ilqr_backward_t16_f64(
    const double* __restrict__ f_x, const double* __restrict__ f_u, const double* __restrict__ c_x,
    const double* __restrict__ c_u, const double* __restrict__ c_xx, const double* __restrict__ c_ux,
    const double* __restrict__ c_uu, const double* __restrict__ vf_x, const double* __restrict__ vf_xx,
    const double* __restrict__ dvec, const double* __restrict__ f_xx, const double* __restrict__ f_ux,
    const double* __restrict__ f_uu, const long svx, const long svxx, const int* __restrict__ active,
    const int shared_h, double* __restrict__ lout, double* __restrict__ Lout, const int T, const int n, const int m,
    const double* __restrict__ c_s, const double* __restrict__ vf_s, double* __restrict__ v_out,
    double* __restrict__ vx_out, double* __restrict__ vxx_out) {
    constexpr int NP = 4 * KS;
    const int lane = threadIdx.x;
    const long traj = blockIdx.x;
    if (active && active[traj] == 0) return;  // whole wave leaves: this trajectory keeps its previous policy
    const int g = lane >> 4, c = lane & 15;
    __shared__ double sm[ILQR_LDS_DOUBLES];
    __shared__ double jA[MODE == 2 ? PK * PLD : 1], jV[MODE == 2 ? PK * PLD : 1], jcs[PK];
    __shared__ int jpq[PK];

    IlqrAddr<KS> a;
    const int nn = n * n, nm = n * m, mm = m * m;
    const bool cA = c < n;
    a.cA = cA;
    const bool cB = (c >= NP) && (c < NP + m);
    const long last = traj * T + (T - 1);
    const double* fxt = f_x + last * nn;
    const double* fut = f_u + last * nm;
    const double* cxt = c_x + last * n;
    const double* cut = c_u + last * m;
    // shared_h: one time-invariant cost Hessian for every trajectory and step (strides 0)
    const double* cxxt = shared_h ? c_xx : c_xx + last * nn;
    const double* cuxt = shared_h ? c_ux : c_ux + last * nm;
    const double* cuut = shared_h ? c_uu : c_uu + last * mm;
    a.sC = shared_h ? 0 : nn;
    const bool row0 = g < n;
    const bool laneA = row0 && cA, laneB = row0 && cB;
    a.dF = laneA ? 4 * n : laneB ? 4 * m : 0;
    a.sF = laneB ? nm : nn;
    a.dC = laneA ? 4 * n : 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int row = 4 * s + g;
        a.rowok[s] = row < n;
        a.vF[s] = (row < n) && (cA || cB);
        a.vC[s] = (row < n) && cA;
    }
    a.pF0 = laneA ? (fxt + g * n + c) : laneB ? (fut + g * m + (c - NP)) : fxt;
    a.pC0 = laneA ? (cxxt + g * n + c) : cxxt;
    // row NP+g of the stacked Hessian: c_ux[g][c] under the state columns, c_uu[g][c-NP] under the control columns
    const bool vux = (g < m) && cA, vuu = (g < m) && cB;
    a.vCu = vux || vuu;
    a.pCu = vux ? (cuxt + g * n + c) : vuu ? (cuut + g * m + (c - NP)) : cuxt;
    a.sCu = shared_h ? 0 : (vuu ? mm : nm);
    a.cu_pad = (g >= m && c == NP + g) ? 1.0 : 0.0;
    a.vcv = cA || cB;
    a.pcv = cA ? (cxt + c) : cB ? (cut + (c - NP)) : cxt;
    a.scv = cB ? m : n;
    if constexpr (MODE == 1) {
        const double* dt_ = dvec + last * n;
        a.pdc = cA ? (dt_ + c) : dt_;
        a.pdr0 = row0 ? (dt_ + g) : dt_;
        a.sd = n;
    }
    if constexpr (MODE == 2) {
        const long nnn = (long)nn * n, nmn = (long)nm * n, nmm = (long)nm * m;
        const double* fxxt = f_xx + last * nnn;
        const double* fuxt = f_ux + last * nmn;
        const double* fuut = f_uu + last * nmm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + g;
            const bool rx = row < n, ru = (row >= NP) && (row < NP + m);
            // compact index: states 0..n-1, controls n..n+m-1
            const int ca = rx ? row : (ru ? n + (row - NP) : -1);
            const int cb = cA ? c : (cB ? n + (c - NP) : -1);
            a.zc[r] = (ca >= 0 && cb >= 0) ? ca * PLD + cb : -1;
            if (rx && cA) {            // f_xx[i][row][c]
                a.pz[r] = fxxt + row * n + c;  a.sz[r] = nn;  a.stz[r] = (int)nnn;
            } else if (rx && cB) {     // (vf_ux)^T: f_ux[i][c-NP][row]
                a.pz[r] = fuxt + (c - NP) * n + row;  a.sz[r] = nm;  a.stz[r] = (int)nmn;
            } else if (ru && cA) {     // f_ux[i][row-NP][c]
                a.pz[r] = fuxt + (row - NP) * n + c;  a.sz[r] = nm;  a.stz[r] = (int)nmn;
            } else if (ru && cB) {     // f_uu[i][row-NP][c-NP]
                a.pz[r] = fuut + (row - NP) * m + (c - NP);  a.sz[r] = mm;  a.stz[r] = (int)nmm;
            } else {                   // padding: finite don't-care data, dropped (zc = -1)
                a.pz[r] = fxxt;  a.sz[r] = 0;  a.stz[r] = (int)nnn;
            }
        }
    }
    a.vL = (g < m) && cA;
    const bool vl = (g < m) && (c == NP);
    a.vOut = a.vL || vl;
    a.pOut = a.vL ? (Lout + last * nm + g * n + c) : (lout + last * m + g);
    a.sOut = vl ? m : nm;

    // LDS addresses of this lane's right-hand side: column c of [Q_ux | .] for c < n, the row q (Q_u) for c == NP
    const bool rhs_qu = (c == NP);
    const int ob0 = rhs_qu ? (64 + NP + 0) : (0 * 16 + c);
    const int ob1 = rhs_qu ? (64 + NP + 1) : (1 * 16 + c);
    const int ob2 = rhs_qu ? (64 + NP + 2) : (2 * 16 + c);
    const int ob3 = rhs_qu ? (64 + NP + 3) : (3 * 16 + c);
    // Q_uu[c][g] as A operand (c < 4): tile row c, column NP+g.  Lanes c >= 4 only feed output rows >= 4 of
    // Q_uu L, which are never used: they read their own (finite) tile element.
    const int oqa = (c < 4) ? (c * 16 + NP + g) : (g * 16 + c);

    // terminal value function
    double Vxx[KS], vxr[KS];
    {
        const double* vxx = shared_h ? vf_xx : vf_xx + traj * svxx;
        const double* vx = vf_x + traj * svx;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int row = 4 * s + g;
            const bool ok = (row < n) && cA;
            const double t = vxx[ok ? row * n + c : 0];
            Vxx[s] = ok ? t : 0.0;
            const double u = vx[row < n ? row : 0];
            vxr[s] = (row < n) ? u : 0.0;
        }
    }

    if (g == 0) sm[80 + c] = cA ? (vf_x + traj * svx)[c] : 0.0;  // v_x (column-indexed), read by MODE 2's contraction
    ilqr_lds_sync();
    IlqrStepRegs<KS> d0, d1;
    ilqr_load_step<KS, MODE>(d0, a);
    if (T >= 2) ilqr_load_step<KS, MODE>(d1, a);
    int k = T - 1;
    while (k >= 3) {
        ilqr_step<KS, MODE, true>(Vxx, vxr, d0, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
        ilqr_step<KS, MODE, true>(Vxx, vxr, d1, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
        k -= 2;
    }
    if (k == 2) {
        ilqr_step<KS, MODE, true>(Vxx, vxr, d0, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
        ilqr_step<KS, MODE, false>(Vxx, vxr, d1, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
        ilqr_step<KS, MODE, false>(Vxx, vxr, d0, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
    } else if (k == 1) {
        ilqr_step<KS, MODE, false>(Vxx, vxr, d0, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
        ilqr_step<KS, MODE, false>(Vxx, vxr, d1, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
    } else {
        ilqr_step<KS, MODE, false>(Vxx, vxr, d0, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
    }
    // optional: the value function the sweep ends with (riccatiStep_ilqr / _ddp return it: ilqrUtils.py:170, :203)
    if (vxx_out) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int row = 4 * s + g;
            if (row < n && cA) vxx_out[traj * nn + row * n + c] = Vxx[s];
        }
    }
    if (vx_out && g == 0 && cA) vx_out[traj * n + c] = sm[80 + c];
    if (v_out) {
        double cs = 0.0;   // sum_k c_k
        if (c_s)
            for (int k2 = lane; k2 < T; k2 += 64) cs += c_s[traj * T + k2];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cs += __shfl_xor(cs, off, 64);
        if (g == 0 && c == NP) v_out[traj] = ((vf_s ? vf_s[traj] : 0.0) + cs) + a.vsum;
    }
}

}  // namespace zm

namespace zm {
struct DdpTensors {
    const double *f_xx, *f_ux, *f_uu;
};
struct ValueIO {   // optional scalar inputs / value-function outputs
    const double *c, *vf;
    double *v_out, *vx_out, *vxx_out;
};

template <int MODE>
static int launch_ilqr(const double* f_x, const double* f_u, const double* c_x, const double* c_u, const double* c_xx,
                       const double* c_ux, const double* c_uu, const double* vf_x, const double* vf_xx, const double* d,
                       long svx, long svxx, const int* act, int sh, double* l, double* L, int64_t batch, int T, int n, int m,
                       hipStream_t st, DdpTensors z = DdpTensors{nullptr, nullptr, nullptr},
                       ValueIO v = ValueIO{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    const dim3 grid((unsigned)batch), block(64);
    if (n <= 4)
        hipLaunchKernelGGL((ilqr_backward_t16_f64<1, MODE>), grid, block, 0, st, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x,
                           vf_xx, d, z.f_xx, z.f_ux, z.f_uu, svx, svxx, act, sh, l, L, T, n, m, v.c, v.vf, v.v_out, v.vx_out, v.vxx_out);
    else if (n <= 8)
        hipLaunchKernelGGL((ilqr_backward_t16_f64<2, MODE>), grid, block, 0, st, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x,
                           vf_xx, d, z.f_xx, z.f_ux, z.f_uu, svx, svxx, act, sh, l, L, T, n, m, v.c, v.vf, v.v_out, v.vx_out, v.vxx_out);
    else
        hipLaunchKernelGGL((ilqr_backward_t16_f64<3, MODE>), grid, block, 0, st, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x,
                           vf_xx, d, z.f_xx, z.f_ux, z.f_uu, svx, svxx, act, sh, l, L, T, n, m, v.c, v.vf, v.v_out, v.vx_out, v.vxx_out);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
}  // namespace zm

static int zm_check_sweep_args(const char* who, int64_t batch, int T, int n, int m) {
    if (batch < 0 || T < 1 || n < 1 || m < 1)
        return zm::set_error(ZM_EINVAL, "%s: bad size batch=%lld T=%d n=%d m=%d", who, (long long)batch, T, n, m);
    if (n > 12 || m > 4) return zm::set_error(ZM_EUNSUPPORTED, "%s: (n=%d, m=%d) not covered (need n<=12, m<=4)", who, n, m);
    if ((int64_t)T * n * n >= (int64_t)1 << 31 || batch >= ((int64_t)1 << 31))
        return zm::set_error(ZM_EUNSUPPORTED, "%s: T*n*n or batch too large", who);
    return ZM_OK;
}

extern "C" int zm_ilqr_backward_ex_f64(const double* f_x, const double* f_u, const double* c_x, const double* c_u,
                                       const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                                       const double* vf_xx, const int32_t* active, int shared_hessian, double* l,
                                       double* L, int64_t batch, int T, int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!f_x || !f_u || !c_x || !c_u || !c_xx || !c_ux || !c_uu || !vf_x || !vf_xx || !l || !L)
        return zm::set_error(ZM_EINVAL, "zm_ilqr_backward_f64: null pointer");
    const int rc = zm_check_sweep_args("zm_ilqr_backward_f64", batch, T, n, m);
    if (rc) return rc;
    if (batch == 0) return ZM_OK;
    return zm::launch_ilqr<0>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n,
                              (const int*)active, shared_hessian ? 1 : 0, l, L, batch, T, n, m, (hipStream_t)stream);
}

extern "C" int zm_ilqr_backward_f64(const double* f_x, const double* f_u, const double* c_x, const double* c_u,
                                    const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                                    const double* vf_xx, double* l, double* L, int64_t batch, int T, int n, int m,
                                    void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    return zm_ilqr_backward_ex_f64(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, 0, l, L, batch, T, n, m,
                                   stream);
}

extern "C" int zm_lqr_backward_affine_f64(const double* A, const double* B, const double* d, const double* Q,
                                          const double* R, const double* H, const double* q, const double* r, double* L,
                                          double* l, int64_t batch, int T, int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!A || !B || !d || !Q || !R || !H || !q || !r || !L || !l)
        return zm::set_error(ZM_EINVAL, "zm_lqr_backward_affine_f64: null pointer");
    const int rc = zm_check_sweep_args("zm_lqr_backward_affine_f64", batch, T, n, m);
    if (rc) return rc;
    if (batch == 0) return ZM_OK;
    // carry (V, v) <- (Q[T-1], q[T-1])   (lqrUtils.py:261): terminal pointers into the last step, trajectory stride T*size
    const double* vf_xx = Q + (int64_t)(T - 1) * n * n;
    const double* vf_x = q + (int64_t)(T - 1) * n;
    return zm::launch_ilqr<1>(A, B, q, r, Q, H, R, vf_x, vf_xx, d, (long)T * n, (long)T * n * n, nullptr, 0, l, L, batch, T,
                              n, m, (hipStream_t)stream);
}

extern "C" int zm_ddp_backward_f64(const double* f_x, const double* f_u, const double* f_xx, const double* f_ux,
                                   const double* f_uu, const double* c_x, const double* c_u, const double* c_xx,
                                   const double* c_ux, const double* c_uu, const double* vf_x, const double* vf_xx,
                                   const int32_t* active, int shared_hessian, double* l, double* L, int64_t batch, int T,
                                   int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!f_x || !f_u || !f_xx || !f_ux || !f_uu || !c_x || !c_u || !c_xx || !c_ux || !c_uu || !vf_x || !vf_xx || !l || !L)
        return zm::set_error(ZM_EINVAL, "zm_ddp_backward_f64: null pointer");
    const int rc = zm_check_sweep_args("zm_ddp_backward_f64", batch, T, n, m);
    if (rc) return rc;
    if ((int64_t)n * n * n >= (int64_t)1 << 31) return zm::set_error(ZM_EUNSUPPORTED, "zm_ddp_backward_f64: n too large");
    if (batch == 0) return ZM_OK;
    return zm::launch_ilqr<2>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n,
                              (const int*)active, shared_hessian ? 1 : 0, l, L, batch, T, n, m, (hipStream_t)stream,
                              zm::DdpTensors{f_xx, f_ux, f_uu});
}

extern "C" int zm_riccati_value_f64(const double* f_x, const double* f_u, const double* f_xx, const double* f_ux,
                                    const double* f_uu, const double* c, const double* c_x, const double* c_u,
                                    const double* c_xx, const double* c_ux, const double* c_uu, const double* vf,
                                    const double* vf_x, const double* vf_xx, double* l, double* L, double* v_out,
                                    double* vx_out, double* vxx_out, int64_t batch, int T, int n, int m, void* stream) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!f_x || !f_u || !c_x || !c_u || !c_xx || !c_ux || !c_uu || !vf_x || !vf_xx || !l || !L)
        return zm::set_error(ZM_EINVAL, "zm_riccati_value_f64: null pointer");
    const bool ddp = f_xx || f_ux || f_uu;
    if (ddp && (!f_xx || !f_ux || !f_uu)) return zm::set_error(ZM_EINVAL, "zm_riccati_value_f64: f_xx, f_ux, f_uu go together");
    const int rc = zm_check_sweep_args("zm_riccati_value_f64", batch, T, n, m);
    if (rc) return rc;
    const zm::ValueIO v{c, vf, v_out, vx_out, vxx_out};
    if (ddp)
        return zm::launch_ilqr<2>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n, nullptr, 0, l,
                                  L, batch, T, n, m, (hipStream_t)stream, zm::DdpTensors{f_xx, f_ux, f_uu}, v);
    return zm::launch_ilqr<0>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n, nullptr, 0, l, L,
                              batch, T, n, m, (hipStream_t)stream, zm::DdpTensors{nullptr, nullptr, nullptr}, v);
}
