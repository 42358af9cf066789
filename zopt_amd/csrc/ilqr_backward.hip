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
#include "dma_ring.h"
#include "jacobi16.h"
#include "models.h"
#include "ns16.h"
#include "tile16_f64.h"
#include "zm_common.h"

#include <cstdlib>

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

// optional compacted list of trajectory ids: block b works on list[b] (grid = count) instead of on trajectory b (grid = batch)
struct TrajList {
    const int* list;
    long count;
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
    bool zlive[4];       // MODE 2: tile element (4r+g, c) is a diagonal entry of the live (state, control) index set
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

// Diagnostic build (-DZM_SWEEP_LAB): s_memtime stamps at the phase boundaries of a step of the ring kernel, summed by wave 0 of block 0
// into zm_sweep_stamps (read back with zm_lab_sweep_stamps).  The stamps' waits forbid overlaps the product has: read SHARES.
#ifdef ZM_SWEEP_LAB
__device__ unsigned long long zm_sweep_stamps[10];
struct SweepLab {
    unsigned long long acc[8], last;
};
#define ZM_SWEEP_STAMP(lab, q)                                                                         \
    do {                                                                                              \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        (lab)->acc[q] += t_ - (lab)->last;                                                            \
        (lab)->last = t_;                                                                             \
    } while (0)
#else
struct SweepLab {};
#define ZM_SWEEP_STAMP(lab, q) do { } while (0)
#endif

__device__ __forceinline__ void ilqr_lds_sync() { wave_lds_sync(); }

// ZIN (MODE 2, the DMA kernel): the caller has contracted vf_zz = sum_i v_x[i] d2f_i/dz2 already (operands in its LDS ring) and
// passes the tile in `zin` (1), or has PD-projected it as well (2)
template <int KS, int MODE, bool PREFETCH, int ZIN = 0>
__device__ __forceinline__ void ilqr_step(double (&Vxx)[KS], double (&vxr)[KS], IlqrStepRegs<KS>& d, IlqrAddr<KS>& a,
                                          double* sm, const int g, const int c, const int ob0, const int ob1,
                                          const int ob2, const int ob3, const int oqa, double* jA, double* jV,
                                          double* jcs, int* jpq, const int n, const int m, const d4 zin = d4{0.0, 0.0, 0.0, 0.0},
                                          SweepLab* lab = nullptr) {
    constexpr int NP = 4 * KS;
    // MODE 2: vf_zz = sum_i v_x[i] * d2f_i/dz2, PD-projected, as extra accumulator init
    d4 pz = zero4();
    if constexpr (MODE == 2 && ZIN == 2) {
        pz = zin;                                    // projected by the caller
    } else if constexpr (MODE == 2 && ZIN == 1) {
        d4 zt = zin;
        psd_project_ns<KS + 1, true>(zt, a.zlive, 1e-3, jA, g, c);   // (the ring kernel's contraction is bitwise symmetric)
        pz = zt;
    } else if constexpr (MODE == 2) {
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
#ifndef ZM_DDP_PSD_JACOBI
        // PD projection on the tile itself by matrix-sign iterations on the fp64 MFMA pipe (ns16.h); jA: transpose buffers
        d4 zt;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a.pz[r] -= a.stz[r];
            zt[r] = (a.zc[r] >= 0) ? z[r] : 0.0;   // padding lanes contracted don't-care data
        }
        psd_project_ns<KS + 1>(zt, a.zlive, 1e-3, jA, g, c);
        pz = zt;
        (void)lane;
#else
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
#endif
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
        vtd = sum_xor16(vtd);
        vtd = sum_xor32(vtd);
        if (g == 0) sm[96 + c] = a.cA ? vtd : 0.0;
        ilqr_lds_sync();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            double vd = Vxx[s] * d.dc;  // (V d)[4s+g]: row reduction over the 16 lanes of the group
            vd = row16_sum(vd);   // (DPP butterflies: the bits of the __shfl_xor ladder over 1, 2, 4, 8 without its eight ds_bpermute)
            const double w = vxr[s] + (a.cA ? vd : sm[96 + 4 * s + g]);
            qp = __builtin_fma(d.F[s], w, qp);
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) qp = __builtin_fma(d.F[s], vxr[s], qp);
    }
    qp = sum_xor16(qp);     // over the four lane groups: gfx950 row / half-wave swaps instead of two ds_bpermute round trips
    qp = sum_xor32(qp);
    const double qv = d.cv + qp;
    if constexpr (PREFETCH) ilqr_load_step<KS, MODE>(d, a);

    ZM_SWEEP_STAMP(lab, 2);   // projection (MODE 2), the six MFMAs of G, the vector terms
    // solve: tile + q row through LDS
    sm[g * 16 + c] = gacc[KS];
    if (g == 0) sm[64 + c] = qv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double S[4][4], b[4], x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) S[i][jj] = sm[i * 16 + NP + jj];
    }
    b[0] = sm[ob0];
    b[1] = sm[ob1];
    b[2] = sm[ob2];
    b[3] = sm[ob3];
    const double quu_a = sm[oqa];  // Q_uu[c][g]: A operand of Q_uu L (0-padded through the address choice below)
    __builtin_amdgcn_wave_barrier();
    const double b0[4] = {b[0], b[1], b[2], b[3]};  // original right-hand side (Sux column) for MODE 1's v' update
    ZM_SWEEP_STAMP(lab, 3);   // the exchange through LDS
    if (__builtin_amdgcn_ballot_w64(!lu_solve4_nopivot(S, b, x)) != 0ull) lu_solve4_fallback(S, b, x);
    {
        double keep = x[0] + x[1] + x[2] + x[3];
        asm volatile("" : "+v"(keep));
    }
    ZM_SWEEP_STAMP(lab, 4);   // the 4 x 4 solve
    const double x01 = (g & 1) ? x[1] : x[0];
    const double x23 = (g & 1) ? x[3] : x[2];
    const double xg = (g & 2) ? x23 : x01;
    const double out = (MODE == 1) ? xg : -xg;  // L_k[g][c] for c < n, l_k[g] for c == NP
    if (a.vOut) *a.pOut = out;
    a.pOut -= a.sOut;
    const double lv = a.vL ? out : 0.0;
    // Q_u (zero beyond m) is read only now: eight registers fewer across the solve; its LDS row is rewritten in the next step
    double qu[4];
    if constexpr (MODE != 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) qu[i] = sm[64 + NP + i];
    }
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
    ZM_SWEEP_STAMP(lab, 5);   // store, value terms, the two MFMAs of V', v_x through LDS
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
    double* __restrict__ vx_out, double* __restrict__ vxx_out, const TrajList tl, const double* __restrict__ Hpk,
    const PairTab ptab) {
    constexpr int NP = 4 * KS;
    const int lane = threadIdx.x;
    const long traj = tl.list ? (long)tl.list[blockIdx.x] : (long)blockIdx.x;
    if (active && active[traj] == 0) return;  // whole wave leaves: this trajectory keeps its previous policy
    const int g = lane >> 4, c = lane & 15;
    __shared__ double sm[ILQR_LDS_DOUBLES];
#ifndef ZM_DDP_PSD_JACOBI
    __shared__ double jA[MODE == 2 ? NS_LDS_DOUBLES : 1], jV[1], jcs[1];
#else
    __shared__ double jA[MODE == 2 ? PK * PLD : 1], jV[MODE == 2 ? PK * PLD : 1], jcs[PK];
#endif
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
            a.zlive[r] = (ca >= 0) && (row == c);
            if (Hpk != nullptr) {
                // packed second derivatives (zm_quadratic_dynamics_pairs_list_f64): element (ca, cb) is pair p = (min, max) of the
                // model's table or structurally zero; H[pt][p][i], i contiguous -> the contraction below runs unchanged with a
                // unit stride over i and the same summation order as over the full tensors
                int pidx = -1;
                if (ca >= 0 && cb >= 0) {
                    const int lo = ca < cb ? ca : cb, hi = ca < cb ? cb : ca;
                    for (int q = 0; q < ptab.n; ++q) pidx = (ptab.ab[q] == (unsigned char)(lo * 16 + hi)) ? q : pidx;
                }
                const double* Hl = Hpk + last * (long)ptab.n * n;
                if (pidx >= 0) {
                    a.pz[r] = Hl + (long)pidx * n;  a.sz[r] = 1;  a.stz[r] = ptab.n * n;
                } else {   // structurally zero (or padding): the lane contracts don't-care data and contributes an exact 0.0
                    a.pz[r] = Hl;  a.sz[r] = 0;  a.stz[r] = ptab.n * n;
                    a.zc[r] = -1;
                }
            } else if (f_ux == nullptr && !(rx && cA)) {   // dynamics affine in the controls: f_ux, f_uu are zero and not materialised
                a.zc[r] = -1;
                a.pz[r] = fxxt;  a.sz[r] = 0;  a.stz[r] = (int)nnn;
            } else if (rx && cA) {     // f_xx[i][row][c]
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
    if constexpr (MODE == 2) {
        // a DDP step is ~100 matrix products long: operand prefetch buys nothing and its registers are needed elsewhere
        for (int k2 = T - 1; k2 >= 0; --k2) {
            IlqrStepRegs<KS> d;
            ilqr_load_step<KS, MODE>(d, a);
            ilqr_step<KS, MODE, false>(Vxx, vxr, d, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, jV, jcs, jpq, n, m);
        }
    } else {
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

// ---------------------------------------------------------------------------------------------------------------------
// K3-DMA: the iLQR sweep (MODE 0) with its per-step operands staged through an LDS ring by DMA, as K1 does
// (lqr_backward_dma.hip): no operand prefetch registers, no per-step address arithmetic for loads, no vector-memory
// latency on the chain.  n in {8, 12}, m = 4, 16-B aligned pointers; every other case runs ilqr_backward_t16_f64.
// Cache policy of the operand DMA (sc0 = 1, nt = 2, sc1 = 16).  Default policy here: unlike K1 (lqr_backward_dma.hip, where
// non-temporal loads are worth 10 %), the sweeps measured 7 % SLOWER with nt at 8192 x 100 steps (per-step Hessians 616 -> 660 us,
// affine 722 -> 754 us, shared Hessians unchanged; gpurun_out/r02_ilqr_nt_ab.txt).  -DZM_ILQR_DMA_AUX=2 builds the nt variant.
#ifndef ZM_ILQR_DMA_AUX
#define ZM_ILQR_DMA_AUX 0
#endif

// Slot image of one step (16-B chunks): f_x | f_u | c_x | c_u [| c_xx | c_ux | c_uu unless the Hessians are shared] | zeros.
namespace zm {

// NJP > 0: [f_x | f_u] arrives PACKED -- NJP doubles per step (the structurally nonzero entries of the model's Jacobian, positions in a
// JacTab; everything else is the identity's 0 or 1, served from two constant chunks in the slot's padding) instead of N (N + M).
struct JacTab {
    unsigned char pos[12 * 16];   // packed position of entry (i, j) of [f_x | f_u], 255: not stored
};
// NHS > 0 (MODE 2): the second derivatives arrive SPARSE -- NHS doubles per step (their structurally nonzero entries, dense positions in
// an HTab), scattered every step into a dense [NPAIR][N] image that lives once in LDS (its other entries stay zero).
struct HTab {
    unsigned short dense[96];   // dense index pair * N + i of sparse entry k
    int n;                      // entries (<= 96)
};
template <int N, int M, bool SHARED, int MODE, int NPAIR = 0, int NJP = 0, int NHS = 0>
struct IlqrDmaGeom {
    static constexpr int KS = N / 4;
    static constexpr int CFX = NJP ? NJP / 2 : N * N / 2, CFU = NJP ? 0 : N * M / 2, CCX = N / 2, CCU = M / 2;
    static constexpr int CXX = SHARED ? 0 : N * N / 2, CUX = SHARED ? 0 : N * M / 2, CUU = SHARED ? 0 : M * M / 2;
    static constexpr int CD = (MODE == 1) ? N / 2 : 0;   // MODE 1: the affine term d of the dynamics
    static constexpr int CH = (MODE == 2) ? (NHS ? NHS / 2 : NPAIR * N / 2) : 0;   // MODE 2: the second derivatives H[pair][i] of the step
    static constexpr int CT = CFX + CFU + CCX + CCU + CXX + CUX + CUU + CD + CH;
    static constexpr int NI = (CT + 63) / 64;
    static constexpr int SLOT = NI * 1024;
    static constexpr int OFX = 0, OFU = CFX * 16, OCX = OFU + CFU * 16, OCU = OCX + CCX * 16;
    static constexpr int OXX = OCU + CCU * 16, OUX = OXX + CXX * 16, OUU = OUX + CUX * 16;
    static constexpr int OD = OUU + CUU * 16;
    static constexpr int OH = OD + CD * 16;
    static constexpr int OZ = CT * 16;
    static constexpr int OONE = OZ + 16;   // NJP: a chunk of ones behind the chunk of zeros
    static_assert(NJP % 2 == 0 && (NJP == 0 || (N == 12 && SHARED)), "packed Jacobians: n = 12, 16-B rows, shared cost Hessians");
    static_assert(NJP == 0 || SLOT - OZ >= 32, "slot needs a zero and a one chunk");
    static_assert(N % 4 == 0 && N >= 8 && N <= 12 && M == 4, "fast path: all K-step rows live, m = 4");
    static_assert(MODE == 0 || (MODE == 1 && !SHARED) || (MODE == 2 && SHARED && N == 12 && NPAIR > 0 && NPAIR <= ZM_MAX_PAIRS),
                  "iLQR, affine LQR, or DDP with shared cost Hessians and packed second derivatives");
    static_assert(SLOT - OZ >= 16, "slot needs zero padding");
};

#ifndef ZM_DDP_DMA_WAVES
#define ZM_DDP_DMA_WAVES 2
#endif
#ifndef ZM_ILQR_DMA_WAVES
#define ZM_ILQR_DMA_WAVES 3
#endif
// W: waves per workgroup, wave w of block b works on slot b * W + w (list form only).  The waves share nothing; what W = 4 buys is
// PLACEMENT when few trajectories are left: the four waves of a workgroup go to the four SIMDs of a CU, whereas single-wave workgroups
// start doubling up on SIMDs beyond three per CU (768 waves: measured 148 against 119 us per sweep at 808 resp. 758 trajectories).
template <int N, int M, int D, bool SHARED, int MODE, int NPAIR = 0, int NJP = 0, int NHS = 0, int W = 1>
__global__ __launch_bounds__(64 * W, MODE == 2 ? ZM_DDP_DMA_WAVES : ZM_ILQR_DMA_WAVES) void ilqr_backward_dma_f64(
    const double* __restrict__ f_x, const double* __restrict__ f_u, const double* __restrict__ c_x,
    const double* __restrict__ c_u, const double* __restrict__ c_xx, const double* __restrict__ c_ux,
    const double* __restrict__ c_uu, const double* __restrict__ vf_x, const double* __restrict__ vf_xx,
    const double* __restrict__ dvec, const long svx, const long svxx, const int* __restrict__ active,
    double* __restrict__ lout, double* __restrict__ Lout, const int T, const TrajList tl, const double* __restrict__ Hpk,
    const PairTab ptab, const JacTab jtab, const HTab htab) {
    using G = IlqrDmaGeom<N, M, SHARED, MODE, NPAIR, NJP, NHS>;
    static_assert(NHS % 2 == 0 && NHS <= 96 && (NHS == 0 || MODE == 2), "sparse second derivatives: DDP mode, 16-B rows");
    constexpr int KS = G::KS, NI = G::NI, SLOT = G::SLOT, NP = N;
    constexpr int nn = N * N, nm = N * M, mm = M * M;
    constexpr int SMO = D * SLOT;
    // MODE 2 with three waves per SIMD (ZM_DDP_DMA_WAVES = 3) keeps state in LDS while the projection runs: X of the sign iteration
    // (256 doubles), the value function V, v (6 doubles per lane) and the shared cost Hessian (4 per lane)
    constexpr bool DIET = (MODE == 2) && (ZM_DDP_DMA_WAVES >= 3);
    constexpr int XST = DIET ? 256 : 0, VST = DIET ? 64 * 2 * KS : 0, CST = DIET ? 64 * (KS + 1) : 0;
#ifndef ZM_DDP_LDS_PAD   // occupancy experiments: extra bytes of LDS per wave in MODE 2
#define ZM_DDP_LDS_PAD 0
#endif
    constexpr int HDN = NHS ? NPAIR * N : 0;   // dense image of the sparse second derivatives
    constexpr int PER_WAVE = SMO + ILQR_LDS_DOUBLES * 8 + (MODE == 2 ? NS_LDS_DOUBLES * 8 + ZM_DDP_LDS_PAD : 0) + (XST + VST + CST + HDN) * 8;
    static_assert(PER_WAVE % 16 == 0, "per-wave LDS slices stay 16-B aligned");
    __shared__ __attribute__((aligned(16))) char lds_all[W * PER_WAVE];
    const int wave = (W == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char* const lds = lds_all + wave * PER_WAVE;
    double* sm = (double*)(lds + SMO);
    double* jA = (double*)(lds + SMO + ILQR_LDS_DOUBLES * 8);   // MODE 2: transpose buffers of the sign iteration (ns16.h)
    double* xst = jA + NS_LDS_DOUBLES;
    double* vst = xst + XST;
    double* cst = vst + VST;
    double* hd = cst + CST;
    const int lane = threadIdx.x & 63;
    const long slot_ = (long)blockIdx.x * W + wave;
    if (W > 1 && slot_ >= tl.count) return;    // (W > 1: list form; whole waves only, no barrier anywhere in this kernel)
    const long traj = tl.list ? (long)tl.list[slot_] : slot_;
    if (active && active[traj] == 0) return;   // whole wave leaves: this trajectory keeps its previous policy
    const int g = lane >> 4, c = lane & 15;
    const bool cA = c < N, cB = (c >= N) && (c < N + M);

    // DMA sources: chunk q of the step image -> (array, offset)
    const char* p[NI];
    int st[NI];
    {
        const long last = traj * T + (T - 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int q = i * 64 + lane;
            const char* src = (NJP && q == G::CT + 1) ? (const char*)zm_one_src : (const char*)zm_zero_src;
            int stride = 0;
            if (q < G::CFX) {
                if constexpr (NJP != 0) {   // f_x points to the packed images
                    src = (const char*)(f_x + last * NJP) + q * 16;  stride = NJP * 8;
                } else {
                    src = (const char*)(f_x + last * nn) + q * 16;  stride = nn * 8;
                }
            } else if ((q -= G::CFX) < G::CFU) {
                src = (const char*)(f_u + last * nm) + q * 16;  stride = nm * 8;
            } else if ((q -= G::CFU) < G::CCX) {
                src = (const char*)(c_x + last * N) + q * 16;   stride = N * 8;
            } else if ((q -= G::CCX) < G::CCU) {
                src = (const char*)(c_u + last * M) + q * 16;   stride = M * 8;
            } else if constexpr (MODE == 2) {
                if ((q -= G::CCU) < G::CH) {
                    constexpr int HROW = NHS ? NHS : NPAIR * N;   // doubles per step in HBM
                    src = (const char*)(Hpk + last * (long)HROW) + q * 16;  stride = HROW * 8;
                }
            } else if constexpr (!SHARED) {
                if ((q -= G::CCU) < G::CXX) {
                    src = (const char*)(c_xx + last * nn) + q * 16;  stride = nn * 8;
                } else if ((q -= G::CXX) < G::CUX) {
                    src = (const char*)(c_ux + last * nm) + q * 16;  stride = nm * 8;
                } else if ((q -= G::CUX) < G::CUU) {
                    src = (const char*)(c_uu + last * mm) + q * 16;  stride = mm * 8;
                } else if constexpr (MODE == 1) {
                    if ((q -= G::CUU) < G::CD) {
                        src = (const char*)(dvec + last * N) + q * 16;  stride = N * 8;
                    }
                }
            }
            p[i] = src;
            st[i] = stride;
        }
    }
    auto dma = [&](char* slot) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)p[i], (lds_void_t*)(slot + i * 1024), 16, 0, ZM_ILQR_DMA_AUX);
            p[i] -= st[i];
        }
    };
    // LDS offsets of this lane's operands inside a slot (lanes outside a matrix read the zero padding)
    const int oF = cA ? (G::OFX + (g * N + c) * 8) : cB ? (G::OFU + (g * M + (c - N)) * 8) : G::OZ;
    const int dF = cA ? 4 * N * 8 : cB ? 4 * M * 8 : 0;
    int oFp[KS];   // NJP: slot offset of F[4s+g][c] -- its packed entry, or the chunk of zeros / ones
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int row = 4 * s + g;
        const int p = NJP ? (int)jtab.pos[row * 16 + c] : 255;
        oFp[s] = (p != 255) ? (G::OFX + p * 8) : ((row == c) ? G::OONE : G::OZ);
    }
    const int ocv = cA ? (G::OCX + c * 8) : cB ? (G::OCU + (c - N) * 8) : G::OZ;
    const int oC = cA ? (G::OXX + (g * N + c) * 8) : G::OZ;       // !SHARED only
    const int dC = cA ? 4 * N * 8 : 0;
    const int oCu = cA ? (G::OUX + (g * N + c) * 8) : cB ? (G::OUU + (g * M + (c - N)) * 8) : G::OZ;   // (g < M always)
    const int odc = cA ? (G::OD + c * 8) : G::OZ;                  // MODE 1: d[c]; d[4 s + g] is read at OD + (4 s + g) 8

    // the pieces of IlqrAddr the step still uses: outputs and lane roles
    IlqrAddr<KS> a;
    a.cA = cA;
    a.vL = cA;                                  // (g < m always: M == 4)
    const bool vl = (c == NP);
    a.vOut = a.vL || vl;
    {
        const long last = traj * T + (T - 1);
        a.pOut = a.vL ? (Lout + last * nm + g * N + c) : (lout + last * M + g);
    }
    a.sOut = vl ? M : nm;
    // MODE 2: tile element (4r+g, c) of vf_zz is pair (min, max) of the model's table (row of the packed image) or structurally zero
    int oH[4];
    bool zok[4];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + g;
            const int lo = row < c ? row : c, hi = row < c ? c : row;
            int pidx = -1;
            for (int q = 0; q < NPAIR; ++q) pidx = (ptab.ab[q] == (unsigned char)(lo * 16 + hi)) ? q : pidx;
            zok[r] = pidx >= 0;
            oH[r] = (NHS ? 0 : G::OH) + (pidx >= 0 ? pidx : 0) * (N * 8);   // NHS: relative to the dense image, else to the slot
            a.zlive[r] = (row == c);   // n + m = 16: every tile index is a live (state or control) index
        }
    }
    const bool rhs_qu = (c == NP);
    const int ob0 = rhs_qu ? (64 + NP + 0) : (0 * 16 + c);
    const int ob1 = rhs_qu ? (64 + NP + 1) : (1 * 16 + c);
    const int ob2 = rhs_qu ? (64 + NP + 2) : (2 * 16 + c);
    const int ob3 = rhs_qu ? (64 + NP + 3) : (3 * 16 + c);
    const int oqa = (c < 4) ? (c * 16 + NP + g) : (g * 16 + c);

    // shared (time-invariant) cost Hessian: once, into registers
    double Csh[KS], Cush = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) Csh[s] = 0.0;
    if constexpr (SHARED) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const double t = c_xx[cA ? (4 * s + g) * N + c : 0];
            Csh[s] = cA ? t : 0.0;
        }
        const double tu = cA ? c_ux[g * N + c] : (cB ? c_uu[g * M + (c - N)] : 0.0);
        Cush = tu;
    }
    // terminal value function
    double Vxx[KS], vxr[KS];
    {
        const double* vxx = SHARED ? vf_xx : vf_xx + traj * svxx;
        const double* vx = vf_x + traj * svx;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int row = 4 * s + g;
            const double t = vxx[cA ? row * N + c : 0];
            Vxx[s] = cA ? t : 0.0;
            vxr[s] = vx[row];
        }
        if (g == 0) sm[80 + c] = cA ? vx[c] : 0.0;
    }
    if constexpr (DIET) {
#pragma unroll
        for (int s = 0; s < KS; ++s) cst[s * 64 + lane] = Csh[s];
        cst[KS * 64 + lane] = Cush;
    }
    int hdst0 = 0, hdst1 = 0;   // NHS: dense positions of sparse entries lane, lane + 64
    if constexpr (NHS != 0) {
        for (int e = lane; e < HDN; e += 64) hd[e] = 0.0;
        hdst0 = (lane < htab.n) ? (int)htab.dense[lane] : -1;
        hdst1 = (lane + 64 < htab.n) ? (int)htab.dense[(lane + 64) % 96] : -1;
    }
    ilqr_lds_sync();

#pragma unroll
    for (int i = 0; i < D; ++i)
        if (T - 1 - i >= 0) dma(lds + i * SLOT);
    int j = T - 1;
#ifdef ZM_SWEEP_LAB
    SweepLab labv;
    for (int q = 0; q < 8; ++q) labv.acc[q] = 0ull;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(labv.last)::"memory");
    SweepLab* const lab = &labv;
#else
    SweepLab* const lab = nullptr;
#endif
    for (;;) {
#pragma unroll
        for (int si = 0; si < D; ++si) {
            char* slot = lds + si * SLOT;
            wait_for_step<NI, D>(j);
            ZM_SWEEP_STAMP(lab, 0);   // waiting for the step's DMA
            IlqrStepRegs<KS> d;
            if constexpr (!DIET) {
#pragma unroll
                for (int s = 0; s < KS; ++s) d.F[s] = *(const double*)(slot + (NJP ? oFp[s] : oF + s * dF));
                d.cv = *(const double*)(slot + ocv);
            }
            if constexpr (DIET) {
                // (operands are read after the projection)
            } else if constexpr (SHARED) {
#pragma unroll
                for (int s = 0; s < KS; ++s) d.C[s] = Csh[s];
                d.Cu = Cush;
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) d.C[s] = *(const double*)(slot + oC + s * dC);
                d.Cu = *(const double*)(slot + oCu);
            }
            if constexpr (MODE == 1) {
                d.dc = *(const double*)(slot + odc);
#pragma unroll
                for (int s = 0; s < KS; ++s) d.dr[s] = *(const double*)(slot + G::OD + (4 * s + g) * 8);
            }
            d4 zin = zero4();
            if constexpr (NHS != 0) {   // this step's nonzero second derivatives into the dense image
                const double* hs = (const double*)(slot + G::OH);
                const double h0 = hs[lane], h1 = hs[(lane + 64 < NHS) ? lane + 64 : 0];
                if (hdst0 >= 0) hd[hdst0] = h0;
                if (hdst1 >= 0) hd[hdst1] = h1;
                ilqr_lds_sync();
            }
            if constexpr (DIET) {
                // (see below) the projection first, with as little as possible alive: V, v wait in LDS, the step's operands are
                // read from the ring afterwards -- the slot is refilled one projection later than otherwise, still a step ahead
                double vx[N];
#pragma unroll
                for (int i = 0; i < N; ++i) vx[i] = sm[80 + i];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double* h = NHS ? (const double*)((const char*)hd + oH[r]) : (const double*)(slot + oH[r]);
                    double z = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) z = __builtin_fma(vx[i], h[i], z);
                    zin[r] = zok[r] ? z : 0.0;
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    vst[s * 64 + lane] = Vxx[s];
                    vst[(KS + s) * 64 + lane] = vxr[s];
                }
                ilqr_lds_sync();
                psd_project_ns<KS + 1>(zin, a.zlive, 1e-3, jA, g, c, xst);
                ilqr_lds_sync();
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    Vxx[s] = vst[s * 64 + lane];
                    vxr[s] = vst[(KS + s) * 64 + lane];
                    d.F[s] = *(const double*)(slot + (NJP ? oFp[s] : oF + s * dF));
                    d.C[s] = cst[s * 64 + lane];
                }
                d.cv = *(const double*)(slot + ocv);
                d.Cu = cst[KS * 64 + lane];
            } else if constexpr (MODE == 2) {
                // vf_zz[4r+g][c] = sum_i v_x[i] H[pair][i], i ascending (the order of the register kernel's contraction); v_x is
                // this wave's LDS row sm[80 ..], written by the previous step
                double vx[N];
#pragma unroll
                for (int i = 0; i < N; ++i) vx[i] = sm[80 + i];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double* h = NHS ? (const double*)((const char*)hd + oH[r]) : (const double*)(slot + oH[r]);
                    double z = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) z = __builtin_fma(vx[i], h[i], z);
                    zin[r] = zok[r] ? z : 0.0;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // operands are in registers: the slot may be refilled
            if (j - D >= 0) dma(slot);
            ZM_SWEEP_STAMP(lab, 1);   // operand reads (and MODE 2's contraction), DMA issue
            if constexpr (DIET)
                ilqr_step<KS, MODE, false, 2>(Vxx, vxr, d, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, nullptr, nullptr, nullptr, N, M, zin);
            else if constexpr (MODE == 2)
                ilqr_step<KS, MODE, false, 1>(Vxx, vxr, d, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, jA, nullptr, nullptr, nullptr, N, M, zin, lab);
            else
                ilqr_step<KS, MODE, false>(Vxx, vxr, d, a, sm, g, c, ob0, ob1, ob2, ob3, oqa, nullptr, nullptr, nullptr, nullptr, N, M, zero4(), lab);
            if (--j < 0) {
#ifdef ZM_SWEEP_LAB
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    for (int q = 0; q < 8; ++q) atomicAdd(&zm_sweep_stamps[q], labv.acc[q]);
                    atomicAdd(&zm_sweep_stamps[8], 1ull);
                }
#endif
                return;
            }
        }
    }
}

// ZM_EUNSUPPORTED unless (n in {8, 12}, m = 4, 16-B aligned pointers): the caller then runs the register-prefetch kernel
template <int MODE>
static int ilqr_backward_dma_dispatch(const double* f_x, const double* f_u, const double* c_x, const double* c_u,
                                      const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                                      const double* vf_xx, const double* dvec, long svx, long svxx, const int* act, int sh,
                                      double* l, double* L, int64_t batch, int T, int n, int m, hipStream_t st,
                                      TrajList tl = TrajList{nullptr, 0}) {
    static const bool off = [] {
        const char* e = zm::fallback_env("ZOPT_AMD_ILQR_PATH");
        return e && e[0] == 'r';   // "reg": force the register-prefetch kernel
    }();
    if (off || m != 4 || (n != 8 && n != 12)) return ZM_EUNSUPPORTED;
    const uintptr_t al = (uintptr_t)f_x | (uintptr_t)f_u | (uintptr_t)c_x | (uintptr_t)c_u | (uintptr_t)dvec |
                         (sh ? 0 : ((uintptr_t)c_xx | (uintptr_t)c_ux | (uintptr_t)c_uu));
    if (al & 15) return ZM_EUNSUPPORTED;
    const dim3 grid((unsigned)(tl.list ? tl.count : batch)), block(64);
    // ring depth: 3 steps in flight when the step image is 2 KB (shared Hessians), 2 when it is 4 KB (measured: profiles/)
    constexpr int DS = 3, DF = 2;
#define ZM_LAUNCH_ILQR_DMA(NN, DD, SH)                                                                                     \
    hipLaunchKernelGGL((ilqr_backward_dma_f64<NN, 4, DD, SH, MODE>), grid, block, 0, st, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, \
                       vf_x, vf_xx, dvec, svx, svxx, act, l, L, T, tl, (const double*)nullptr, PairTab{}, JacTab{}, HTab{})
    if constexpr (MODE == 0) {
        if (n == 12) {
            if (sh) ZM_LAUNCH_ILQR_DMA(12, DS, true); else ZM_LAUNCH_ILQR_DMA(12, DF, false);
        } else {
            if (sh) ZM_LAUNCH_ILQR_DMA(8, DS, true); else ZM_LAUNCH_ILQR_DMA(8, DF, false);
        }
    } else {
        if (n == 12) ZM_LAUNCH_ILQR_DMA(12, DF, false); else ZM_LAUNCH_ILQR_DMA(8, DF, false);
    }
#undef ZM_LAUNCH_ILQR_DMA
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// K4-DMA: the DDP sweep with shared cost Hessians and the PACKED second derivatives (the solver's form) on the same LDS ring -- the
// step image grows by the pairs' rows (28 x 12 doubles for the quadcopter: 4.3 KB per step), the contraction reads them from LDS.
// ZM_EUNSUPPORTED unless (n = 12, m = 4, 28 declared pairs, 16-B aligned pointers): the caller then runs ilqr_backward_t16_f64.
static int ddp_backward_dma_dispatch(const double* f_x, const double* f_u, const double* Hpk, const PairTab& ptab, const double* c_x,
                                     const double* c_u, const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                                     const double* vf_xx, long svx, const int* act, int sh, double* l, double* L, int64_t batch, int T,
                                     int n, int m, hipStream_t st, TrajList tl) {
    static const bool off = [] {
        const char* e = zm::fallback_env("ZOPT_AMD_ILQR_PATH");
        return e && e[0] == 'r';   // "reg": force the register kernel
    }();
    if (off || !sh || n != 12 || m != 4 || ptab.n != 28 || !Hpk) return ZM_EUNSUPPORTED;
    if (((uintptr_t)f_x | (uintptr_t)f_u | (uintptr_t)c_x | (uintptr_t)c_u | (uintptr_t)Hpk) & 15) return ZM_EUNSUPPORTED;
    const dim3 grid((unsigned)(tl.list ? tl.count : batch)), block(64);
    hipLaunchKernelGGL((ilqr_backward_dma_f64<12, 4, 2, true, 2, 28>), grid, block, 0, st, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x,
                       vf_xx, (const double*)nullptr, svx, 0L, act, l, L, T, tl, Hpk, ptab, JacTab{}, HTab{});
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// Solver-internal (ilqr_solve.hip): the iLQR (Hpk == nullptr) or DDP sweep of the listed trajectories with PACKED Jacobians Fp
// (njp doubles per step: 56 or 60, positions `pos`; written by linearize.hip's packed expansion), shared cost Hessians, n = 12, m = 4.
// nhs > 0: Hpk holds the SPARSE second derivatives, nhs doubles per step (70 or 86), dense positions `hdense` (nh of them).
int sweep_packed_jacobians(const double* Fp, int njp, const unsigned char* pos, const double* Hpk, const PairTab& ptab, const double* c_x,
                           const double* c_u, const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                           const double* vf_xx, const int* act, double* l, double* L, int64_t batch, int T, hipStream_t st, TrajList tl,
                           int nhs, const unsigned short* hdense, int nh) {
    if (((uintptr_t)Fp | (uintptr_t)c_x | (uintptr_t)c_u | (uintptr_t)Hpk) & 15) return ZM_EUNSUPPORTED;
    if ((njp != 56 && njp != 60) || (Hpk && ptab.n != 28)) return ZM_EUNSUPPORTED;
    if (nhs && !(Hpk && ((njp == 56 && nhs == 70) || (njp == 60 && nhs == 86)) && hdense && nh > 0 && nh <= nhs)) return ZM_EUNSUPPORTED;
    JacTab jt;
    for (int e = 0; e < 192; ++e) jt.pos[e] = pos[e];
    HTab ht{};
    for (int e = 0; e < nh && nhs; ++e) ht.dense[e] = hdense[e];
    ht.n = nhs ? nh : 0;
    const long nslot = tl.list ? tl.count : batch;
    // few trajectories left (at most one wave per SIMD): four-wave workgroups, one wave per SIMD of a CU
    static const long wg4_max = [] {   // ZOPT_AMD_SWEEP_WG4=<n>: four-wave workgroups up to n listed trajectories (A/B)
        const char* e = zm::lab_env("ZOPT_AMD_SWEEP_WG4");
        return e ? atol(e) : 2048L;   // (round 3, same-box A/B of the 8192-problem solve: 34.8-34.9 ms at 1024, 34.3-34.5 at 2048, 34.2-34.7 at 3072 / 4096)
    }();
    const bool quad_wg = tl.list && nslot <= wg4_max;
    const dim3 grid((unsigned)(quad_wg ? (nslot + 3) / 4 : nslot)), block(quad_wg ? 256 : 64);
#define ZM_LAUNCH_PACKED(DD, MODE_, NPAIR_, NJP_, NHS_)                                                                               \
    do {                                                                                                                              \
        if (quad_wg)                                                                                                                  \
            hipLaunchKernelGGL((ilqr_backward_dma_f64<12, 4, DD, true, MODE_, NPAIR_, NJP_, NHS_, 4>), grid, block, 0, st, Fp,         \
                               (const double*)nullptr, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, (const double*)nullptr, 12L, 0L, act, l, \
                               L, T, tl, Hpk, ptab, jt, ht);                                                                          \
        else                                                                                                                          \
            hipLaunchKernelGGL((ilqr_backward_dma_f64<12, 4, DD, true, MODE_, NPAIR_, NJP_, NHS_>), grid, block, 0, st, Fp,            \
                               (const double*)nullptr, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, (const double*)nullptr, 12L, 0L, act, l, \
                               L, T, tl, Hpk, ptab, jt, ht);                                                                          \
    } while (0)
    if (Hpk && nhs) {
        if (njp == 56) ZM_LAUNCH_PACKED(2, 2, 28, 56, 70); else ZM_LAUNCH_PACKED(2, 2, 28, 60, 86);
    } else if (Hpk) {
        if (njp == 56) ZM_LAUNCH_PACKED(2, 2, 28, 56, 0); else ZM_LAUNCH_PACKED(2, 2, 28, 60, 0);
    } else {
        if (njp == 56) ZM_LAUNCH_PACKED(3, 0, 0, 56, 0); else ZM_LAUNCH_PACKED(3, 0, 0, 60, 0);
    }
#undef ZM_LAUNCH_PACKED
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

}  // namespace zm

namespace zm {
struct DdpTensors {
    const double *f_xx, *f_ux, *f_uu;
    const double* Hpk = nullptr;   // packed second derivatives of the model's declared pairs (instead of the three tensors)
    PairTab ptab = PairTab{};
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
                       ValueIO v = ValueIO{nullptr, nullptr, nullptr, nullptr, nullptr}, TrajList tl = TrajList{nullptr, 0}) {
    const dim3 grid((unsigned)(tl.list ? tl.count : batch)), block(64);
    if (n <= 4)
        hipLaunchKernelGGL((ilqr_backward_t16_f64<1, MODE>), grid, block, 0, st, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x,
                           vf_xx, d, z.f_xx, z.f_ux, z.f_uu, svx, svxx, act, sh, l, L, T, n, m, v.c, v.vf, v.v_out, v.vx_out, v.vxx_out, tl, z.Hpk, z.ptab);
    else if (n <= 8)
        hipLaunchKernelGGL((ilqr_backward_t16_f64<2, MODE>), grid, block, 0, st, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x,
                           vf_xx, d, z.f_xx, z.f_ux, z.f_uu, svx, svxx, act, sh, l, L, T, n, m, v.c, v.vf, v.v_out, v.vx_out, v.vxx_out, tl, z.Hpk, z.ptab);
    else
        hipLaunchKernelGGL((ilqr_backward_t16_f64<3, MODE>), grid, block, 0, st, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x,
                           vf_xx, d, z.f_xx, z.f_ux, z.f_uu, svx, svxx, act, sh, l, L, T, n, m, v.c, v.vf, v.v_out, v.vx_out, v.vxx_out, tl, z.Hpk, z.ptab);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
}  // namespace zm

namespace zm {
// sweep_tiled_f64.hip: the iLQR (mode 0) / bilinear-affine (mode 1) sweeps on fp64 MFMA tiles for n <= 48, m <= 16
int sweep_tiled_f64_dispatch(int mode, const double* f_x, const double* f_u, const double* c_x, const double* c_u, const double* c_xx,
                             const double* c_ux, const double* c_uu, const double* vf_x, const double* vf_xx, const double* dvec,
                             long svx, long svxx, const int* active, int shared_hessian, double* l, double* L, int64_t batch, int T,
                             int n, int m, hipStream_t st, const int* list, long count, const double* cs = nullptr,
                             const double* vf = nullptr, double* v_out = nullptr, double* vx_out = nullptr, double* vxx_out = nullptr);
}  // namespace zm

// tiled: the entry point also has the large-state tile kernel (n <= 48, m <= 16) behind it
static int zm_check_sweep_args(const char* who, int64_t batch, int T, int n, int m, bool tiled = false) {
    if (batch < 0 || T < 1 || n < 1 || m < 1)
        return zm::set_error(ZM_EINVAL, "%s: bad size batch=%lld T=%d n=%d m=%d", who, (long long)batch, T, n, m);
    if (tiled ? (n > 48 || m > 16) : (n > 12 || m > 4))
        return zm::set_error(ZM_EUNSUPPORTED, "%s: (n=%d, m=%d) not covered (need %s)", who, n, m,
                             tiled ? "n<=48, m<=16" : "n<=12, m<=4");
    if ((int64_t)T * n * n >= (int64_t)1 << 31 || batch >= ((int64_t)1 << 31))
        return zm::set_error(ZM_EUNSUPPORTED, "%s: T*n*n or batch too large", who);
    return ZM_OK;
}

extern "C" int zm_ilqr_backward_list_f64(const double* f_x, const double* f_u, const double* c_x, const double* c_u,
                                         const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                                         const double* vf_xx, const int32_t* list, int64_t count, const int32_t* active,
                                         int shared_hessian, double* l, double* L, int64_t batch, int T, int n, int m,
                                         void* stream) {
    if (batch == 0 || (list && count == 0)) return ZM_OK;   /* nothing to do (pointers of empty arrays may be NULL) */
    if (list && (count < 0 || count > batch)) return zm::set_error(ZM_EINVAL, "zm_ilqr_backward_list_f64: bad list length");
    const zm::TrajList tl{(const int*)list, (long)count};
    if (!f_x || !f_u || !c_x || !c_u || !c_xx || !c_ux || !c_uu || !vf_x || !vf_xx || !l || !L)
        return zm::set_error(ZM_EINVAL, "zm_ilqr_backward_f64: null pointer");
    const int rc = zm_check_sweep_args("zm_ilqr_backward_f64", batch, T, n, m, true);
    if (rc) return rc;
    if (n > 12 || m > 4)   // large states: fp64 MFMA tile sweep (sweep_tiled_f64.hip)
        return zm::sweep_tiled_f64_dispatch(0, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n,
                                            (const int*)active, shared_hessian ? 1 : 0, l, L, batch, T, n, m, (hipStream_t)stream,
                                            (const int*)list, (long)count);
    if (zm::ilqr_backward_dma_dispatch<0>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n,
                                       (const int*)active, shared_hessian ? 1 : 0, l, L, batch, T, n, m,
                                       (hipStream_t)stream, tl) == ZM_OK)
        return ZM_OK;
    return zm::launch_ilqr<0>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n,
                              (const int*)active, shared_hessian ? 1 : 0, l, L, batch, T, n, m, (hipStream_t)stream,
                              zm::DdpTensors{nullptr, nullptr, nullptr}, zm::ValueIO{nullptr, nullptr, nullptr, nullptr, nullptr}, tl);
}

extern "C" int zm_ilqr_backward_ex_f64(const double* f_x, const double* f_u, const double* c_x, const double* c_u,
                                       const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                                       const double* vf_xx, const int32_t* active, int shared_hessian, double* l,
                                       double* L, int64_t batch, int T, int n, int m, void* stream) {
    return zm_ilqr_backward_list_f64(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, 0, active, shared_hessian, l, L,
                                     batch, T, n, m, stream);
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
    const int rc = zm_check_sweep_args("zm_lqr_backward_affine_f64", batch, T, n, m, true);
    if (rc) return rc;
    if (batch == 0) return ZM_OK;
    // carry (V, v) <- (Q[T-1], q[T-1])   (lqrUtils.py:261): terminal pointers into the last step, trajectory stride T*size
    const double* vf_xx = Q + (int64_t)(T - 1) * n * n;
    const double* vf_x = q + (int64_t)(T - 1) * n;
    if (n > 12 || m > 4)   // large states: fp64 MFMA tile sweep (sweep_tiled_f64.hip)
        return zm::sweep_tiled_f64_dispatch(1, A, B, q, r, Q, H, R, vf_x, vf_xx, d, (long)T * n, (long)T * n * n, nullptr, 0, l, L, batch,
                                            T, n, m, (hipStream_t)stream, nullptr, 0);
    if (zm::ilqr_backward_dma_dispatch<1>(A, B, q, r, Q, H, R, vf_x, vf_xx, d, (long)T * n, (long)T * n * n, nullptr, 0, l, L,
                                          batch, T, n, m, (hipStream_t)stream) == ZM_OK)
        return ZM_OK;
    return zm::launch_ilqr<1>(A, B, q, r, Q, H, R, vf_x, vf_xx, d, (long)T * n, (long)T * n * n, nullptr, 0, l, L, batch, T,
                              n, m, (hipStream_t)stream);
}

extern "C" int zm_ddp_backward_list_f64(const double* f_x, const double* f_u, const double* f_xx, const double* f_ux,
                                        const double* f_uu, const double* c_x, const double* c_u, const double* c_xx,
                                        const double* c_ux, const double* c_uu, const double* vf_x, const double* vf_xx,
                                        const int32_t* list, int64_t count, const int32_t* active, int shared_hessian,
                                        double* l, double* L, int64_t batch, int T, int n, int m, void* stream) {
    if (batch == 0 || (list && count == 0)) return ZM_OK;   /* nothing to do (pointers of empty arrays may be NULL) */
    if (list && (count < 0 || count > batch)) return zm::set_error(ZM_EINVAL, "zm_ddp_backward_list_f64: bad list length");
    if (!f_x || !f_u || !f_xx || (!f_ux != !f_uu) || !c_x || !c_u || !c_xx || !c_ux || !c_uu || !vf_x || !vf_xx || !l || !L)
        return zm::set_error(ZM_EINVAL, "zm_ddp_backward_f64: null pointer (f_ux and f_uu may be NULL together: identically zero)");
    const int rc = zm_check_sweep_args("zm_ddp_backward_f64", batch, T, n, m);
    if (rc) return rc;
    if ((int64_t)n * n * n >= (int64_t)1 << 31) return zm::set_error(ZM_EUNSUPPORTED, "zm_ddp_backward_f64: n too large");
    if (batch == 0) return ZM_OK;
    return zm::launch_ilqr<2>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n,
                              (const int*)active, shared_hessian ? 1 : 0, l, L, batch, T, n, m, (hipStream_t)stream,
                              zm::DdpTensors{f_xx, f_ux, f_uu}, zm::ValueIO{nullptr, nullptr, nullptr, nullptr, nullptr},
                              zm::TrajList{(const int*)list, (long)count});
}

extern "C" int zm_ddp_backward_pairs_list_f64(const zm_model_t* model, const double* f_x, const double* f_u, const double* H,
                                              const double* c_x, const double* c_u, const double* c_xx, const double* c_ux,
                                              const double* c_uu, const double* vf_x, const double* vf_xx, const int32_t* list,
                                              int64_t count, const int32_t* active, int shared_hessian, double* l, double* L,
                                              int64_t batch, int T, void* stream) {
    if (batch == 0 || (list && count == 0)) return ZM_OK;
    if (list && (count < 0 || count > batch)) return zm::set_error(ZM_EINVAL, "zm_ddp_backward_pairs_list_f64: bad list length");
    if (!model || !f_x || !f_u || !H || !c_x || !c_u || !c_xx || !c_ux || !c_uu || !vf_x || !vf_xx || !l || !L)
        return zm::set_error(ZM_EINVAL, "zm_ddp_backward_pairs_list_f64: null pointer");
#ifdef ZM_DDP_PSD_JACOBI
    return zm::set_error(ZM_EUNSUPPORTED, "zm_ddp_backward_pairs_list_f64: not available in the Jacobi-projection build");
#endif
    const int n = model->n, m = model->m;
    const int rc = zm_check_sweep_args("zm_ddp_backward_pairs_list_f64", batch, T, n, m);
    if (rc) return rc;
    const zm::PairTab pt = zm::model_pair_table(model->kind);
    if (pt.n < 1)
        return zm::set_error(ZM_EUNSUPPORTED, "zm_ddp_backward_pairs_list_f64: the model declares no Hessian pairs");
    const zm::TrajList tl{(const int*)list, (long)count};
    {   // the LDS-ring kernel where it applies (quadcopter shapes, shared cost Hessians); the register kernel otherwise
        const int rd = zm::ddp_backward_dma_dispatch(f_x, f_u, H, pt, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, (long)n, (const int*)active,
                                                     shared_hessian ? 1 : 0, l, L, batch, T, n, m, (hipStream_t)stream, tl);
        if (rd != ZM_EUNSUPPORTED) return rd;
    }
    zm::DdpTensors z{nullptr, nullptr, nullptr, H, pt};
    return zm::launch_ilqr<2>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n, (const int*)active,
                              shared_hessian ? 1 : 0, l, L, batch, T, n, m, (hipStream_t)stream, z,
                              zm::ValueIO{nullptr, nullptr, nullptr, nullptr, nullptr}, tl);
}

extern "C" int zm_ddp_backward_f64(const double* f_x, const double* f_u, const double* f_xx, const double* f_ux,
                                   const double* f_uu, const double* c_x, const double* c_u, const double* c_xx,
                                   const double* c_ux, const double* c_uu, const double* vf_x, const double* vf_xx,
                                   const int32_t* active, int shared_hessian, double* l, double* L, int64_t batch, int T,
                                   int n, int m, void* stream) {
    return zm_ddp_backward_list_f64(f_x, f_u, f_xx, f_ux, f_uu, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, 0, active,
                                    shared_hessian, l, L, batch, T, n, m, stream);
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
    if (ddp && (!f_xx || (!f_ux != !f_uu)))
        return zm::set_error(ZM_EINVAL, "zm_riccati_value_f64: f_xx goes with f_ux and f_uu (the latter two may be NULL together: zero)");
    const int rc = zm_check_sweep_args("zm_riccati_value_f64", batch, T, n, m, !ddp);
    if (rc) return rc;
    if (!ddp && (n > 12 || m > 4)) {   // large states: the tile sweep with its value-function outputs (sweep_tiled_f64.hip)
        if (!v_out || !vx_out || !vxx_out) return zm::set_error(ZM_EINVAL, "zm_riccati_value_f64: the value outputs are required at n > 12 or m > 4");
        return zm::sweep_tiled_f64_dispatch(0, f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n, nullptr, 0,
                                            l, L, batch, T, n, m, (hipStream_t)stream, nullptr, 0, c, vf, v_out, vx_out, vxx_out);
    }
    const zm::ValueIO v{c, vf, v_out, vx_out, vxx_out};
    if (ddp)
        return zm::launch_ilqr<2>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n, nullptr, 0, l,
                                  L, batch, T, n, m, (hipStream_t)stream, zm::DdpTensors{f_xx, f_ux, f_uu}, v);
    return zm::launch_ilqr<0>(f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, nullptr, (long)n, (long)n * n, nullptr, 0, l, L,
                              batch, T, n, m, (hipStream_t)stream, zm::DdpTensors{nullptr, nullptr, nullptr}, v);
}

#ifdef ZM_SWEEP_LAB
extern "C" int zm_lab_sweep_stamps(unsigned long long* out, int reset) {
    ZM_HIP_CHECK(hipDeviceSynchronize());
    ZM_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(zm::zm_sweep_stamps), sizeof(unsigned long long) * 10));
    if (reset) {
        unsigned long long z[10] = {0};
        ZM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(zm::zm_sweep_stamps), z, sizeof(z)));
    }
    return ZM_OK;
}
#endif
