// Wave-level symmetric eigen-decomposition / PD projection of a k x k matrix (k <= 16) held in LDS.
// Cyclic two-sided Jacobi, round-robin parallel ordering (K-1 rounds per sweep, K/2 disjoint rotations per round).
// Used by psd.hip (K5) and by the DDP backward step (K4), where the projection sits inside the sequential sweep
// (zopt/ilqrUtils.py:217-219, 237-251).
#pragma once
#include "zm_common.h"
#include <hip/hip_runtime.h>

namespace zm {

constexpr int PK = 16;   // max matrix size
constexpr int PLD = 17;  // padded leading dimension in LDS

// (wave_lds_sync: zm_common.h)

__device__ __forceinline__ double jac_rsqrt(const double x) {   // 1/sqrt(x), x > 0: hardware estimate + 2 Newton steps
    double y = __builtin_amdgcn_rsq(x);
    y = y * __builtin_fma(-0.5 * x, y * y, 1.5);
    y = y * __builtin_fma(-0.5 * x, y * y, 1.5);
    return y;
}
__device__ __forceinline__ double jac_rcp(const double a) {     // 1/a: hardware estimate + 2 Newton steps
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    return r;
}

// In: As (PK*PLD doubles) holds the SYMMETRISED matrix in its leading k x k block.  Scratch: Vs (PK*PLD), cs (PK), pq (PK).
// Out: As[i*PLD + j] = (V max(w, eps) V^T)[i][j].  Must be called by all 64 lanes of one wave.
//
// Work split: 8 lanes per rotation pair (pair = lane >> 3, sub = lane & 7), every lane computes its pair's rotation itself
// from (a_pp, a_qq, a_pq) -- no exchange of rotation parameters -- and applies it to rows sub and sub+8 of the column pair
// (A and V), then, after one wave-level sync, to columns sub and sub+8 of the row pair (A).  Two syncs per round, no integer
// division, reciprocal / reciprocal-square-root by Newton steps.  A rotation is skipped when
// a_pq^2 <= 2^-104 |a_pp a_qq|; the sweeps stop after the first one that rotated nothing (or after 20).
//
// warm = true: Vs already holds an orthogonal matrix V0 (the eigenvectors this routine left behind for a NEARBY matrix, e.g.
// the previous step of a DDP sweep).  The iteration then starts from V0^T A V0 -- nearly diagonal -- and accumulates onto V0:
// same result, typically 2-3 sweeps instead of 6-8.
__device__ __forceinline__ void psd_project_lds(double* As, double* Vs, double* cs, int* pq, const int k,
                                                const double eps, const int lane, const bool warm = false) {
    (void)pq;
    const int K = (k + 1) & ~1;  // even number of round-robin players; index k (if k is odd) is a bye
    const int pr = lane >> 3, sub = lane & 7;
    if (!warm) {   // V <- I on the full padded tile
        const int i = lane & 15, jb = lane >> 4;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) Vs[i * PLD + jb + 4 * jj] = (i == jb + 4 * jj) ? 1.0 : 0.0;
    } else {       // A <- V0^T A V0  (two k^3 products through registers; lane owns row i, columns jb + 4 jj)
        const int i = lane & 15, jb = lane >> 4;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int t = 0; t < k; ++t) {      // T = A V0
            const double ait = (i < k) ? As[i * PLD + t] : 0.0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_fma(ait, Vs[t * PLD + jb + 4 * jj], acc[jj]);
        }
        wave_lds_sync();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) As[i * PLD + jb + 4 * jj] = acc[jj];
        wave_lds_sync();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = 0.0;
        for (int t = 0; t < k; ++t) {      // V0^T T
            const double vti = (i < k) ? Vs[t * PLD + i] : 0.0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_fma(vti, As[t * PLD + jb + 4 * jj], acc[jj]);
        }
        wave_lds_sync();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) As[i * PLD + jb + 4 * jj] = acc[jj];
        // the products leave rounding-level asymmetry; the rotations read a_pq from the upper triangle only, both
        // triangles are transformed identically afterwards
    }
    wave_lds_sync();
    // Round-robin schedule of K players (player K-1 stays, the others rotate), tabulated once per call: entry (r, pair) =
    // p | q << 8 | active << 16 with p < q.  (Measured: the routine is bound by LDS bandwidth -- 27 b64 accesses per lane and
    // round through the one LDS pipe of the CU -- not by these integer instructions or by latency.)
    __shared__ int jtab[(PK - 1) * (PK / 2)];
    for (int e = lane; e < (K - 1) * (PK / 2); e += 64) {
        const int r = e >> 3, pp = e & 7;
        int p, q;
        if (pp == 0) {
            p = K - 1;
            q = r;
        } else {
            p = r + pp;
            p = p >= K - 1 ? p - (K - 1) : p;
            q = r - pp + (K - 1);
            q = q >= K - 1 ? q - (K - 1) : q;
        }
        if (p > q) {
            const int t = p;
            p = q;
            q = t;
        }
        const bool on = (pp < (K >> 1)) && (q < k);   // q == k: bye
        jtab[e] = on ? (p | (q << 8) | (1 << 16)) : 0;
    }
    wave_lds_sync();
    for (int sweep = 0; sweep < 20; ++sweep) {
        bool rotated = false, big = false;
        for (int r = 0; r < K - 1; ++r) {
            const int ent = jtab[(r << 3) + pr];
            const int p = ent & 0xff, q = (ent >> 8) & 0xff;
            const bool on = (ent >> 16) != 0;
            const double app = As[p * PLD + p], aqq = As[q * PLD + q], apq = As[p * PLD + q];
            // the column data the rotation will be applied to does not depend on the angle: fetch it now, so that its LDS
            // latency runs under the rotation's arithmetic instead of after it
            double ca[2][4];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int i = sub + 8 * h2;
                ca[h2][0] = As[i * PLD + p];
                ca[h2][1] = As[i * PLD + q];
                ca[h2][2] = Vs[i * PLD + p];
                ca[h2][3] = Vs[i * PLD + q];
            }
            const double apq2 = apq * apq, dd = __builtin_fabs(app * aqq);
            const bool rot = on && (apq2 > 0x1p-104 * dd);
            // quadratic convergence: if every rotation of this sweep started from |a_pq| <= 2^-26 sqrt|a_pp a_qq|, the sweep
            // leaves off-diagonals <= ~2^-52 relative -- the next sweep would rotate nothing and need not be run
            big |= (__ballot(on && (apq2 > 0x1p-52 * dd)) != 0ull);
            double c = 1.0, s = 0.0;
            if (rot) {
                // t = sgn(d) 2 a_pq / (|d| + sqrt(d^2 + 4 a_pq^2)),  d = a_qq - a_pp   (the smaller root of t^2 + 2 tau t - 1).
                // The angle only steers convergence, so t takes one Newton step per reciprocal (~1e-9); orthogonality needs
                // c^2 + s^2 = 1 to rounding, so c = 1/sqrt(1 + t^2) takes two.
                const double d = aqq - app, b2 = 2.0 * apq;
                const double x = __builtin_fma(d, d, b2 * b2);
                double y = __builtin_amdgcn_rsq(x);
                y = y * __builtin_fma(-0.5 * x, y * y, 1.5);
                const double den = __builtin_fabs(d) + x * y;
                double rc = __builtin_amdgcn_rcp(den);
                rc = __builtin_fma(rc, __builtin_fma(-den, rc, 1.0), rc);
                const double t = (d >= 0.0 ? b2 : -b2) * rc;
                c = jac_rsqrt(__builtin_fma(t, t, 1.0));
                s = t * c;
            }
            rotated |= (__ballot(rot) != 0ull);
            // columns p, q of A and V, rows sub and sub+8:  X[:,p] <- c X[:,p] - s X[:,q],  X[:,q] <- s X[:,p] + c X[:,q]
            if (rot) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int i = sub + 8 * h2;
                    As[i * PLD + p] = c * ca[h2][0] - s * ca[h2][1];
                    As[i * PLD + q] = s * ca[h2][0] + c * ca[h2][1];
                    Vs[i * PLD + p] = c * ca[h2][2] - s * ca[h2][3];
                    Vs[i * PLD + q] = s * ca[h2][2] + c * ca[h2][3];
                }
            }
            wave_lds_sync();
            // rows p, q of A, columns sub and sub+8
            if (rot) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int j = sub + 8 * h2;
                    const double ap = As[p * PLD + j], aq = As[q * PLD + j];
                    As[p * PLD + j] = c * ap - s * aq;
                    As[q * PLD + j] = s * ap + c * aq;
                }
            }
            wave_lds_sync();
        }
        if (!rotated || !big) break;
    }
    // A <- V max(w, eps) V^T   (the clamped eigenvalues are copied out of the diagonal first)
    if (lane < PK) cs[lane] = (lane < k) ? (As[lane * PLD + lane] > eps ? As[lane * PLD + lane] : eps) : 0.0;
    wave_lds_sync();
    {
        const int i = lane & 15, jb = lane >> 4;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int t = 0; t < k; ++t) {
            const double vi = Vs[i * PLD + t] * cs[t];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_fma(vi, Vs[(jb + 4 * jj) * PLD + t], acc[jj]);
        }
        wave_lds_sync();
        if (i < k) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (jb + 4 * jj < k) As[i * PLD + jb + 4 * jj] = acc[jj];
        }
    }
    wave_lds_sync();
}

}  // namespace zm
