// Wave-level symmetric eigen-decomposition / PD projection of a k x k matrix (k <= 16) held in LDS.
// Cyclic two-sided Jacobi, round-robin parallel ordering (K-1 rounds per sweep, K/2 disjoint rotations per round).
// Used by psd.hip (K5) and by the DDP backward step (K4), where the projection sits inside the sequential sweep
// (zopt/ilqrUtils.py:217-219, 237-251).
#pragma once
#include <hip/hip_runtime.h>

namespace zm {

constexpr int PK = 16;   // max matrix size
constexpr int PLD = 17;  // padded leading dimension in LDS

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In: As (PK*PLD doubles) holds the SYMMETRISED matrix in its leading k x k block.  Scratch: Vs (PK*PLD), cs (PK), pq (PK).
// Out: As[i*PLD + j] = (V max(w, eps) V^T)[i][j].  Must be called by all 64 lanes of one wave.
__device__ __forceinline__ void psd_project_lds(double* As, double* Vs, double* cs, int* pq, const int k,
                                                const double eps, const int lane) {
    const int K = (k + 1) & ~1;  // even number of round-robin players; index k (if k is odd) is a bye
    for (int e = lane; e < k * k; e += 64) Vs[(e / k) * PLD + (e % k)] = ((e / k) == (e % k)) ? 1.0 : 0.0;
    wave_lds_sync();
    for (int sweep = 0; sweep < 20; ++sweep) {
        bool rotated = false;
        for (int r = 0; r < K - 1; ++r) {
            bool rot = false;
            if (lane < K / 2) {
                int p, q;
                if (lane == 0) {
                    p = K - 1;
                    q = r;
                } else {
                    p = (r + lane) % (K - 1);
                    q = (r - lane + (K - 1)) % (K - 1);
                }
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                double c = 1.0, s = 0.0;
                if (q < k) {
                    const double app = As[p * PLD + p], aqq = As[q * PLD + q], apq = As[p * PLD + q];
                    if (apq != 0.0 && __builtin_fabs(apq) > 0x1p-52 * __builtin_sqrt(__builtin_fabs(app * aqq))) {
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (__builtin_fabs(tau) + __builtin_sqrt(1.0 + tau * tau));
                        c = 1.0 / __builtin_sqrt(1.0 + t * t);
                        s = t * c;
                        rot = true;
                    }
                } else {
                    q = p;  // bye: identity on a single index
                }
                pq[2 * lane] = p;
                pq[2 * lane + 1] = q;
                cs[2 * lane] = c;
                cs[2 * lane + 1] = s;
            }
            rotated |= (__ballot(rot) != 0ull);
            wave_lds_sync();
            // columns p,q of A and V:  X[:, p] <- c X[:,p] - s X[:,q],  X[:, q] <- s X[:,p] + c X[:,q]
            for (int e = lane; e < (K / 2) * k; e += 64) {
                const int pr = e / k, i = e % k;
                const int p = pq[2 * pr], q = pq[2 * pr + 1];
                const double c = cs[2 * pr], s = cs[2 * pr + 1];
                if (p != q) {
                    const double ap = As[i * PLD + p], aq = As[i * PLD + q];
                    As[i * PLD + p] = c * ap - s * aq;
                    As[i * PLD + q] = s * ap + c * aq;
                    const double vp = Vs[i * PLD + p], vq = Vs[i * PLD + q];
                    Vs[i * PLD + p] = c * vp - s * vq;
                    Vs[i * PLD + q] = s * vp + c * vq;
                }
            }
            wave_lds_sync();
            // rows p,q of A
            for (int e = lane; e < (K / 2) * k; e += 64) {
                const int pr = e / k, j = e % k;
                const int p = pq[2 * pr], q = pq[2 * pr + 1];
                const double c = cs[2 * pr], s = cs[2 * pr + 1];
                if (p != q) {
                    const double ap = As[p * PLD + j], aq = As[q * PLD + j];
                    As[p * PLD + j] = c * ap - s * aq;
                    As[q * PLD + j] = s * ap + c * aq;
                }
            }
            wave_lds_sync();
        }
        if (!rotated) break;
    }
    // A <- V max(w, eps) V^T   (the clamped eigenvalues are copied out of the diagonal first)
    if (lane < k) cs[lane] = As[lane * PLD + lane] > eps ? As[lane * PLD + lane] : eps;
    wave_lds_sync();
    for (int e = lane; e < k * k; e += 64) {
        const int i = e / k, j = e % k;
        double acc = 0.0;
        for (int t = 0; t < k; ++t) acc = __builtin_fma(Vs[i * PLD + t] * cs[t], Vs[j * PLD + t], acc);
        As[i * PLD + j] = acc;
    }
    wave_lds_sync();
}

}  // namespace zm
