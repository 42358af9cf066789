/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's discrete finite-horizon
 * LQR backward Riccati recursion (zopt/lqrUtils.py:144-173) used (a) as a second, independent
 * checker next to oracle/zopt_oracle.py and (b) as bench.py's timed `cpu_baseline` ("port").
 * The product library (zopt_amd/csrc) never links or calls this file.
 *
 * Per trajectory, following lqrUtils.py:167-172 literally:
 *     V <- Q[T-1]                                                       (:172)
 *     for k = T-1 .. 0:
 *         L_k = solve(R_k + B_k^T V B_k,  B_k^T V A_k)                  (:168)  LU, partial pivoting
 *         V   = Q_k + L_k^T R_k L_k + (A_k-B_k L_k)^T V (A_k-B_k L_k)   (:169)  Joseph form
 * Layout: C-contiguous, time-first, one trajectory after another:
 *     A (batch,T,n,n)  B (batch,T,n,m)  Q (batch,T,n,n)  R (batch,T,m,m)  ->  L (batch,T,m,n)
 *
 * Pinning: checked in tests/test_oracle.py against the reference's known answer
 * (tests/test_lqrUtils.py:61-69: L[1]=0.5 I, L[0]=0.6 I) and against the NumPy oracle.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ZO_MAXN 64
#define ZO_MAXM 16

/* C[p x r] = A^T[p x q] * B[q x r], A stored q x p.  Loop order k-i-j keeps the inner loop contiguous so gcc
 * vectorises it; each C[i][j] is still the k-ordered sum the reference's matmul forms. */
static void atb(const double *A, const double *B, double *C, int q, int p, int r) {
    for (int i = 0; i < p * r; ++i) C[i] = 0.0;
    for (int k = 0; k < q; ++k)
        for (int i = 0; i < p; ++i) {
            const double a = A[k * p + i];
            for (int j = 0; j < r; ++j) C[i * r + j] += a * B[k * r + j];
        }
}
/* C[p x r] = A[p x q] * B[q x r] */
static void ab(const double *A, const double *B, double *C, int p, int q, int r) {
    for (int i = 0; i < p; ++i) {
        for (int j = 0; j < r; ++j) C[i * r + j] = 0.0;
        for (int k = 0; k < q; ++k) {
            const double a = A[i * q + k];
            for (int j = 0; j < r; ++j) C[i * r + j] += a * B[k * r + j];
        }
    }
}
/* Solve S X = Y in place (S m x m, Y m x n): LU with partial pivoting (first max wins, as
 * LAPACK idamax), forward elimination applied to Y, then back substitution. */
static void lu_solve(double *S, double *Y, int m, int n) {
    for (int k = 0; k < m; ++k) {
        int p = k;
        double best = fabs(S[k * m + k]);
        for (int i = k + 1; i < m; ++i) {
            double a = fabs(S[i * m + k]);
            if (a > best) { best = a; p = i; }
        }
        if (p != k) {
            for (int j = 0; j < m; ++j) { double t = S[k * m + j]; S[k * m + j] = S[p * m + j]; S[p * m + j] = t; }
            for (int j = 0; j < n; ++j) { double t = Y[k * n + j]; Y[k * n + j] = Y[p * n + j]; Y[p * n + j] = t; }
        }
        const double piv = S[k * m + k];
        for (int i = k + 1; i < m; ++i) {
            const double f = S[i * m + k] / piv;
            for (int j = k + 1; j < m; ++j) S[i * m + j] -= f * S[k * m + j];
            for (int j = 0; j < n; ++j) Y[i * n + j] -= f * Y[k * n + j];
        }
    }
    for (int k = m - 1; k >= 0; --k) {
        const double piv = S[k * m + k];
        for (int j = 0; j < n; ++j) {
            double s = Y[k * n + j];
            for (int i = k + 1; i < m; ++i) s -= S[k * m + i] * Y[i * n + j];
            Y[k * n + j] = s / piv;
        }
    }
}

static void one_trajectory(const double *A, const double *B, const double *Q, const double *R, double *L,
                           int T, int n, int m) {
    double V[ZO_MAXN * ZO_MAXN], BtV[ZO_MAXM * ZO_MAXN], Suu[ZO_MAXM * ZO_MAXM], X[ZO_MAXM * ZO_MAXN];
    double Acl[ZO_MAXN * ZO_MAXN], W[ZO_MAXN * ZO_MAXN], LtR[ZO_MAXN * ZO_MAXM], T1[ZO_MAXN * ZO_MAXN];
    memcpy(V, Q + (size_t)(T - 1) * n * n, sizeof(double) * n * n);
    for (int k = T - 1; k >= 0; --k) {
        const double *Ak = A + (size_t)k * n * n, *Bk = B + (size_t)k * n * m;
        const double *Qk = Q + (size_t)k * n * n, *Rk = R + (size_t)k * m * m;
        double *Lk = L + (size_t)k * m * n;
        atb(Bk, V, BtV, n, m, n);                    /* B^T V            (m x n) */
        ab(BtV, Bk, Suu, m, n, m);                   /* (B^T V) B        (m x m) */
        for (int i = 0; i < m * m; ++i) Suu[i] = Rk[i] + Suu[i];
        ab(BtV, Ak, X, m, n, n);                     /* (B^T V) A        (m x n) */
        lu_solve(Suu, X, m, n);                      /* L_k */
        memcpy(Lk, X, sizeof(double) * m * n);
        ab(Bk, X, Acl, n, m, n);                     /* B L */
        for (int i = 0; i < n * n; ++i) Acl[i] = Ak[i] - Acl[i];
        atb(X, Rk, LtR, m, n, m);                    /* L^T R            (n x m) */
        ab(LtR, X, T1, n, m, n);                     /* (L^T R) L        (n x n) */
        atb(Acl, V, W, n, n, n);                     /* Acl^T V */
        ab(W, Acl, V, n, n, n);                      /* (Acl^T V) Acl -> V (W, Acl distinct from V) */
        for (int i = 0; i < n * n; ++i) V[i] = Qk[i] + T1[i] + V[i];
    }
}

/* returns 0 on success, -1 on unsupported shape; nthreads<=0 -> OpenMP default */
int zo_lqr_backward_f64(const double *A, const double *B, const double *Q, const double *R, double *L,
                        int64_t batch, int T, int n, int m, int nthreads) {
    if (n < 1 || m < 1 || n > ZO_MAXN || m > ZO_MAXM || T < 1 || batch < 0) return -1;
    /* the thread count is a clause of THIS parallel region: the process-wide OpenMP default stays untouched (a one-thread
       checker call must not turn a later all-cores baseline run into a one-thread run) */
#ifdef _OPENMP
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    const int nt = 1;
#endif
    (void)nt;
    const size_t sA = (size_t)T * n * n, sB = (size_t)T * n * m, sR = (size_t)T * m * m, sL = (size_t)T * m * n;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t b = 0; b < batch; ++b)
        one_trajectory(A + b * sA, B + b * sB, Q + b * sA, R + b * sR, L + b * sL, T, n, m);
    return 0;
}

int zo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
