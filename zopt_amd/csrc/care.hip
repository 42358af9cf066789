// Continuous-time LQR design, batched: one wave per design, matrices (n <= 16) zero-padded to 16 x 16 in LDS.
//
//   care_sda_kernel     infiniteHorizonLqr / infiniteHorizonIntegralLqr (reference zopt/lqrUtils.py:13-36, 101-141): the
//                       stabilising solution of A^T P + P A - P G P + Q = 0, G = B R^-1 B^T, by the structure-preserving
//                       doubling algorithm (Chu, Fan, Lin 2005) -- the reference calls scipy.linalg.solve_continuous_are
//                       (a Schur method); both return the same P, the stabilising solution is unique.
//   riccati_ode_kernel  finiteHorizonLqr (lqrUtils.py:39-98): dV/dt = -Q + V S V - V A - A^T V backwards from V(T) = Qf
//                       with an adaptive Dormand-Prince 5(4) pair -- the scheme of jax.experimental.ode.odeint the
//                       reference calls -- stepping exactly onto the N output times.
//
// These are design-time operations (a few hundred matrix products per design), far from any roofline: the code is
// written for clarity, every lane owns the entries (4 r + g, c), r = 0..3, of each 16 x 16 matrix.
#include "zm_common.h"

namespace zm {

constexpr int WN = 16, WLD = 17, WSZ = WN * WLD;

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc += op(X) op(Y) over the inner index k < kd (X, Y in LDS, zero-padded)
template <bool TX, bool TY>
__device__ __forceinline__ void wmm_acc(double (&acc)[4], const double* X, const double* Y, const int kd, const int g,
                                        const int c) {
    for (int k = 0; k < kd; ++k) {
        const double y = TY ? Y[c * WLD + k] : Y[k * WLD + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + g;
            const double x = TX ? X[k * WLD + i] : X[i * WLD + k];
            acc[r] = __builtin_fma(x, y, acc[r]);
        }
    }
}

__device__ __forceinline__ void wget(double (&v)[4], const double* M, const int g, const int c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = M[(4 * r + g) * WLD + c];
}

__device__ __forceinline__ void wput(double* M, const double (&v)[4], const int g, const int c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) M[(4 * r + g) * WLD + c] = v[r];
}

// (rows x cols) row-major global matrix -> zero-padded LDS tile
__device__ __forceinline__ void wload(double* M, const double* src, const int rows, const int cols, const int g, const int c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        M[i * WLD + c] = (i < rows && c < cols) ? src[i * cols + c] : 0.0;
    }
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// X1 <- M^-1 X1, X2 <- M^-1 X2 (X2 optional) by Gauss-Jordan elimination with partial pivoting; M (n x n) is destroyed.
// Returns false when a pivot column is exactly zero.
__device__ __forceinline__ bool wsolve(double* M, double* X1, double* X2, const int n, const int g, const int c) {
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        // pivot row: first maximum of |M[i][k]|, i >= k   (every 16-lane group does the same search)
        double v = (c >= k && c < n) ? fabs(M[c * WLD + k]) : -1.0;
        int p = c;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(v, off, 64);
            const int op = __shfl_xor(p, off, 64);
            if (ov > v || (ov == v && op < p)) {
                v = ov;
                p = op;
            }
        }
        if (!(v > 0.0)) {
            ok = false;
            break;
        }
        if (p != k) {   // swap rows k and p: group 0 -> M, 1 -> X1, 2 -> X2
            double* S = (g == 0) ? M : (g == 1) ? X1 : (g == 2) ? X2 : nullptr;
            if (S) {
                const double a = S[k * WLD + c], b = S[p * WLD + c];
                S[k * WLD + c] = b;
                S[p * WLD + c] = a;
            }
        }
        wsync();
        const double inv = 1.0 / M[k * WLD + k];
        const double pm = M[k * WLD + c] * inv, p1 = X1[k * WLD + c] * inv, p2 = X2 ? X2[k * WLD + c] * inv : 0.0;
        double f[4], m_[4], x1[4], x2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + g;
            f[r] = M[i * WLD + k];
            m_[r] = M[i * WLD + c];
            x1[r] = X1[i * WLD + c];
            x2[r] = X2 ? X2[i * WLD + c] : 0.0;
        }
        wsync();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + g;
            const bool piv = (i == k);
            M[i * WLD + c] = piv ? pm : __builtin_fma(-f[r], pm, m_[r]);
            X1[i * WLD + c] = piv ? p1 : __builtin_fma(-f[r], p1, x1[r]);
            if (X2) X2[i * WLD + c] = piv ? p2 : __builtin_fma(-f[r], p2, x2[r]);
        }
        wsync();
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void care_sda_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                      const double* __restrict__ Q, const double* __restrict__ R,
                                                      double* __restrict__ Kout, double* __restrict__ Pout,
                                                      int* __restrict__ info, const int n, const int m, const double tol,
                                                      const int max_iter) {
    __shared__ double sA[WSZ], sG[WSZ], sH[WSZ], sT1[WSZ], sT2[WSZ], sT3[WSZ], sRB[WSZ], sM[WSZ];
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const long inst = blockIdx.x;
    double id[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) id[r] = (4 * r + g == c && c < n) ? 1.0 : 0.0;
    int status = 0;

    // G = B R^-1 B^T;  sRB = R^-1 B^T (kept for K = R^-1 B^T P)
    wload(sA, A + inst * n * n, n, n, g, c);
    wload(sH, Q + inst * n * n, n, n, g, c);
    wload(sT1, B + inst * n * m, n, m, g, c);
    wload(sM, R + inst * m * m, m, m, g, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // sRB <- B^T
        const int i = 4 * r + g;
        sRB[i * WLD + c] = (i < m && c < n) ? B[inst * n * m + c * m + i] : 0.0;
    }
    wsync();
    if (!wsolve(sM, sRB, nullptr, m, g, c)) status = -2;
    {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        wmm_acc<false, false>(acc, sT1, sRB, m, g, c);
        wput(sG, acc, g, c);
    }
    // gamma = 1.25 |A|_inf (> spectral radius: A - gamma I is nonsingular), 1 for A = 0
    double rs = 0.0;
    if (g == 0 && c < n)
        for (int j = 0; j < n; ++j) rs += fabs(sA[c * WLD + j]);
    double gamma = 1.25 * wave_max(rs);
    if (!(gamma > 0.0)) gamma = 1.0;
    wsync();

    // Agi = (A - gamma I)^-1,  AgiG = Agi G,  W = Ag^T + H AgiG,  Wi = W^-1
    double a[4], ag[4];
    wget(a, sA, g, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) ag[r] = a[r] - gamma * id[r];
    wput(sM, ag, g, c);
    wput(sT1, id, g, c);
    wsync();
    if (!wsolve(sM, sT1, nullptr, n, g, c)) status = -2;   // sT1 = Agi
    {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        wmm_acc<false, false>(acc, sT1, sG, n, g, c);
        wput(sT2, acc, g, c);                              // sT2 = AgiG
        wsync();
        double w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = sA[c * WLD + (4 * r + g)] - gamma * id[r];   // Ag^T
        wmm_acc<false, false>(w, sH, sT2, n, g, c);
        wput(sM, w, g, c);
        wput(sT3, id, g, c);
        wsync();
    }
    if (!wsolve(sM, sT3, nullptr, n, g, c)) status = -2;   // sT3 = Wi
    {
        // A0 = I + 2 gamma Wi^T;  G0 = 2 gamma AgiG Wi;  H0 = 2 gamma Wi (H Agi)
        double a0[4], g0[4] = {0.0, 0.0, 0.0, 0.0}, hag[4] = {0.0, 0.0, 0.0, 0.0}, h0[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) a0[r] = id[r] + 2.0 * gamma * sT3[c * WLD + (4 * r + g)];
        wmm_acc<false, false>(g0, sT2, sT3, n, g, c);
        wmm_acc<false, false>(hag, sH, sT1, n, g, c);
        wsync();
        wput(sM, hag, g, c);
        wsync();
        wmm_acc<false, false>(h0, sT3, sM, n, g, c);
        wsync();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            g0[r] *= 2.0 * gamma;
            h0[r] *= 2.0 * gamma;
        }
        wput(sA, a0, g, c);
        wput(sG, g0, g, c);
        wput(sH, h0, g, c);
        wsync();
    }
    // doubling:  M = I + G H;  A' = A M^-1 A;  G' = G + A (M^-1 G) A^T;  H' = H + A^T (H M^-1 A)
    int it = 0;
    bool conv = false;
    while (status == 0 && it < max_iter && !conv) {
        double mm_[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mm_[r] = id[r];
        wmm_acc<false, false>(mm_, sG, sH, n, g, c);
        double ak[4], gk[4];
        wget(ak, sA, g, c);
        wget(gk, sG, g, c);
        wput(sM, mm_, g, c);
        wput(sT1, ak, g, c);
        wput(sT2, gk, g, c);
        wsync();
        if (!wsolve(sM, sT1, sT2, n, g, c)) {   // sT1 = M^-1 A, sT2 = M^-1 G
            status = -2;
            break;
        }
        double a1[4] = {0.0, 0.0, 0.0, 0.0}, t[4] = {0.0, 0.0, 0.0, 0.0}, u[4] = {0.0, 0.0, 0.0, 0.0};
        wmm_acc<false, false>(a1, sA, sT1, n, g, c);   // A (M^-1 A)
        wmm_acc<false, false>(t, sA, sT2, n, g, c);    // A (M^-1 G)
        wmm_acc<false, false>(u, sH, sT1, n, g, c);    // H (M^-1 A)
        wsync();
        wput(sT2, t, g, c);
        wput(sT3, u, g, c);
        wsync();
        double g1[4], h1[4], h0[4];
        wget(g1, sG, g, c);
        wget(h0, sH, g, c);
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[r] = h0[r];
        wmm_acc<false, true>(g1, sT2, sA, n, g, c);    // + (A M^-1 G) A^T
        wmm_acc<true, false>(h1, sA, sT3, n, g, c);    // + A^T (H M^-1 A)
        double d = 0.0, s = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d = fmax(d, fabs(h1[r] - h0[r]));
            s = fmax(s, fabs(h1[r]));
        }
        d = wave_max(d);
        s = wave_max(s);
        wsync();
        wput(sA, a1, g, c);
        wput(sG, g1, g, c);
        wput(sH, h1, g, c);
        wsync();
        ++it;
        conv = d <= tol * s;
        if (!(s < 1e300) || !(d == d)) status = -2;   // overflow / NaN: no stabilising solution
    }
    if (status == 0 && !conv) status = -1;
    // P = (H + H^T) / 2 as solve_continuous_are returns it;  K = R^-1 B^T P            (lqrUtils.py:34-35)
    double p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = 0.5 * (sH[(4 * r + g) * WLD + c] + sH[c * WLD + (4 * r + g)]);
    wsync();
    wput(sT1, p, g, c);
    wsync();
    double k_[4] = {0.0, 0.0, 0.0, 0.0};
    wmm_acc<false, false>(k_, sRB, sT1, n, g, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        if (i < m && c < n) Kout[inst * m * n + i * n + c] = k_[r];
        if (Pout && i < n && c < n) Pout[inst * n * n + i * n + c] = p[r];
    }
    if (info && lane == 0) info[inst] = status ? status : it;
}

// ---------------------------------------------------------------------------------------------------------------------
// Coefficients at time tau in [0, T]: samples at linspace(0, T, ns), linear in between (ns = 1: time-invariant).
struct OdeCoef {
    const double *A, *B, *Ri, *Q;   // this design's (ns, n, n), (ns, n, m), (ns, m, m), (ns, n, n) samples
    int n, m, ns;
    double T;
};

// S = B R_inv B^T from sB (n x m), sRi (m x m); sT is scratch
__device__ __forceinline__ void ode_form_s(const OdeCoef& k, double* sS, double* sB, double* sRi, double* sT, const int g,
                                           const int c) {
    double t[4] = {0.0, 0.0, 0.0, 0.0}, u[4] = {0.0, 0.0, 0.0, 0.0};
    wmm_acc<false, false>(t, sB, sRi, k.m, g, c);
    wput(sT, t, g, c);
    wsync();
    wmm_acc<false, true>(u, sT, sB, k.m, g, c);
    wput(sS, u, g, c);
    wsync();
}

__device__ __forceinline__ void ode_coef(const OdeCoef& k, const double tau, double* sA, double* sS, double* sQ, double* sB,
                                         double* sRi, double* sT, const int g, const int c) {
    if (k.ns == 1) return;   // loaded once
    double u = tau / k.T * (k.ns - 1);
    u = fmin(fmax(u, 0.0), (double)(k.ns - 1));
    int i0 = (int)u;
    if (i0 > k.ns - 2) i0 = k.ns - 2;
    const double w = u - i0;
    const int nn = k.n * k.n, nm = k.n * k.m, mm = k.m * k.m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        if (i < k.n && c < k.n) {
            const int e = i0 * nn + i * k.n + c;
            sA[i * WLD + c] = __builtin_fma(w, k.A[e + nn] - k.A[e], k.A[e]);
            sQ[i * WLD + c] = __builtin_fma(w, k.Q[e + nn] - k.Q[e], k.Q[e]);
        }
        if (i < k.n && c < k.m) {
            const int e = i0 * nm + i * k.m + c;
            sB[i * WLD + c] = __builtin_fma(w, k.B[e + nm] - k.B[e], k.B[e]);
        }
        if (i < k.m && c < k.m) {
            const int e = i0 * mm + i * k.m + c;
            sRi[i * WLD + c] = __builtin_fma(w, k.Ri[e + mm] - k.Ri[e], k.Ri[e]);
        }
    }
    wsync();
    ode_form_s(k, sS, sB, sRi, sT, g, c);
}

// f = Q - (V S) V + V A + A^T V at reversed time s (actual time T - s); V is taken from sV (LDS)         (lqrUtils.py:50, :90)
__device__ __forceinline__ void ode_rhs(double (&f)[4], const OdeCoef& k, const double s, double* sV, double* sT, double* sA,
                                        double* sS, double* sQ, double* sB, double* sRi, const int g, const int c) {
    ode_coef(k, k.T - s, sA, sS, sQ, sB, sRi, sT, g, c);
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    wmm_acc<false, false>(t, sV, sS, k.n, g, c);
    wput(sT, t, g, c);
    wget(f, sQ, g, c);
    wmm_acc<false, false>(f, sV, sA, k.n, g, c);
    wmm_acc<true, false>(f, sA, sV, k.n, g, c);
    wsync();
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = 0.0;
    wmm_acc<false, false>(t, sT, sV, k.n, g, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) f[r] -= t[r];
    wsync();
}

__global__ __launch_bounds__(64) void riccati_ode_kernel(const double* __restrict__ As, const double* __restrict__ Bs,
                                                         const double* __restrict__ Ris, const double* __restrict__ Qs,
                                                         const double* __restrict__ Qf, double* __restrict__ Vout,
                                                         int* __restrict__ info, const int n, const int m, const int ns,
                                                         const int N, const double T, const double rtol, const double atol,
                                                         const int max_steps) {
    __shared__ double sA[WSZ], sS[WSZ], sQ[WSZ], sV[WSZ], sT[WSZ], sB[WSZ], sRi[WSZ];
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const long inst = blockIdx.x;
    const int nn = n * n;
    OdeCoef k{As + inst * ns * nn, Bs + inst * ns * n * m, Ris + inst * ns * m * m, Qs + inst * ns * nn, n, m, ns, T};
    wload(sA, k.A, n, n, g, c);
    wload(sB, k.B, n, m, g, c);
    wload(sRi, k.Ri, m, m, g, c);
    wload(sQ, k.Q, n, n, g, c);
    wsync();
    ode_form_s(k, sS, sB, sRi, sT, g, c);
    double y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        y[r] = (i < n && c < n) ? Qf[inst * nn + i * n + c] : 0.0;
    }
    auto store = [&](const int j) {   // V at time t_j = j T / (N - 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + g;
            if (i < n && c < n) Vout[(inst * N + j) * nn + i * n + c] = y[r];
        }
    };
    store(N - 1);
    if (N == 1) {
        if (info && lane == 0) info[inst] = 0;
        return;
    }
    // Dormand-Prince 5(4) (the tableau of jax.experimental.ode), FSAL
    constexpr double a21 = 1.0 / 5, a31 = 3.0 / 40, a32 = 9.0 / 40, a41 = 44.0 / 45, a42 = -56.0 / 15, a43 = 32.0 / 9,
                     a51 = 19372.0 / 6561, a52 = -25360.0 / 2187, a53 = 64448.0 / 6561, a54 = -212.0 / 729,
                     a61 = 9017.0 / 3168, a62 = -355.0 / 33, a63 = 46732.0 / 5247, a64 = 49.0 / 176, a65 = -5103.0 / 18656,
                     b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84,
                     e1 = 35.0 / 384 - 1951.0 / 21600, e3 = 500.0 / 1113 - 22642.0 / 50085, e4 = 125.0 / 192 - 451.0 / 720,
                     e5 = -2187.0 / 6784 + 12231.0 / 42400, e6 = 11.0 / 84 - 649.0 / 6300, e7 = -1.0 / 60;
    const double hgrid = T / (N - 1);
    const double inv_cnt = 1.0 / (double)nn;
    auto rms = [&](const double (&v)[4]) {   // sqrt(mean(v^2)) over the n x n entries (padding holds zeros)
        double q = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) q = __builtin_fma(v[r], v[r], q);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
        return sqrt(q * inv_cnt);
    };
    double k1[4], k2[4], k3[4], k4[4], k5[4], k6[4], k7[4], yt[4], yn[4];
    wput(sV, y, g, c);
    wsync();
    ode_rhs(k1, k, 0.0, sV, sT, sA, sS, sQ, sB, sRi, g, c);
    // first step (Hairer, Norsett, Wanner II.4): h0 = 0.01 |y| / |f| in the tolerance-scaled norm, one Euler probe
    double h;
    {
        double sc[4], v0[4], v1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = atol + rtol * fabs(y[r]);
            v0[r] = y[r] / sc[r];
            v1[r] = k1[r] / sc[r];
        }
        const double d0 = rms(v0), d1 = rms(v1);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
#pragma unroll
        for (int r = 0; r < 4; ++r) yt[r] = __builtin_fma(h0, k1[r], y[r]);
        wput(sV, yt, g, c);
        wsync();
        ode_rhs(k2, k, h0, sV, sT, sA, sS, sQ, sB, sRi, g, c);
#pragma unroll
        for (int r = 0; r < 4; ++r) v1[r] = (k2[r] - k1[r]) / sc[r];
        const double d2 = rms(v1) / h0;
        const double dm = fmax(d1, d2);
        const double h1 = (dm <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / dm, 0.2);
        h = fmin(100.0 * h0, h1);
    }
    double s = 0.0;
    int steps = 0, status = 0, jfail = N;
    for (int j = 1; j < N && status == 0; ++j) {
        const double s_end = (j == N - 1) ? T : j * hgrid;
        while (s < s_end) {
            if (++steps > max_steps) {
                status = -1;
                break;
            }
            double hs = fmin(h, s_end - s);
            const bool last = (hs >= (s_end - s));
#define ZM_STAGE(expr, tt, kk)                                    \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) yt[r] = (expr); \
    wput(sV, yt, g, c);                                           \
    wsync();                                                      \
    ode_rhs(kk, k, s + (tt) * hs, sV, sT, sA, sS, sQ, sB, sRi, g, c);
            ZM_STAGE(y[r] + hs * (a21 * k1[r]), 1.0 / 5, k2)
            ZM_STAGE(y[r] + hs * (a31 * k1[r] + a32 * k2[r]), 3.0 / 10, k3)
            ZM_STAGE(y[r] + hs * (a41 * k1[r] + a42 * k2[r] + a43 * k3[r]), 4.0 / 5, k4)
            ZM_STAGE(y[r] + hs * (a51 * k1[r] + a52 * k2[r] + a53 * k3[r] + a54 * k4[r]), 8.0 / 9, k5)
            ZM_STAGE(y[r] + hs * (a61 * k1[r] + a62 * k2[r] + a63 * k3[r] + a64 * k4[r] + a65 * k5[r]), 1.0, k6)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                yn[r] = y[r] + hs * (b1 * k1[r] + b3 * k3[r] + b4 * k4[r] + b5 * k5[r] + b6 * k6[r]);
            wput(sV, yn, g, c);
            wsync();
            ode_rhs(k7, k, s + hs, sV, sT, sA, sS, sQ, sB, sRi, g, c);
#undef ZM_STAGE
            double er[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double e = hs * (e1 * k1[r] + e3 * k3[r] + e4 * k4[r] + e5 * k5[r] + e6 * k6[r] + e7 * k7[r]);
                er[r] = e / (atol + rtol * fmax(fabs(y[r]), fabs(yn[r])));
            }
            const double err = rms(er);
            if (!(err == err)) {   // NaN: finite escape time of the Riccati flow
                status = -2;
                break;
            }
            if (err <= 1.0) {
                s = last ? s_end : s + hs;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[r] = yn[r];
                    k1[r] = k7[r];
                }
            }
            // step-size controller of jax's odeint: safety 0.9, growth <= 10, shrink >= 0.2, order 5
            const double fac = (err == 0.0) ? 10.0 : fmin(10.0, fmax(err < 1.0 ? 1.0 : 0.2, 0.9 * pow(err, -0.2)));
            if (!(last && err <= 1.0 && fac >= 1.0)) h = hs * fac;   // a step clipped at an output time does not shrink h
        }
        if (status == 0) store(N - 1 - j);   // reversed time s_j is actual time T - s_j = t_{N-1-j}
        else jfail = j;
    }
    if (status != 0) {   // outputs not reached are NaN, as a failed integration's are
        const double qnan = __builtin_nan("");
        for (int j2 = 0; j2 <= N - 1 - jfail; ++j2)
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * r + g;
                if (i < n && c < n) Vout[(inst * N + j2) * nn + i * n + c] = qnan;
            }
    }
    if (info && lane == 0) info[inst] = status ? status : steps;
}

}  // namespace zm

extern "C" int zm_care_f64(const double* A, const double* B, const double* Q, const double* R, double* K, double* P,
                           int32_t* info, int64_t batch, int n, int m, double tol, int max_iter, void* stream) {
    if (batch == 0) return ZM_OK;
    if (!A || !B || !Q || !R || !K) return zm::set_error(ZM_EINVAL, "zm_care_f64: null pointer");
    if (batch < 0 || n < 1 || m < 1 || max_iter < 1 || !(tol >= 0.0)) return zm::set_error(ZM_EINVAL, "zm_care_f64: bad argument");
    if (n > 16 || m > 16) return zm::set_error(ZM_EUNSUPPORTED, "zm_care_f64: (n=%d, m=%d) not covered (need n<=16, m<=16)", n, m);
    if (batch >= ((int64_t)1 << 31)) return zm::set_error(ZM_EUNSUPPORTED, "zm_care_f64: batch too large");
    hipLaunchKernelGGL(zm::care_sda_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, A, B, Q, R, K, P,
                       (int*)info, n, m, tol, max_iter);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

extern "C" int zm_riccati_ode_f64(const double* A_s, const double* B_s, const double* Rinv_s, const double* Q_s,
                                  const double* Qf, double* V, int32_t* info, int64_t batch, int n, int m, int n_samples, int N,
                                  double T, double rtol, double atol, int max_steps, void* stream) {
    if (batch == 0) return ZM_OK;
    if (!A_s || !B_s || !Rinv_s || !Q_s || !Qf || !V) return zm::set_error(ZM_EINVAL, "zm_riccati_ode_f64: null pointer");
    if (batch < 0 || n < 1 || m < 1 || n_samples < 1 || N < 1 || max_steps < 1 || !(T > 0.0) || !(rtol >= 0.0) || !(atol >= 0.0) ||
        !(rtol + atol > 0.0))
        return zm::set_error(ZM_EINVAL, "zm_riccati_ode_f64: bad argument");
    if (n > 16 || m > 16) return zm::set_error(ZM_EUNSUPPORTED, "zm_riccati_ode_f64: (n=%d, m=%d) not covered (need n<=16, m<=16)", n, m);
    if (batch >= ((int64_t)1 << 31) || (int64_t)n_samples * n * n >= ((int64_t)1 << 30) || (int64_t)N * n * n >= ((int64_t)1 << 30))
        return zm::set_error(ZM_EUNSUPPORTED, "zm_riccati_ode_f64: batch / sample count too large");
    hipLaunchKernelGGL(zm::riccati_ode_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, A_s, B_s, Rinv_s, Q_s, Qf,
                       V, (int*)info, n, m, n_samples, N, T, rtol, atol, max_steps);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
