// K1-G  lqr_backward_lds_f64 -- finite-horizon LQR backward Riccati sweep, fp64, any 1 <= n <= 64, 1 <= m <= 16.
//
// Replaces zopt/lqrUtils.py:144-173 discreteFiniteHorizonLqr for the shapes the tile-16 kernels (n <= 12, m <= 4) do not
// cover:  V <- Q[T-1];  L_k = solve(R_k + B_k^T V B_k, B_k^T V A_k) (:168);  V = Q_k + L_k^T R_k L_k + Acl^T V Acl (:169).
// Coverage kernel, not a tuned one: one 256-thread block per trajectory, V / A_k / V Acl resident in LDS (135 kB at
// n = 64, m = 16), plain fp64 FMAs in the reference's own operation order, LU with partial pivoting by the first wave.
#include <hip/hip_runtime.h>

#include "zm_common.h"

namespace zm {

__global__ __launch_bounds__(256) void lqr_backward_lds_f64(const double* __restrict__ A, const double* __restrict__ B,
                                                            const double* __restrict__ Q, const double* __restrict__ R,
                                                            double* __restrict__ L, const long batch, const int T, const int n,
                                                            const int m) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int nn = n * n, nm = n * m, mm = m * m, SLD = n + m;
    double* Vs = sm;             // n x n   value matrix
    double* As = Vs + nn;        // n x n   A_k, then Acl in place
    double* Ts = As + nn;        // n x n   V Acl
    double* Bs = Ts + nn;        // n x m
    double* Gs = Bs + nm;        // m x n   B^T V
    double* Ss = Gs + nm;        // m x (n+m)  [Sux | Suu]  ->  L in its first n columns
    double* Hs = Ss + m * SLD;   // m x n   R L
    double* Rs = Hs + nm;        // m x m
    __shared__ int piv_s;
    const int t = threadIdx.x, NTH = blockDim.x;
    const long traj = blockIdx.x;
    if (traj >= batch) return;
    const double* Ab = A + traj * T * nn;
    const double* Bb = B + traj * T * nm;
    const double* Qb = Q + traj * T * nn;
    const double* Rb = R + traj * T * mm;
    double* Lb = L + traj * T * nm;

    for (int e = t; e < nn; e += NTH) Vs[e] = Qb[(long)(T - 1) * nn + e];   // terminal value = last stage cost (:172)
    for (int k = T - 1; k >= 0; --k) {
        for (int e = t; e < nn; e += NTH) As[e] = Ab[(long)k * nn + e];
        for (int e = t; e < nm; e += NTH) Bs[e] = Bb[(long)k * nm + e];
        for (int e = t; e < mm; e += NTH) Rs[e] = Rb[(long)k * mm + e];
        __syncthreads();
        for (int e = t; e < nm; e += NTH) {   // G = B^T V
            const int u = e / n, j = e % n;
            double acc = 0.0;
            for (int i = 0; i < n; ++i) acc = __builtin_fma(Bs[i * m + u], Vs[i * n + j], acc);
            Gs[e] = acc;
        }
        __syncthreads();
        for (int e = t; e < m * SLD; e += NTH) {   // Sux = G A,  Suu = R + G B
            const int u = e / SLD, j = e % SLD;
            double acc = (j < n) ? 0.0 : Rs[u * m + (j - n)];
            if (j < n)
                for (int i = 0; i < n; ++i) acc = __builtin_fma(Gs[u * n + i], As[i * n + j], acc);
            else
                for (int i = 0; i < n; ++i) acc = __builtin_fma(Gs[u * n + i], Bs[i * m + (j - n)], acc);
            Ss[e] = acc;
        }
        __syncthreads();
        // L = solve(Suu, Sux): LU with partial pivoting (jnp.linalg.solve); every column of [Sux | Suu] has ONE owner thread
        for (int kk = 0; kk < m; ++kk) {
            if (t == 0) {
                int p = kk;
                double best = __builtin_fabs(Ss[kk * SLD + n + kk]);
                for (int r = kk + 1; r < m; ++r) {
                    const double v = __builtin_fabs(Ss[r * SLD + n + kk]);
                    if (v > best) {
                        best = v;
                        p = r;
                    }
                }
                piv_s = p;
            }
            __syncthreads();
            const int p = piv_s;
            const double inv = 1.0 / Ss[p * SLD + n + kk];      // pivot (row p before the swap)
            if (t < SLD) {
                const int j = t;
                const double a = Ss[kk * SLD + j], b = Ss[p * SLD + j];
                Ss[kk * SLD + j] = b;
                Ss[p * SLD + j] = a;
            }
            __syncthreads();
            if (t < SLD && (t < n || t > n + kk)) {   // multipliers live in column n+kk: it is read, not updated
                const int j = t;
                const double pj = Ss[kk * SLD + j];
                for (int r = kk + 1; r < m; ++r) Ss[r * SLD + j] -= (Ss[r * SLD + n + kk] * inv) * pj;
            }
            __syncthreads();
        }
        if (t < n) {   // back substitution, in place per column
            const int j = t;
            for (int kk = m - 1; kk >= 0; --kk) {
                double acc = Ss[kk * SLD + j];
                for (int r = kk + 1; r < m; ++r) acc -= Ss[kk * SLD + n + r] * Ss[r * SLD + j];
                Ss[kk * SLD + j] = acc / Ss[kk * SLD + n + kk];
            }
        }
        __syncthreads();
        for (int e = t; e < nm; e += NTH) {   // L_k out;  H = R L
            const int u = e / n, j = e % n;
            Lb[(long)k * nm + e] = Ss[u * SLD + j];
            double acc = 0.0;
            for (int v = 0; v < m; ++v) acc = __builtin_fma(Rs[u * m + v], Ss[v * SLD + j], acc);
            Hs[e] = acc;
        }
        for (int e = t; e < nn; e += NTH) {   // Acl = A - B L
            const int i = e / n, j = e % n;
            double acc = As[e];
            for (int u = 0; u < m; ++u) acc = __builtin_fma(-Bs[i * m + u], Ss[u * SLD + j], acc);
            Ts[e] = acc;
        }
        __syncthreads();
        for (int e = t; e < nn; e += NTH) As[e] = Ts[e];
        __syncthreads();
        for (int e = t; e < nn; e += NTH) {   // T = V Acl
            const int i = e / n, j = e % n;
            double acc = 0.0;
            for (int l = 0; l < n; ++l) acc = __builtin_fma(Vs[i * n + l], As[l * n + j], acc);
            Ts[e] = acc;
        }
        __syncthreads();
        for (int e = t; e < nn; e += NTH) {   // V' = Q + L^T (R L) + Acl^T (V Acl)
            const int i = e / n, j = e % n;
            double acc = Qb[(long)k * nn + e];
            for (int u = 0; u < m; ++u) acc = __builtin_fma(Ss[u * SLD + i], Hs[u * n + j], acc);
            for (int l = 0; l < n; ++l) acc = __builtin_fma(As[l * n + i], Ts[l * n + j], acc);
            Vs[e] = acc;
        }
        __syncthreads();
    }
}

int lqr_backward_lds_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L, int64_t batch,
                              int T, int n, int m, hipStream_t st) {
    const size_t doubles = 3 * (size_t)n * n + 3 * (size_t)n * m + (size_t)m * (n + m) + (size_t)m * m;
    const size_t bytes = doubles * sizeof(double);
    // per launch (cheap): the attribute is per device, and several devices may be driven from one process
    ZM_HIP_CHECK(hipFuncSetAttribute((const void*)lqr_backward_lds_f64, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    hipLaunchKernelGGL(lqr_backward_lds_f64, dim3((unsigned)batch), dim3(256), bytes, st, A, B, Q, R, L, (long)batch, T, n, m);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

}  // namespace zm
