// Element accessors of the symmetric matrices the PD projections work on (psd.hip: one 16 x 16 tile; psd_tiled.hip: up to 64 x 64).
#pragma once
#include <hip/hip_runtime.h>

namespace zm {

// element accessors of the (possibly block-stored) symmetric matrix
struct PlainMat {
    double* A;
    int k;
    __device__ __forceinline__ double load(long mat, int i, int j) const { return A[(mat * k + i) * k + j]; }
    __device__ __forceinline__ void store(long mat, int i, int j, double v) const { A[(mat * k + i) * k + j] = v; }
};
struct StackedCost {  // [[c_xx, c_ux^T],[c_ux, c_uu]]   (ilqrUtils.py:230)
    double *cxx, *cux, *cuu;
    int n, m;
    __device__ __forceinline__ double load(long mat, int i, int j) const {
        if (i < n) return (j < n) ? cxx[(mat * n + i) * n + j] : cux[(mat * m + (j - n)) * n + i];
        return (j < n) ? cux[(mat * m + (i - n)) * n + j] : cuu[(mat * m + (i - n)) * m + (j - n)];
    }
    __device__ __forceinline__ void store(long mat, int i, int j, double v) const {
        // ilqrUtils.py:232: c_xx = zz[:n,:n], c_uu = zz[-m:,-m:], c_ux = zz[-m:,:n]   (the upper-right block is dropped)
        if (i < n) {
            if (j < n) cxx[(mat * n + i) * n + j] = v;
        } else {
            if (j < n)
                cux[(mat * m + (i - n)) * n + j] = v;
            else
                cuu[(mat * m + (i - n)) * m + (j - n)] = v;
        }
    }
};

// vf_zz = sum_l v_x[l] * d2f_l/dz2, stacked as [[vf_xx, vf_ux^T],[vf_ux, vf_uu]]   (ilqrUtils.py:240-247); the projected blocks go
// to separate outputs (:249)
struct ContractedDynamics {
    const double *f_xx, *f_ux, *f_uu, *v_x;
    double *o_xx, *o_ux, *o_uu;
    int n, m;
    __device__ __forceinline__ double load(long mat, int i, int j) const {
        const double* vx = v_x + mat * n;
        const double* p;
        long st;
        if (i < n && j < n) {
            p = f_xx + mat * (long)n * n * n + (long)i * n + j;
            st = (long)n * n;
        } else if (i >= n && j >= n) {
            p = f_uu + mat * (long)n * m * m + (long)(i - n) * m + (j - n);
            st = (long)m * m;
        } else {
            const int u = (i >= n) ? i - n : j - n, x = (i >= n) ? j : i;   // f_ux[l][u][x], also under the transposed block
            p = f_ux + mat * (long)n * m * n + (long)u * n + x;
            st = (long)m * n;
        }
        double acc = 0.0;
        for (int l = 0; l < n; ++l) acc = __builtin_fma(vx[l], p[l * st], acc);
        return acc;
    }
    __device__ __forceinline__ void store(long mat, int i, int j, double v) const {
        if (i < n) {
            if (j < n) o_xx[(mat * n + i) * n + j] = v;
        } else {
            if (j < n)
                o_ux[(mat * m + (i - n)) * n + j] = v;
            else
                o_uu[(mat * m + (i - n)) * m + (j - n)] = v;
        }
    }
};

}  // namespace zm
