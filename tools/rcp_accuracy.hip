// Accuracy of v_rcp_f64 and of one / two Newton steps on it (relative error vs IEEE division), gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* a, double* e0, double* e1, double* e2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], ex = 1.0 / x;
    double r = __builtin_amdgcn_rcp(x);
    e0[i] = fabs(r - ex) / fabs(ex);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    e1[i] = fabs(r - ex) / fabs(ex);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    e2[i] = fabs(r - ex) / fabs(ex);
}
int main() {
    const int n = 1 << 20;
    double *a, *e0, *e1, *e2;
    hipMallocManaged(&a, n * 8); hipMallocManaged(&e0, n * 8); hipMallocManaged(&e1, n * 8); hipMallocManaged(&e2, n * 8);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20) * ((s & 1) ? 1 : -1); }
    k<<<n / 256, 256>>>(a, e0, e1, e2, n);
    hipDeviceSynchronize();
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m0 = fmax(m0, e0[i]); m1 = fmax(m1, e1[i]); m2 = fmax(m2, e2[i]); }
    printf("max rel err: v_rcp_f64 %.3e   +1 Newton %.3e   +2 Newton %.3e   (eps = %.3e)\n", m0, m1, m2, ldexp(1.0, -52));
    return 0;
}
