// Sustained fp64 MFMA rate of one MI355X: v_mfma_f64_16x16x4_f64 from 1 / 2 / 4 waves per SIMD (4 independent accumulators
// per wave), launch after launch for about a second per configuration, so that the power-managed shader clock has settled.
// Prints ns per MFMA per SIMD and the chip-wide TFLOP/s of each launch.   hipcc --offload-arch=gfx950 -O3 -o ubench ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k(double* out, int iters) {
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    double* out;
    CHK(hipMalloc(&out, 8192 * 64 * 8));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int simds = 256 * 4;
    for (int waves : {1, 2, 4}) {
        const int blocks = simds * waves, iters = 100000;   // 400 000 MFMAs per wave and launch
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(e0);
            k<<<blocks, 64>>>(out, iters);
            hipEventRecord(e1);
            CHK(hipDeviceSynchronize());
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n_per_simd = 4.0 * iters * waves;
            printf("waves/SIMD=%d launch %2d: %7.2f ms  %6.2f ns per MFMA per SIMD  %6.1f TFLOP/s\n", waves, rep, ms,
                   ms * 1e6 / n_per_simd, 2048.0 * n_per_simd * simds / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
