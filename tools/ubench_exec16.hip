// Does a wave with only 16 (or 32) of its 64 lanes enabled issue fp64 VALU instructions faster?  (If the SIMD skipped the disabled
// quarter-waves, a lone 16-lane wave would run the MPC chain up to 4x faster than with four instances packed into one wave.)
// hipcc --offload-arch=gfx950 -O3 tools/ubench_exec16.hip -o tools/ubench_exec16 && tools/ubench_exec16
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, int nact) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 0.5 + l * 1e-4;
    double r0 = a, r1 = b, r2 = a + 1, r3 = b + 1;
    if (l < nact) {
        for (int i = 0; i < iters; ++i) {
            if constexpr (MODE == 0) { REP16(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b));) }
            if constexpr (MODE == 1) { asm volatile(REP16("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b)); }
            if constexpr (MODE == 2) { asm volatile(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b)); }
        }
    }
    out[blockIdx.x * 64 + l] = r0 + r1 + r2 + r3;
}

template <int MODE>
int run(const char* name, int per_iter, int blocks, int nact) {
    double* out;
    CHK(hipMalloc(&out, blocks * 64 * 8));
    const int iters = 20000;
    k<MODE><<<blocks, 64>>>(out, 10, nact);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    k<MODE><<<blocks, 64>>>(out, iters, nact);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s blocks=%5d active lanes=%2d  wall ns per instruction (per wave) = %6.2f\n", name, blocks, nact, ms * 1e6 / iters / per_iter);
    CHK(hipFree(out));
    return 0;
}

int main() {
    for (int blocks : {256, 1024}) {
        for (int nact : {64, 32, 16}) {
            run<0>("v_fma_f64 dependent", 16, blocks, nact);
            run<1>("v_fma_f64 4 chains", 64, blocks, nact);
            run<2>("v_fmac_f64_dpp 4 chains", 64, blocks, nact);
        }
    }
    return 0;
}
