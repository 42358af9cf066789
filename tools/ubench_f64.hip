// Micro-benchmarks of the gfx950 instructions the Riccati step is built from (fp64 MFMA, fp64 VALU, division).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_f64.hip -o tools/ubench_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, long long* cyc) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 0.5 + l * 1e-4;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {  // dependent MFMA chain (accumulate)
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        } else if constexpr (MODE == 1) {  // 4 independent accumulators
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if constexpr (MODE == 2) {  // MFMA whose B operand is the previous result (D -> B chain)
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[0], c1, 0, 0, 0);
        } else if constexpr (MODE == 3) {  // dependent fma chain
            x0 = __builtin_fma(x0, a, b);
        } else if constexpr (MODE == 4) {  // 4 independent fma
            x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
        } else if constexpr (MODE == 5) {  // dependent division chain
            x0 = 1.0 / (x0 + 1.5);
        } else if constexpr (MODE == 6) {  // 4 independent divisions
            x0 = 1.0 / (x0 + 1.5); x1 = 1.0 / (x1 + 1.5); x2 = 1.0 / (x2 + 1.5); x3 = 1.0 / (x3 + 1.5);
        } else if constexpr (MODE == 7) {  // v_rcp_f64 only, independent x4
            x0 = __builtin_amdgcn_rcp(x0 + 1.5); x1 = __builtin_amdgcn_rcp(x1 + 1.5); x2 = __builtin_amdgcn_rcp(x2 + 1.5); x3 = __builtin_amdgcn_rcp(x3 + 1.5);
        } else if constexpr (MODE == 8) {  // cndmask-heavy: 4 selects on doubles
            bool s = x0 > x1; double t = s ? x0 : x1; x1 = s ? x1 : x0; x0 = t + 1e-9; bool s2 = x2 > x3; double t2 = s2 ? x2 : x3; x3 = s2 ? x3 : x2; x2 = t2 + 1e-9;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3;
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
int run(const char* name, int per_iter, int blocks) {
    double* out; long long* cyc; long long h;
    CHK(hipMalloc(&out, blocks * 64 * 8)); CHK(hipMalloc(&cyc, 8));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, 100, cyc);
    CHK(hipDeviceSynchronize());
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, iters, cyc);
    hipEventRecord(e1);
    CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s blocks=%5d  memtime-ticks/op = %8.2f   wall ns/op(wave) = %8.2f\n", name, blocks, (double)h / iters / per_iter, ms * 1e6 / iters / per_iter);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    // 1 wave on the chip: pure latency; 1024 = 1 wave per SIMD; 4096 = 4 waves per SIMD (throughput / sharing)
    for (int blocks : {1, 1024, 4096}) {
        run<0>("mfma_f64_16x16x4 dependent acc chain", 1, blocks);
        run<1>("mfma_f64_16x16x4 4 independent", 4, blocks);
        run<2>("mfma_f64_16x16x4 D->B operand chain", 1, blocks);
        run<3>("v_fma_f64 dependent", 1, blocks);
        run<4>("v_fma_f64 4 independent", 4, blocks);
        run<5>("fp64 division (1/x) dependent", 1, blocks);
        run<6>("fp64 division (1/x) 4 independent", 4, blocks);
        run<7>("v_rcp_f64 4 independent", 4, blocks);
        run<8>("f64 compare+2 selects x2 (+adds)", 2, blocks);
    }
    return 0;
}
