// fp32 MFMA issue rate on gfx950 from 1 / 2 / 4 waves per SIMD: independent accumulators vs one dependent chain.
// Answers: can ONE wave per SIMD keep the fp32 matrix pipe full (needed by the n=64 wave-per-trajectory Riccati kernel)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    const int l = threadIdx.x;
    float a = 1.0f + l * 1e-3f, b = 0.5f + l * 1e-4f;
    if constexpr (MODE == 0 || MODE == 1) {            // 16x16x4 f32: 8 passes
        f4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = MODE == 0 ? i : 0;       // MODE 1: every MFMA accumulates into the same tile (dependent chain)
                c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        out[blockIdx.x * 64 + l] = s;
    } else {                                           // 32x32x2 f32: 16 passes
        f16 c[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = MODE == 2 ? (i & 3) : 0;
                c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
        out[blockIdx.x * 64 + l] = s;
    }
}

template <int MODE>
int run(const char* name, double flop_per_mfma, int blocks) {
    float* out;
    CHK(hipMalloc(&out, blocks * 64 * 4));
    const int iters = 20000;
    k<MODE><<<blocks, 64>>>(out, 10);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, iters);
    hipEventRecord(e1);
    CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8;
    printf("%-36s waves/SIMD=%d  ns per MFMA per SIMD = %6.2f   chip TFLOP/s = %7.1f\n", name, blocks / 1024,
           ms * 1e6 / (n * (blocks / 1024)), n * blocks * flop_per_mfma / (ms * 1e-3) / 1e12);
    hipFree(out);
    return 0;
}

int main() {
    for (int blocks : {1024, 2048, 4096}) {
        run<0>("mfma_f32_16x16x4f32 8 independent", 2048, blocks);
        run<1>("mfma_f32_16x16x4f32 dependent chain", 2048, blocks);
        run<2>("mfma_f32_32x32x2f32 4 independent", 4096, blocks);
        run<3>("mfma_f32_32x32x2f32 dependent chain", 4096, blocks);
    }
    return 0;
}
