// What does an fp64 MFMA / VALU instruction cost in SHADER CYCLES, and what clock does the chip hold under that load?
// Every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop: in-kernel clock = d(memtime) /
// d(memrealtime) * 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6), cycles per instruction = d(memtime) * waves-per-SIMD /
// instructions issued per SIMD.  Launches are repeated for ~1 s per configuration so that the power-managed clock has settled.
// Bodies: (0) v_mfma_f64_16x16x4 only, 4 accumulators; (1) v_fma_f64 only, 8 chains; (2) the K1 mix: 9 big + 3 block MFMAs + 80 fp64
// VALU per "step"; (3) v_mfma_f64_4x4x4_4b only; (4) big MFMAs with a dependent chain on ONE accumulator.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_clock_f64 tools/ubench_clock_f64.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BODY>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* stamps, int iters) {
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double s0 = a, s1 = b, s2 = a + 1, s3 = b + 1, s4 = a + 2, s5 = b + 2, s6 = a + 3, s7 = b + 3;
    double t0 = 0.0;
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (BODY == 0) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if constexpr (BODY == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0 = __builtin_fma(s0, a, b); s1 = __builtin_fma(s1, a, b); s2 = __builtin_fma(s2, a, b); s3 = __builtin_fma(s3, a, b);
                s4 = __builtin_fma(s4, a, b); s5 = __builtin_fma(s5, a, b); s6 = __builtin_fma(s6, a, b); s7 = __builtin_fma(s7, a, b);
            }
        } else if constexpr (BODY == 2) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c0[0], b, t0, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c0[1], b, t0, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c0[2], b, t0, 0, 0, 0);
            s0 += t0;
#pragma unroll
            for (int r = 0; r < 10; ++r) {   // 80 dependent-ish fp64 VALU ops (8 chains x 10)
                s0 = __builtin_fma(s0, a, b); s1 = __builtin_fma(s1, a, s0); s2 = __builtin_fma(s2, a, s1); s3 = __builtin_fma(s3, a, s2);
                s4 = __builtin_fma(s4, a, s3); s5 = __builtin_fma(s5, a, s4); s6 = __builtin_fma(s6, a, s5); s7 = __builtin_fma(s7, a, s6);
            }
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s7, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(s7, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1[0], c2[0], c3, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1[1], c2[1], c3, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1[2], c2[2], c3, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1[3], c2[3], c3, 0, 0, 0);
        } else if constexpr (BODY == 3) {
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + t0;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = m1 - m0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int BODY>
static int run(const char* name, double insts_per_iter, int iters, double* out, unsigned long long* stamps) {
    const int simds = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves : {1, 2, 4}) {
        const int blocks = simds * waves;
        float ms = 0, total = 0;
        int reps = 0;
        while (total < 900.f && reps < 400) {          // ~1 s of back-to-back launches: settled clock
            hipEventRecord(e0);
            k<BODY><<<blocks, 64>>>(out, stamps, iters);
            hipEventRecord(e1);
            CHK(hipDeviceSynchronize());
            hipEventElapsedTime(&ms, e0, e1);
            total += ms;
            ++reps;
        }
        std::vector<unsigned long long> h(2 * blocks);
        CHK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> clk(blocks), cyc(blocks);
        for (int i = 0; i < blocks; ++i) {
            clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;   // GHz
            cyc[i] = (double)h[2 * i];
        }
        std::nth_element(clk.begin(), clk.begin() + blocks / 2, clk.end());
        std::nth_element(cyc.begin(), cyc.begin() + blocks / 2, cyc.end());
        const double per_simd = insts_per_iter * iters * waves;
        printf("%-28s waves/SIMD=%d  last launch %8.3f ms (%d launches)  in-kernel clock %.3f GHz  %7.2f shader cycles and %6.2f ns per "
               "instruction(-group) per SIMD\n", name, waves, ms, reps, clk[blocks / 2], cyc[blocks / 2] * waves / per_simd,
               ms * 1e6 / per_simd);
    }
    return 0;
}

int main() {
    double* out;
    unsigned long long* stamps;
    CHK(hipMalloc(&out, 8192 * 64 * 8));
    CHK(hipMalloc(&stamps, 8192 * 16));
    if (run<0>("mfma_f64_16x16x4 (4 acc)", 4, 20000, out, stamps)) return 1;
    if (run<4>("mfma_f64_16x16x4 (1 chain)", 4, 20000, out, stamps)) return 1;
    if (run<3>("mfma_f64_4x4x4_4b (4 acc)", 4, 80000, out, stamps)) return 1;
    if (run<1>("v_fma_f64 (8 chains)", 32, 40000, out, stamps)) return 1;
    if (run<2>("K1 mix: step = 9+3 MFMA + 80 VALU", 1, 8000, out, stamps)) return 1;
    return 0;
}
