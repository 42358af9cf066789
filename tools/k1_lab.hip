// K1 laboratory: the product kernel lqr_backward_dma_f64 (included as source) under diagnostic template bits, timed A/B in ONE
// process with interleaved rounds on random resident inputs (BASELINE configs[1]: 4096 x T=50, n=12, m=4), plus a per-segment
// s_memtime profile at 1..4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZM_K1_LAB -Iinclude -Izopt_amd/csrc
//   -o tools/k1_lab tools/k1_lab.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../zopt_amd/csrc/lqr_backward_dma.hip"

namespace zm {
char* last_error_buf() { static char b[256]; return b; }
int set_error(int code, const char*, ...) { return code; }
}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill(double* p, size_t n, unsigned seed, double scale, int diag_n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        double u = ((z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
        double v = u * scale;
        if (diag_n) { size_t e = i % ((size_t)diag_n * diag_n); if (e / diag_n == e % diag_n) v += 1.0 + 0.5 * diag_n * scale; }
        p[i] = v;
    }
}

typedef void (*kern_t)(const double*, const double*, const double*, const double*, double*, int, long);
struct Variant { const char* name; kern_t k; int W; };

int main(int argc, char** argv) {
    const int batch = 4096, T = 50, n = 12, m = 4;
    const size_t nA = (size_t)batch * T * n * n, nB = (size_t)batch * T * n * m, nR = (size_t)batch * T * m * m;
    double *A[2], *B[2], *Q[2], *R[2], *L, *Lr[4];
    int rot = 0;
    for (int s = 0; s < 2; ++s) {
        CHK(hipMalloc(&A[s], nA * 8)); CHK(hipMalloc(&B[s], nB * 8)); CHK(hipMalloc(&Q[s], nA * 8)); CHK(hipMalloc(&R[s], nR * 8));
        fill<<<2048, 256>>>(A[s], nA, 11 + s, 0.9 / 3.4641 * 1.7, 0);
        fill<<<2048, 256>>>(B[s], nB, 21 + s, 1.0, 0);
        fill<<<2048, 256>>>(Q[s], nA, 31 + s, 0.05, n);    // diagonally dominant, nonsymmetric
        fill<<<2048, 256>>>(R[s], nR, 41 + s, 0.05, m);
    }
    CHK(hipMalloc(&L, nB * 8));
    for (int i = 0; i < 4; ++i) CHK(hipMalloc(&Lr[i], nB * 8));
    if (argc > 1) rot = atoi(argv[1]);   // rotate the output over 4 buffers (314 MB > Infinity Cache)
    int lcount = 0;
    CHK(hipDeviceSynchronize());
    using namespace zm;
#define XAUX(aux) (16 | ((aux) << 8))
#define SM(k) ((k) << 13)
#define SP(k) ((k) << 16)
    std::vector<Variant> vs = {
        {"product (nt loads, D=3, 4 w/SIMD)", lqr_backward_dma_f64<12, 4, 3, true, 0, 1, 4>, 1},
        {"compiler-only LDS sync (X bit 21)", lqr_backward_dma_f64<12, 4, 3, true, (1 << 21), 1, 4>, 1},
        {"L2-resident                      ", lqr_backward_dma_f64<12, 4, 3, true, 1, 1, 4>, 1},
        {"L2-resident, compiler-only sync  ", lqr_backward_dma_f64<12, 4, 3, true, 1 | (1 << 21), 1, 4>, 1},
    };
    auto launch = [&](kern_t k, int blocks, int set, size_t dyn, int W = 1) {
        hipLaunchKernelGGL(k, dim3(blocks / W), dim3(64 * W), dyn, 0, A[set], B[set], Q[set], R[set], rot ? Lr[(lcount++) & 3] : L, T, (long)blocks);
    };
    {   // the variants that keep the arithmetic must reproduce the product's output bit for bit
        std::vector<double> ref(nB), got(nB);
        launch(vs[0].k, batch, 0, 0, 1);
        CHK(hipMemcpy(ref.data(), L, nB * 8, hipMemcpyDeviceToHost));
        for (size_t v : {(size_t)1}) {
            if (rot) break;
            CHK(hipMemset(L, 0xff, nB * 8));
            launch(vs[v].k, batch, 0, 0, vs[v].W);
            CHK(hipMemcpy(got.data(), L, nB * 8, hipMemcpyDeviceToHost));
            size_t bad = 0, nan = 0;
            for (size_t i = 0; i < nB; ++i) { bad += (memcmp(&ref[i], &got[i], 8) != 0); nan += (ref[i] != ref[i]); }
            printf("bitwise check %s: %zu of %zu differ (reference has %zu NaN)\n", vs[v].name, bad, nB, nan);
        }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // warm the clock
    for (int i = 0; i < 600; ++i) launch(vs[0].k, batch, i & 1, 0);
    CHK(hipDeviceSynchronize());
    const int rounds = 7, per = 60;
    std::vector<std::vector<float>> t(vs.size());
    for (int r = 0; r < rounds; ++r)
        for (size_t v = 0; v < vs.size(); ++v) {
            for (int i = 0; i < 10; ++i) launch(vs[v].k, batch, i & 1, 0, vs[v].W);
            hipEventRecord(e0);
            for (int i = 0; i < per; ++i) launch(vs[v].k, batch, i & 1, 0, vs[v].W);
            hipEventRecord(e1);
            CHK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            t[v].push_back(ms * 1e3f / per);
        }
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(t[v].begin(), t[v].end());
        printf("%s  us/launch: min %7.2f  median %7.2f  max %7.2f   (frac of 8 TB/s at median: %.3f)\n", vs[v].name, t[v][0], t[v][rounds / 2],
               t[v][rounds - 1], 655.36 / t[v][rounds / 2] / 8000.0 * 1e3);
    }
    // occupancy sweep: 1..4 waves per SIMD = 4..16 blocks per CU (dynamic LDS padding), grid = exactly one resident round
    printf("\noccupancy sweep (grid = 256 CUs x 4 SIMDs x w blocks, one resident round; time per launch and per step-round)\n");
    for (int w = 1; w <= 4; ++w) {
        const size_t lds_static = 3 * 3072 + 512 + 384;
        const size_t want = 160 * 1024 / (4 * w);                    // per-block LDS so that exactly 4w blocks fit a CU
        const size_t dyn = want > lds_static + 64 ? ((want - lds_static) & ~(size_t)15) - (w == 4 ? 0 : 0) : 0;
        for (int x : {0, 1}) {
            kern_t k = x ? (kern_t)lqr_backward_dma_f64<12, 4, 3, true, 1, 1> : (kern_t)lqr_backward_dma_f64<12, 4, 3, true, 0, 1>;
            CHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            const int blocks = 1024 * w;
            for (int i = 0; i < 30; ++i) launch(k, blocks, i & 1, dyn);
            hipEventRecord(e0);
            for (int i = 0; i < 100; ++i) launch(k, blocks, i & 1, dyn);
            hipEventRecord(e1);
            CHK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  waves/SIMD=%d %s  %7.2f us/launch  %6.3f us per step-round  dynLDS=%zu\n", w, x ? "L2-resident" : "HBM        ",
                   ms * 10.f, ms * 10.f / T, dyn);
        }
    }
    // per-segment stamps (diagnostic build: its waits forbid overlaps -- read SHARES)
    printf("\nsegment stamps, cycles per wave-step: [wait+operand reads+DMA issue | Y, Y_B round trip, 4x4x4 | exch round trip | solve+store | 6 MFMA]\n");
    for (int w : {1, 4}) {
        const size_t lds_static = 3 * 3072 + 512 + 384;
        const size_t want = 160 * 1024 / (4 * w);
        const size_t dyn = want > lds_static + 64 ? ((want - lds_static) & ~(size_t)15) : 0;
        kern_t k = (kern_t)lqr_backward_dma_f64<12, 4, 3, true, 4, 1>;
        CHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        unsigned long long z[8] = {0};
        for (int i = 0; i < 20; ++i) launch(k, 1024 * w, i & 1, dyn);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpyToSymbol(HIP_SYMBOL(zm_k1_stamps), z, sizeof(z)));
        hipEventRecord(e0);
        launch(k, 1024 * w, 0, dyn);
        hipEventRecord(e1);
        CHK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        CHK(hipMemcpyFromSymbol(z, HIP_SYMBOL(zm_k1_stamps), sizeof(z)));
        double tot = 0;
        for (int q = 0; q < 5; ++q) tot += (double)z[q];
        printf("  waves/SIMD=%d  (%.1f us)  ", w, ms * 1e3);
        for (int q = 0; q < 5; ++q) printf("%8.0f ", (double)z[q] / ((double)z[5] * T));
        printf(" total %8.0f\n", tot / ((double)z[5] * T));
    }
    return 0;
}
