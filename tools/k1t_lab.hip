// K1-T laboratory: the tiled fp32 sweep (n = 64, m = 16) with per-phase s_memtime stamps, one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZM_TILED_LAB -Iinclude -Izopt_amd/csrc -o tools/k1t_lab tools/k1t_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../zopt_amd/csrc/lqr_tiled_core.h"
namespace zm {
char* last_error_buf() { static char b[256]; return b; }
int set_error(int code, const char*, ...) { return code; }
}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fillf(float* p, size_t n, unsigned seed, float scale, int diag_n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        float u = (float)(((z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0) * scale;
        if (diag_n) { size_t e = i % ((size_t)diag_n * diag_n); if (e / diag_n == e % diag_n) u += 1.0f + 0.5f * diag_n * scale; }
        p[i] = u;
    }
}
int main() {
    const int batch = 1024, T = 60, n = 64, m = 16;
    const size_t nA = (size_t)batch * T * n * n, nB = (size_t)batch * T * n * m, nR = (size_t)batch * T * m * m;
    float *A, *B, *Q, *R, *L;
    CHK(hipMalloc(&A, nA * 4)); CHK(hipMalloc(&B, nB * 4)); CHK(hipMalloc(&Q, nA * 4)); CHK(hipMalloc(&R, nR * 4)); CHK(hipMalloc(&L, nB * 4));
    fillf<<<2048, 256>>>(A, nA, 1, 0.9f / 8.0f * 1.7f, 0);
    fillf<<<2048, 256>>>(B, nB, 2, 1.0f, 0);
    fillf<<<2048, 256>>>(Q, nA, 3, 0.01f, n);
    fillf<<<2048, 256>>>(R, nR, 4, 0.02f, m);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto k = zm::lqr_backward_tiled<zm::TileF32, 4, true>;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(batch), dim3(64), 0, 0, A, B, Q, R, L, (long)batch, T, n, m);
    CHK(hipDeviceSynchronize());
    unsigned long long z[12] = {0};
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(zm::zm_tiled_stamps), z, sizeof(z)));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(batch), dim3(64), 0, 0, A, B, Q, R, L, (long)batch, T, n, m);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpyFromSymbol(z, HIP_SYMBOL(zm::zm_tiled_stamps), sizeof(z)));
    const char* names[7] = {"wait for operands (loop head)", "Y_B + S (144 MFMA) + LDS writes", "solve operands from LDS", "solve || Y_A (256 MFMA)",
                            "L store + -L via LDS", "issue next operand loads", "-RL, Acl, W, V' (464 MFMA)"};
    const int mf[7] = {0, 144, 0, 256, 0, 0, 464};
    double tot = 0;
    for (int q = 0; q < 7; ++q) tot += (double)z[q];
    printf("stamped launch: %.3f ms for %d x T=%d (%.0f cycles per step; MFMA issue 27648)\n", ms, batch, T, tot / ((double)z[7] * T));
    for (int q = 0; q < 7; ++q)
        printf("  %-36s %8.0f cycles per step  (%4.1f %%)   MFMA issue cycles %5d\n", names[q], (double)z[q] / ((double)z[7] * T), 100.0 * z[q] / tot, mf[q] * 32);
    return 0;
}
