// Probes v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result lane layout (by one-hot inputs) and issue cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
__global__ __launch_bounds__(64) void timing(double* out, int iters, int mode) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 0.5 + l * 1e-4;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {   // independent
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        } else {           // dependent chain
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        }
    }
    out[blockIdx.x * 64 + l] = c0 + c1 + c2 + c3;
}

int main() {
    double *a, *b, *d, ha[64], hb[64], hd[64];
    CHK(hipMalloc(&a, 512)); CHK(hipMalloc(&b, 512)); CHK(hipMalloc(&d, 512));
    // A one-hot at lane la (value 1), B all lanes = 100 + lane: which D lanes become nonzero, and with which B lane's value?
    printf("A lane -> (D lane : B lane whose value it picked up)\n");
    for (int la = 0; la < 64; ++la) {
        for (int i = 0; i < 64; ++i) { ha[i] = (i == la) ? 1.0 : 0.0; hb[i] = 100.0 + i; }
        hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(a, b, d); CHK(hipDeviceSynchronize());
        hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
        printf("A%02d:", la);
        for (int i = 0; i < 64; ++i) if (hd[i] != 0.0) printf(" D%02d<-B%02d", i, (int)(hd[i] - 100.0));
        printf("\n");
    }
    double* out; CHK(hipMalloc(&out, 4096 * 64 * 8));
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks : {1024, 4096}) {
            timing<<<blocks, 64>>>(out, 10, mode); CHK(hipDeviceSynchronize());
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 20000;
            hipEventRecord(e0); timing<<<blocks, 64>>>(out, iters, mode); hipEventRecord(e1); CHK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s  waves/SIMD=%d: ns per mfma_4x4x4 per SIMD = %.2f\n", mode ? "dependent  " : "independent", blocks / 1024,
                   ms * 1e6 / ((double)iters * 4 * (blocks / 1024)));
        }
    return 0;
}
