// What bounds the expansion kernel (linearize.hip)?  The quadcopter's closed-form Jacobian column per lane, 16 lanes per point, with
// parts switched off one at a time.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izopt_amd/csrc -o tools/lin_lab tools/lin_lab.hip
// FLAGS: 1 no stores, 2 no loads (state from the lane id), 4 no arithmetic, 8 column-strided stores (no LDS transpose), 16 no sincos
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "models.h"
#include "quad_derivs_gen.h"
#include "zm_common.h"
using namespace zm;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PTS, int WAVES, int FLAGS, int MINW>
__global__ __launch_bounds__(64 * WAVES, MINW) void lab(const double* __restrict__ xTraj, const double* __restrict__ uTraj,
                                                        double* __restrict__ f_x, double* __restrict__ f_u, const long batch, const int T,
                                                        const double dt, const double never) {
    __shared__ double tile[4 * WAVES][192];
    const int lane = threadIdx.x, j = lane & 15, q = lane >> 4;
    const int gpt = (T + PTS - 1) / PTS;
    const long gi = (long)blockIdx.x * (4 * WAVES) + q;
    if (gi >= batch * gpt) return;
    const long traj = gi / gpt;
    const int k0 = (int)(gi - traj * gpt) * PTS;
    const int k1 = (k0 + PTS < T) ? k0 + PTS : T;
#pragma unroll 1
    for (int k = k0; k < k1; ++k) {
        const long pt = traj * T + k;
        QuadAtoms a;
        if constexpr (FLAGS & 2) {
#pragma unroll
            for (int i = 0; i < 12; ++i) a.x[i] = 0.01 * (i + 1) + 1e-4 * k;
            a.u0 = 9.8;
        } else {
            const double* xk = xTraj + (traj * (T + 1) + k) * 12;
#pragma unroll
            for (int i = 0; i < 12; ++i) a.x[i] = xk[i];
            a.u0 = uTraj[pt * 4];
        }
        a.w[0] = a.w[1] = a.w[2] = 0.0;
        if constexpr (FLAGS & 16) {
            a.s6 = a.x[6]; a.c6 = 1 - a.x[6]; a.s7 = a.x[7]; a.c7 = 1 - a.x[7]; a.s8 = a.x[8]; a.c8 = 1 - a.x[8];
        } else {
            zm_sincos(a.x[6], &a.s6, &a.c6);
            zm_sincos(a.x[7], &a.s7, &a.c7);
            zm_sincos(a.x[8], &a.s8, &a.c8);
        }
        a.ic7 = 1.0 / a.c7;
        double col[12];
        if constexpr (FLAGS & 4) {
#pragma unroll
            for (int i = 0; i < 12; ++i) col[i] = a.x[i] + j;
        } else {
            double o[12];
            quad_jac_column<false>(j, a, o);
#pragma unroll
            for (int i = 0; i < 12; ++i) col[i] = __builtin_fma(dt, o[i], (i == j) ? 1.0 : 0.0);
        }
        const bool st = (FLAGS & 1) ? (col[0] == never) : true;
        if constexpr (FLAGS & 8) {
            double* dst = (j < 12) ? f_x + pt * 144 + j : f_u + pt * 48 + (j - 12);
            const int sd = (j < 12) ? 12 : 4;
            if (st) {
#pragma unroll
                for (int i = 0; i < 12; ++i) dst[i * sd] = col[i];
            }
        } else {
            wave_lds_sync();
            double* t = tile[q] + ((j < 12) ? j : 144 + (j - 12));
            const int sd = (j < 12) ? 12 : 4;
#pragma unroll
            for (int i = 0; i < 12; ++i) t[i * sd] = col[i];
            wave_lds_sync();
            if (st) {
                double* ox = f_x + pt * 144;
                double* ou = f_u + pt * 48;
#pragma unroll
                for (int e = 0; e < 9; ++e) ox[j + 16 * e] = tile[q][j + 16 * e];
#pragma unroll
                for (int e = 0; e < 3; ++e) ou[j + 16 * e] = tile[q][144 + j + 16 * e];
            }
        }
    }
}

template <int PTS, int WAVES, int FLAGS, int MINW>
static int run(const char* name, const double* x, const double* u, double* fx, double* fu, long batch, int T) {
    const long ngrp = batch * ((T + PTS - 1) / PTS);
    const dim3 grid((unsigned)((ngrp + 4 * WAVES - 1) / (4 * WAVES))), block(64 * WAVES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        lab<PTS, WAVES, FLAGS, MINW><<<grid, block>>>(x, u, fx, fu, batch, T, 0.1, -12345.678);
        hipEventRecord(e1);
        CHK(hipDeviceSynchronize());
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    printf("  batch %5ld  PTS %d waves/WG %d minw %d flags %2d  %-42s %8.1f us   (%6.0f waves/us, %5.2f TB/s written)\n", batch, PTS, WAVES, MINW, FLAGS, name,
           best * 1e3, (double)grid.x * WAVES / (best * 1e3), (FLAGS & 1) ? 0.0 : batch * T * 1536.0 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int T = 100;
    const long B = 8192;
    double *x, *u, *fx, *fu;
    CHK(hipMalloc(&x, B * (T + 1) * 12 * 8));
    CHK(hipMalloc(&u, B * T * 4 * 8));
    CHK(hipMalloc(&fx, B * T * 144 * 8));
    CHK(hipMalloc(&fu, B * T * 48 * 8));
    std::vector<double> hx(B * (T + 1) * 12), hu(B * T * 4);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0.3 * ((i * 2654435761u) % 1000) / 1000.0 - 0.15;
    for (size_t i = 0; i < hu.size(); ++i) hu[i] = 9.8 + 0.1 * ((i * 40503u) % 100) / 100.0;
    CHK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(u, hu.data(), hu.size() * 8, hipMemcpyHostToDevice));
    for (long b : {8192L, 736L}) {
        run<1, 1, 0, 1>("as the product was (1 wave/WG)", x, u, fx, fu, b, T);
        run<1, 1, 8, 1>("  column-strided stores", x, u, fx, fu, b, T);
        run<1, 4, 0, 1>("4 waves/WG", x, u, fx, fu, b, T);
        run<1, 4, 1, 1>("  no stores", x, u, fx, fu, b, T);
        run<1, 4, 2, 1>("  no loads", x, u, fx, fu, b, T);
        run<1, 4, 3, 1>("  no loads, no stores", x, u, fx, fu, b, T);
        run<1, 4, 4, 1>("  no arithmetic (sincos kept)", x, u, fx, fu, b, T);
        run<1, 4, 20, 1>("  no arithmetic, no sincos", x, u, fx, fu, b, T);
        run<1, 4, 23, 1>("  nothing but the launch", x, u, fx, fu, b, T);
        run<5, 4, 0, 1>("5 points per group", x, u, fx, fu, b, T);
        run<5, 4, 0, 2>("5 points per group", x, u, fx, fu, b, T);
        run<5, 4, 0, 3>("5 points per group", x, u, fx, fu, b, T);
        run<10, 4, 0, 2>("10 points per group", x, u, fx, fu, b, T);
        run<25, 4, 0, 2>("25 points per group", x, u, fx, fu, b, T);
        run<5, 1, 0, 2>("5 points per group", x, u, fx, fu, b, T);
    }
    return 0;
}
