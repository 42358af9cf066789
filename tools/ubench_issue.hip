// Per-instruction SIMD issue cost on gfx950 at 1 and 4 waves per SIMD (inline asm, 16 independent instrs per iter).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, long long* cyc) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 0.5 + l * 1e-4, c = 0.25;
    double r0 = a, r1 = b, r2 = a + 1, r3 = b + 1;
    int i0 = l, i1 = l + 1, i2 = l + 2;
    unsigned long long m = (l & 1) ? ~0ull : 0ull;
    m = __builtin_amdgcn_readfirstlane((int)m) | 0x5555555555555555ull;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) { REP16(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b));) }
        if constexpr (MODE == 1) { asm volatile(REP16("v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n") : "+v"(r0), "+v"(r1) : "v"(a), "v"(b)); }
        if constexpr (MODE == 2) { asm volatile(REP16("v_cndmask_b32_e64 %0, %2, %3, %4\n v_cndmask_b32_e64 %1, %3, %2, %4\n") : "+v"(i0), "+v"(i1) : "v"(i2), "v"(l), "s"(m)); }
        if constexpr (MODE == 3) { asm volatile(REP16("v_mov_b64_e32 %0, %2\n v_mov_b64_e32 %1, %3\n") : "+v"(r0), "+v"(r1) : "v"(a), "v"(b)); }
        if constexpr (MODE == 4) { asm volatile(REP16("v_mov_b32_e32 %0, %2\n v_mov_b32_e32 %1, %3\n") : "+v"(i0), "+v"(i1) : "v"(i2), "v"(l)); }
        if constexpr (MODE == 5) { asm volatile(REP16("v_mul_f64 %0, %2, %3\n v_mul_f64 %1, %3, %2\n") : "+v"(r0), "+v"(r1) : "v"(a), "v"(b)); }
        if constexpr (MODE == 6) { asm volatile(REP16("v_cmp_gt_f64_e64 s[20:21], |%0|, |%1|\n v_cmp_gt_f64_e64 s[22:23], |%1|, |%0|\n") :: "v"(a), "v"(b) : "s20", "s21", "s22", "s23"); }
        if constexpr (MODE == 7) { asm volatile(REP16("v_rcp_f64_e32 %0, %2\n v_rcp_f64_e32 %1, %3\n") : "+v"(r0), "+v"(r1) : "v"(a), "v"(b)); }
        if constexpr (MODE == 8) { asm volatile(REP16("v_div_scale_f64 %0, vcc, %2, %3, %2\n v_div_fixup_f64 %1, %2, %3, %2\n") : "+v"(r0), "+v"(r1) : "v"(a), "v"(b) : "vcc"); }
        if constexpr (MODE == 9) { asm volatile(REP16("v_add_f64 %0, %2, %3\n v_add_f64 %1, %3, %2\n") : "+v"(r0), "+v"(r1) : "v"(a), "v"(b)); }
        if constexpr (MODE == 10) { asm volatile(REP16("v_mov_b32_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n") : "+v"(i0), "+v"(i1) : "v"(i2), "v"(l)); }
        if constexpr (MODE == 11) { asm volatile(REP16("ds_bpermute_b32 %0, %2, %3\n ds_bpermute_b32 %1, %2, %3\n") "s_waitcnt lgkmcnt(0)" : "+v"(i0), "+v"(i1) : "v"(i2), "v"(l)); }
        if constexpr (MODE == 12) { asm volatile(REP16("v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 7\n") :: "v"(i2), "v"(l) : "s20", "s21"); }
        if constexpr (MODE == 13) { asm volatile(REP16("v_max_f64 %0, %2, %3\n v_max_f64 %1, %3, %2\n") : "+v"(r0), "+v"(r1) : "v"(a), "v"(b)); }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = r0 + r1 + r2 + r3 + i0 + i1 + c;
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
int run(const char* name, int per_iter, int blocks) {
    double* out; long long* cyc; long long h;
    CHK(hipMalloc(&out, blocks * 64 * 8)); CHK(hipMalloc(&cyc, 8));
    const int iters = 4000;
    k<MODE><<<blocks, 64>>>(out, 10, cyc);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, iters, cyc);
    hipEventRecord(e1);
    CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double t = (double)h / iters / per_iter;
    const int wps = blocks / 1024 > 0 ? blocks / 1024 : 1;
    printf("%-34s waves/SIMD=%d  ticks/instr(wave) = %7.2f  SIMD cycles/instr = %6.2f  wall ns/instr = %6.2f\n", name, wps, t, t / wps, ms * 1e6 / iters / per_iter);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    for (int blocks : {1024, 4096}) {
        run<0>("v_fma_f64 dependent (same acc)", 16, blocks);
        run<1>("v_fma_f64 2 chains", 32, blocks);
        run<5>("v_mul_f64", 32, blocks);
        run<9>("v_add_f64", 32, blocks);
        run<13>("v_max_f64", 32, blocks);
        run<2>("v_cndmask_b32_e64 (sgpr mask)", 32, blocks);
        run<3>("v_mov_b64", 32, blocks);
        run<4>("v_mov_b32", 32, blocks);
        run<6>("v_cmp_gt_f64_e64 |a|,|b|", 32, blocks);
        run<7>("v_rcp_f64", 32, blocks);
        run<8>("v_div_scale/v_div_fixup_f64", 32, blocks);
        run<10>("v_mov_b32_dpp row_newbcast", 32, blocks);
        run<11>("ds_bpermute_b32", 32, blocks);
        run<12>("v_readlane_b32", 32, blocks);
    }
    return 0;
}
