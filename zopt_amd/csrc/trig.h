// sin and cos of one fp64 argument together, for the model code (models.h, rollout_fast.hip): the library sincos is ~150 VALU
// instructions (it carries a Payne-Hanek path for huge arguments), three of them per model evaluation.  Here:
// Cody-Waite reduction by pi/2 in two FMA steps (exact cancellation by the fused product; the neglected third term is
// k * 1.5e-33) and the minimax kernels of fdlibm's __kernel_sin / __kernel_cos on [-pi/4, pi/4] (Sun Microsystems' public
// coefficient tables): ~35 instructions; measured against 200-bit arithmetic (the same operation sequence in Python): <= 1.4 ulp
// for |x| <= 1e6, except right next to the zeros of sin / cos at large arguments, where the error is <= 3e-22 absolute.  Beyond
// that the absolute error grows like 1.5e-33 |x| (the neglected third term of pi/2) up to |x| ~ 2^52; inf and NaN give NaN.
#pragma once
#include <hip/hip_runtime.h>

namespace zm {

__device__ __forceinline__ void zm_sincos(const double x, double* sn, double* cs) {
#pragma clang fp contract(off)   // every FMA below is explicit: the same bits in whichever kernel this is inlined (see quad_step.h)
#ifdef ZM_LIBM_SINCOS   // A/B builds: the library everywhere
    sincos(x, sn, cs);
    return;
#endif
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);        // x * 2/pi
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);            // fl(pi/2)
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);                   // pi/2 - fl(pi/2)
    const double z = r * r;
    // sin(r) = r + r^3 (S1 + z (S2 + ...))
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double s = __builtin_fma(r * z, ps, r);
    // cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ...)), the leading terms summed without cancellation error
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + __builtin_fma(z * z, pc, (1.0 - w) - hz);
    // quadrant k mod 4:  0: (s, c)   1: (c, -s)   2: (-s, -c)   3: (-c, s)
    const int q = (int)__builtin_fma(-4.0, __builtin_rint(0.25 * k), k);   // k mod 4 in {-2..2}, no integer overflow for huge k
    const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

}  // namespace zm
