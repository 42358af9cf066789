// One explicit-Euler step of the quadcopter's inertial dynamics in still air, x+ = x + dt * inertialDynamics(x, u), given the sines
// and cosines of its Euler angles (quad_inertial_dynamics<double> of models.h with tan = sin / cos; reference quadcopter.py:23-144).
// Shared by the rollout kernels (rollout_fast.hip, rollout_quad.hip) so that they compute the same bits.
#pragma once

namespace zm {

// Every product-sum below is written out as the FMA it should be, under `fp contract(off)`: left to the compiler's contraction, which
// of two products of an `a b + c d` gets fused depends on how many uses each product has AFTER inlining, i.e. on the kernel the
// function is inlined into -- and two kernels then differ by an ulp now and then (seen when rollout_fast.hip changed in round 3).
__device__ __forceinline__ void quad_euler_step_trig(const double (&x)[12], const double (&u)[4], const double dt, const double sphi,
                                                 const double cphi, const double sth, const double cth, const double spsi,
                                                 const double cpsi, double (&xn)[12]) {
#pragma clang fp contract(off)
    constexpr int RN = 12;
    constexpr double g = 9.807, mass = 2.5;
    const double icth = 1.0 / cth;   // one division: tan(theta) = sin * (1 / cos), and the two quotients of the psi-dot row
    const double tth = sth * icth;
    const double fa0 = __builtin_fma(-0.05, x[0] * x[0], -0.2 * x[0]);
    const double fa1 = __builtin_fma(-0.05, x[1] * x[1], -0.2 * x[1]);
    const double fa2 = __builtin_fma(-0.1, x[2] * x[2], -0.3 * x[2]);
    const double sc = sphi * cth, cc = cphi * cth;
    const double ft0 = __builtin_fma(mass * g, -sth, fa0);
    const double ft1 = __builtin_fma(mass * g, sc, fa1);
    const double ft2 = __builtin_fma(mass * g, cc, __builtin_fma(mass, -u[0], fa2));
    const double c0 = __builtin_fma(x[4], x[2], -(x[5] * x[1]));
    const double c1 = __builtin_fma(x[5], x[0], -(x[3] * x[2]));
    const double c2 = __builtin_fma(x[3], x[1], -(x[4] * x[0]));
    double xd[RN];
    xd[0] = (1.0 / mass) * (ft0 - c0);
    xd[1] = (1.0 / mass) * (ft1 - c1);
    xd[2] = (1.0 / mass) * (ft2 - c2);
    xd[3] = __builtin_fma(-0.1, x[3], u[1]);
    xd[4] = __builtin_fma(-0.1, x[4], u[2]);
    xd[5] = __builtin_fma(-0.05, x[5], u[3]);
    xd[6] = __builtin_fma(cphi * tth, x[5], __builtin_fma(sphi * tth, x[4], x[3]));
    xd[7] = __builtin_fma(cphi, x[4], -(sphi * x[5]));
    xd[8] = __builtin_fma(cphi * icth, x[5], (sphi * icth) * x[4]);
    const double ss = sphi * sth, cs = cphi * sth;
    const double r01 = __builtin_fma(ss, cpsi, -(cphi * spsi)), r02 = __builtin_fma(cs, cpsi, -(sphi * spsi));
    const double r11 = __builtin_fma(ss, spsi, cphi * cpsi), r12 = __builtin_fma(cs, spsi, -(sphi * cpsi));
    xd[9] = __builtin_fma(r02, x[2], __builtin_fma(r01, x[1], (cth * cpsi) * x[0]));
    xd[10] = __builtin_fma(r12, x[2], __builtin_fma(r11, x[1], (cth * spsi) * x[0]));
    xd[11] = __builtin_fma(cc, x[2], __builtin_fma(sc, x[1], (-sth) * x[0]));
#pragma unroll
    for (int i = 0; i < RN; ++i) xn[i] = __builtin_fma(dt, xd[i], x[i]);
}

}  // namespace zm
