// One explicit-Euler step of the quadcopter's inertial dynamics in still air, x+ = x + dt * inertialDynamics(x, u), given the sines
// and cosines of its Euler angles (quad_inertial_dynamics<double> of models.h with tan = sin / cos; reference quadcopter.py:23-144).
// Shared by the rollout kernels (rollout_fast.hip, rollout_quad.hip) so that they compute the same bits.
#pragma once

namespace zm {

__device__ __forceinline__ void quad_euler_step_trig(const double (&x)[12], const double (&u)[4], const double dt, const double sphi,
                                                 const double cphi, const double sth, const double cth, const double spsi,
                                                 const double cpsi, double (&xn)[12]) {
    constexpr int RN = 12;
    constexpr double g = 9.807, mass = 2.5;
    const double icth = 1.0 / cth;   // one division: tan(theta) = sin * (1 / cos), and the two quotients of the psi-dot row
    const double tth = sth * icth;
    const double fa0 = -0.2 * x[0] + -0.05 * (x[0] * x[0]);
    const double fa1 = -0.2 * x[1] + -0.05 * (x[1] * x[1]);
    const double fa2 = -0.3 * x[2] + -0.1 * (x[2] * x[2]);
    const double ft0 = fa0 + (mass * g) * (-sth);
    const double ft1 = fa1 + (mass * g) * (sphi * cth);
    const double ft2 = ((mass * (-u[0])) + fa2) + (mass * g) * (cphi * cth);
    const double c0 = x[4] * x[2] - x[5] * x[1];
    const double c1 = x[5] * x[0] - x[3] * x[2];
    const double c2 = x[3] * x[1] - x[4] * x[0];
    double xd[RN];
    xd[0] = (1.0 / mass) * (ft0 - c0);
    xd[1] = (1.0 / mass) * (ft1 - c1);
    xd[2] = (1.0 / mass) * (ft2 - c2);
    xd[3] = u[1] + -0.1 * x[3];
    xd[4] = u[2] + -0.1 * x[4];
    xd[5] = u[3] + -0.05 * x[5];
    xd[6] = (x[3] + (sphi * tth) * x[4]) + (cphi * tth) * x[5];
    xd[7] = cphi * x[4] - sphi * x[5];
    xd[8] = (sphi * icth) * x[4] + (cphi * icth) * x[5];
    xd[9] = ((cth * cpsi) * x[0] + (sphi * sth * cpsi - cphi * spsi) * x[1]) + (cphi * sth * cpsi - sphi * spsi) * x[2];
    xd[10] = ((cth * spsi) * x[0] + (sphi * sth * spsi + cphi * cpsi) * x[1]) + (cphi * sth * spsi - sphi * cpsi) * x[2];
    xd[11] = ((-sth) * x[0] + (sphi * cth) * x[1]) + (cphi * cth) * x[2];
#pragma unroll
    for (int i = 0; i < RN; ++i) xn[i] = x[i] + dt * xd[i];
}

}  // namespace zm
