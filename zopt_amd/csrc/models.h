// Registered device models for the rollout / linearisation kernels (reference: arbitrary Python callables
// differentiated by JAX; a HIP kernel needs the model in device code -- SURVEY section 7 "registered device models").
//
//   ZM_MODEL_LINEAR     x+ = A x + B u            (time-invariant A (n,n), B (n,m) shared by the batch; n <= 12, m <= 4)
//                       -- the LQ problem of the reference's own iLQR test (tests/test_ilqrUtils.py:167-196)
//   ZM_MODEL_QUADCOPTER x+ = x + dt * inertialDynamics(x, u)   (zopt/quadcopter.py:116-144 via :70-113, :23-67;
//                       demos/iterativeLqr.py:35), n = 12, m = 4;  dt = 0: x+ := inertialDynamics(x, u) itself
//   ZM_MODEL_QUADCOPTER_RB  the 8-state rigidBodyDynamics (:70-113) the reference trims and linearises (:146-201), same dt rule
//   cost                c(x,u) = x^T Q x + u^T R u,  c_f(x) = x^T Qf x   (no 1/2: demos/iterativeLqr.py:12-13,37)
//
// The dynamics are templated on the scalar type so that the same code evaluated on dual numbers gives the
// Jacobians (linearize kernel), as jax.jacobian does for the reference (pytrees.py:139-153).
#pragma once
#include "trig.h"
#include <hip/hip_runtime.h>

#include "../../include/zopt_amd.h"

namespace zm {

constexpr int MAXN = 12;
constexpr int MAXM = 4;

// ---- forward-mode dual number (value + one directional derivative) --------------------------------------------
struct Dual {
    double v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const double q = a.v / b.v;
    return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator+(Dual a, double b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(double a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, double b) { return {a.v / b, a.d / b}; }
__device__ __forceinline__ Dual operator/(double a, Dual b) {
    const double q = a / b.v;
    return {q, -q * b.d / b.v};
}
// sin, cos and tan = sin / cos all come from one zm_sincos (trig.h); after inlining the compiler keeps a single evaluation per angle
__device__ __forceinline__ Dual zsin(Dual a) { double s_, c_; zm_sincos(a.v, &s_, &c_); return {s_, c_ * a.d}; }
__device__ __forceinline__ Dual zcos(Dual a) { double s_, c_; zm_sincos(a.v, &s_, &c_); return {c_, -s_ * a.d}; }
__device__ __forceinline__ Dual ztan(Dual a) {
    double s_, c_;
    zm_sincos(a.v, &s_, &c_);
    const double t = s_ / c_;
    return {t, (1.0 + t * t) * a.d};
}
__device__ __forceinline__ double zsin(double a) { double s_, c_; zm_sincos(a, &s_, &c_); return s_; }
__device__ __forceinline__ double zcos(double a) { double s_, c_; zm_sincos(a, &s_, &c_); return c_; }
__device__ __forceinline__ double ztan(double a) { double s_, c_; zm_sincos(a, &s_, &c_); return s_ / c_; }

// ---- hyper-dual number: value, two first-order parts and the mixed second-order part; f(x + e_a eps1 + e_b eps2)
//      carries d2 f / dz_a dz_b in .d12 exactly (no truncation error) -- the role of jax.hessian (pytrees.py:180-186)
struct Hyper {
    double v, d1, d2, d12;
};
__device__ __forceinline__ Hyper operator+(Hyper a, Hyper b) { return {a.v + b.v, a.d1 + b.d1, a.d2 + b.d2, a.d12 + b.d12}; }
__device__ __forceinline__ Hyper operator-(Hyper a, Hyper b) { return {a.v - b.v, a.d1 - b.d1, a.d2 - b.d2, a.d12 - b.d12}; }
__device__ __forceinline__ Hyper operator-(Hyper a) { return {-a.v, -a.d1, -a.d2, -a.d12}; }
__device__ __forceinline__ Hyper operator*(Hyper a, Hyper b) {
    return {a.v * b.v, a.d1 * b.v + a.v * b.d1, a.d2 * b.v + a.v * b.d2,
            (a.d12 * b.v + a.d1 * b.d2) + (a.d2 * b.d1 + a.v * b.d12)};
}
__device__ __forceinline__ Hyper hyper_fn(Hyper a, double f, double fp, double fpp) {  // chain rule for a unary function
    return {f, fp * a.d1, fp * a.d2, fp * a.d12 + fpp * (a.d1 * a.d2)};
}
__device__ __forceinline__ Hyper hyper_inv(Hyper b) {
    const double g = 1.0 / b.v;
    return hyper_fn(b, g, -g * g, 2.0 * g * g * g);
}
__device__ __forceinline__ Hyper operator/(Hyper a, Hyper b) { return a * hyper_inv(b); }
__device__ __forceinline__ Hyper operator+(Hyper a, double b) { return {a.v + b, a.d1, a.d2, a.d12}; }
__device__ __forceinline__ Hyper operator+(double a, Hyper b) { return {a + b.v, b.d1, b.d2, b.d12}; }
__device__ __forceinline__ Hyper operator-(Hyper a, double b) { return {a.v - b, a.d1, a.d2, a.d12}; }
__device__ __forceinline__ Hyper operator-(double a, Hyper b) { return {a - b.v, -b.d1, -b.d2, -b.d12}; }
__device__ __forceinline__ Hyper operator*(Hyper a, double b) { return {a.v * b, a.d1 * b, a.d2 * b, a.d12 * b}; }
__device__ __forceinline__ Hyper operator*(double a, Hyper b) { return {a * b.v, a * b.d1, a * b.d2, a * b.d12}; }
__device__ __forceinline__ Hyper operator/(Hyper a, double b) { return {a.v / b, a.d1 / b, a.d2 / b, a.d12 / b}; }
__device__ __forceinline__ Hyper operator/(double a, Hyper b) { return a * hyper_inv(b); }
__device__ __forceinline__ Hyper zsin(Hyper a) { double s_, c_; zm_sincos(a.v, &s_, &c_); return hyper_fn(a, s_, c_, -s_); }
__device__ __forceinline__ Hyper zcos(Hyper a) { double s_, c_; zm_sincos(a.v, &s_, &c_); return hyper_fn(a, c_, -s_, -c_); }
__device__ __forceinline__ Hyper ztan(Hyper a) {
    double s_, c_;
    zm_sincos(a.v, &s_, &c_);
    const double t = s_ / c_, q = 1.0 + t * t;
    return hyper_fn(a, t, q, 2.0 * t * q);
}

// ---- quadcopter (zopt/quadcopter.py) ---------------------------------------------------------------------------
// state [u,v,w,p,q,r,phi,theta,psi,x,y,z], control [thrust,mx,my,mz]; g = 9.807, mass = 2.5, I = eye(3) (:15-18).
// rigidBodyDynamics (:70-113): 8 states [u,v,w,p,q,r,phi,theta], wind given in the BODY frame (the aero forces see uvw - wind, :64)
// (the trigonometric values are passed in: inertialDynamics needs them too, and on dual / hyper-dual numbers they are the
// expensive part of the model)
template <typename S>
__device__ __forceinline__ void quad_rigid_body_trig(const S (&x)[8], const S (&u)[4], const S (&wb)[3], const S cphi, const S sphi,
                                                     const S cth, const S sth, const S tth, S (&xd)[8]) {
    constexpr double g = 9.807, mass = 2.5;
    const S va0 = x[0] - wb[0], va1 = x[1] - wb[1], va2 = x[2] - wb[2];
    // _getAeroForceMomemnts (:51-67): force = lin * uvw_aero + quad * uvw_aero^2, moment = lin * pqr
    const S fa0 = -0.2 * va0 + -0.05 * (va0 * va0);
    const S fa1 = -0.2 * va1 + -0.05 * (va1 * va1);
    const S fa2 = -0.3 * va2 + -0.1 * (va2 * va2);
    const S ma0 = -0.1 * x[3], ma1 = -0.1 * x[4], ma2 = -0.05 * x[5];
    const S d2x = -sth, d2y = sphi * cth, d2z = cphi * cth;                       // :94
    const S ft0 = (fa0 + (mass * g) * d2x);                                       // force_control = m*[0,0,-thrust]
    const S ft1 = (fa1 + (mass * g) * d2y);
    const S ft2 = ((mass * (-u[0])) + fa2) + (mass * g) * d2z;                    // :98-100
    // -cross(pqr, uvw) + force_total, times 1/m   (:106)
    const S c0 = x[4] * x[2] - x[5] * x[1];
    const S c1 = x[5] * x[0] - x[3] * x[2];
    const S c2 = x[3] * x[1] - x[4] * x[0];
    xd[0] = (1.0 / mass) * (ft0 - c0);
    xd[1] = (1.0 / mass) * (ft1 - c1);
    xd[2] = (1.0 / mass) * (ft2 - c2);
    xd[3] = u[1] + ma0;                                                           // :107 (I = eye: cross(pqr,pqr) = 0)
    xd[4] = u[2] + ma1;
    xd[5] = u[3] + ma2;
    // body rates -> Euler rates, first two rows (:41-48, :108)
    xd[6] = (x[3] + (sphi * tth) * x[4]) + (cphi * tth) * x[5];
    xd[7] = cphi * x[4] - sphi * x[5];
}
template <typename S>
__device__ __forceinline__ void quad_rigid_body(const S (&x)[8], const S (&u)[4], const S (&wb)[3], S (&xd)[8]) {
    quad_rigid_body_trig<S>(x, u, wb, zcos(x[6]), zsin(x[6]), zcos(x[7]), zsin(x[7]), ztan(x[7]), xd);
}

// inertialDynamics (:116-144): wind = constant wind in the NED frame (:117), seen by the rigid body as R_b2i^T wind (:138).  The
// iLQR / MPC demos roll out without wind (demos/iterativeLqr.py:35); their closed-loop simulation uses (3,1,0) (:48).
template <typename S>
__device__ __forceinline__ void quad_inertial_dynamics(const S (&x)[12], const S (&u)[4], const double (&wind)[3], S (&xd)[12]) {
    const S cphi = zcos(x[6]), sphi = zsin(x[6]);
    const S cth = zcos(x[7]), sth = zsin(x[7]), tth = ztan(x[7]);
    const S cpsi = zcos(x[8]), spsi = zsin(x[8]);
    // R_b2i as written in the reference (:31-37; [0][2] = cphi*sth*cpsi - sphi*spsi, quirk Q4)
    const S r00 = cth * cpsi, r01 = sphi * sth * cpsi - cphi * spsi, r02 = cphi * sth * cpsi - sphi * spsi;
    const S r10 = cth * spsi, r11 = sphi * sth * spsi + cphi * cpsi, r12 = cphi * sth * spsi - sphi * cpsi;
    const S r20 = -sth, r21 = sphi * cth, r22 = cphi * cth;
    const S wb[3] = {(r00 * wind[0] + r10 * wind[1]) + r20 * wind[2], (r01 * wind[0] + r11 * wind[1]) + r21 * wind[2],
                     (r02 * wind[0] + r12 * wind[1]) + r22 * wind[2]};
    const S x8[8] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
    S xd8[8];
    quad_rigid_body_trig<S>(x8, u, wb, cphi, sphi, cth, sth, tth, xd8);           // state[:9] -> the 8 states it reads (quirk Q5)
#pragma unroll
    for (int i = 0; i < 8; ++i) xd[i] = xd8[i];
    xd[8] = (sphi / cth) * x[4] + (cphi / cth) * x[5];                            // psiDot (:141)
    // xyzDot = R_b2i @ uvw (:142)
    xd[9] = (r00 * x[0] + r01 * x[1]) + r02 * x[2];
    xd[10] = (r10 * x[0] + r11 * x[1]) + r12 * x[2];
    xd[11] = (r20 * x[0] + r21 * x[1]) + r22 * x[2];
}

// One discrete step x+ = f(x, u) of a registered model.  n, m are the model's dimensions (<= MAXN, MAXM).
// Variables (bit i: state i; bit n + j: control j) in which the model is NOT affine: only pairs of these can have a nonzero
// second derivative.  The quadcopter's controls enter linearly (thrust along body z and the three moments, quadcopter.py:94-105)
// and its inertial position does not enter at all; a linear model has no such variable.
__host__ __device__ __forceinline__ unsigned model_nonlinear_mask(const zm_model_t& md) {
    if (md.kind == ZM_MODEL_QUADCOPTER) return 0x1FFu;      // u v w p q r phi theta psi
    if (md.kind == ZM_MODEL_QUADCOPTER_RB) return 0xFFu;    // u v w p q r phi theta
    return 0u;
}

// The unordered variable pairs (a <= b) that can have a nonzero second derivative, for models that declare them: a tighter
// statement than model_nonlinear_mask (whose V (V + 1) / 2 pairs are the fallback).  Quadcopter, states u v w p q r phi theta psi
// = 0..8 (quadcopter.py:23-144): the aerodynamic force is quadratic in each body velocity alone; omega x v couples six
// velocity / rate pairs; gravity, the Euler-angle kinematics and the rotation of the body velocity into the inertial frame couple
// the angles with each other, with q and r, and with u, v, w (this also covers a constant wind, which enters through the same
// rotation).  28 pairs instead of 45: two trajectory points share a wave.  tests/test_ddp_gpu.py checks against autograd that
// every other second derivative is exactly zero.
constexpr int ZM_MAX_PAIRS = 32;
// The same table by value (host and device): ab[p] = a * 16 + b, n pairs (0: the model declares none).  Kernels that consume
// the packed second derivatives of zm_quadratic_dynamics_pairs_list_f64 take it as an argument.
struct PairTab {
    int n;
    unsigned char ab[ZM_MAX_PAIRS];
};
__host__ __device__ inline PairTab model_pair_table(const int kind) {
    PairTab t{};
    if (kind == ZM_MODEL_QUADCOPTER) {
        const unsigned char tab[28] = {0x00, 0x11, 0x22, 0x24, 0x15, 0x05, 0x23, 0x13, 0x04, 0x66, 0x67, 0x77, 0x68, 0x78,
                                       0x88, 0x06, 0x07, 0x08, 0x16, 0x17, 0x18, 0x26, 0x27, 0x28, 0x46, 0x47, 0x56, 0x57};
        t.n = 28;
        for (int q = 0; q < 28; ++q) t.ab[q] = tab[q];
    }
    return t;
}
__device__ __forceinline__ int model_hessian_pairs(const zm_model_t& md, const int p, int& a, int& b) {
    if (md.kind == ZM_MODEL_QUADCOPTER) {
        // packed as a * 16 + b
        constexpr unsigned char tab[28] = {0x00, 0x11, 0x22,                                       // (u,u) (v,v) (w,w)
                                           0x24, 0x15, 0x05, 0x23, 0x13, 0x04,                     // omega x v
                                           0x66, 0x67, 0x77, 0x68, 0x78, 0x88,                     // angle, angle
                                           0x06, 0x07, 0x08, 0x16, 0x17, 0x18, 0x26, 0x27, 0x28,   // velocity, angle
                                           0x46, 0x47, 0x56, 0x57};                                // q / r, phi / theta
        int e = 0;
#pragma unroll
        for (int q = 0; q < 28; ++q) e = (q == p) ? tab[q] : e;   // select chain: no dynamic indexing of a local table
        a = e >> 4;
        b = e & 15;
        return 28;
    }
    return 0;   // not declared: the caller enumerates the pairs of model_nonlinear_mask
}

template <typename S>
__device__ __forceinline__ void model_step(const zm_model_t& md, const S (&x)[MAXN], const S (&u)[MAXM], S (&xn)[MAXN]) {
    if (md.kind == ZM_MODEL_QUADCOPTER) {
        S xd[12];
        quad_inertial_dynamics<S>(x, u, md.wind_ned, xd);
#pragma unroll
        for (int i = 0; i < 12; ++i) xn[i] = (md.dt == 0.0) ? xd[i] : x[i] + md.dt * xd[i];   // dt = 0: the derivative itself
    } else if (md.kind == ZM_MODEL_QUADCOPTER_RB) {
        const S x8[8] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
        const S wb[3] = {x[0] * 0.0 + md.wind_ned[0], x[0] * 0.0 + md.wind_ned[1], x[0] * 0.0 + md.wind_ned[2]};   // body-frame wind
        S xd8[8];
        quad_rigid_body<S>(x8, u, wb, xd8);
#pragma unroll
        for (int i = 0; i < 8; ++i) xn[i] = (md.dt == 0.0) ? xd8[i] : x[i] + md.dt * xd8[i];
#pragma unroll
        for (int i = 8; i < MAXN; ++i) xn[i] = x[i] * 0.0;
    } else {  // ZM_MODEL_LINEAR: A @ x + B @ u
        const int n = md.n, m = md.m;
#pragma unroll
        for (int i = 0; i < MAXN; ++i) {
            if (i < n) {
                S ax = x[0] * md.A[i * n];
#pragma unroll
                for (int j = 1; j < MAXN; ++j)
                    if (j < n) ax = ax + x[j] * md.A[i * n + j];
                S bu = u[0] * md.B[i * m];
#pragma unroll
                for (int j = 1; j < MAXM; ++j)
                    if (j < m) bu = bu + u[j] * md.B[i * m + j];
                xn[i] = ax + bu;
            } else {
                xn[i] = x[i] * 0.0;
            }
        }
    }
}

// c(x,u) = (x^T Q) x + (u^T R) u      (demos/iterativeLqr.py:12-13)
__device__ __forceinline__ double running_cost(const zm_quadcost_t& cs, const int n, const int m, const double (&x)[MAXN],
                                               const double (&u)[MAXM]) {
    double cx = 0.0, cu = 0.0;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
        if (j < n) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < MAXN; ++i)
                if (i < n) s = __builtin_fma(x[i], cs.Q[i * n + j], s);
            cx = __builtin_fma(s, x[j], cx);
        }
    }
#pragma unroll
    for (int j = 0; j < MAXM; ++j) {
        if (j < m) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < MAXM; ++i)
                if (i < m) s = __builtin_fma(u[i], cs.R[i * m + j], s);
            cu = __builtin_fma(s, u[j], cu);
        }
    }
    return cx + cu;
}

// c_f(x) = (x^T Qf) x
__device__ __forceinline__ double terminal_cost(const zm_quadcost_t& cs, const int n, const double (&x)[MAXN]) {
    double cx = 0.0;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
        if (j < n) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < MAXN; ++i)
                if (i < n) s = __builtin_fma(x[i], cs.Qf[i * n + j], s);
            cx = __builtin_fma(s, x[j], cx);
        }
    }
    return cx;
}

}  // namespace zm
