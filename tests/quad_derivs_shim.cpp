// Host build of the generated closed-form quadcopter derivatives (zopt_amd/csrc/quad_derivs_gen.h) for tests/test_quad_derivs.py:
// the same header the kernels include, compiled by g++ so that the formulas can be checked against the oracle without a GPU.
#include <cmath>

#include "../zopt_amd/csrc/quad_derivs_gen.h"

static zm::QuadAtoms atoms(const double* x, const double* u, const double* w) {
    zm::QuadAtoms a;
    for (int i = 0; i < 12; ++i) a.x[i] = x[i];
    a.u0 = u[0];
    for (int i = 0; i < 3; ++i) a.w[i] = w[i];
    a.s6 = std::sin(x[6]); a.c6 = std::cos(x[6]);
    a.s7 = std::sin(x[7]); a.c7 = std::cos(x[7]);
    a.s8 = std::sin(x[8]); a.c8 = std::cos(x[8]);
    a.ic7 = 1.0 / a.c7;
    return a;
}

// still_air != 0: the WIND = false instantiation (the caller passes w = 0)
extern "C" void quad_jacobian(const double* x, const double* u, const double* w, int still_air, double* J /* (12, 16) */) {
    const zm::QuadAtoms a = atoms(x, u, w);
    for (int j = 0; j < 16; ++j) {
        double o[12];
        if (still_air) zm::quad_jac_column<false>(j, a, o);
        else zm::quad_jac_column<true>(j, a, o);
        for (int i = 0; i < 12; ++i) J[i * 16 + j] = o[i];
    }
}

extern "C" void quad_hessian_pairs(const double* x, const double* u, const double* w, int still_air, int npairs,
                                   double* H /* (npairs, 12) */) {
    const zm::QuadAtoms a = atoms(x, u, w);
    for (int p = 0; p < npairs; ++p) {
        double o[12];
        if (still_air) zm::quad_hess_pair<false>(p, a, o);
        else zm::quad_hess_pair<true>(p, a, o);
        for (int i = 0; i < 12; ++i) H[p * 12 + i] = o[i];
    }
}

// the packed image of [f_x | f_u] = I + dt d xd / d z (quad_jac_column_packed) and the position table that goes with it
extern "C" int quad_jacobian_packed(const double* x, const double* u, const double* w, int still_air, double dt, double* t /* >= 60 */,
                                    unsigned char* pos /* (12, 16) */) {
    const zm::QuadAtoms a = atoms(x, u, w);
    for (int j = 0; j < 16; ++j) {
        if (still_air) zm::quad_jac_column_packed<false>(j, a, dt, t);
        else zm::quad_jac_column_packed<true>(j, a, dt, t);
    }
    const unsigned char* p = still_air ? zm::QUAD_JPOS_STILL : zm::QUAD_JPOS_WIND;
    for (int e = 0; e < 192; ++e) pos[e] = p[e];
    return still_air ? zm::QUAD_NJ_STILL : zm::QUAD_NJ_WIND;
}

// the sparse image of the second derivatives (quad_hess_pair2_packed) and the dense index table that goes with it
extern "C" int quad_hessian_sparse(const double* x, const double* u, const double* w, int still_air, double dt, double* t /* >= 86 */,
                                   unsigned short* dense /* >= 86 */) {
    const zm::QuadAtoms a = atoms(x, u, w);
    for (int j = 0; j < 14; ++j) {
        if (still_air) zm::quad_hess_pair2_packed<false>(j, a, dt, t);
        else zm::quad_hess_pair2_packed<true>(j, a, dt, t);
    }
    const int nh = still_air ? zm::QUAD_NH_STILL : zm::QUAD_NH_WIND;
    const unsigned short* p = still_air ? zm::QUAD_HDENSE_STILL : zm::QUAD_HDENSE_WIND;
    for (int e = 0; e < nh; ++e) dense[e] = p[e];
    return nh;
}

// the straight-line forms (one lane per trajectory point on the device): the same images by quad_jac_all_packed / quad_hess_all_packed
extern "C" void quad_all_packed(const double* x, const double* u, const double* w, int still_air, double dt, double* tj /* >= 60 */,
                                double* th /* >= 86 */) {
    const zm::QuadAtoms a = atoms(x, u, w);
    if (still_air) {
        zm::quad_jac_all_packed<false>(a, dt, tj);
        zm::quad_hess_all_packed<false>(a, dt, th);
    } else {
        zm::quad_jac_all_packed<true>(a, dt, tj);
        zm::quad_hess_all_packed<true>(a, dt, th);
    }
}
