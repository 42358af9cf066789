// PD projection  a -> V max(w, eps) V^T  (reference ilqrUtils.py:217-225: jnp.linalg.eigh, clip, reassemble) of a symmetric
// 16 x 16 tile held in MFMA registers, WITHOUT an eigen-decomposition: with X = a - eps I,
//
//     V max(w, eps) V^T = eps I + (X + |X|) / 2,      |X| = sign(X) X,
//
// and sign(X) comes from matrix iterations that are nothing but 16x16 products -- the work fp64 MFMA is built for:
//   cubic   Newton-Schulz step  Z <- Z (3 I - Z^2) / 2                           (quadratically convergent near |x| = 1)
//   quintic booster        step  Z <- Z (a I + b Z^2 + c Z^4), a = 3.4445        (x -> 3.44 x for small x, [0.7, 1.2] invariant)
// A (quintic, quintic, cubic) group runs while F = |I - Z^2|_F^2 > 0.9, i.e. while some eigenvalue may still be below 0.23 (after a
// group every eigenvalue that has reached the band contributes < 0.052 to F, 16 of them < 0.9); cubic steps, two per convergence test,
// finish.  One reduction (F) and one LDS transpose (the symmetrisation) per group / per two cubic steps: they cost a wave about as much
// as two of the products, which is why the steps come in groups.  An eigenvalue of X below ~1e-12 |X|_F is not resolved
// within the iteration caps; it then contributes an error of at most its own size.
// Measured against eigh on 900 adversarial spectra (tools/ns_psd_model.py): <= 4e-12 relative, 3e-15 on dense random matrices.
//
// The tile algebra gives X^T Y for free (tile16_f64.h); Z is symmetric, so X^T Y = X Y -- but rounding makes Z' = Z^T W slightly
// nonsymmetric and Z^T (instead of Z) then amplifies the antisymmetric part 3x per step: every update is symmetrised through an
// LDS transpose (Z^T Z and (Z^T Z)^T (Z^T Z) are bitwise symmetric by construction and need none).
//
// Rows / columns outside the live index set are zero and stay zero (live[r] marks the diagonal entries of the live set).
#pragma once
#include "tile16_f64.h"
#include "zm_common.h"

namespace zm {

constexpr int NS_LD = 17;                 // padded leading dimension of the transpose buffers
constexpr int NS_LDS_DOUBLES = 2 * 16 * NS_LD;

__device__ __forceinline__ void ns_sync() { wave_lds_sync(); }

// (dpp_mov64 of zm_common.h under its older name)
template <int CTRL>
__device__ __forceinline__ double ns_dpp(const double v) {
    return dpp_mov64<CTRL>(v);
}

__device__ __forceinline__ double ns_readlane(const double v, const int l) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// sum over the 64 lanes, the same (wave-uniform) value in every lane: butterflies inside each row of 16 by DPP
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row sums through readlane
__device__ __forceinline__ double ns_wave_sum(double v) {
    v += ns_dpp<0xB1>(v);
    v += ns_dpp<0x4E>(v);
    v += ns_dpp<0x141>(v);
    v += ns_dpp<0x140>(v);
    return (ns_readlane(v, 0) + ns_readlane(v, 16)) + (ns_readlane(v, 32) + ns_readlane(v, 48));
}

// X^T Y over the first KSZ row groups (the others hold zeros)
template <int KSZ>
__device__ __forceinline__ d4 ns_op(const d4& X, const d4& Y) {
    d4 acc = zero4();
#pragma unroll
    for (int s = 0; s < KSZ; ++s) acc = mfma(X[s], Y[s], acc);
    return acc;
}

// v <- (v + v^T) / 2 through one of two alternating LDS buffers (one barrier per call)
__device__ __forceinline__ d4 ns_symmetrise(const d4& v, double* T, int& flip, const int g, const int c) {
    double* buf = T + (flip ? 16 * NS_LD : 0);
    flip ^= 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(4 * r + g) * NS_LD + c] = v[r];
    ns_sync();
    d4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = 0.5 * (v[r] + buf[c * NS_LD + 4 * r + g]);
    return o;
}

// thresholds on F = ||I - Z^2||_F^2 above which one / two further quintics ride along in a booster pair (see below)
#ifndef NS_RIDER1
#define NS_RIDER1 0.9
#endif
#ifndef NS_RIDER2
#define NS_RIDER2 1e300
#endif

// sign iteration on Z (scaled X) over K row groups; returns |X| = sign(X) X
template <int K>
__device__ __forceinline__ d4 ns_sign_times(d4 z, d4& x, const d4& idr, double* T, int& flip, const int g, const int c,
                                            double* xstash = nullptr) {
    if (xstash) {   // X leaves the registers for the duration of the iteration
#pragma unroll
        for (int r = 0; r < 4; ++r) xstash[r * 64 + g * 16 + c] = x[r];
        ns_sync();
    }
    constexpr double QA = 3.4445, QB = -4.7750, QC = 2.0315;
    constexpr int MAX_PAIRS = 18, MAX_CUBIC = 14;
    int pairs = 0, cubic = 0;
    for (;;) {
        d4 z2 = ns_op<K>(z, z);
        double f = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double e = idr[r] - z2[r];
            f = __builtin_fma(e, e, f);
        }
        f = ns_wave_sum(f);
        d4 w;
        if (f > 0.9 && pairs < MAX_PAIRS) {
            // while eigenvalues are still far below the band further quintics ride along before the pair's own (each: gain 3.44 for
            // 3 products and NO reduction / transpose; a pair alone is gain 5.17 for 5 products and one of each); their results stay
            // unsymmetrised like the pair's quintic (tools/ns_psd_model.py: accuracy and counts on the DDP sweeps' spectra)
            const int riders = (f > NS_RIDER1 ? 1 : 0) + (f > NS_RIDER2 ? 1 : 0);
            for (int e = 0; e < riders; ++e) {
                const d4 y4 = ns_op<K>(z2, z2);
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = __builtin_fma(QC, y4[r], __builtin_fma(QB, z2[r], QA * idr[r]));
                z = ns_op<K>(w, z);
                z2 = ns_op<K>(z, z);
            }
            const d4 z4 = ns_op<K>(z2, z2);
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = __builtin_fma(QC, z4[r], __builtin_fma(QB, z2[r], QA * idr[r]));
            // W is bitwise symmetric (built from Z^T Z and its square): W^T Z = W Z needs no transpose of Z, and the booster's
            // result may stay unsymmetrised until the cubic step that follows (tools/ns_psd_model.py: same accuracy)
            z = ns_op<K>(w, z);
            z2 = ns_op<K>(z, z);
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = 1.5 * idr[r] - 0.5 * z2[r];
            z = ns_symmetrise(ns_op<K>(w, z), T, flip, g, c);
            ++pairs;
        } else {
            // Cubic steps go in twos: the norm reduction and the symmetrising LDS transpose are paid once per two steps (an
            // unsymmetrised intermediate amplifies its rounding asymmetry 3x -- harmless once; tools/ns_psd_model.py: same accuracy).
            // F' = 0.5625 F^2 after a step, so the second step of a double is the final one when 0.5625 F^2 is far below 1e-16.
            const bool last = (f < 1e-16) || (cubic + 1 >= MAX_CUBIC);
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = 1.5 * idr[r] - 0.5 * z2[r];
            if (last) {
                z = ns_symmetrise(ns_op<K>(z, w), T, flip, g, c);
                ++cubic;
                break;
            }
            z = ns_op<K>(z, w);
            z2 = ns_op<K>(z, z);
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = 1.5 * idr[r] - 0.5 * z2[r];
            z = ns_symmetrise(ns_op<K>(z, w), T, flip, g, c);
            cubic += 2;
            if (0.5625 * f * f < 1e-18 || cubic >= MAX_CUBIC) break;
        }
    }
    if (xstash) {
        ns_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = xstash[r * 64 + g * 16 + c];
    }
    return ns_op<K>(z, x);
}

// a (D layout: a[r] = A[4r+g][c]) is replaced by its projection.  T: NS_LDS_DOUBLES of LDS private to the wave.
// KSZ: row groups that can hold live indices; the products run over the row groups that actually do (wave-uniform).
// xstash: optional 256 doubles of wave-private LDS: X = a - eps I waits there during the iteration instead of in 8 registers (the
// sweep kernel that wants a third wave per SIMD passes it)
// SYMIN: the caller guarantees a bitwise symmetric tile (the DDP ring sweep contracts element (i, j) and (j, i) from the same packed row
// in the same order): the input symmetrisation -- an identity then -- and its LDS round trip are skipped.
template <int KSZ, bool SYMIN = false>
__device__ __forceinline__ void psd_project_ns(d4& a, const bool (&live)[4], const double eps, double* T, const int g,
                                               const int c, double* xstash = nullptr) {
    int flip = 0;
    d4 x;
    if constexpr (!SYMIN) a = ns_symmetrise(a, T, flip, g, c);        // jnp.linalg.eigh symmetrises its input
    // An index whose row and column are exactly zero is an eigenvector with eigenvalue 0, decoupled from the rest: its projection
    // is eps on the diagonal, and it stays out of the iteration (idr).  Otherwise X would carry the eigenvalue -eps once per such
    // index -- 1e-4..1e-5 of |X| for the Hessians of a model that is affine in some of its variables -- and the sign iteration
    // needs ~log(|X| / eps) booster steps to resolve it.
    const bool nzl = (a[0] != 0.0) | (a[1] != 0.0) | (a[2] != 0.0) | (a[3] != 0.0);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(nzl);
    const unsigned colmask = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull);
    const bool colnz = (colmask >> c) & 1u;
    d4 idr;
#pragma unroll
    for (int r = 0; r < 4; ++r) idr[r] = (live[r] && colnz) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = a[r] - eps * idr[r];
    double ss = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) ss = __builtin_fma(x[r], x[r], ss);
    ss = ns_wave_sum(ss);
    d4 ax = zero4();
    if (ss > 0.0) {   // X = 0: |X| = 0, nothing to iterate
        const double inv = 1.0 / sqrt(ss);
        d4 z;
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = x[r] * inv;
        // nonzero columns all below row group `need`: the higher groups of every iterate stay zero and are skipped
        const int need = (colmask >> 12) ? 4 : (colmask >> 8) ? 3 : (colmask >> 4) ? 2 : 1;
        if (KSZ >= 4 && need == 4) ax = ns_sign_times<(KSZ >= 4 ? 4 : KSZ)>(z, x, idr, T, flip, g, c, xstash);
        else if (KSZ >= 3 && need == 3) ax = ns_sign_times<(KSZ >= 3 ? 3 : KSZ)>(z, x, idr, T, flip, g, c, xstash);
        else if (KSZ >= 2 && need == 2) ax = ns_sign_times<(KSZ >= 2 ? 2 : KSZ)>(z, x, idr, T, flip, g, c, xstash);
        else ax = ns_sign_times<1>(z, x, idr, T, flip, g, c, xstash);
    }
    d4 p;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = __builtin_fma(0.5, x[r] + ax[r], eps * (live[r] ? 1.0 : 0.0));
    a = ns_symmetrise(p, T, flip, g, c);
}

}  // namespace zm
