// Radial functions as functions of s = rho^2 (device side).
// phi(s) and psi(s) = phi'(rho)/rho for the five kernels of Morbit.RbfKernels
// (/root/reference/src/models/RbfModel.jl:48-54); formulas restate the published
// RadialBasisFunctionModels.jl radial functions (see oracle/rbf_oracle.py header).
#pragma once
#include "common.hpp"

namespace mrbf {

__device__ __forceinline__ double ipow_small(double x, int e) {
    double r = 1.0;
    for (int i = 0; i < e; ++i) r *= x;
    return r;
}

template <int KID>
__device__ __forceinline__ double rbf_phi(double s, const KP &p) {
    if constexpr (KID == MRBF_GAUSSIAN) {
        return exp(-p.a2 * s);
    } else if constexpr (KID == MRBF_MULTIQUADRIC) {
        double t = fma(p.a2, s, 1.0);
        return p.fast ? -sqrt(t) : p.sgn * pow(t, p.b);
    } else if constexpr (KID == MRBF_INV_MULTIQUADRIC) {
        double t = fma(p.a2, s, 1.0);
        return p.fast ? 1.0 / sqrt(t) : pow(t, -p.b);
    } else if constexpr (KID == MRBF_CUBIC) {
        double r = sqrt(s);
        return p.fast ? p.sgn * s * r : p.sgn * pow(r, p.a);
    } else {  // thin plate spline: sgn * rho^(2k) log(rho), 0 at 0
        if (s <= 0.0) return 0.0;
        return p.sgn * ipow_small(s, p.ik) * (0.5 * log(s));
    }
}

template <int KID>
__device__ __forceinline__ void rbf_phi_psi(double s, const KP &p, double &phi, double &psi) {
    if constexpr (KID == MRBF_GAUSSIAN) {
        phi = exp(-p.a2 * s);
        psi = -2.0 * p.a2 * phi;
    } else if constexpr (KID == MRBF_MULTIQUADRIC) {
        double t = fma(p.a2, s, 1.0);
        if (p.fast) {
            double r = sqrt(t);
            phi = -r;
            psi = -p.a2 / r;
        } else {
            double pw = pow(t, p.b);
            phi = p.sgn * pw;
            psi = p.sgn * 2.0 * p.a2 * p.b * pw / t;
        }
    } else if constexpr (KID == MRBF_INV_MULTIQUADRIC) {
        double t = fma(p.a2, s, 1.0);
        if (p.fast) {
            double r = 1.0 / sqrt(t);
            phi = r;
            psi = -p.a2 * r / t;
        } else {
            double pw = pow(t, -p.b);
            phi = pw;
            psi = -2.0 * p.a2 * p.b * pw / t;
        }
    } else if constexpr (KID == MRBF_CUBIC) {
        double r = sqrt(s);
        if (p.fast) {
            phi = p.sgn * s * r;
            psi = p.sgn * 3.0 * r;
        } else {
            phi = p.sgn * pow(r, p.a);
            psi = (s <= 0.0 && p.a < 2.0) ? 0.0 : p.sgn * p.a * pow(r, p.a - 2.0);
        }
    } else {
        if (s <= 0.0) {
            phi = 0.0;
            psi = 0.0;
        } else {
            double lg = 0.5 * log(s);
            double sk1 = ipow_small(s, p.ik - 1);
            phi = p.sgn * s * sk1 * lg;
            psi = p.sgn * sk1 * (2.0 * p.ik * lg + 1.0);
        }
    }
}

// sqrt(t) and 1/sqrt(t) for t > 0 from the hardware reciprocal-square-root estimate + two coupled Newton steps
// (Goldschmidt): ~8 fp64 ops instead of the ~30 of the correctly rounded library sqrt + divide.  Relative error
// <= ~2e-16 (not correctly rounded); the epilogues that use it are VALU-bound otherwise.
__device__ __forceinline__ void fast_sqrt_rsqrt(double t, double &sq, double &rs) {
    double y = __builtin_amdgcn_rsq(t);          // ~2^-26 accurate
    double g = t * y;                            // ~sqrt(t)
    double h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    // one correction of the square root itself
    const double e = fma(-g, g, t);
    sq = fma(e, h, g);
    rs = 2.0 * h;
}

template <int KID, bool FAST>
__device__ __forceinline__ double rbf_phi_t(double s, const KP &p) {
    if constexpr (KID == MRBF_MULTIQUADRIC && FAST) {
        double sq, rs;
        fast_sqrt_rsqrt(fma(p.a2, s, 1.0), sq, rs);
        return -sq;
    } else if constexpr (KID == MRBF_INV_MULTIQUADRIC && FAST) {
        double sq, rs;
        fast_sqrt_rsqrt(fma(p.a2, s, 1.0), sq, rs);
        return rs;
    } else if constexpr (KID == MRBF_CUBIC && FAST) {
        double sq, rs;  // branch-free: the root of a stand-in for s <= 0, the result selected afterwards
        fast_sqrt_rsqrt(s > 0.0 ? s : 1.0, sq, rs);
        return s > 0.0 ? p.sgn * s * sq : 0.0;
    } else {
        return rbf_phi<KID>(s, p);
    }
}

// compile-time variant of the fast/general switch (keeps pow() out of the register budget of the hot kernels)
template <int KID, bool FAST>
__device__ __forceinline__ void rbf_phi_psi_t(double s, const KP &p, double &phi, double &psi) {
    if constexpr (KID == MRBF_MULTIQUADRIC && FAST) {
        double sq, rs;
        fast_sqrt_rsqrt(fma(p.a2, s, 1.0), sq, rs);
        phi = -sq;
        psi = -p.a2 * rs;
    } else if constexpr (KID == MRBF_INV_MULTIQUADRIC && FAST) {
        double sq, rs;
        fast_sqrt_rsqrt(fma(p.a2, s, 1.0), sq, rs);
        phi = rs;
        psi = -p.a2 * rs * rs * rs;
    } else if constexpr (KID == MRBF_CUBIC && FAST) {
        double sq, rs;  // branch-free (see rbf_phi_t)
        fast_sqrt_rsqrt(s > 0.0 ? s : 1.0, sq, rs);
        phi = s > 0.0 ? p.sgn * s * sq : 0.0;
        psi = s > 0.0 ? p.sgn * 3.0 * sq : 0.0;
    } else {
        rbf_phi_psi<KID>(s, p, phi, psi);
    }
}

// host-side dispatch over the kernel id
#define MRBF_DISPATCH_KID(kid, ...)                                        \
    switch (kid) {                                                         \
        case MRBF_CUBIC: { constexpr int KID = MRBF_CUBIC; __VA_ARGS__; } break;                       \
        case MRBF_INV_MULTIQUADRIC: { constexpr int KID = MRBF_INV_MULTIQUADRIC; __VA_ARGS__; } break; \
        case MRBF_MULTIQUADRIC: { constexpr int KID = MRBF_MULTIQUADRIC; __VA_ARGS__; } break;         \
        case MRBF_THIN_PLATE_SPLINE: { constexpr int KID = MRBF_THIN_PLATE_SPLINE; __VA_ARGS__; } break; \
        default: { constexpr int KID = MRBF_GAUSSIAN; __VA_ARGS__; } break;                            \
    }

}  // namespace mrbf
