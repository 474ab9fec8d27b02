// Device helpers for the hemispheric two-stream coefficients (reference
// source/kernels.cu:109-290).  Shared by the per-stage kernels and the fused path.
#pragma once
#include "hx_common.h"

namespace hx {

// CUDA's min/max on doubles ignore a NaN operand (fmin/fmax); the reference relies on
// min(0.0, NaN) == 0.0 to discard NaN beam terms (kernels.cu:1449, :1654), so do we.
__device__ __forceinline__ double dmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double dmax(double a, double b) { return fmax(a, b); }

// improved-two-stream correction factor E (Heng, Malik & Kitzmann 2018 fit), kernels.cu:109-124
__device__ __forceinline__ double E_factor(double w0, double g0, int scat_corr, double i2s) {
    if (scat_corr == 1 && w0 > i2s && g0 >= 0.0)
        return dmax(1.0, 1.225 - 0.1582 * g0 - 0.1777 * w0 - 0.07465 * (g0 * g0) +
                             0.2351 * w0 * g0 - 0.05582 * (w0 * w0));
    return 1.0;
}

__device__ __forceinline__ double clip_G(double G) {  // kernels.cu:218-231
    return fabs(G) < 1e8 ? G : 1e8 * G / fabs(G);
}

struct Slab {
    double w0, E, trans, M, N, P, Gp, Gm;
};

// everything calc_trans_* derives from (w0, total optical depth, g0) for one (half-)layer
__device__ __forceinline__ Slab slab_coeffs(double w0, double dtau, double g0, double epsi,
                                            double epsi2, double mu_star, int scat_corr,
                                            double i2s, bool need_G = true) {
    Slab s;
    const double E = E_factor(w0, g0, scat_corr, i2s);
    const double omg = 1.0 - w0 * g0;
    s.w0 = w0;
    s.E = E;
    s.trans = exp(-1.0 / epsi * sqrt(E * omg * (E - w0)) * dtau);  // :144
    const double r = sqrt((E - w0) / (E * omg));
    const double zm = 0.5 * (1.0 - r), zp = 0.5 * (1.0 + r);       // :272, :289
    const double t2 = s.trans * s.trans;
    s.M = (zm * zm) * t2 - (zp * zp);
    s.N = zp * zm * (1.0 - t2);
    s.P = ((zm * zm) - (zp * zp)) * s.trans;
    s.Gp = 0.0;
    s.Gm = 0.0;
    if (!need_G) return s;  // only the direct-beam terms use G+-
    // G+-, :149-213
    const double num = w0 * (E * omg + g0 * epsi / epsi2);
    const double den = E * (1.0 / (epsi * epsi)) * (E - w0) * omg - 1.0 / (mu_star * mu_star);
    const double third = epsi * w0 * g0 * mu_star / (epsi2 * E * omg);
    const double inv = 1.0 / (mu_star * E * omg);
    s.Gp = clip_G(0.5 * (num / den * (1.0 / epsi + inv) + third));
    s.Gm = clip_G(0.5 * (num / den * (1.0 / epsi - inv) - third));
    return s;
}

// slab_coeffs for E = 1 and g0 = 0 (no I2S correction, isotropic scattering, no clouds -- BASELINE config 2), without G+-:
// the same operations in the same order minus those that are exact no-ops there -- E * omg = 1.0, so both divisions by it
// return their numerator and the two square roots have the same argument (1 - w0).  Bit-identical to slab_coeffs(w0,
// dtau, 0.0, ..., scat_corr = 0, need_G = false); saves a square root and a division of the eight per half-layer.
__device__ __forceinline__ Slab slab_coeffs_plain(double w0, double dtau, double epsi) {
    Slab s;
    s.w0 = w0;
    s.E = 1.0;
    const double r = sqrt(1.0 - w0);                 // = sqrt(E * omg * (E - w0)) = sqrt((E - w0) / (E * omg))
    s.trans = exp(-1.0 / epsi * r * dtau);           // :144
    const double zm = 0.5 * (1.0 - r), zp = 0.5 * (1.0 + r);
    const double t2 = s.trans * s.trans;
    s.M = (zm * zm) * t2 - (zp * zp);
    s.N = zp * zm * (1.0 - t2);
    s.P = ((zm * zm) - (zp * zp)) * s.trans;
    s.Gp = 0.0;
    s.Gm = 0.0;
    return s;
}

__device__ __forceinline__ double single_scat_albedo(double scat, double absorb, double limit) {
    return dmin(scat / (scat + absorb), limit);  // :249-256
}

__device__ __forceinline__ double tiny_abs(double F) { return fabs(F) < 1e-100 ? fabs(F) : F; }

// fractional table index on a uniform (T, log10 P) grid; margin 0.001 (premixed, :545-559) or 0
// (per-species tables, :3228-3241)
struct TPIndex {
    double t, p;
    int tdown, tup, pdown, pup;
};

__device__ __forceinline__ TPIndex locate_tp(double temp, double press, const double* tgrid,
                                             int ntemp, const double* pgrid, int npress,
                                             bool margin, bool log_t) {
    TPIndex k;
    double t;
    if (log_t) {
        const double dt = (log10(tgrid[ntemp - 1]) - log10(tgrid[0])) / (ntemp - 1.0);
        t = (log10(temp) - log10(tgrid[0])) / dt;
    } else {
        const double dt = (tgrid[ntemp - 1] - tgrid[0]) / (ntemp - 1.0);
        t = (temp - tgrid[0]) / dt;
    }
    const double dp = (log10(pgrid[npress - 1]) - log10(pgrid[0])) / (npress - 1.0);
    double p = (log10(press) - log10(pgrid[0])) / dp;
    if (margin) {
        t = dmin(ntemp - 1.001, dmax(0.001, t));
        p = dmin(npress - 1.001, dmax(0.001, p));
    } else {
        t = dmin(ntemp - 1.0, dmax(0.0, t));
        p = dmin(npress - 1.0, dmax(0.0, p));
    }
    k.t = t;
    k.p = p;
    k.tdown = (int)floor(t);
    k.tup = (int)ceil(t);
    k.pdown = (int)floor(p);
    k.pup = (int)ceil(p);
    return k;
}

// four-case bilinear blend (:561-608); `species` selects the term order of :637-640
__device__ __forceinline__ double blend_tp(double dd, double ud, double du, double uu,
                                           const TPIndex& k, bool species) {
    if (k.pdown != k.pup && k.tdown != k.tup)
        return dd * (k.pup - k.p) * (k.tup - k.t) + ud * (k.p - k.pdown) * (k.tup - k.t) +
               du * (k.pup - k.p) * (k.t - k.tdown) + uu * (k.p - k.pdown) * (k.t - k.tdown);
    if (k.tdown == k.tup && k.pdown != k.pup) return dd * (k.pup - k.p) + ud * (k.p - k.pdown);
    if (k.pdown == k.pup && k.tdown != k.tup)
        return species ? du * (k.t - k.tdown) + dd * (k.tup - k.t)
                       : dd * (k.tup - k.t) + du * (k.t - k.tdown);
    return dd;
}

// Planck-table look-up (:956-974): rows at T = 1 + step*r, clamp to [0.001, dim-1.001]
__device__ __forceinline__ double planck_lookup(const double* planck_grid, double T, int x,
                                                int nbin, int dim, int step) {
    double t = (T - 1.0) / step;
    t = dmax(0.001, dmin(dim - 1.001, t));
    const int tdown = (int)floor(t), tup = (int)ceil(t);
    if (tdown != tup)
        return planck_grid[x + (size_t)tdown * nbin] * (tup - t) +
               planck_grid[x + (size_t)tup * nbin] * (t - tdown);
    return planck_grid[x + (size_t)tdown * nbin];
}


// Rayleigh scattering cross-section of water vapour per molecule (calc_index_h2o + calc_h2o_scat,
// kernels.cu:3174-3205, :3404-3440): refractive index of water at the vapour's density, Lorentz-Lorenz
// factor and King correction; zero beyond 2.5 micron.
__device__ __forceinline__ double h2o_rayleigh_cross(double T, double P, double f, double lam_cm, double mass_h2o) {
    const double dens = f * P * mass_h2o / (HX_KBOLTZMANN * T);
    const double lamda = lam_cm / 0.589e-4;
    const double delta = dmin(1.0, dens) / 1.0;
    const double theta = T / 273.15;
    const double lamda_UV = 0.229202, lamda_IR = 5.432937;
    const double a0 = 0.244257733, a1 = 0.974634476e-2, a2 = -0.373234996e-2, a3 = 0.268678472e-3,
                 a4 = 0.158920570e-2, a5 = 0.245934259e-2, a6 = 0.900704920, a7 = -0.166626219e-1;
    const double l2 = lamda * lamda;
    const double A = delta * (a0 + a1 * delta + a2 * theta + a3 * l2 * theta + a4 / l2 +
                              a5 / (l2 - lamda_UV * lamda_UV) + a6 / (l2 - lamda_IR * lamda_IR) +
                              a7 * (delta * delta));
    const double index = sqrt((2.0 * A + 1.0) / (1.0 - A));
    const double n_ref = f * P / (HX_KBOLTZMANN * T);
    const double King = (6.0 + 3.0 * 3e-4) / (6.0 - 7.0 * 3e-4);
    double sc = 0.0;
    if (lam_cm < 2.5e-4) {
        const double n2 = index * index;
        const double lor = (n2 - 1.0) / (n2 + 2.0);
        const double lam2 = lam_cm * lam_cm;
        sc = 24.0 * (HX_PI * HX_PI * HX_PI) / ((n_ref * n_ref) * (lam2 * lam2)) * (lor * lor) * King;
    }
    return sc;
}

}  // namespace hx
