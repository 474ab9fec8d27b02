/* TEST INFRASTRUCTURE ONLY -- see helios_oracle.h.
 *
 * CPU restatement (plain C, OpenMP over wavelength bins) of the reference's hot-path arithmetic.
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Expressions keep the reference's operand order so that the results agree with the reference's
 * own kernels to the last few ulp (host build) / to what FMA contraction does to cancelling terms
 * (the gfx950 build that generated tests/golden/: tolerances in tests/golden_checks.py); compile with
 * -ffp-contract=off.
 */
#include "helios_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* device-side constants, source/kernels.cu:36-41 (values are physical constants) */
static const double PI = 3.141592653589793;
static const double HCONST = 6.62607004e-27;
static const double CSPEED = 29979245800.0;
static const double KBOLTZMANN = 1.38064852e-16;
static const double STEFANBOLTZMANN = 5.6703669999999995e-5;
static const double AMU = 1.6605390666e-24;

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* CUDA's min/max on doubles are fmin/fmax: a NaN operand is ignored.  This matters: with the default
 * mu_star = -0.5 and epsi = 0.5 the G+- denominator (kernels.cu:168) is exactly 0 for w0 = 0, G+- become
 * NaN, and it is `min(0.0, NaN) = 0.0` (kernels.cu:1449, :1654, ...) that keeps the fluxes finite. */
static inline double dmin(double a, double b) { return fmin(a, b); }
static inline double dmax(double a, double b) { return fmax(a, b); }

/* x^i by repeated multiplication, source/kernels.cu:235-245 */
static double ipow(double x, int i) {
    double r = 1.0;
    for (int j = 1; j <= i; j++) r *= x;
    return r;
}

/* ============================================================================================
 * Planck table                                            source/kernels.cu:95-105, 362-416
 * ============================================================================================ */
static double planck_series_term(int n, double y1, double y2) {
    double dn = n;
    return exp(-dn * y2) * ((y2 * y2 * y2) / dn + 3.0 * (y2 * y2) / (dn * dn) +
                            6.0 * y2 / (dn * dn * dn) + 6.0 / (dn * dn * dn * dn)) -
           exp(-dn * y1) * ((y1 * y1 * y1) / dn + 3.0 * (y1 * y1) / (dn * dn) +
                            6.0 * y1 / (dn * dn * dn) + 6.0 / (dn * dn * dn * dn));
}

void orc_planck_table(double* planck_grid, const double* lambda_edge, const double* deltalambda,
                      int nbin, double T_star, int dim, int step) {
    /* The reference fills rows r = 0 .. 10*(dim/10)-1 with T = r*step + 1 in ten launches and the
     * row 10*(dim/10) (= dim when dim % 10 == 0) with T_star (kernels.cu:386-393). */
    const int d10 = dim / 10;
    const int nrow = 10 * d10 + 1;
#pragma omp parallel for schedule(dynamic, 16)
    for (int r = 0; r < nrow; r++) {
        double T = (r < 10 * d10) ? (double)(r * step + 1) : T_star;
        for (int x = 0; x < nbin; x++) {
            double acc = 0.0;
            if (T > 0.01) {
                double D = 2.0 * (ipow(KBOLTZMANN / HCONST, 3) * KBOLTZMANN * ipow(T, 4)) /
                           (CSPEED * CSPEED);
                double y_top = HCONST * CSPEED / (lambda_edge[x + 1] * KBOLTZMANN * T);
                double y_bot = HCONST * CSPEED / (lambda_edge[x] * KBOLTZMANN * T);
                if (y_bot < y_top) {
                    double s = y_top;
                    y_top = y_bot;
                    y_bot = s;
                }
                for (int n = 1; n < 200; n++) acc += D * planck_series_term(n, y_bot, y_top);
            }
            planck_grid[x + r * nbin] = acc / deltalambda[x];
        }
    }
}

/* source/kernels.cu:420-468 */
void orc_corr_inc_energy(double* planck_grid, double* starflux, const double* deltalambda,
                         int realstar, int nbin, double T_star, int dim) {
    double num_flux = 0.0;
    if (realstar == 1) {
        for (int x = 0; x < nbin; x++) num_flux += deltalambda[x] * starflux[x];
    } else {
        for (int x = 0; x < nbin; x++) num_flux += deltalambda[x] * PI * planck_grid[x + dim * nbin];
    }
    double corr = STEFANBOLTZMANN * pow(T_star, 4.0) / num_flux;
    for (int x = 0; x < nbin; x++) {
        if (realstar == 1)
            starflux[x] *= corr;
        else
            planck_grid[x + dim * nbin] *= corr;
    }
}

/* ============================================================================================
 * Temperatures and Planck interpolation          source/kernels.cu:496-520, 923-1011
 * ============================================================================================ */
void orc_temp_inter(const double* T_lay, double* T_int, int ninterface) {
    for (int i = 0; i < ninterface; i++) {
        if (i == 0)
            T_int[i] = T_lay[i] - 0.5 * (T_lay[i + 1] - T_lay[i]);
        else if (i == ninterface - 1)
            T_int[i] = T_lay[i - 1] + 0.5 * (T_lay[i - 1] - T_lay[i - 2]);
        else
            T_int[i] = T_lay[i - 1] + 0.5 * (T_lay[i] - T_lay[i - 1]);
    }
}

static double planck_lookup(const double* planck_grid, double T, int x, int nbin, int dim, int step) {
    double t = (T - 1.0) / step;
    t = dmax(0.001, dmin(dim - 1.001, t));
    int tdown = (int)floor(t);
    int tup = (int)ceil(t);
    if (tdown != tup)
        return planck_grid[x + tdown * nbin] * (tup - t) + planck_grid[x + tup * nbin] * (t - tdown);
    return planck_grid[x + tdown * nbin];
}

void orc_planck_interpol_layer(const double* T_lay, double* planckband_lay, const double* planck_grid,
                               const double* starflux, int realstar, int nlayer, int nbin, int dim,
                               int step) {
#pragma omp parallel for
    for (int x = 0; x < nbin; x++) {
        double* out = planckband_lay + (size_t)x * (nlayer + 2);
        for (int i = 0; i < nlayer; i++) out[i] = planck_lookup(planck_grid, T_lay[i], x, nbin, dim, step);
        out[nlayer] = (realstar == 1) ? starflux[x] / PI : planck_grid[x + dim * nbin];
        out[nlayer + 1] = planck_lookup(planck_grid, T_lay[nlayer], x, nbin, dim, step);
    }
}

void orc_planck_interpol_interface(const double* T_int, double* planckband_int,
                                   const double* planck_grid, int ninterface, int nbin, int dim,
                                   int step) {
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int i = 0; i < ninterface; i++)
            planckband_int[i + (size_t)x * ninterface] =
                planck_lookup(planck_grid, T_int[i], x, nbin, dim, step);
}

/* ============================================================================================
 * Table look-ups in (T, log10 P)
 * ============================================================================================ */
typedef struct {
    double t, p;
    int tdown, tup, pdown, pup;
} tp_index;

/* fractional index on a uniform grid derived from the first and last node; the margin is 0.001 for
 * the premixed/mean-mass/kappa tables (kernels.cu:545-559, 665-679) and 0 for the per-species
 * tables (kernels.cu:3228-3241). `log_t` selects log10(T) spacing (cp_interpol, kernels.cu:777). */
static tp_index locate_tp(double temp, double press, const double* tgrid, int ntemp,
                          const double* pgrid, int npress, double margin, int log_t) {
    tp_index k;
    double dt, t;
    if (log_t) {
        dt = (log10(tgrid[ntemp - 1]) - log10(tgrid[0])) / (ntemp - 1.0);
        t = (log10(temp) - log10(tgrid[0])) / dt;
    } else {
        dt = (tgrid[ntemp - 1] - tgrid[0]) / (ntemp - 1.0);
        t = (temp - tgrid[0]) / dt;
    }
    double dp = (log10(pgrid[npress - 1]) - log10(pgrid[0])) / (npress - 1.0);
    double p = (log10(press) - log10(pgrid[0])) / dp;
    if (margin > 0.0) {
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

/* four-case bilinear blend, kernels.cu:561-608 (same term order) */
static double blend_pt(double dd, double ud, double du, double uu, const tp_index* k) {
    double p = k->p, t = k->t;
    int pdown = k->pdown, pup = k->pup, tdown = k->tdown, tup = k->tup;
    if (pdown != pup && tdown != tup)
        return dd * (pup - p) * (tup - t) + ud * (p - pdown) * (tup - t) + du * (pup - p) * (t - tdown) +
               uu * (p - pdown) * (t - tdown);
    if (tdown == tup && pdown != pup) return dd * (pup - p) + ud * (p - pdown);
    if (pdown == pup && tdown != tup) return dd * (tup - t) + du * (t - tdown);
    return dd;
}

/* species variant, kernels.cu:613-645: the T-only case adds the upper term first */
static double blend_pt_species(double dd, double ud, double du, double uu, const tp_index* k) {
    double p = k->p, t = k->t;
    int pdown = k->pdown, pup = k->pup, tdown = k->tdown, tup = k->tup;
    if (pdown != pup && tdown != tup)
        return dd * (pup - p) * (tup - t) + ud * (p - pdown) * (tup - t) + du * (pup - p) * (t - tdown) +
               uu * (p - pdown) * (t - tdown);
    if (tdown == tup && pdown != pup) return dd * (pup - p) + ud * (p - pdown);
    if (pdown == pup && tdown != tup) return du * (t - tdown) + dd * (tup - t);
    return dd;
}

/* source/kernels.cu:524-610 */
void orc_opac_interpol(const double* temp, const double* opactemp, const double* press,
                       const double* opacpress, const double* ktable, double* opac,
                       const double* crosstable, double* scat_cross, int npress, int ntemp, int ny,
                       int nbin, int nlev) {
    const size_t sp = (size_t)ny * nbin, st = sp * npress;
    for (int i = 0; i < nlev; i++) {
        tp_index k = locate_tp(temp[i], press[i], opactemp, ntemp, opacpress, npress, 0.001, 0);
#pragma omp parallel for
        for (int x = 0; x < nbin; x++) {
            for (int y = 0; y < ny; y++) {
                size_t c = (size_t)y + (size_t)ny * x;
                opac[c + sp * i] = blend_pt(ktable[c + sp * k.pdown + st * k.tdown],
                                            ktable[c + sp * k.pup + st * k.tdown],
                                            ktable[c + sp * k.pdown + st * k.tup],
                                            ktable[c + sp * k.pup + st * k.tup], &k);
            }
            size_t cp = (size_t)nbin, ct = (size_t)nbin * npress;
            scat_cross[x + (size_t)nbin * i] =
                blend_pt(crosstable[x + cp * k.pdown + ct * k.tdown], crosstable[x + cp * k.pup + ct * k.tdown],
                         crosstable[x + cp * k.pdown + ct * k.tup], crosstable[x + cp * k.pup + ct * k.tup], &k);
        }
    }
}

static void scalar_table_interpol(const double* temp, const double* tgrid, const double* press,
                                  const double* pgrid, double* out, const double* table, int npress,
                                  int ntemp, int nlev, int log_t) {
    for (int i = 0; i < nlev; i++) {
        tp_index k = locate_tp(temp[i], press[i], tgrid, ntemp, pgrid, npress, 0.001, log_t);
        /* kernels.cu:681-697: here the T-only case is listed before the P-only one, same terms */
        out[i] = blend_pt(table[k.pdown + npress * k.tdown], table[k.pup + npress * k.tdown],
                          table[k.pdown + npress * k.tup], table[k.pup + npress * k.tup], &k);
    }
}

/* source/kernels.cu:649-699 */
void orc_meanmolmass_interpol(const double* temp, const double* opactemp, double* meanmolmass,
                              const double* opac_meanmass, const double* press,
                              const double* opacpress, int npress, int ntemp, int nlev) {
    scalar_table_interpol(temp, opactemp, press, opacpress, meanmolmass, opac_meanmass, npress, ntemp,
                          nlev, 0);
}
/* source/kernels.cu:703-757 (linear T) */
void orc_kappa_interpol(const double* temp, const double* entr_temp, const double* press,
                        const double* entr_press, double* kappa, const double* entr_kappa,
                        int entr_npress, int entr_ntemp, int nlev) {
    scalar_table_interpol(temp, entr_temp, press, entr_press, kappa, entr_kappa, entr_npress,
                          entr_ntemp, nlev, 0);
}
/* source/kernels.cu:761-811 (log10 T) */
void orc_cp_interpol(const double* temp, const double* entr_temp, const double* press,
                     const double* entr_press, double* cp, const double* entr_cp, int entr_npress,
                     int entr_ntemp, int nlev) {
    scalar_table_interpol(temp, entr_temp, press, entr_press, cp, entr_cp, entr_npress, entr_ntemp,
                          nlev, 1);
}
/* source/kernels.cu:815-865 (log10 T) */
void orc_entropy_interpol(const double* temp, const double* entr_temp, const double* press,
                          const double* entr_press, double* entropy, const double* entr_entropy,
                          int entr_npress, int entr_ntemp, int nlayer) {
    scalar_table_interpol(temp, entr_temp, press, entr_press, entropy, entr_entropy, entr_npress,
                          entr_ntemp, nlayer, 1);
}
/* source/kernels.cu:869-919 (linear T) */
void orc_phase_number_interpol(const double* temp, const double* entr_temp, const double* press,
                               const double* entr_press, double* state, const double* entr_state,
                               int entr_npress, int entr_ntemp, int nlayer) {
    scalar_table_interpol(temp, entr_temp, press, entr_press, state, entr_state, entr_npress, entr_ntemp,
                          nlayer, 0);
}

/* source/kernels.cu:3209-3259 */
void orc_opac_species_interpol(const double* temp, const double* opactemp, const double* press,
                               const double* opacpress, const double* pretab, double* opac_spec,
                               int npress, int ntemp, int ny, int nbin, int nlev) {
    const size_t sp = (size_t)ny * nbin, st = sp * npress;
    for (int i = 0; i < nlev; i++) {
        tp_index k = locate_tp(temp[i], press[i], opactemp, ntemp, opacpress, npress, 0.0, 0);
#pragma omp parallel for
        for (int x = 0; x < nbin; x++)
            for (int y = 0; y < ny; y++) {
                size_t c = (size_t)y + (size_t)ny * x;
                opac_spec[c + sp * i] = blend_pt_species(
                    pretab[c + sp * k.pdown + st * k.tdown], pretab[c + sp * k.pup + st * k.tdown],
                    pretab[c + sp * k.pdown + st * k.tup], pretab[c + sp * k.pup + st * k.tup], &k);
            }
    }
}

/* ============================================================================================
 * k-coefficient mixing: correlated-k and random overlap      source/kernels.cu:3143-3171, 3263-3399
 * ============================================================================================ */
enum { RO_NY = 20, RO_N = RO_NY * RO_NY };

/* ascending stable sort of (key, weight) pairs: the reference repeats adjacent-swap passes with a
 * strict '<' until a pass makes no swap (kernels.cu:3152-3171), i.e. a stable sort.  A stable
 * insertion sort produces the identical permutation. */
static void stable_sort_pairs(double* key, double* w, int n) {
    for (int a = 1; a < n; a++) {
        double k = key[a], g = w[a];
        int b = a - 1;
        while (b >= 0 && k < key[b]) {
            key[b + 1] = key[b];
            w[b + 1] = w[b];
            b--;
        }
        key[b + 1] = k;
        w[b + 1] = g;
    }
}

void orc_add_to_mixed_opac(const double* vmr, const double* opac_spec, double* opac_wg,
                           const double* meanmolmass, const double* gauss_weight,
                           const double* gauss_y, double mass_spec, int s, int ro_method, int ny,
                           int nbin, int nlev) {
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
    for (int i = 0; i < nlev; i++)
        for (int x = 0; x < nbin; x++) {
            double* mixp = opac_wg + (size_t)ny * x + (size_t)ny * nbin * i;
            const double* specp = opac_spec + (size_t)ny * x + (size_t)ny * nbin * i;
            double mix[RO_NY], add[RO_NY];
            int nloc = ny < RO_NY ? ny : RO_NY;
            if (ny > RO_NY) { /* only the correlated-k branch is defined for ny > 20 */
                for (int y = 0; y < ny; y++) mixp[y] += vmr[i] * mass_spec / meanmolmass[i] * specp[y];
                continue;
            }
            for (int y = 0; y < nloc; y++) {
                mix[y] = mixp[y];
                add[y] = vmr[i] * mass_spec / meanmolmass[i] * specp[y];
            }
            /* kernels.cu:3297-3302 */
            int negligible = (0.01 * mix[0] > add[ny - 1]) || (0.01 * add[0] > mix[ny - 1]);
            int corrk = (ro_method == 0) || (s == 0) || negligible || (ny == 1);
            if (corrk) {
                for (int y = 0; y < ny; y++) mixp[y] += add[y];
                continue;
            }
            /* random overlap: requires ny == 20 (kernels.cu:3315-3317) */
            double K[RO_N], G[RO_N], Y[RO_N];
            int yx = ny; /* first index after the LAST crossing of the two curves, :3321-3329 */
            for (int y = 1; y < ny; y++)
                if ((mix[y] > add[y]) != (mix[y - 1] > add[y - 1])) yx = y;
            /* fill order, kernels.cu:3332-3365 (matters only for the order of equal keys) */
            if (mix[0] > add[0]) {
                for (int y1 = 0; y1 < ny; y1++)
                    for (int y2 = 0; y2 < yx; y2++) {
                        K[y2 + yx * y1] = mix[y1] + add[y2];
                        G[y2 + yx * y1] = (0.5 * gauss_weight[y1]) * (0.5 * gauss_weight[y2]);
                    }
                for (int y2 = yx; y2 < ny; y2++)
                    for (int y1 = 0; y1 < ny; y1++) {
                        K[y1 + ny * y2] = mix[y1] + add[y2];
                        G[y1 + ny * y2] = (0.5 * gauss_weight[y1]) * (0.5 * gauss_weight[y2]);
                    }
            } else {
                for (int y2 = 0; y2 < ny; y2++)
                    for (int y1 = 0; y1 < yx; y1++) {
                        K[y1 + yx * y2] = mix[y1] + add[y2];
                        G[y1 + yx * y2] = (0.5 * gauss_weight[y1]) * (0.5 * gauss_weight[y2]);
                    }
                for (int y1 = yx; y1 < ny; y1++)
                    for (int y2 = 0; y2 < ny; y2++) {
                        K[y2 + ny * y1] = mix[y1] + add[y2];
                        G[y2 + ny * y1] = (0.5 * gauss_weight[y1]) * (0.5 * gauss_weight[y2]);
                    }
            }
            stable_sort_pairs(K, G, RO_N);
            /* cumulative mid-point abscissae, kernels.cu:3371-3376 */
            Y[0] = 0.5 * G[0];
            for (int w = 1; w < RO_N; w++) Y[w] = Y[w - 1] + 0.5 * G[w - 1] + 0.5 * G[w];
            /* re-binning to the Gauss abscissae, kernels.cu:3379-3396 */
            int q = 0;
            for (int w = 1; w < RO_N; w++) {
                if (Y[w] > gauss_y[q]) {
                    mixp[q] = (K[w - 1] * (Y[w] - gauss_y[q]) + K[w] * (gauss_y[q] - Y[w - 1])) /
                              (Y[w] - Y[w - 1]);
                    if (q < 19)
                        q++;
                    else
                        break;
                }
            }
        }
}

/* source/kernels.cu:3174-3205, 3404-3440 */
static double h2o_refractive_index(double wave, double press, double temp, double f_h2o,
                                   double mass_h2o) {
    double dens = f_h2o * press * mass_h2o / (KBOLTZMANN * temp);
    double lamda = wave / 0.589e-4;
    double delta = dmin(1.0, dens) / 1.0;
    double theta = temp / 273.15;
    const double lamda_UV = 0.229202, lamda_IR = 5.432937;
    const double a0 = 0.244257733, a1 = 0.974634476e-2, a2 = -0.373234996e-2, a3 = 0.268678472e-3,
                 a4 = 0.158920570e-2, a5 = 0.245934259e-2, a6 = 0.900704920, a7 = -0.166626219e-1;
    double A = delta * (a0 + a1 * delta + a2 * theta + a3 * pow(1.0 * lamda, 2.0) * theta +
                        a4 * pow(1.0 * lamda, -2.0) +
                        a5 / (pow(1.0 * lamda, 2.0) - pow(1.0 * lamda_UV, 2.0)) +
                        a6 / (pow(1.0 * lamda, 2.0) - pow(1.0 * lamda_IR, 2.0)) +
                        a7 * pow(1.0 * delta, 2.0));
    return pow((2.0 * A + 1.0) / (1.0 - A), 0.5);
}

void orc_calc_h2o_scat(const double* temp, const double* press, const double* wave,
                       double* scat_cross, const double* vmr, double mass_h2o, int nbin, int nlev) {
    for (int i = 0; i < nlev; i++)
        for (int x = 0; x < nbin; x++) {
            double index = h2o_refractive_index(wave[x], press[i], temp[i], vmr[i], mass_h2o);
            double n_ref = vmr[i] * press[i] / (KBOLTZMANN * temp[i]);
            double King = (6.0 + 3.0 * 3e-4) / (6.0 - 7.0 * 3e-4);
            double sc = 0.0;
            if (wave[x] < 2.5e-4)
                sc = 24.0 * pow(1.0 * PI, 3.0) / (pow(1.0 * n_ref, 2.0) * pow(1.0 * wave[x], 4.0)) *
                     pow((pow(1.0 * index, 2.0) - 1.0) / (pow(1.0 * index, 2.0) + 2.0), 2.0) * King;
            scat_cross[x + (size_t)nbin * i] = sc;
        }
}

/* source/kernels.cu:3444-3459 */
void orc_add_to_mixed_scat(const double* vmr, const double* scat_cross_spec, double* scat_cross,
                           int nbin, int nlev) {
    for (int i = 0; i < nlev; i++)
        for (int x = 0; x < nbin; x++)
            scat_cross[x + (size_t)nbin * i] += vmr[i] * scat_cross_spec[x + (size_t)nbin * i];
}

/* source/kernels.cu:472-492 */
void orc_calc_total_g0(const double* scat_cross, const double* g_0_all_clouds,
                       const double* scat_cross_all_clouds, double* g_0_tot, double g_0, int nbin,
                       int nlev) {
    for (size_t k = 0; k < (size_t)nbin * nlev; k++) {
        double num = g_0 * scat_cross[k] + g_0_all_clouds[k] * scat_cross_all_clouds[k];
        double den = scat_cross[k] + scat_cross_all_clouds[k];
        g_0_tot[k] = num / den;
    }
}

/* ============================================================================================
 * Two-stream coefficients                                  source/kernels.cu:109-290
 * ============================================================================================ */
static double E_param(double w0, double g0, double i2s_transition) {
    if (w0 > i2s_transition && g0 >= 0)
        return dmax(1.0, 1.225 - 0.1582 * g0 - 0.1777 * w0 - 0.07465 * pow(1.0 * g0, 2.0) +
                             0.2351 * w0 * g0 - 0.05582 * pow(w0, 2.0));
    return 1.0;
}
static double E_of(double w0, double g0, int scat_corr, double i2s) {
    return scat_corr == 1 ? E_param(w0, g0, i2s) : 1.0;
}
static double G_clip(double G) { return fabs(G) < 1e8 ? G : 1e8 * G / fabs(G); }

typedef struct {
    double w0, trans, M, N, P, Gp, Gm;
} coeffs;

/* everything that calc_trans_{iso,noniso} derive from (w0, total delta_tau, g0) */
static coeffs slab_coeffs(double w0, double del_tau, double g0, double epsi, double epsi2,
                          double mu_star, int scat_corr, double i2s) {
    coeffs c;
    double E = E_of(w0, g0, scat_corr, i2s);
    c.w0 = w0;
    c.trans = exp(-1.0 / epsi * sqrt(E * (1.0 - w0 * g0) * (E - w0)) * del_tau); /* :144 */
    double zm = 0.5 * (1.0 - sqrt((E - w0) / (E * (1.0 - w0 * g0))));             /* :272 */
    double zp = 0.5 * (1.0 + sqrt((E - w0) / (E * (1.0 - w0 * g0))));             /* :289 */
    c.M = (zm * zm) * (c.trans * c.trans) - (zp * zp);
    c.N = zp * zm * (1.0 - (c.trans * c.trans));
    c.P = ((zm * zm) - (zp * zp)) * c.trans;
    /* G+-, :149-213 */
    double num = w0 * (E * (1.0 - w0 * g0) + g0 * epsi / epsi2);
    double den = E * pow(epsi, -2.0) * (E - w0) * (1.0 - w0 * g0) - pow(mu_star, -2.0);
    double third = epsi * w0 * g0 * mu_star / (epsi2 * E * (1.0 - w0 * g0));
    double second_p = 1.0 / epsi + 1.0 / (mu_star * E * (1.0 - w0 * g0));
    double second_m = 1.0 / epsi - 1.0 / (mu_star * E * (1.0 - w0 * g0));
    c.Gp = G_clip(0.5 * (num / den * second_p + third));
    c.Gm = G_clip(0.5 * (num / den * second_m - third));
    return c;
}

/* source/kernels.cu:1015-1104 */
void orc_calc_trans_iso(double* trans_wg, double* delta_tau_wg, double* M_term, double* N_term,
                        double* P_term, double* G_plus, double* G_minus, const double* delta_colmass,
                        const double* opac_wg_lay, const double* meanmolmass_lay,
                        const double* scat_cross_lay, const double* abs_cross_all_clouds_lay,
                        const double* scat_cross_all_clouds_lay, double* delta_tau_all_clouds,
                        double* w_0, const double* g_0_tot_lay, int* scat_trigger, double g_0,
                        double epsi, double epsi2, double mu_star, double w_0_limit,
                        double w_0_scat_limit, int scat, int nbin, int ny, int nlayer, int clouds,
                        int scat_corr, double i2s_transition) {
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int i = 0; i < nlayer; i++) {
            size_t b = x + (size_t)nbin * i;
            double g0 = clouds == 1 ? g_0_tot_lay[b] : g_0;
            double ray = scat == 1 ? scat_cross_lay[b] : 0.0;
            double cl_sc = scat == 1 ? scat_cross_all_clouds_lay[b] : 0.0;
            double cl_abs = abs_cross_all_clouds_lay[b];
            double mu = meanmolmass_lay[i];
            delta_tau_all_clouds[b] = delta_colmass[i] * (cl_abs + cl_sc) / mu;
            for (int y = 0; y < ny; y++) {
                size_t k = (size_t)y + (size_t)ny * x + (size_t)ny * nbin * i;
                double kap = opac_wg_lay[k];
                double w0 = dmin((ray + cl_sc) / ((ray + cl_sc) + (kap * mu + cl_abs)), w_0_limit);
                double dtau = delta_colmass[i] * (kap + ray / mu);
                coeffs c = slab_coeffs(w0, dtau + delta_tau_all_clouds[b], g0, epsi, epsi2, mu_star,
                                       scat_corr, i2s_transition);
                w_0[k] = w0;
                delta_tau_wg[k] = dtau;
                trans_wg[k] = c.trans;
                M_term[k] = c.M;
                N_term[k] = c.N;
                P_term[k] = c.P;
                G_plus[k] = c.Gp;
                G_minus[k] = c.Gm;
                if (w0 > w_0_scat_limit) scat_trigger[y + ny * x] = 1;
            }
        }
}

/* source/kernels.cu:1107-1243 */
void orc_calc_trans_noniso(
    double* trans_wg_upper, double* trans_wg_lower, double* delta_tau_wg_upper,
    double* delta_tau_wg_lower, double* M_upper, double* M_lower, double* N_upper, double* N_lower,
    double* P_upper, double* P_lower, double* G_plus_upper, double* G_plus_lower,
    double* G_minus_upper, double* G_minus_lower, const double* delta_col_upper,
    const double* delta_col_lower, const double* opac_wg_lay, const double* opac_wg_int,
    const double* meanmolmass_lay, const double* meanmolmass_int, const double* scat_cross_lay,
    const double* scat_cross_int, const double* abs_cross_all_clouds_lay,
    const double* abs_cross_all_clouds_int, const double* scat_cross_all_clouds_lay,
    const double* scat_cross_all_clouds_int, double* delta_tau_all_clouds_upper,
    double* delta_tau_all_clouds_lower, double* w_0_upper, double* w_0_lower,
    const double* g_0_tot_lay, const double* g_0_tot_int, int* scat_trigger, double g_0, double epsi,
    double epsi2, double mu_star, double w_0_limit, double w_0_scat_limit, int scat, int nbin, int ny,
    int nlayer, int clouds, int scat_corr, double i2s_transition) {
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int i = 0; i < nlayer; i++) {
            size_t b = x + (size_t)nbin * i, bu = x + (size_t)nbin * (i + 1);
            double g0_up = g_0, g0_low = g_0;
            if (clouds == 1) {
                g0_up = (g_0_tot_lay[b] + g_0_tot_int[bu]) / 2.0;
                g0_low = (g_0_tot_int[b] + g_0_tot_lay[b]) / 2.0;
            }
            double ray_up = 0, ray_low = 0, csc_up = 0, csc_low = 0;
            if (scat == 1) {
                ray_up = (scat_cross_lay[b] + scat_cross_int[bu]) / 2.0;
                ray_low = (scat_cross_int[b] + scat_cross_lay[b]) / 2.0;
                csc_up = (scat_cross_all_clouds_lay[b] + scat_cross_all_clouds_int[bu]) / 2.0;
                csc_low = (scat_cross_all_clouds_int[b] + scat_cross_all_clouds_lay[b]) / 2.0;
            }
            double cab_up = (abs_cross_all_clouds_lay[b] + abs_cross_all_clouds_int[bu]) / 2.0;
            double cab_low = (abs_cross_all_clouds_int[b] + abs_cross_all_clouds_lay[b]) / 2.0;
            double mu_up = (meanmolmass_lay[i] + meanmolmass_int[i + 1]) / 2.0;
            double mu_low = (meanmolmass_int[i] + meanmolmass_lay[i]) / 2.0;
            delta_tau_all_clouds_upper[b] = delta_col_upper[i] * (cab_up + csc_up) / mu_up;
            delta_tau_all_clouds_lower[b] = delta_col_lower[i] * (cab_low + csc_low) / mu_low;
            for (int y = 0; y < ny; y++) {
                size_t k = (size_t)y + (size_t)ny * x + (size_t)ny * nbin * i;
                size_t ku = k + (size_t)ny * nbin;
                double kap_up = (opac_wg_lay[k] + opac_wg_int[ku]) / 2.0;
                double kap_low = (opac_wg_int[k] + opac_wg_lay[k]) / 2.0;
                double w_up = dmin((ray_up + csc_up) / ((ray_up + csc_up) + (kap_up * mu_up + cab_up)),
                                   w_0_limit);
                double w_low = dmin(
                    (ray_low + csc_low) / ((ray_low + csc_low) + (kap_low * mu_low + cab_low)), w_0_limit);
                double dt_up = delta_col_upper[i] * (kap_up + ray_up / mu_up);
                double dt_low = delta_col_lower[i] * (kap_low + ray_low / mu_low);
                coeffs cu = slab_coeffs(w_up, dt_up + delta_tau_all_clouds_upper[b], g0_up, epsi, epsi2,
                                        mu_star, scat_corr, i2s_transition);
                coeffs cl = slab_coeffs(w_low, dt_low + delta_tau_all_clouds_lower[b], g0_low, epsi,
                                        epsi2, mu_star, scat_corr, i2s_transition);
                w_0_upper[k] = w_up;
                w_0_lower[k] = w_low;
                delta_tau_wg_upper[k] = dt_up;
                delta_tau_wg_lower[k] = dt_low;
                trans_wg_upper[k] = cu.trans;
                trans_wg_lower[k] = cl.trans;
                M_upper[k] = cu.M;
                M_lower[k] = cl.M;
                N_upper[k] = cu.N;
                N_lower[k] = cl.N;
                P_upper[k] = cu.P;
                P_lower[k] = cl.P;
                G_plus_upper[k] = cu.Gp;
                G_plus_lower[k] = cl.Gp;
                G_minus_upper[k] = cu.Gm;
                G_minus_lower[k] = cl.Gm;
                if (w_up > w_0_scat_limit) scat_trigger[y + ny * x] = 1;
                if (w_low > w_0_scat_limit) scat_trigger[y + ny * x] = 1;
            }
        }
}

/* source/kernels.cu:1247-1261 */
void orc_calc_delta_z(const double* T_lay, const double* p_int, const double* meanmolmass_lay,
                      double* delta_z_lay, double g, int nlayer) {
    for (int i = 0; i < nlayer; i++)
        delta_z_lay[i] = KBOLTZMANN * T_lay[i] / (meanmolmass_lay[i] * g) * log(p_int[i] / p_int[i + 1]);
}

/* ============================================================================================
 * Direct stellar beam                                      source/kernels.cu:1265-1362
 * ============================================================================================ */
static double slant_mu(double mu_star, double R_planet, const double* z_lay, int i, int j,
                       int geom_zenith_corr) {
    if (geom_zenith_corr == 1)
        return -sqrt(1.0 - pow((R_planet + z_lay[i]) / (R_planet + z_lay[j]), 2.0) *
                               (1.0 - pow(mu_star, 2.0)));
    return mu_star;
}

void orc_fdir_iso(double* F_dir_wg, const double* planckband_lay, const double* delta_tau_wg,
                  const double* z_lay, double mu_star, double R_planet, double R_star, double a,
                  int dir_beam, int geom_zenith_corr, int ninterface, int nbin, int ny) {
    const size_t sl = (size_t)ny * nbin;
#pragma omp parallel for
    for (int x = 0; x < nbin; x++) {
        double I_dir = ((R_star / a) * (R_star / a)) * PI *
                       planckband_lay[(ninterface - 1) + (size_t)x * (ninterface - 1 + 2)];
        for (int y = 0; y < ny; y++)
            for (int i = 0; i < ninterface; i++) {
                size_t c = (size_t)y + (size_t)ny * x;
                double F = -dir_beam * mu_star * I_dir;
                for (int j = ninterface - 2; j >= i; j--)
                    F *= exp(delta_tau_wg[c + sl * j] /
                             slant_mu(mu_star, R_planet, z_lay, i, j, geom_zenith_corr));
                F_dir_wg[c + sl * i] = F;
            }
    }
}

void orc_fdir_noniso(double* F_dir_wg, double* Fc_dir_wg, const double* planckband_lay,
                     const double* delta_tau_wg_upper, const double* delta_tau_wg_lower,
                     const double* z_lay, double mu_star, double R_planet, double R_star, double a,
                     int dir_beam, int geom_zenith_corr, int ninterface, int nbin, int ny) {
    const size_t sl = (size_t)ny * nbin;
#pragma omp parallel for
    for (int x = 0; x < nbin; x++) {
        double I_dir = ((R_star / a) * (R_star / a)) * PI *
                       planckband_lay[(ninterface - 1) + (size_t)x * (ninterface - 1 + 2)];
        for (int y = 0; y < ny; y++)
            for (int i = 0; i < ninterface; i++) {
                size_t c = (size_t)y + (size_t)ny * x;
                double F = -dir_beam * mu_star * I_dir;
                for (int j = ninterface - 2; j >= i; j--) {
                    double mu_j = slant_mu(mu_star, R_planet, z_lay, i, j, geom_zenith_corr);
                    double dtau = delta_tau_wg_upper[c + sl * j] + delta_tau_wg_lower[c + sl * j];
                    /* Fc_dir keeps the value of the LAST pass (j == i): everything above layer i
                     * times the upper half of layer i.  Fc_dir[TOA] is never written (:1346). */
                    Fc_dir_wg[c + sl * i] = F * exp(delta_tau_wg_upper[c + sl * j] / mu_j);
                    F *= exp(dtau / mu_j);
                }
                F_dir_wg[c + sl * i] = F;
            }
    }
}

/* ============================================================================================
 * Two-stream sweeps                                        source/kernels.cu:1366-1799
 * ============================================================================================ */
static inline double tiny_abs(double F) { return fabs(F) < 1e-100 ? fabs(F) : F; }

void orc_fband_iso(double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                   const double* planckband_lay, const double* w_0, const double* M_term,
                   const double* N_term, const double* P_term, const double* G_plus,
                   const double* G_minus, const double* surf_albedo, const double* g_0_tot_lay,
                   double g_0, double Rstar, double a, int ninterface, int nbin, double f_factor,
                   double mu_star, int ny, double epsi, int dir_beam, int clouds, int scat_corr,
                   double i2s_transition) {
    const size_t sl = (size_t)ny * nbin;
    const int npl = ninterface - 1 + 2;
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int y = 0; y < ny; y++) {
            const size_t c = (size_t)y + (size_t)ny * x;
            const double* B = planckband_lay + (size_t)x * npl;
            double w0 = 0, E = 1.0;
            /* down, TOA -> BOA (:1416-1461) */
            F_down_wg[c + sl * (ninterface - 1)] =
                (1.0 - dir_beam) * f_factor * ((Rstar / a) * (Rstar / a)) * PI * B[ninterface - 1];
            for (int i = ninterface - 2; i >= 0; i--) {
                size_t k = c + sl * i;
                w0 = w_0[k];
                double M = M_term[k], N = N_term[k], P = P_term[k], Gp = G_plus[k], Gm = G_minus[k];
                double g0 = clouds == 1 ? g_0_tot_lay[x + (size_t)nbin * i] : g_0;
                E = E_of(w0, g0, scat_corr, i2s_transition);
                double flux = P * F_down_wg[k + sl] - N * F_up_wg[k];
                double planck = B[i] * (N + M - P);
                double direct = F_dir_wg[k] / (-mu_star) * (Gm * M + Gp * N) -
                                F_dir_wg[k + sl] / (-mu_star) * P * Gm;
                direct = dmin(0.0, direct);
                F_down_wg[k] =
                    tiny_abs(1.0 / M * (flux + 2.0 * PI * epsi * (1.0 - w0) / (E - w0) * planck + direct));
            }
            /* up, BOA -> TOA (:1464-1515); the BOA term re-uses w0/E of layer 0 (Q8) */
            F_up_wg[c] = surf_albedo[x] * (F_dir_wg[c] + F_down_wg[c]) +
                         (1.0 - surf_albedo[x]) * PI * (1.0 - w0) / (E - w0) * B[ninterface];
            for (int i = 1; i < ninterface; i++) {
                size_t k = c + sl * (i - 1);
                w0 = w_0[k];
                double M = M_term[k], N = N_term[k], P = P_term[k], Gp = G_plus[k], Gm = G_minus[k];
                double g0 = clouds == 1 ? g_0_tot_lay[x + (size_t)nbin * (i - 1)] : g_0;
                E = E_of(w0, g0, scat_corr, i2s_transition);
                double flux = P * F_up_wg[k] - N * F_down_wg[k + sl];
                double planck = B[i - 1] * (N + M - P);
                double direct = F_dir_wg[k + sl] / (-mu_star) * (Gm * N + Gp * M) -
                                F_dir_wg[k] / (-mu_star) * P * Gp;
                direct = dmin(0.0, direct);
                F_up_wg[k + sl] =
                    tiny_abs(1.0 / M * (flux + 2.0 * PI * epsi * (1.0 - w0) / (E - w0) * planck + direct));
            }
        }
}

void orc_fband_noniso(double* F_down_wg, double* F_up_wg, double* Fc_down_wg, double* Fc_up_wg,
                      const double* F_dir_wg, const double* Fc_dir_wg, const double* planckband_lay,
                      const double* planckband_int, const double* w_0_upper, const double* w_0_lower,
                      const double* delta_tau_wg_upper, const double* delta_tau_wg_lower,
                      const double* delta_tau_all_clouds_upper,
                      const double* delta_tau_all_clouds_lower, const double* M_upper,
                      const double* M_lower, const double* N_upper, const double* N_lower,
                      const double* P_upper, const double* P_lower, const double* G_plus_upper,
                      const double* G_plus_lower, const double* G_minus_upper,
                      const double* G_minus_lower, const double* surf_albedo,
                      const double* g_0_tot_lay, const double* g_0_tot_int, double g_0, double Rstar,
                      double a, int ninterface, int nbin, double f_factor, double mu_star, int ny,
                      double epsi, double delta_tau_limit, int dir_beam, int clouds, int scat_corr,
                      double i2s_transition) {
    const size_t sl = (size_t)ny * nbin;
    const int nlayer = ninterface - 1, npl = nlayer + 2;
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int y = 0; y < ny; y++) {
            const size_t c = (size_t)y + (size_t)ny * x;
            const double* Bl = planckband_lay + (size_t)x * npl;
            const double* Bi = planckband_int + (size_t)x * ninterface;
            double w_low = 0, E_low = 1.0;

            /* ---- down, TOA -> BOA (:1597-1693) ---- */
            F_down_wg[c + sl * nlayer] =
                (1.0 - dir_beam) * f_factor * ((Rstar / a) * (Rstar / a)) * PI * Bl[nlayer];
            for (int i = nlayer - 1; i >= 0; i--) {
                size_t k = c + sl * i, b = x + (size_t)nbin * i;
                double w_up = w_0_upper[k];
                double dt_up = delta_tau_wg_upper[k] + delta_tau_all_clouds_upper[b];
                double M_up = M_upper[k], N_up = N_upper[k], P_up = P_upper[k];
                double Gp_up = G_plus_upper[k], Gm_up = G_minus_upper[k];
                w_low = w_0_lower[k];
                double dt_low = delta_tau_wg_lower[k] + delta_tau_all_clouds_lower[b];
                double M_low = M_lower[k], N_low = N_lower[k], P_low = P_lower[k];
                double Gp_low = G_plus_lower[k], Gm_low = G_minus_lower[k];
                double g0_up = g_0, g0_low = g_0;
                if (clouds == 1) {
                    g0_up = (g_0_tot_lay[b] + g_0_tot_int[b + nbin]) / 2.0;
                    g0_low = (g_0_tot_int[b] + g_0_tot_lay[b]) / 2.0;
                }
                double E_up = E_of(w_up, g0_up, scat_corr, i2s_transition);
                E_low = E_of(w_low, g0_low, scat_corr, i2s_transition);
                double planck, flux, direct;
                /* upper half: interface i+1 -> layer centre i */
                if (dt_up < delta_tau_limit) {
                    planck = (Bi[i + 1] + Bl[i]) / 2.0 * (N_up + M_up - P_up);
                } else {
                    double pgrad = (Bl[i] - Bi[i + 1]) / dt_up;
                    planck = Bl[i] * (M_up + N_up) - Bi[i + 1] * P_up +
                             epsi / (E_up * (1.0 - w_up * g0_up)) * (P_up - M_up + N_up) * pgrad;
                }
                flux = P_up * F_down_wg[k + sl] - N_up * Fc_up_wg[k];
                direct = Fc_dir_wg[k] / (-mu_star) * (Gm_up * M_up + Gp_up * N_up) -
                         F_dir_wg[k + sl] / (-mu_star) * Gm_up * P_up;
                direct = dmin(0.0, direct);
                Fc_down_wg[k] = tiny_abs(
                    1.0 / M_up * (flux + 2.0 * PI * epsi * (1.0 - w_up) / (E_up - w_up) * planck + direct));
                /* lower half: layer centre i -> interface i */
                if (dt_low < delta_tau_limit) {
                    planck = (Bi[i] + Bl[i]) / 2.0 * (N_low + M_low - P_low);
                } else {
                    double pgrad = (Bi[i] - Bl[i]) / dt_low;
                    planck = Bi[i] * (M_low + N_low) - Bl[i] * P_low +
                             epsi / (E_low * (1.0 - w_low * g0_low)) * (P_low - M_low + N_low) * pgrad;
                }
                flux = P_low * Fc_down_wg[k] - N_low * F_up_wg[k];
                direct = F_dir_wg[k] / (-mu_star) * (Gm_low * M_low + Gp_low * N_low) -
                         Fc_dir_wg[k] / (-mu_star) * P_low * Gm_low;
                direct = dmin(0.0, direct);
                F_down_wg[k] = tiny_abs(1.0 / M_low *
                                        (flux + 2.0 * PI * epsi * (1.0 - w_low) / (E_low - w_low) * planck + direct));
            }

            /* ---- up, BOA -> TOA (:1696-1797); BOA uses w0/E of layer 0's lower half (Q8) ---- */
            F_up_wg[c] = surf_albedo[x] * (F_dir_wg[c] + F_down_wg[c]) +
                         (1.0 - surf_albedo[x]) * PI * (1.0 - w_low) / (E_low - w_low) * Bl[ninterface];
            for (int i = 1; i < ninterface; i++) {
                size_t k = c + sl * (i - 1), b = x + (size_t)nbin * (i - 1);
                w_low = w_0_lower[k];
                double dt_low = delta_tau_wg_lower[k] + delta_tau_all_clouds_lower[b];
                double M_low = M_lower[k], N_low = N_lower[k], P_low = P_lower[k];
                double Gp_low = G_plus_lower[k], Gm_low = G_minus_lower[k];
                double w_up = w_0_upper[k];
                double dt_up = delta_tau_wg_upper[k] + delta_tau_all_clouds_upper[b];
                double M_up = M_upper[k], N_up = N_upper[k], P_up = P_upper[k];
                double Gp_up = G_plus_upper[k], Gm_up = G_minus_upper[k];
                double g0_up = g_0, g0_low = g_0;
                if (clouds == 1) {
                    g0_low = (g_0_tot_int[b] + g_0_tot_lay[b]) / 2.0;
                    g0_up = (g_0_tot_lay[b] + g_0_tot_int[b + nbin]) / 2.0;
                }
                double E_up = E_of(w_up, g0_up, scat_corr, i2s_transition);
                E_low = E_of(w_low, g0_low, scat_corr, i2s_transition);
                double planck, flux, direct;
                /* lower half: interface i-1 -> layer centre i-1 */
                if (dt_low < delta_tau_limit) {
                    planck = (Bi[i - 1] + Bl[i - 1]) / 2.0 * (N_low + M_low - P_low);
                } else {
                    double pgrad = (Bi[i - 1] - Bl[i - 1]) / dt_low;
                    planck = Bl[i - 1] * (M_low + N_low) - Bi[i - 1] * P_low +
                             epsi / (E_low * (1.0 - w_low * g0_low)) * pgrad * (M_low - P_low - N_low);
                }
                flux = P_low * F_up_wg[k] - N_low * Fc_down_wg[k];
                direct = Fc_dir_wg[k] / (-mu_star) * (Gm_low * N_low + Gp_low * M_low) -
                         F_dir_wg[k] / (-mu_star) * P_low * Gp_low;
                direct = dmin(0.0, direct);
                Fc_up_wg[k] = 1.0 / M_low *
                              (flux + 2.0 * PI * epsi * (1.0 - w_low) / (E_low - w_low) * planck + direct);
                /* the reference's tiny-value patch addresses index i, not i-1 (:1763) */
                Fc_up_wg[k + sl] = tiny_abs(Fc_up_wg[k + sl]);
                /* upper half: layer centre i-1 -> interface i */
                if (dt_up < delta_tau_limit) {
                    planck = (Bi[i] + Bl[i - 1]) / 2.0 * (N_up + M_up - P_up);
                } else {
                    double pgrad = (Bl[i - 1] - Bi[i]) / dt_up;
                    planck = Bi[i] * (M_up + N_up) - Bl[i - 1] * P_up +
                             epsi / (E_up * (1.0 - w_up * g0_up)) * pgrad * (M_up - P_up - N_up);
                }
                flux = P_up * Fc_up_wg[k] - N_up * F_down_wg[k + sl];
                direct = F_dir_wg[k + sl] / (-mu_star) * (Gm_up * N_up + Gp_up * M_up) -
                         Fc_dir_wg[k] / (-mu_star) * P_up * Gp_up;
                direct = dmin(0.0, direct);
                F_up_wg[k + sl] = tiny_abs(
                    1.0 / M_up * (flux + 2.0 * PI * epsi * (1.0 - w_up) / (E_up - w_up) * planck + direct));
            }
        }
}

/* ============================================================================================
 * Quadrature and totals                                    source/kernels.cu:2428-2513
 * The reference accumulates with CAS-loop atomics in arbitrary order; this sums in index order.
 * ============================================================================================ */
void orc_integrate_flux(const double* deltalambda, double* F_down_tot, double* F_up_tot,
                        double* F_net, const double* F_down_wg, const double* F_up_wg,
                        const double* F_dir_wg, double* F_down_band, double* F_up_band,
                        double* F_dir_band, const double* gauss_weight, int nbin, int ninterface,
                        int ny) {
    const size_t sl = (size_t)ny * nbin;
#pragma omp parallel for
    for (int i = 0; i < ninterface; i++) {
        double up = 0.0, down = 0.0;
        for (int x = 0; x < nbin; x++) {
            double d = 0.0, u = 0.0, dn = 0.0;
            for (int y = 0; y < ny; y++) {
                size_t k = (size_t)y + (size_t)ny * x + sl * i;
                d += 0.5 * gauss_weight[y] * F_dir_wg[k];
                u += 0.5 * gauss_weight[y] * F_up_wg[k];
                dn += 0.5 * gauss_weight[y] * F_down_wg[k];
            }
            size_t b = x + (size_t)nbin * i;
            F_dir_band[b] = d;
            F_up_band[b] = u;
            F_down_band[b] = dn;
            up += u * deltalambda[x];
            down += (d + dn) * deltalambda[x];
        }
        F_up_tot[i] = up;
        F_down_tot[i] = down;
        F_net[i] = up - down;
    }
}

/* ============================================================================================
 * Temperature steps                                        source/kernels.cu:2606-2884
 * ============================================================================================ */
void orc_rad_temp_iter(const double* F_down_tot, const double* F_up_tot, const double* F_net,
                       double* F_net_diff, double* T_lay, const double* p_lay, const double* p_int,
                       int* abrt, double* T_store, double* deltat_prefactor,
                       const double* F_add_heat_lay, const double* F_add_heat_sum, double* F_smooth,
                       double* F_smooth_sum, const double* c_p_lay, const double* meanmolmass_lay,
                       int itervalue, int foreplay, double g, int nlayer, double physical_tstep,
                       double local_limit, int adapt_interval, int smooth, int dim, int step,
                       double F_intern, int no_atmo) {
    (void)F_up_tot;
    /* smoothing flux from the temperatures BEFORE this step, then its prefix sum -- done
     * deterministically here (the reference's version is racy) */
    if (smooth == 1) {
        for (int i = 0; i < nlayer; i++) {
            double t_mid = T_lay[i];
            if (p_lay[i] < 1e6 && i < nlayer - 1 && i > 0) t_mid = (T_lay[i - 1] + T_lay[i + 1]) / 2.0;
            F_smooth[i] = pow((t_mid - T_lay[i]), 7.0);
        }
        for (int i = 0; i < nlayer; i++) {
            double s = 0;
            for (int j = 0; j <= i; j++) s += F_smooth[j];
            F_smooth_sum[i] = s;
        }
    }
    const double F_toa = F_down_tot[nlayer];
    for (int i = 0; i <= nlayer; i++) {
        double dF, delta_T = 0.0;
        if (i < nlayer) {
            F_net_diff[i] = F_net[i] - F_net[i + 1] + F_add_heat_lay[i];
            dF = F_net_diff[i] + F_smooth[i];
        } else {
            dF = F_intern - F_net[0];
            if (fabs(F_intern - F_net[1]) / (F_toa + F_intern) > 0.5 * local_limit)
                dF = F_intern - F_net[1];
        }
        if (physical_tstep == 0) {
            if (itervalue == foreplay) deltat_prefactor[i] = 1e0;
            if (itervalue == 10000) deltat_prefactor[i] = 1e-1;
            if (dF != 0) {
                double delta_t = deltat_prefactor[i] * p_lay[0] / pow(fabs(dF), 0.9);
                delta_T = dF / (p_int[0] - p_int[1]) * delta_t;
            } /* dF == 0: the reference multiplies 0 by an uninitialised value; defined as 0 here */
            if (fabs(delta_T) > 500.0) delta_T = 500.0 * dF / fabs(dF);
            if (itervalue % adapt_interval == 0) T_store[i] = T_lay[i];
            if (itervalue % adapt_interval == adapt_interval - 1) {
                if (fabs(T_lay[i] - T_store[i]) < adapt_interval / 2.0 * fabs(delta_T))
                    deltat_prefactor[i] /= 1.5;
                else
                    deltat_prefactor[i] *= 1.1;
            }
        } else {
            int j = i < nlayer ? i : 0;
            delta_T = g / (c_p_lay[j] / (meanmolmass_lay[j] / AMU)) * dF / (p_int[j] - p_int[j + 1]) *
                      physical_tstep;
        }
        double T = T_lay[i] + delta_T;
        if (no_atmo == 1 && i != nlayer) T = 1.001;
        T_lay[i] = dmin(dmax(T, 1.001), dim * step - 1.001);
        int ok;
        if (i < nlayer)
            ok = fabs(F_intern + F_add_heat_sum[i] + F_smooth_sum[i] - F_net[i + 1]) / (F_toa + F_intern) <
                 local_limit;
        else
            ok = fabs(F_intern - F_net[0]) / (F_toa + F_intern) < local_limit;
        abrt[i] = ok ? 1 : 0;
    }
}

void orc_conv_temp_iter(const double* F_net, double* F_net_diff, double* T_lay, const double* p_lay,
                        const double* p_int, double* T_store, double* deltat_prefactor,
                        const int* marked_red, const double* F_add_heat_lay, double* F_smooth,
                        double* F_smooth_sum, int nlayer, int itervalue, int adapt_interval,
                        int smooth, double F_intern) {
    if (smooth == 1) {
        for (int i = 0; i < nlayer; i++) {
            double t_mid = T_lay[i];
            /* :2808 has no i > 0 guard; i == 0 would read T_lay[-1] -- guarded here */
            if (p_lay[i] < 1e6 && i < nlayer - 1 && i > 0) t_mid = (T_lay[i - 1] + T_lay[i + 1]) / 2.0;
            F_smooth[i] = pow((t_mid - T_lay[i]), 7.0);
        }
        for (int i = 0; i < nlayer; i++) {
            double s = 0;
            for (int j = 0; j <= i; j++) s += F_smooth[j];
            F_smooth_sum[i] = s;
        }
    }
    for (int i = 0; i <= nlayer; i++) {
        double dF;
        if (i < nlayer) {
            F_net_diff[i] = F_net[i] - F_net[i + 1] + F_add_heat_lay[i];
            dF = F_net_diff[i] + F_smooth[i];
        } else {
            dF = F_intern - F_net[0];
            for (int j = 0; j < nlayer; j++)
                if (marked_red[j] == 1) {
                    dF = F_intern - F_net[j + 1];
                    break;
                }
        }
        if (itervalue == 0) deltat_prefactor[i] = 1e-2;
        if (itervalue == 6000) deltat_prefactor[i] = 1e-3;
        double delta_T = 0.0;
        if (dF != 0) {
            double delta_t = deltat_prefactor[i] * p_lay[0] / pow(fabs(dF), 0.5);
            delta_T = dF / (p_int[0] - p_int[1]) * delta_t;
        }
        if (fabs(delta_T) > 20.0) delta_T = 20.0 * dF / fabs(dF);
        if (itervalue % adapt_interval == 0) T_store[i] = T_lay[i];
        if (itervalue % adapt_interval == adapt_interval - 1) {
            if (fabs(T_lay[i] - T_store[i]) < adapt_interval / 2.0 * fabs(delta_T))
                deltat_prefactor[i] /= 1.5;
            else
                deltat_prefactor[i] *= 1.1;
        }
        T_lay[i] = dmax(T_lay[i] + delta_T, 1.001);
    }
}

/* ============================================================================================
 * Post-loop diagnostics                                    source/kernels.cu:2888-3139
 * ============================================================================================ */
void orc_integrate_optdepth_transmission_iso(const double* trans_wg, double* trans_band,
                                             const double* delta_tau_wg, double* delta_tau_band,
                                             const double* gauss_weight, int nbin, int nlayer,
                                             int ny) {
    for (int i = 0; i < nlayer; i++)
        for (int x = 0; x < nbin; x++) {
            double dt = 0, tr = 0;
            for (int y = 0; y < ny; y++) {
                size_t k = (size_t)y + (size_t)ny * x + (size_t)ny * nbin * i;
                dt += 0.5 * gauss_weight[y] * delta_tau_wg[k];
                tr += 0.5 * gauss_weight[y] * trans_wg[k];
            }
            delta_tau_band[x + (size_t)nbin * i] = dt;
            trans_band[x + (size_t)nbin * i] = tr;
        }
}

void orc_integrate_optdepth_transmission_noniso(
    const double* trans_wg_upper, const double* trans_wg_lower, double* trans_band,
    const double* delta_tau_wg_upper, const double* delta_tau_wg_lower, double* delta_tau_band,
    const double* gauss_weight, double* delta_tau_all_clouds,
    const double* delta_tau_all_clouds_upper, const double* delta_tau_all_clouds_lower, int nbin,
    int nlayer, int ny) {
    for (int i = 0; i < nlayer; i++)
        for (int x = 0; x < nbin; x++) {
            double dt = 0, tr = 0;
            for (int y = 0; y < ny; y++) {
                size_t k = (size_t)y + (size_t)ny * x + (size_t)ny * nbin * i;
                dt += 0.5 * gauss_weight[y] * (delta_tau_wg_upper[k] + delta_tau_wg_lower[k]);
                tr += 0.5 * gauss_weight[y] * (trans_wg_upper[k] * trans_wg_lower[k]);
            }
            size_t b = x + (size_t)nbin * i;
            delta_tau_band[b] = dt;
            trans_band[b] = tr;
            delta_tau_all_clouds[b] = delta_tau_all_clouds_lower[b] + delta_tau_all_clouds_upper[b];
        }
}

/* NB the reference accumulates INTO trans_weight_band (+=) without zeroing it (:2978, :3015) */
void orc_calc_contr_func_iso(const double* trans_wg, double* trans_weight_band,
                             double* contr_func_band, const double* gauss_weight,
                             const double* planckband_lay, double epsi, int nbin, int nlayer, int ny) {
    const size_t sl = (size_t)ny * nbin;
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int i = 0; i < nlayer; i++) {
            size_t b = x + (size_t)nbin * i;
            for (int y = 0; y < ny; y++) {
                size_t c = (size_t)y + (size_t)ny * x;
                double to_top = 1.0;
                for (int j = i + 1; j < nlayer; j++) to_top = to_top * trans_wg[c + sl * j];
                trans_weight_band[b] += 0.5 * gauss_weight[y] * (1.0 - trans_wg[c + sl * i]) * to_top;
            }
            contr_func_band[b] =
                2.0 * PI * epsi * planckband_lay[i + (size_t)x * (nlayer + 2)] * trans_weight_band[b];
        }
}

void orc_calc_contr_func_noniso(const double* trans_wg_upper, const double* trans_wg_lower,
                                double* trans_weight_band, double* contr_func_band,
                                const double* gauss_weight, const double* planckband_lay, double epsi,
                                int nbin, int nlayer, int ny) {
    const size_t sl = (size_t)ny * nbin;
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int i = 0; i < nlayer; i++) {
            size_t b = x + (size_t)nbin * i;
            for (int y = 0; y < ny; y++) {
                size_t c = (size_t)y + (size_t)ny * x;
                double to_top = 1.0;
                for (int j = i + 1; j < nlayer; j++)
                    to_top = to_top * trans_wg_upper[c + sl * j] * trans_wg_lower[c + sl * j];
                trans_weight_band[b] += 0.5 * gauss_weight[y] *
                                        (1.0 - trans_wg_upper[c + sl * i] * trans_wg_lower[c + sl * i]) *
                                        to_top;
            }
            contr_func_band[b] =
                2.0 * PI * epsi * planckband_lay[i + (size_t)x * (nlayer + 2)] * trans_weight_band[b];
        }
}

/* source/kernels.cu:294-329 */
static double dB_dT(double lambda, double T) {
    double D = 2.0 * HCONST * ipow(CSPEED, 3) * HCONST / (ipow(lambda, 6) * KBOLTZMANN * (T * T));
    double num = exp(HCONST * CSPEED / (lambda * KBOLTZMANN * T));
    double den = (exp(HCONST * CSPEED / (lambda * KBOLTZMANN * T)) - 1.0) *
                 (exp(HCONST * CSPEED / (lambda * KBOLTZMANN * T)) - 1.0);
    return D * num / den;
}
static double integrated_dB_dT(const double* kw, const double* ky, int ny, double lb, double lt,
                               double T) {
    double r = 0;
    for (int y = 0; y < ny; y++) {
        double xx = (ky[y] - 0.5) * 2.0;
        double arg = (lt - lb) / 2.0 * xx + (lt + lb) / 2.0;
        r += (lt - lb) / 2.0 * kw[y] * dB_dT(arg, T);
    }
    return r;
}

/* source/kernels.cu:3024-3115 */
void orc_calc_mean_opacities(double* planck_opac_T_pl, double* ross_opac_T_pl,
                             double* planck_opac_T_star, double* ross_opac_T_star,
                             const double* opac_wg_lay, const double* abs_cross_all_clouds_lay,
                             const double* meanmolmass_lay, const double* planckband_lay,
                             const double* opac_interwave, const double* opac_deltawave,
                             const double* T_lay, const double* gauss_weight, const double* gauss_y,
                             double* opac_band_lay, int nlayer, int nbin, int ny, double T_star) {
#pragma omp parallel for
    for (int i = 0; i < nlayer; i++) {
        double npl = 0, dpl = 0, nrl = 0, drl = 0, nps = 0, dps = 0, nrs = 0, drs = 0;
        for (int x = 0; x < nbin; x++) {
            double ob = 0;
            for (int y = 0; y < ny; y++)
                ob += 0.5 * gauss_weight[y] * opac_wg_lay[(size_t)y + (size_t)ny * x + (size_t)ny * nbin * i];
            opac_band_lay[x + (size_t)nbin * i] = ob;
        }
        for (int x = 0; x < nbin; x++) {
            size_t b = x + (size_t)nbin * i;
            double ext = opac_band_lay[b] + abs_cross_all_clouds_lay[b] / meanmolmass_lay[i];
            double Bp = planckband_lay[i + (size_t)x * (nlayer + 2)];
            double Bs = planckband_lay[nlayer + (size_t)x * (nlayer + 2)];
            npl += ext * Bp * opac_deltawave[x];
            dpl += Bp * opac_deltawave[x];
            double dbp = integrated_dB_dT(gauss_weight, gauss_y, ny, opac_interwave[x],
                                          opac_interwave[x + 1], T_lay[i]);
            nrl += dbp;
            if (ext > 0) drl += dbp / ext;
            nps += ext * Bs * opac_deltawave[x];
            dps += Bs * opac_deltawave[x];
            double dbs = integrated_dB_dT(gauss_weight, gauss_y, ny, opac_interwave[x],
                                          opac_interwave[x + 1], T_star);
            nrs += dbs;
            if (ext > 0) drs += dbs / ext;
        }
        planck_opac_T_pl[i] = npl / dpl;
        ross_opac_T_pl[i] = T_lay[i] < 70 ? -3 : nrl / drl;
        planck_opac_T_star[i] = T_star < 70 ? -3 : nps / dps;
        ross_opac_T_star[i] = T_star < 70 ? -3 : nrs / drs;
    }
}

/* source/kernels.cu:3119-3139 */
void orc_integrate_beamflux(double* F_dir_tot, const double* F_dir_band, const double* deltalambda,
                            int nbin, int ninterface) {
    for (int i = 0; i < ninterface; i++) {
        double s = 0;
        for (int x = 0; x < nbin; x++) s += F_dir_band[x + (size_t)nbin * i] * deltalambda[x];
        F_dir_tot[i] = s;
    }
}

/* ============================================================================================
 * Flux solve as one tridiagonal system (Thomas algorithm)      source/kernels.cu:1803-2424
 * Optional method of the reference (`flux calculation method = matrix`).  Unknowns, non-isothermal:
 * x = [F_dn[0], F_up[0], Fc_dn[0], Fc_up[0], F_dn[1], ...], n = 4*ninterface - 2; isothermal:
 * x = [F_dn[0], F_up[0], F_dn[1], ...], n = 2*ninterface.  Spectral points whose
 * scat_trigger is 0 take a pure-absorption sweep instead.
 * ============================================================================================ */
static void thomas_solve(int n, double albedo, double src_boa, double src_toa, const double* alpha,
                         const double* beta, const double* s_down, const double* s_up, size_t stride,
                         double* c_prime, double* d_prime) {
    /* rows: 0 = BOA; odd r = 2j+1: down equation of slab j; even r = 2j+2: up equation of slab j;
     * n-1 = TOA.  The sub-diagonal of a row equals the super-diagonal of the previous one. */
    double b = -albedo, c = 1.0, d = src_boa;
    c_prime[0] = c / b;
    d_prime[0] = d / b;
    for (int i = 1; i < n - 1; i++) {
        const double c_prev = c;
        if (i % 2 == 0) {
            const size_t j = (size_t)(i / 2 - 1) * stride;
            b = -beta[j];
            c = 1.0;
            d = s_up[j];
        } else {
            const size_t j = (size_t)((i - 1) / 2) * stride;
            b = -beta[j];
            c = -alpha[j];
            d = s_down[j];
        }
        const double den = b - c_prev * c_prime[(size_t)(i - 1) * stride];
        c_prime[(size_t)i * stride] = c / den;
        d_prime[(size_t)i * stride] = (d - c_prev * d_prime[(size_t)(i - 1) * stride]) / den;
    }
    d_prime[(size_t)(n - 1) * stride] = (src_toa - c * d_prime[(size_t)(n - 2) * stride]) /
                                        (0.0 - c * c_prime[(size_t)(n - 2) * stride]);
}

void orc_fband_matrix_iso(double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                          const double* planckband_lay, const double* w_0, const double* M_term,
                          const double* N_term, const double* P_term, const double* G_plus,
                          const double* G_minus, const double* g_0_tot_lay, double* alpha, double* beta,
                          double* source_term_down, double* source_term_up, double* c_prime,
                          double* d_prime, const int* scat_trigger, const double* trans_wg,
                          const double* surf_albedo, double g_0, double Rstar, double a, int ninterface,
                          int nbin, double f_factor, double mu_star, int ny, double epsi, int dir_beam,
                          int clouds, int scat_corr, double i2s_transition) {
    const size_t sl = (size_t)ny * nbin;
    const int nl = ninterface - 1, npl = nl + 2;
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int y = 0; y < ny; y++) {
            const size_t c = (size_t)y + (size_t)ny * x;
            const double* B = planckband_lay + (size_t)x * npl;
            const double src_toa = (1.0 - dir_beam) * f_factor * ((Rstar / a) * (Rstar / a)) * PI * B[nl];
            if (scat_trigger[c] == 1) {
                double E = 1.0, g0 = g_0; /* both persist across layers exactly as in the reference */
                for (int j = 0; j < nl; j++) {
                    const size_t k = c + sl * j;
                    const double M = M_term[k], N = N_term[k], P = P_term[k], w0 = w_0[k];
                    const double Gm = G_minus[k], Gp = G_plus[k];
                    if (clouds == 1) g0 = g_0_tot_lay[x + (size_t)nbin * j];
                    if (scat_corr == 1) E = E_param(w0, g0, i2s_transition);
                    alpha[k] = P / M;
                    beta[k] = -N / M;
                    const double planck = 2.0 * PI * epsi * (1.0 - w0) / (E - w0) * (N + M - P) * B[j];
                    double dd = F_dir_wg[k] / (-mu_star) * (Gm * M + Gp * N) - F_dir_wg[k + sl] / (-mu_star) * P * Gm;
                    double du = F_dir_wg[k + sl] / (-mu_star) * (Gm * N + Gp * M) - F_dir_wg[k] / (-mu_star) * P * Gp;
                    dd = dmin(0.0, dd);
                    du = dmin(0.0, du);
                    source_term_down[k] = 1.0 / M * (planck + dd);
                    source_term_up[k] = 1.0 / M * (planck + du);
                }
                const double w0 = w_0[c];
                if (clouds == 1) g0 = g_0_tot_lay[x];
                if (scat_corr == 1) E = E_param(w0, g0, i2s_transition);
                const double src_boa = surf_albedo[x] * F_dir_wg[c] +
                                       (1.0 - surf_albedo[x]) * PI * (1.0 - w0) / (E - w0) * B[ninterface];
                const int n = 2 * ninterface;
                thomas_solve(n, surf_albedo[x], src_boa, src_toa, alpha + c, beta + c, source_term_down + c,
                             source_term_up + c, sl, c_prime + c, d_prime + c);
                double xi = d_prime[c + sl * (n - 1)];
                F_up_wg[c + sl * ((n - 2) / 2)] = xi;
                for (int i = n - 2; i >= 0; i--) {
                    xi = d_prime[c + sl * i] - c_prime[c + sl * i] * xi;
                    if (i % 2 == 0)
                        F_down_wg[c + sl * (i / 2)] = xi;
                    else
                        F_up_wg[c + sl * ((i - 1) / 2)] = xi;
                }
            } else {
                F_down_wg[c + sl * nl] = src_toa;
                for (int i = nl - 1; i >= 0; i--) {
                    const double t = trans_wg[c + sl * i];
                    F_down_wg[c + sl * i] =
                        tiny_abs(t * F_down_wg[c + sl * (i + 1)] + 2.0 * PI * epsi * (1.0 - t) * B[i]);
                }
                F_up_wg[c] = surf_albedo[x] * (F_dir_wg[c] + F_down_wg[c]) + (1.0 - surf_albedo[x]) * PI * B[ninterface];
                for (int i = 1; i < ninterface; i++) {
                    const double t = trans_wg[c + sl * (i - 1)];
                    F_up_wg[c + sl * i] =
                        tiny_abs(t * F_up_wg[c + sl * (i - 1)] + 2.0 * PI * epsi * (1.0 - t) * B[i - 1]);
                }
            }
        }
}

void orc_fband_matrix_noniso(
    double* F_down_wg, double* F_up_wg, double* Fc_down_wg, double* Fc_up_wg, const double* F_dir_wg,
    const double* Fc_dir_wg, const double* planckband_lay, const double* planckband_int,
    const double* w_0_upper, const double* w_0_lower, const double* delta_tau_wg_upper,
    const double* delta_tau_wg_lower, const double* delta_tau_all_clouds_upper,
    const double* delta_tau_all_clouds_lower, const double* M_upper, const double* M_lower,
    const double* N_upper, const double* N_lower, const double* P_upper, const double* P_lower,
    const double* G_plus_upper, const double* G_plus_lower, const double* G_minus_upper,
    const double* G_minus_lower, const double* g_0_tot_lay, const double* g_0_tot_int, double* alpha,
    double* beta, double* source_term_down, double* source_term_up, double* c_prime, double* d_prime,
    const int* scat_trigger, const double* trans_wg_upper, const double* trans_wg_lower,
    const double* surf_albedo, double g_0, double Rstar, double a, int ninterface, int nbin,
    double f_factor, double mu_star, int ny, double epsi, double delta_tau_limit, int dir_beam, int clouds,
    int scat_corr, double i2s_transition) {
    const size_t sl = (size_t)ny * nbin;
    const int nl = ninterface - 1, npl = nl + 2;
#pragma omp parallel for
    for (int x = 0; x < nbin; x++)
        for (int y = 0; y < ny; y++) {
            const size_t c = (size_t)y + (size_t)ny * x;
            const double* Bl = planckband_lay + (size_t)x * npl;
            const double* Bi = planckband_int + (size_t)x * ninterface;
            const double src_toa = (1.0 - dir_beam) * f_factor * ((Rstar / a) * (Rstar / a)) * PI * Bl[nl];
            if (scat_trigger[c] == 1) {
                double E = 1.0, g0 = g_0;
                for (int j = 0; j < 2 * nl; j++) { /* slab j: even = lower half of layer j/2, odd = upper half */
                    const int i = j / 2;
                    const int lower = (j % 2 == 0);
                    const size_t k = c + sl * i, b = x + (size_t)nbin * i;
                    double M, N, P, w0, Gm, Gp, dtau, pd, pu, dd, du;
                    if (lower) {
                        M = M_lower[k]; N = N_lower[k]; P = P_lower[k]; w0 = w_0_lower[k];
                        Gm = G_minus_lower[k]; Gp = G_plus_lower[k];
                        dtau = delta_tau_wg_lower[k] + delta_tau_all_clouds_lower[b];
                        if (clouds == 1) g0 = (g_0_tot_int[b] + g_0_tot_lay[b]) / 2.0;
                    } else {
                        M = M_upper[k]; N = N_upper[k]; P = P_upper[k]; w0 = w_0_upper[k];
                        Gm = G_minus_upper[k]; Gp = G_plus_upper[k];
                        dtau = delta_tau_wg_upper[k] + delta_tau_all_clouds_upper[b];
                        if (clouds == 1) g0 = (g_0_tot_int[b + nbin] + g_0_tot_lay[b]) / 2.0;
                    }
                    if (scat_corr == 1) E = E_param(w0, g0, i2s_transition);
                    /* B_bot / B_top: Planck function at the bottom / top node of the slab */
                    const double B_bot = lower ? Bi[i] : Bl[i], B_top = lower ? Bl[i] : Bi[i + 1];
                    if (dtau < delta_tau_limit) {
                        pu = lower ? (N + M - P) * (Bi[i] + Bl[i]) / 2.0 : (N + M - P) * (Bl[i] + Bi[i + 1]) / 2.0;
                        pd = pu;
                    } else {
                        const double pgrad = (B_bot - B_top) / dtau;
                        pd = (M + N) * B_bot - P * B_top + epsi / (E * (1.0 - w0 * g0)) * (P - M + N) * pgrad;
                        pu = (M + N) * B_top - P * B_bot + epsi / (E * (1.0 - w0 * g0)) * (M - N - P) * pgrad;
                    }
                    const double F_bot = lower ? F_dir_wg[k] : Fc_dir_wg[k];
                    const double F_top = lower ? Fc_dir_wg[k] : F_dir_wg[k + sl];
                    dd = F_bot / (-mu_star) * (Gm * M + Gp * N) - F_top / (-mu_star) * P * Gm;
                    du = F_top / (-mu_star) * (Gm * N + Gp * M) - F_bot / (-mu_star) * P * Gp;
                    dd = dmin(0.0, dd);
                    du = dmin(0.0, du);
                    const size_t kj = c + sl * j;
                    alpha[kj] = P / M;
                    beta[kj] = -N / M;
                    source_term_down[kj] = 1.0 / M * (2.0 * PI * epsi * (1.0 - w0) / (E - w0) * pd + dd);
                    source_term_up[kj] = 1.0 / M * (2.0 * PI * epsi * (1.0 - w0) / (E - w0) * pu + du);
                }
                const double w0 = w_0_lower[c];
                if (clouds == 1) g0 = (g_0_tot_int[x] + g_0_tot_lay[x]) / 2.0;
                if (scat_corr == 1) E = E_param(w0, g0, i2s_transition);
                const double src_boa = surf_albedo[x] * F_dir_wg[c] +
                                       (1.0 - surf_albedo[x]) * PI * (1.0 - w0) / (E - w0) * Bl[ninterface];
                const int n = 4 * ninterface - 2;
                thomas_solve(n, surf_albedo[x], src_boa, src_toa, alpha + c, beta + c, source_term_down + c,
                             source_term_up + c, sl, c_prime + c, d_prime + c);
                double xi = d_prime[c + sl * (n - 1)];
                F_up_wg[c + sl * (ninterface - 1)] = xi;
                for (int i = n - 2; i >= 0; i--) {
                    xi = d_prime[c + sl * i] - c_prime[c + sl * i] * xi;
                    if (xi < 1e-100) xi = fabs(xi);
                    switch (i % 4) {
                        case 0: F_down_wg[c + sl * (i / 4)] = xi; break;
                        case 1: F_up_wg[c + sl * ((i - 1) / 4)] = xi; break;
                        case 2: Fc_down_wg[c + sl * ((i - 2) / 4)] = xi; break;
                        default: Fc_up_wg[c + sl * ((i - 3) / 4)] = xi; break;
                    }
                }
            } else {
                /* pure absorption, SURVEY.md 10.3 last paragraph (kernels.cu:2286-2421) */
                F_down_wg[c + sl * nl] = src_toa;
                for (int i = nl - 1; i >= 0; i--) {
                    const size_t k = c + sl * i, b = x + (size_t)nbin * i;
                    const double tu = trans_wg_upper[k], tl = trans_wg_lower[k];
                    const double du_ = delta_tau_wg_upper[k] + delta_tau_all_clouds_upper[b];
                    const double dl_ = delta_tau_wg_lower[k] + delta_tau_all_clouds_lower[b];
                    double pt;
                    if (du_ < delta_tau_limit) pt = (Bi[i + 1] + Bl[i]) / 2.0 * (1.0 - tu);
                    else pt = Bl[i] - tu * Bi[i + 1] + epsi * (tu - 1.0) * ((Bl[i] - Bi[i + 1]) / du_);
                    Fc_down_wg[k] = tiny_abs(tu * F_down_wg[k + sl] + 2.0 * PI * epsi * pt);
                    if (dl_ < delta_tau_limit) pt = (Bi[i] + Bl[i]) / 2.0 * (1.0 - tl);
                    else pt = Bi[i] - tl * Bl[i] + epsi * (tl - 1.0) * ((Bi[i] - Bl[i]) / dl_);
                    F_down_wg[k] = tiny_abs(tl * Fc_down_wg[k] + 2.0 * PI * epsi * pt);
                }
                F_up_wg[c] = surf_albedo[x] * (F_dir_wg[c] + F_down_wg[c]) + (1.0 - surf_albedo[x]) * PI * Bl[ninterface];
                for (int i = 1; i < ninterface; i++) {
                    const size_t k = c + sl * (i - 1), b = x + (size_t)nbin * (i - 1);
                    const double tu = trans_wg_upper[k], tl = trans_wg_lower[k];
                    const double du_ = delta_tau_wg_upper[k] + delta_tau_all_clouds_upper[b];
                    const double dl_ = delta_tau_wg_lower[k] + delta_tau_all_clouds_lower[b];
                    double pt;
                    if (dl_ < delta_tau_limit) pt = (Bi[i - 1] + Bl[i - 1]) / 2.0 * (1.0 - tl);
                    else pt = Bl[i - 1] - tl * Bi[i - 1] + epsi * ((Bi[i - 1] - Bl[i - 1]) / dl_) * (1.0 - tl);
                    Fc_up_wg[k] = tl * F_up_wg[k] + 2.0 * PI * epsi * pt;
                    /* the reference patches entry i, not i-1 (:2394); at i = nlayer that is past the end
                     * of the array, so the restatement stops one short */
                    if (i < nl) Fc_up_wg[k + sl] = tiny_abs(Fc_up_wg[k + sl]);
                    if (du_ < delta_tau_limit) pt = (Bi[i] + Bl[i - 1]) / 2.0 * (1.0 - tu);
                    else pt = Bi[i] - tu * Bl[i - 1] + epsi * ((Bl[i - 1] - Bi[i]) / du_) * (1.0 - tu);
                    F_up_wg[k + sl] = tiny_abs(tu * Fc_up_wg[k] + 2.0 * PI * epsi * pt);
                }
            }
        }
}
