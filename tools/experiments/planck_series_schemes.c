// Round 5 experiment: how far do cheaper evaluation schemes of the Planck series move the table?
// gcc -O2 -ffp-contract=off -mfma planck_series_schemes.c -lm && ./a.out NBIN DIM STEP ROWSTRIDE
// noise experiment for the Planck table (CPU): present formula vs cheaper evaluation schemes
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#define KB 1.380649e-16
#define HC 6.62607015e-27
#define CS 29979245800.0
static double term(int n, double y1, double y2) {
    const double dn = n;
    return exp(-dn * y2) * ((y2 * y2 * y2) / dn + 3.0 * (y2 * y2) / (dn * dn) + 6.0 * y2 / (dn * dn * dn) + 6.0 / (dn * dn * dn * dn)) -
           exp(-dn * y1) * ((y1 * y1 * y1) / dn + 3.0 * (y1 * y1) / (dn * dn) + 6.0 * y1 / (dn * dn * dn) + 6.0 / (dn * dn * dn * dn));
}
static double A(double yb, double yt, double D) {
    double acc = 0;
    for (int n = 1; n < 200; n++) acc += D * term(n, yb, yt);
    return acc;
}
static double rc1[200], rc2[200], rc3[200], rc4[200], d1[200], d2[200], d3[200], d4[200];
static inline double mdiv(double a, double d, double r) {  // Markstein: correctly rounded a / d from r = RN(1 / d)
    double q = a * r;
    double rem = fma(-q, d, a);
    return fma(rem, r, q);
}
// phi_n(y) for all n by scheme; mode 0: libm exp per term, 1: double-double powers, 2: recurrence re-anchored every 8
static void phis(double y, double* out, int mode) {
    const double y2 = y * y, y3 = y2 * y, a2 = 3.0 * y2, a1 = 6.0 * y;
    double eh = exp(-y), el = 0.0, Eh = 0, El = 0;
    if (mode == 1) {  // low part of exp(-y): exp(-y) = eh + el, el from a Newton step on log: el = eh * (-(y + log(eh)))
        // log(eh) is not exact either; use expm1 of the residual in long double for the experiment's sake
        long double t = expl(-(long double)y);
        el = (double)(t - (long double)eh);
    }
    double ph = 1.0, pl = 0.0;
    for (int n = 1; n < 200; n++) {
        double e;
        if (mode == 0) e = exp(-(double)n * y);
        else if (mode == 1) {
            // (ph, pl) *= (eh, el)
            double p = ph * eh;
            double err = fma(ph, eh, -p);
            err += ph * el + pl * eh;
            ph = p + err;
            pl = err - (ph - p);
            e = ph;
        } else {
            if ((n & 7) == 1) Eh = exp(-(double)n * y); else Eh = Eh * eh;
            e = Eh;
        }
        const double poly = mdiv(y3, d1[n], rc1[n]) + mdiv(a2, d2[n], rc2[n]) + mdiv(a1, d3[n], rc3[n]) + mdiv(6.0, d4[n], rc4[n]);
        out[n] = e * poly;
    }
    (void)El;
}
int main(int argc, char** argv) {
    int nbin = atoi(argv[1]), dim = atoi(argv[2]), step = atoi(argv[3]), rstride = atoi(argv[4]);
    for (int n = 1; n < 200; n++) {
        double dn = n;
        d1[n] = dn; d2[n] = dn * dn; d3[n] = dn * dn * dn; d4[n] = dn * dn * dn * dn;
        rc1[n] = 1.0 / d1[n]; rc2[n] = 1.0 / d2[n]; rc3[n] = 1.0 / d3[n]; rc4[n] = 1.0 / d4[n];
    }
    double* edge = malloc((nbin + 1) * sizeof(double));
    for (int x = 0; x <= nbin; x++) edge[x] = 0.3e-4 * pow(500.0 / 0.3, (double)x / nbin);
    double worst[3] = {0, 0, 0};
    long ndiff[3] = {0, 0, 0}, ntot = 0;
    double (*ph)[200] = malloc((nbin + 1) * sizeof(*ph));
    for (int r = 0; r < dim; r += rstride) {
        double T = r * step + 1;
        const double kh = KB / HC;
        const double D = 2.0 * (kh * kh * kh * KB * (T * T * T * T)) / (CS * CS);
        for (int mode = 0; mode < 3; mode++) {
            for (int x = 0; x <= nbin; x++) phis(HC * CS / (edge[x] * KB * T), ph[x], mode);
            for (int x = 0; x < nbin; x++) {
                double ytop = HC * CS / (edge[x + 1] * KB * T), ybot = HC * CS / (edge[x] * KB * T);
                double ref = A(ybot, ytop, D);
                double acc = 0;
                for (int n = 1; n < 200; n++) acc += D * (ph[x + 1][n] - ph[x][n]);
                if (mode == 0) ntot++;
                if (acc != ref) ndiff[mode]++;
                double rel = fabs(acc - ref) / (fabs(ref) + 1e-290);
                if (rel > worst[mode]) worst[mode] = rel;
            }
        }
    }
    printf("nbin %d dim %d step %d: entries %ld; libm-exp+markstein: differ %ld worst %.3e | dd-power: differ %ld worst %.3e | anchor8: differ %ld worst %.3e\n",
           nbin, dim, step, ntot, ndiff[0], worst[0], ndiff[1], worst[1], ndiff[2], worst[2]);
    return 0;
}
