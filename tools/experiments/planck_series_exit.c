// Round 5 experiment: does leaving the Planck series early change a bit?  CPU replay of k_plancktable.
// gcc -O2 -ffp-contract=off planck_series_exit.c -lm && ./a.out NBIN DIM STEP ROWSTRIDE
// does the early exit change a bit?  CPU replay of the kernel's arithmetic against the full 199-term sum
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#define KB 1.380649e-16
#define HC 6.62607015e-27
#define CS 29979245800.0
static double phi(int n, double y) { const double dn = n; return exp(-dn * y) * ((y * y * y) / dn + 3.0 * (y * y) / (dn * dn) + 6.0 * y / (dn * dn * dn) + 6.0 / (dn * dn * dn * dn)); }
int main(int argc, char** argv) {
    int nbin = atoi(argv[1]), dim = atoi(argv[2]), step = atoi(argv[3]), rstride = atoi(argv[4]);
    double* edge = malloc((nbin + 1) * sizeof(double));
    for (int x = 0; x <= nbin; x++) edge[x] = 0.3e-4 * pow(500.0 / 0.3, (double)x / nbin);
    long ndiff = 0, ntot = 0; double terms = 0;
    for (int r = 0; r < dim; r += rstride) {
        double T = r * step + 1;
        const double kh = KB / HC, D = 2.0 * (kh * kh * kh * KB * (T * T * T * T)) / (CS * CS);
        for (int x0 = 0; x0 < nbin; x0 += 63) {   // one wavefront: bins x0 .. x0+62
            int nb = nbin - x0 < 63 ? nbin - x0 : 63;
            double full[63], acc[63]; int done = 0, nterm = 199;
            for (int b = 0; b < nb; b++) { full[b] = 0; acc[b] = 0; }
            for (int n = 1; n < 200; n++) {
                int live = 0;
                for (int b = 0; b < nb; b++) {
                    double yt = HC * CS / (edge[x0 + b + 1] * KB * T), yb = HC * CS / (edge[x0 + b] * KB * T);
                    double pt = phi(n, yt), pb = phi(n, yb), d = pt - pb;
                    full[b] += D * d;
                    if (!done) { acc[b] += D * d; if (!(D * (fabs(d) + 0x1p-47 * fmax(pt, pb)) < 0x1p-55 * fabs(acc[b]))) live = 1; }
                }
                if (!done && !live) { done = 1; nterm = n; }
            }
            terms += nterm; ntot++;
            for (int b = 0; b < nb; b++) if (full[b] != acc[b]) ndiff++;
        }
    }
    printf("nbin %d: %ld wavefront-rows, %ld entries differ, mean terms %.1f of 199\n", nbin, ntot, ndiff, terms / ntot);
    return 0;
}
