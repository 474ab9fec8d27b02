// Per-stage entry points, part 3b: the flux solve as one tridiagonal system per spectral point
// (`flux calculation method = matrix`; reference kernels.cu:1803-2424, launched from
// computation.py:625-710).  One thread owns one spectral point c = y + ny*x: it assembles the 2-per-
// slab coefficients into the caller's work arrays, runs the Thomas elimination bottom-up and
// back-substitutes top-down.  All work arrays are strided by ny*nbin, so a wavefront's accesses are
// contiguous.  Points whose scat_trigger is 0 take the pure-absorption sweep.
#include "two_stream.h"

using namespace hx;

namespace {

struct Thomas {
    double* c_prime;
    double* d_prime;
    size_t c, nc;
    double sup;  // super-diagonal of the previous row == sub-diagonal of the current one
    int row;
    double cp, dp;  // c', d' of the previous row (kept in registers: read back from the work arrays they were two
                    // dependent memory round trips per row)

    // row 0 (BOA): -A x0 + x1 = src
    __device__ void first(double albedo, double src) {
        sup = 1.0;
        cp = sup / (-albedo);
        dp = src / (-albedo);
        c_prime[c] = cp;
        d_prime[c] = dp;
        row = 1;
    }
    __device__ void push(double b, double sup_new, double d) {
        const size_t k = c + nc * row;
        const double den = b - sup * cp;
        const double cn = sup_new / den, dn = (d - sup * dp) / den;
        c_prime[k] = cn;
        d_prime[k] = dn;
        cp = cn;
        dp = dn;
        sup = sup_new;
        row++;
    }
    // last row (TOA): sub-diagonal only
    __device__ double last(double src) {
        const size_t k = c + nc * row;
        const double x = (src - sup * dp) / (0.0 - sup * cp);
        d_prime[k] = x;
        return x;
    }
};

// Back-substitution x[i] = d'[i] - c'[i] x[i+1] for i = top ... 0: the rows' coefficients are requested eight at a time
// (one at a time, every step waited for its own two loads); `emit(i, x)` stores x[i] where it belongs.
template <bool ABS_TINY, class Emit>
__device__ __forceinline__ void back_substitute(const double* __restrict__ c_prime, const double* __restrict__ d_prime,
                                                size_t c, size_t nc, int top, double xi, Emit emit) {
    int i = top;
    while (i >= 0) {
        double cpv[8], dpv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = max(i - u, 0);
            cpv[u] = c_prime[c + nc * r];
            dpv[u] = d_prime[c + nc * r];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (i - u < 0) break;
            xi = dpv[u] - cpv[u] * xi;
            if (ABS_TINY && xi < 1e-100) xi = fabs(xi);
            emit(i - u, xi);
        }
        i -= 8;
    }
}

// LEAN: the call of the device-resident loop -- the work arrays alpha ... s_up, which nothing reads, are not written, and
// without a direct beam its (zero) arrays and G+- are not read: traffic, the kernel runs at the memory system's rate
template <bool LEAN>
__global__ void __launch_bounds__(256)
k_fband_matrix_iso(double* __restrict__ F_down, double* __restrict__ F_up,
                   const double* __restrict__ F_dir, const double* __restrict__ planckband_lay,
                   const double* __restrict__ w_0, const double* __restrict__ M_term,
                   const double* __restrict__ N_term, const double* __restrict__ P_term,
                   const double* __restrict__ G_plus, const double* __restrict__ G_minus,
                   const double* __restrict__ g_0_tot_lay, double* __restrict__ alpha,
                   double* __restrict__ beta, double* __restrict__ s_down, double* __restrict__ s_up,
                   double* __restrict__ c_prime, double* __restrict__ d_prime,
                   const int* __restrict__ scat_trigger, const double* __restrict__ trans,
                   const double* __restrict__ surf_albedo, double g_0, double Rstar, double a, int ni,
                   int nbin, double f_factor, double mu_star, int ny, double epsi, int dir_beam,
                   int clouds, int scat_corr, double i2s, const int* __restrict__ skip) {
    const size_t nc = (size_t)ny * nbin;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc || (skip && *skip)) return;
    const int x = (int)(c / ny), nl = ni - 1;
    const double* B = planckband_lay + (size_t)x * (ni + 1);
    const double A = surf_albedo[x];
    const double src_toa = (1.0 - dir_beam) * f_factor * ((Rstar / a) * (Rstar / a)) * HX_PI * B[nl];
    if (scat_trigger[c] == 1) {
        Thomas th = {c_prime, d_prime, c, nc, 0.0, 0};
        {
            const double w0 = w_0[c];
            const double g0 = clouds == 1 ? g_0_tot_lay[x] : g_0;
            const double E = E_factor(w0, g0, scat_corr, i2s);
            th.first(A, A * F_dir[c] + (1.0 - A) * HX_PI * (1.0 - w0) / (E - w0) * B[ni]);
        }
        double Fdir_bot = F_dir[c];
        for (int j = 0; j < nl; j++) {
            const size_t k = c + nc * j;
            const double M = M_term[k], N = N_term[k], P = P_term[k], w0 = w_0[k];
            const double g0 = clouds == 1 ? g_0_tot_lay[x + (size_t)nbin * j] : g_0;
            const double E = E_factor(w0, g0, scat_corr, i2s);
            const double Fdir_top = (!LEAN || dir_beam == 1) ? F_dir[k + nc] : 0.0;
            const double al = P / M, be = -N / M;
            const double planck = 2.0 * HX_PI * epsi * (1.0 - w0) / (E - w0) * (N + M - P) * B[j];
            double dd = 0.0, du = 0.0;
            if (!LEAN || dir_beam == 1) {
                const double Gm = G_minus[k], Gp = G_plus[k];
                dd = dmin(0.0, Fdir_bot / (-mu_star) * (Gm * M + Gp * N) - Fdir_top / (-mu_star) * P * Gm);
                du = dmin(0.0, Fdir_top / (-mu_star) * (Gm * N + Gp * M) - Fdir_bot / (-mu_star) * P * Gp);
            }
            const double sd = 1.0 / M * (planck + dd), su = 1.0 / M * (planck + du);
            if (!LEAN) {
                alpha[k] = al;
                beta[k] = be;
                s_down[k] = sd;
                s_up[k] = su;
            }
            th.push(-be, -al, sd);  // down equation of slab j
            th.push(-be, 1.0, su);  // up equation of slab j
            Fdir_bot = Fdir_top;
        }
        double xi = th.last(src_toa);
        F_up[c + nc * nl] = xi;
        back_substitute<false>(c_prime, d_prime, c, nc, 2 * ni - 2, xi, [&](int i, double x) {
            if (i % 2 == 0)
                F_down[c + nc * (i / 2)] = x;
            else
                F_up[c + nc * ((i - 1) / 2)] = x;
        });
    } else {
        double Fd = src_toa;
        F_down[c + nc * nl] = Fd;
        for (int i = nl - 1; i >= 0; i--) {
            const double t = trans[c + nc * i];
            Fd = tiny_abs(t * Fd + 2.0 * HX_PI * epsi * (1.0 - t) * B[i]);
            F_down[c + nc * i] = Fd;
        }
        double Fu = A * (F_dir[c] + Fd) + (1.0 - A) * HX_PI * B[ni];
        F_up[c] = Fu;
        for (int i = 1; i < ni; i++) {
            const double t = trans[c + nc * (i - 1)];
            Fu = tiny_abs(t * Fu + 2.0 * HX_PI * epsi * (1.0 - t) * B[i - 1]);
            F_up[c + nc * i] = Fu;
        }
    }
}

struct MatrixNoniso {
    const double *w0u, *w0l, *dtu, *dtl, *dcu, *dcl, *Mu, *Ml, *Nu, *Nl, *Pu, *Pl, *Gpu, *Gpl, *Gmu, *Gml;
    const double *g0_lay, *g0_int, *trans_u, *trans_l;
    double *alpha, *beta, *s_down, *s_up, *c_prime, *d_prime;
};

template <bool LEAN>
__global__ void __launch_bounds__(256)
k_fband_matrix_noniso(double* __restrict__ F_down, double* __restrict__ F_up,
                      double* __restrict__ Fc_down, double* __restrict__ Fc_up,
                      const double* __restrict__ F_dir, const double* __restrict__ Fc_dir,
                      const double* __restrict__ planckband_lay,
                      const double* __restrict__ planckband_int, MatrixNoniso q,
                      const int* __restrict__ scat_trigger, const double* __restrict__ surf_albedo,
                      double g_0, double Rstar, double a, int ni, int nbin, double f_factor,
                      double mu_star, int ny, double epsi, double dtau_limit, int dir_beam, int clouds,
                      int scat_corr, double i2s, const int* __restrict__ skip) {
    const size_t nc = (size_t)ny * nbin;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc || (skip && *skip)) return;
    const int x = (int)(c / ny), nl = ni - 1;
    const double* Bl = planckband_lay + (size_t)x * (ni + 1);
    const double* Bi = planckband_int + (size_t)x * ni;
    const double A = surf_albedo[x];
    const double src_toa = (1.0 - dir_beam) * f_factor * ((Rstar / a) * (Rstar / a)) * HX_PI * Bl[nl];
    if (scat_trigger[c] == 1) {
        Thomas th = {q.c_prime, q.d_prime, c, nc, 0.0, 0};
        {
            const double w0 = q.w0l[c];
            const double g0 = clouds == 1 ? (q.g0_int[x] + q.g0_lay[x]) / 2.0 : g_0;
            const double E = E_factor(w0, g0, scat_corr, i2s);
            th.first(A, A * F_dir[c] + (1.0 - A) * HX_PI * (1.0 - w0) / (E - w0) * Bl[ni]);
        }
        for (int j = 0; j < 2 * nl; j++) {  // slab j: even = lower half of layer j/2, odd = upper half
            const int i = j >> 1;
            const bool lower = (j & 1) == 0;
            const size_t k = c + nc * i, b = x + (size_t)nbin * i;
            const double M = lower ? q.Ml[k] : q.Mu[k], N = lower ? q.Nl[k] : q.Nu[k];
            const double P = lower ? q.Pl[k] : q.Pu[k], w0 = lower ? q.w0l[k] : q.w0u[k];
            const double dtau = lower ? q.dtl[k] + q.dcl[b] : q.dtu[k] + q.dcu[b];
            double g0 = g_0;
            if (clouds == 1) g0 = ((lower ? q.g0_int[b] : q.g0_int[b + nbin]) + q.g0_lay[b]) / 2.0;
            const double E = E_factor(w0, g0, scat_corr, i2s);
            const double B_bot = lower ? Bi[i] : Bl[i], B_top = lower ? Bl[i] : Bi[i + 1];
            double pd, pu;
            if (dtau < dtau_limit) {
                pu = lower ? (N + M - P) * (Bi[i] + Bl[i]) / 2.0 : (N + M - P) * (Bl[i] + Bi[i + 1]) / 2.0;
                pd = pu;
            } else {
                const double pgrad = (B_bot - B_top) / dtau;
                pd = (M + N) * B_bot - P * B_top + epsi / (E * (1.0 - w0 * g0)) * (P - M + N) * pgrad;
                pu = (M + N) * B_top - P * B_bot + epsi / (E * (1.0 - w0 * g0)) * (M - N - P) * pgrad;
            }
            double dd = 0.0, du = 0.0;
            if (!LEAN || dir_beam == 1) {
                const double Gm = lower ? q.Gml[k] : q.Gmu[k], Gp = lower ? q.Gpl[k] : q.Gpu[k];
                const double F_bot = lower ? F_dir[k] : Fc_dir[k];
                const double F_top = lower ? Fc_dir[k] : F_dir[k + nc];
                dd = dmin(0.0, F_bot / (-mu_star) * (Gm * M + Gp * N) - F_top / (-mu_star) * P * Gm);
                du = dmin(0.0, F_top / (-mu_star) * (Gm * N + Gp * M) - F_bot / (-mu_star) * P * Gp);
            }
            const double al = P / M, be = -N / M;
            const double sd = 1.0 / M * (2.0 * HX_PI * epsi * (1.0 - w0) / (E - w0) * pd + dd);
            const double su = 1.0 / M * (2.0 * HX_PI * epsi * (1.0 - w0) / (E - w0) * pu + du);
            const size_t kj = c + nc * j;
            if (!LEAN) {
                q.alpha[kj] = al;
                q.beta[kj] = be;
                q.s_down[kj] = sd;
                q.s_up[kj] = su;
            }
            th.push(-be, -al, sd);
            th.push(-be, 1.0, su);
        }
        double xi = th.last(src_toa);
        F_up[c + nc * nl] = xi;
        back_substitute<true>(q.c_prime, q.d_prime, c, nc, 4 * ni - 4, xi, [&](int i, double x) {
            const size_t k = c + nc * (i >> 2);
            switch (i & 3) {
                case 0: F_down[k] = x; break;
                case 1: F_up[k] = x; break;
                case 2: Fc_down[k] = x; break;
                default: Fc_up[k] = x; break;
            }
        });
    } else {
        // pure absorption (kernels.cu:2286-2421)
        double Fd = src_toa;
        F_down[c + nc * nl] = Fd;
        for (int i = nl - 1; i >= 0; i--) {
            const size_t k = c + nc * i, b = x + (size_t)nbin * i;
            const double tu = q.trans_u[k], tl = q.trans_l[k];
            const double du_ = q.dtu[k] + q.dcu[b], dl_ = q.dtl[k] + q.dcl[b];
            double pt;
            if (du_ < dtau_limit) pt = (Bi[i + 1] + Bl[i]) / 2.0 * (1.0 - tu);
            else pt = Bl[i] - tu * Bi[i + 1] + epsi * (tu - 1.0) * ((Bl[i] - Bi[i + 1]) / du_);
            const double Fcd = tiny_abs(tu * Fd + 2.0 * HX_PI * epsi * pt);
            Fc_down[k] = Fcd;
            if (dl_ < dtau_limit) pt = (Bi[i] + Bl[i]) / 2.0 * (1.0 - tl);
            else pt = Bi[i] - tl * Bl[i] + epsi * (tl - 1.0) * ((Bi[i] - Bl[i]) / dl_);
            Fd = tiny_abs(tl * Fcd + 2.0 * HX_PI * epsi * pt);
            F_down[k] = Fd;
        }
        double Fu = A * (F_dir[c] + Fd) + (1.0 - A) * HX_PI * Bl[ni];
        F_up[c] = Fu;
        for (int i = 1; i < ni; i++) {
            const size_t k = c + nc * (i - 1), b = x + (size_t)nbin * (i - 1);
            const double tu = q.trans_u[k], tl = q.trans_l[k];
            const double du_ = q.dtu[k] + q.dcu[b], dl_ = q.dtl[k] + q.dcl[b];
            double pt;
            if (dl_ < dtau_limit) pt = (Bi[i - 1] + Bl[i - 1]) / 2.0 * (1.0 - tl);
            else pt = Bl[i - 1] - tl * Bi[i - 1] + epsi * ((Bi[i - 1] - Bl[i - 1]) / dl_) * (1.0 - tl);
            const double Fcu = tl * Fu + 2.0 * HX_PI * epsi * pt;
            Fc_up[k] = Fcu;
            // the reference applies its |.| patch to entry i, not i-1 (kernels.cu:2394); nl-1 at most
            if (i < nl) Fc_up[k + nc] = tiny_abs(Fc_up[k + nc]);
            if (du_ < dtau_limit) pt = (Bi[i] + Bl[i - 1]) / 2.0 * (1.0 - tu);
            else pt = Bi[i] - tu * Bl[i - 1] + epsi * ((Bl[i - 1] - Bi[i]) / du_) * (1.0 - tu);
            Fu = tiny_abs(tu * Fcu + 2.0 * HX_PI * epsi * pt);
            F_up[k + nc] = Fu;
        }
    }
}

}  // namespace

extern "C" {

// `skip`: device flag of the fused loop (a column whose loop has ended keeps its fluxes), or null
int hx_internal_fband_matrix_iso(hx_context* ctx, const int* skip, double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                        const double* planckband_lay, const double* w_0, const double* M_term,
                        const double* N_term, const double* P_term, const double* G_plus,
                        const double* G_minus, const double* g_0_tot_lay, double* alpha, double* beta,
                        double* source_term_down, double* source_term_up, double* c_prime,
                        double* d_prime, const int* scat_trigger, const double* trans_wg,
                        const double* surf_albedo, double g_0, int singlewalk, double Rstar, double a,
                        int numinterfaces, int nbin, double f_factor, double mu_star, int ny,
                        double epsi, int dir_beam, int clouds, int scat_corr, int debug,
                        double i2s_transition) {
    (void)singlewalk;
    (skip ? k_fband_matrix_iso<true> : k_fband_matrix_iso<false>)<<<hx_cdiv((long long)ny * nbin, 256), 256, 0, ctx->stream>>>(
        F_down_wg, F_up_wg, F_dir_wg, planckband_lay, w_0, M_term, N_term, P_term, G_plus, G_minus,
        g_0_tot_lay, alpha, beta, source_term_down, source_term_up, c_prime, d_prime, scat_trigger,
        trans_wg, surf_albedo, g_0, Rstar, a, numinterfaces, nbin, f_factor, mu_star, ny, epsi,
        dir_beam, clouds, scat_corr, i2s_transition, skip);
    HX_LAUNCH_CHECK(ctx);
    if (debug == 1) {  // kernels.cu:1990, :2018 (the solution vector of :2268 lands in these arrays)
        const size_t n = (size_t)ny * nbin * numinterfaces;
        int rc = hx_internal_count_negative(ctx, F_down_wg, n, HX_DIAG_NEG_DOWN);
        if (!rc) rc = hx_internal_count_negative(ctx, F_up_wg, n, HX_DIAG_NEG_UP);
        if (rc) return rc;
    }
    return 0;
}

int hx_internal_fband_matrix_noniso(
    hx_context* ctx, const int* skip, double* F_down_wg, double* F_up_wg, double* Fc_down_wg, double* Fc_up_wg,
    const double* F_dir_wg, const double* Fc_dir_wg, const double* planckband_lay,
    const double* planckband_int, const double* w_0_upper, const double* w_0_lower,
    const double* delta_tau_wg_upper, const double* delta_tau_wg_lower,
    const double* delta_tau_all_clouds_upper, const double* delta_tau_all_clouds_lower,
    const double* M_upper, const double* M_lower, const double* N_upper, const double* N_lower,
    const double* P_upper, const double* P_lower, const double* G_plus_upper,
    const double* G_plus_lower, const double* G_minus_upper, const double* G_minus_lower,
    const double* g_0_tot_lay, const double* g_0_tot_int, double* alpha, double* beta,
    double* source_term_down, double* source_term_up, double* c_prime, double* d_prime,
    const int* scat_trigger, const double* trans_wg_upper, const double* trans_wg_lower,
    const double* surf_albedo, double g_0, int singlewalk, double Rstar, double a, int numinterfaces,
    int nbin, double f_factor, double mu_star, int ny, double epsi, double delta_tau_limit,
    int dir_beam, int clouds, int scat_corr, int debug, double i2s_transition) {
    (void)singlewalk;
    MatrixNoniso q = {w_0_upper, w_0_lower, delta_tau_wg_upper, delta_tau_wg_lower,
                      delta_tau_all_clouds_upper, delta_tau_all_clouds_lower, M_upper, M_lower,
                      N_upper, N_lower, P_upper, P_lower, G_plus_upper, G_plus_lower, G_minus_upper,
                      G_minus_lower, g_0_tot_lay, g_0_tot_int, trans_wg_upper, trans_wg_lower,
                      alpha, beta, source_term_down, source_term_up, c_prime, d_prime};
    (skip ? k_fband_matrix_noniso<true> : k_fband_matrix_noniso<false>)<<<hx_cdiv((long long)ny * nbin, 256), 256, 0, ctx->stream>>>(
        F_down_wg, F_up_wg, Fc_down_wg, Fc_up_wg, F_dir_wg, Fc_dir_wg, planckband_lay, planckband_int,
        q, scat_trigger, surf_albedo, g_0, Rstar, a, numinterfaces, nbin, f_factor, mu_star, ny, epsi,
        delta_tau_limit, dir_beam, clouds, scat_corr, i2s_transition, skip);
    HX_LAUNCH_CHECK(ctx);
    if (debug == 1) {  // kernels.cu:2268, :2329, :2351, :2397, :2418
        const size_t nc = (size_t)ny * nbin;
        int rc = hx_internal_count_negative(ctx, F_down_wg, nc * numinterfaces, HX_DIAG_NEG_DOWN);
        if (!rc) rc = hx_internal_count_negative(ctx, Fc_down_wg, nc * (numinterfaces - 1), HX_DIAG_NEG_DOWN);
        if (!rc) rc = hx_internal_count_negative(ctx, F_up_wg, nc * numinterfaces, HX_DIAG_NEG_UP);
        if (!rc) rc = hx_internal_count_negative(ctx, Fc_up_wg, nc * (numinterfaces - 1), HX_DIAG_NEG_UP);
        if (rc) return rc;
    }
    return 0;
}

int hx_fband_matrix_iso(hx_context* ctx, double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                        const double* planckband_lay, const double* w_0, const double* M_term,
                        const double* N_term, const double* P_term, const double* G_plus,
                        const double* G_minus, const double* g_0_tot_lay, double* alpha, double* beta,
                        double* source_term_down, double* source_term_up, double* c_prime,
                        double* d_prime, const int* scat_trigger, const double* trans_wg,
                        const double* surf_albedo, double g_0, int singlewalk, double Rstar, double a,
                        int numinterfaces, int nbin, double f_factor, double mu_star, int ny,
                        double epsi, int dir_beam, int clouds, int scat_corr, int debug,
                        double i2s_transition) {
    return hx_internal_fband_matrix_iso(ctx, nullptr, F_down_wg, F_up_wg, F_dir_wg, planckband_lay, w_0, M_term, N_term,
                                        P_term, G_plus, G_minus, g_0_tot_lay, alpha, beta, source_term_down,
                                        source_term_up, c_prime, d_prime, scat_trigger, trans_wg, surf_albedo, g_0,
                                        singlewalk, Rstar, a, numinterfaces, nbin, f_factor, mu_star, ny, epsi,
                                        dir_beam, clouds, scat_corr, debug, i2s_transition);
}

int hx_fband_matrix_noniso(
    hx_context* ctx, double* F_down_wg, double* F_up_wg, double* Fc_down_wg, double* Fc_up_wg,
    const double* F_dir_wg, const double* Fc_dir_wg, const double* planckband_lay,
    const double* planckband_int, const double* w_0_upper, const double* w_0_lower,
    const double* delta_tau_wg_upper, const double* delta_tau_wg_lower,
    const double* delta_tau_all_clouds_upper, const double* delta_tau_all_clouds_lower,
    const double* M_upper, const double* M_lower, const double* N_upper, const double* N_lower,
    const double* P_upper, const double* P_lower, const double* G_plus_upper,
    const double* G_plus_lower, const double* G_minus_upper, const double* G_minus_lower,
    const double* g_0_tot_lay, const double* g_0_tot_int, double* alpha, double* beta,
    double* source_term_down, double* source_term_up, double* c_prime, double* d_prime,
    const int* scat_trigger, const double* trans_wg_upper, const double* trans_wg_lower,
    const double* surf_albedo, double g_0, int singlewalk, double Rstar, double a, int numinterfaces,
    int nbin, double f_factor, double mu_star, int ny, double epsi, double delta_tau_limit,
    int dir_beam, int clouds, int scat_corr, int debug, double i2s_transition) {
    return hx_internal_fband_matrix_noniso(
        ctx, nullptr, F_down_wg, F_up_wg, Fc_down_wg, Fc_up_wg, F_dir_wg, Fc_dir_wg, planckband_lay, planckband_int,
        w_0_upper, w_0_lower, delta_tau_wg_upper, delta_tau_wg_lower, delta_tau_all_clouds_upper,
        delta_tau_all_clouds_lower, M_upper, M_lower, N_upper, N_lower, P_upper, P_lower, G_plus_upper, G_plus_lower,
        G_minus_upper, G_minus_lower, g_0_tot_lay, g_0_tot_int, alpha, beta, source_term_down, source_term_up, c_prime,
        d_prime, scat_trigger, trans_wg_upper, trans_wg_lower, surf_albedo, g_0, singlewalk, Rstar, a, numinterfaces,
        nbin, f_factor, mu_star, ny, epsi, delta_tau_limit, dir_beam, clouds, scat_corr, debug, i2s_transition);
}

}  // extern "C"
