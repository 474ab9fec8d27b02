// Per-stage entry points, part 3: the two-stream sweeps on the reference's layouts, quadrature and
// totals, temperature steps.  (The shipped radiation_loop uses the fused path in rt_fused.hip; these
// keep every reference stage callable on its own and are the on-device baseline the fused path is
// checked against.)
#include "two_stream.h"

using namespace hx;

namespace {

// fband_iso (kernels.cu:1366-1517): thread per c = y + ny*x, serial over interfaces
__global__ void __launch_bounds__(256)
k_fband_iso(double* __restrict__ F_down, double* __restrict__ F_up, const double* __restrict__ F_dir,
            const double* __restrict__ planckband_lay, const double* __restrict__ w_0,
            const double* __restrict__ M_term, const double* __restrict__ N_term,
            const double* __restrict__ P_term, const double* __restrict__ G_plus,
            const double* __restrict__ G_minus, const double* __restrict__ surf_albedo,
            const double* __restrict__ g_0_tot_lay, double g_0, double Rstar, double a, int ni,
            int nbin, double f_factor, double mu_star, int ny, double epsi, int dir_beam, int clouds,
            int scat_corr, double i2s) {
    const size_t nc = (size_t)ny * nbin;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const int x = (int)(c / ny);
    const double* B = planckband_lay + (size_t)x * (ni + 1);
    double w0 = 0.0, E = 1.0;
    double Fd = (1.0 - dir_beam) * f_factor * ((Rstar / a) * (Rstar / a)) * HX_PI * B[ni - 1];
    F_down[c + nc * (ni - 1)] = Fd;
    double Fdir_above = F_dir[c + nc * (ni - 1)];
    for (int i = ni - 2; i >= 0; i--) {
        const size_t k = c + nc * i;
        w0 = w_0[k];
        const double M = M_term[k], N = N_term[k], P = P_term[k], Gp = G_plus[k], Gm = G_minus[k];
        const double g0 = clouds == 1 ? g_0_tot_lay[x + (size_t)nbin * i] : g_0;
        E = E_factor(w0, g0, scat_corr, i2s);
        const double Fdir_here = F_dir[k];
        const double flux = P * Fd - N * F_up[k];
        const double planck = B[i] * (N + M - P);
        double direct = Fdir_here / (-mu_star) * (Gm * M + Gp * N) - Fdir_above / (-mu_star) * P * Gm;
        direct = dmin(0.0, direct);
        Fd = tiny_abs(1.0 / M * (flux + 2.0 * HX_PI * epsi * (1.0 - w0) / (E - w0) * planck + direct));
        F_down[k] = Fd;
        Fdir_above = Fdir_here;
    }
    // BOA: w0/E of layer 0 left over from the down sweep
    double Fu = surf_albedo[x] * (F_dir[c] + Fd) +
                (1.0 - surf_albedo[x]) * HX_PI * (1.0 - w0) / (E - w0) * B[ni];
    F_up[c] = Fu;
    double Fdir_below = F_dir[c];
    for (int i = 1; i < ni; i++) {
        const size_t k = c + nc * (i - 1);
        w0 = w_0[k];
        const double M = M_term[k], N = N_term[k], P = P_term[k], Gp = G_plus[k], Gm = G_minus[k];
        const double g0 = clouds == 1 ? g_0_tot_lay[x + (size_t)nbin * (i - 1)] : g_0;
        E = E_factor(w0, g0, scat_corr, i2s);
        const double Fdir_here = F_dir[k + nc];
        const double flux = P * Fu - N * F_down[k + nc];
        const double planck = B[i - 1] * (N + M - P);
        double direct = Fdir_here / (-mu_star) * (Gm * N + Gp * M) - Fdir_below / (-mu_star) * P * Gp;
        direct = dmin(0.0, direct);
        Fu = tiny_abs(1.0 / M * (flux + 2.0 * HX_PI * epsi * (1.0 - w0) / (E - w0) * planck + direct));
        F_up[k + nc] = Fu;
        Fdir_below = Fdir_here;
    }
}

struct NonisoCoef {
    const double *w0_u, *w0_l, *dtau_u, *dtau_l, *dtc_u, *dtc_l, *M_u, *M_l, *N_u, *N_l, *P_u, *P_l,
        *Gp_u, *Gp_l, *Gm_u, *Gm_l;
};

// one half-layer step of SURVEY.md 10.3: F_out = 1/M (P F_in - N F_opp + K*planck + min(0,direct))
__device__ __forceinline__ double half_step(double M, double N, double P, double w0, double E,
                                            double g0, double dtau, double B_exit, double B_entry,
                                            double F_in, double F_opp, double direct, double epsi,
                                            double dtau_limit, bool up) {
    double planck;
    if (dtau < dtau_limit) {
        planck = (B_entry + B_exit) / 2.0 * (N + M - P);
    } else if (!up) {
        const double pgrad = (B_exit - B_entry) / dtau;
        planck = B_exit * (M + N) - B_entry * P + epsi / (E * (1.0 - w0 * g0)) * (P - M + N) * pgrad;
    } else {
        const double pgrad = (B_entry - B_exit) / dtau;
        planck = B_exit * (M + N) - B_entry * P + epsi / (E * (1.0 - w0 * g0)) * pgrad * (M - P - N);
    }
    direct = dmin(0.0, direct);
    return 1.0 / M * (P * F_in - N * F_opp + 2.0 * HX_PI * epsi * (1.0 - w0) / (E - w0) * planck + direct);
}

// fband_noniso (kernels.cu:1521-1799)
__global__ void __launch_bounds__(256)
k_fband_noniso(double* __restrict__ F_down, double* __restrict__ F_up, double* __restrict__ Fc_down,
               double* __restrict__ Fc_up, const double* __restrict__ F_dir,
               const double* __restrict__ Fc_dir, const double* __restrict__ planckband_lay,
               const double* __restrict__ planckband_int, NonisoCoef q,
               const double* __restrict__ surf_albedo, const double* __restrict__ g_0_tot_lay,
               const double* __restrict__ g_0_tot_int, double g_0, double Rstar, double a, int ni,
               int nbin, double f_factor, double mu_star, int ny, double epsi, double dtau_limit,
               int dir_beam, int clouds, int scat_corr, double i2s) {
    const size_t nc = (size_t)ny * nbin;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const int x = (int)(c / ny);
    const int nl = ni - 1;
    const double* Bl = planckband_lay + (size_t)x * (nl + 2);
    const double* Bi = planckband_int + (size_t)x * ni;
    const double nmu = -mu_star;
    double w_low = 0.0, E_low = 1.0;

    double Fd = (1.0 - dir_beam) * f_factor * ((Rstar / a) * (Rstar / a)) * HX_PI * Bl[nl];
    F_down[c + nc * nl] = Fd;
    double Fdir_top = F_dir[c + nc * nl];
    for (int i = nl - 1; i >= 0; i--) {
        const size_t k = c + nc * i, b = x + (size_t)nbin * i;
        double g0_up = g_0, g0_low = g_0;
        if (clouds == 1) {
            g0_up = (g_0_tot_lay[b] + g_0_tot_int[b + nbin]) / 2.0;
            g0_low = (g_0_tot_int[b] + g_0_tot_lay[b]) / 2.0;
        }
        const double w_up = q.w0_u[k];
        w_low = q.w0_l[k];
        const double E_up = E_factor(w_up, g0_up, scat_corr, i2s);
        E_low = E_factor(w_low, g0_low, scat_corr, i2s);
        const double Fcdir = Fc_dir[k], Fdir_bot = F_dir[k];
        {  // upper half: interface i+1 -> centre i
            const double M = q.M_u[k], N = q.N_u[k], P = q.P_u[k], Gp = q.Gp_u[k], Gm = q.Gm_u[k];
            const double direct = Fcdir / nmu * (Gm * M + Gp * N) - Fdir_top / nmu * Gm * P;
            Fd = tiny_abs(half_step(M, N, P, w_up, E_up, g0_up, q.dtau_u[k] + q.dtc_u[b], Bl[i],
                                    Bi[i + 1], Fd, Fc_up[k], direct, epsi, dtau_limit, false));
            Fc_down[k] = Fd;
        }
        {  // lower half: centre i -> interface i
            const double M = q.M_l[k], N = q.N_l[k], P = q.P_l[k], Gp = q.Gp_l[k], Gm = q.Gm_l[k];
            const double direct = Fdir_bot / nmu * (Gm * M + Gp * N) - Fcdir / nmu * P * Gm;
            Fd = tiny_abs(half_step(M, N, P, w_low, E_low, g0_low, q.dtau_l[k] + q.dtc_l[b], Bi[i],
                                    Bl[i], Fd, F_up[k], direct, epsi, dtau_limit, false));
            F_down[k] = Fd;
        }
        Fdir_top = Fdir_bot;
    }

    // BOA boundary with w0/E of layer 0's lower half
    double Fu = surf_albedo[x] * (F_dir[c] + Fd) +
                (1.0 - surf_albedo[x]) * HX_PI * (1.0 - w_low) / (E_low - w_low) * Bl[ni];
    F_up[c] = Fu;
    for (int i = 1; i < ni; i++) {
        const size_t k = c + nc * (i - 1), b = x + (size_t)nbin * (i - 1);
        double g0_up = g_0, g0_low = g_0;
        if (clouds == 1) {
            g0_low = (g_0_tot_int[b] + g_0_tot_lay[b]) / 2.0;
            g0_up = (g_0_tot_lay[b] + g_0_tot_int[b + nbin]) / 2.0;
        }
        const double w_up = q.w0_u[k];
        w_low = q.w0_l[k];
        const double E_up = E_factor(w_up, g0_up, scat_corr, i2s);
        E_low = E_factor(w_low, g0_low, scat_corr, i2s);
        const double Fcdir = Fc_dir[k], Fdir_bot = F_dir[k], Fdir_up = F_dir[k + nc];
        {  // lower half: interface i-1 -> centre i-1
            const double M = q.M_l[k], N = q.N_l[k], P = q.P_l[k], Gp = q.Gp_l[k], Gm = q.Gm_l[k];
            const double direct = Fcdir / nmu * (Gm * N + Gp * M) - Fdir_bot / nmu * P * Gp;
            // NB no tiny-value patch here: the reference's addresses the wrong index (:1763)
            Fu = half_step(M, N, P, w_low, E_low, g0_low, q.dtau_l[k] + q.dtc_l[b], Bl[i - 1],
                           Bi[i - 1], Fu, Fc_down[k], direct, epsi, dtau_limit, true);
            Fc_up[k] = Fu;
        }
        {  // upper half: centre i-1 -> interface i
            const double M = q.M_u[k], N = q.N_u[k], P = q.P_u[k], Gp = q.Gp_u[k], Gm = q.Gm_u[k];
            const double direct = Fdir_up / nmu * (Gm * N + Gp * M) - Fcdir / nmu * P * Gp;
            Fu = tiny_abs(half_step(M, N, P, w_up, E_up, g0_up, q.dtau_u[k] + q.dtc_u[b], Bi[i],
                                    Bl[i - 1], Fu, F_down[k + nc], direct, epsi, dtau_limit, true));
            F_up[k + nc] = Fu;
        }
    }
}

// Quadrature over the Gauss points (kernels.cu:2474-2476).  grid (ceil(nbin / QBINS), ninterface), 256 threads: the
// workgroup reads the ny*QBINS spectral points of its bins as one contiguous run, weights them into LDS, and one thread per
// bin adds its Gauss points in the reference's order (a thread per bin walking its own ny values: 1.0 ms at
// 10 000 x 101 x 20, one 64-byte sector per double)
constexpr int QBINS = 32;
__global__ void __launch_bounds__(256)
k_band_quadrature(const double* __restrict__ F_down_wg, const double* __restrict__ F_up_wg,
                  const double* __restrict__ F_dir_wg, double* __restrict__ F_down_band,
                  double* __restrict__ F_up_band, double* __restrict__ F_dir_band,
                  const double* __restrict__ gauss_weight, int nbin, int ni, int ny) {
    extern __shared__ __align__(16) double smem[];
    const int x0 = blockIdx.x * QBINS, i = blockIdx.y;
    const int nb = min(QBINS, nbin - x0), pitch = ny + 1;
    double* sdir = smem;
    double* sup = smem + QBINS * pitch;
    double* sdn = sup + QBINS * pitch;
    const size_t base = (size_t)ny * x0 + (size_t)ny * nbin * i;
    for (int t = threadIdx.x; t < nb * ny; t += blockDim.x) {
        const int xl = t / ny, y = t - xl * ny;
        const double w = 0.5 * gauss_weight[y];
        sdir[xl * pitch + y] = w * F_dir_wg[base + t];
        sup[xl * pitch + y] = w * F_up_wg[base + t];
        sdn[xl * pitch + y] = w * F_down_wg[base + t];
    }
    __syncthreads();
    if ((int)threadIdx.x >= nb) return;
    const int xl = threadIdx.x;
    double d = 0.0, u = 0.0, dn = 0.0;
    for (int y = 0; y < ny; y++) {
        d += sdir[xl * pitch + y];
        u += sup[xl * pitch + y];
        dn += sdn[xl * pitch + y];
    }
    const size_t b = x0 + xl + (size_t)nbin * i;
    F_dir_band[b] = d;
    F_up_band[b] = u;
    F_down_band[b] = dn;
}

// Totals over wavelength (kernels.cu:2494-2509): one block per interface, fixed-order tree
__global__ void __launch_bounds__(1024)
k_band_totals(const double* __restrict__ dlambda, const double* __restrict__ F_down_band,
              const double* __restrict__ F_up_band, const double* __restrict__ F_dir_band,
              double* __restrict__ F_down_tot, double* __restrict__ F_up_tot, double* __restrict__ F_net,
              int nbin) {
    __shared__ double su[1024], sd[1024];
    const int i = blockIdx.x;
    double up = 0.0, down = 0.0;
    for (int x = threadIdx.x; x < nbin; x += blockDim.x) {
        const size_t b = x + (size_t)nbin * i;
        up += F_up_band[b] * dlambda[x];
        down += (F_dir_band[b] + F_down_band[b]) * dlambda[x];
    }
    su[threadIdx.x] = up;
    sd[threadIdx.x] = down;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            su[threadIdx.x] += su[threadIdx.x + w];
            sd[threadIdx.x] += sd[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        F_up_tot[i] = su[0];
        F_down_tot[i] = sd[0];
        F_net[i] = su[0] - sd[0];
    }
}

}  // namespace

// temperature steps live in temp_step.h so that the fused path shares them
#include "temp_step.h"

namespace {

__global__ void __launch_bounds__(1024)
k_rad_temp_iter(hx::RadTempArgs a) { hx::rad_temp_step(a, threadIdx.x, blockDim.x); }

__global__ void __launch_bounds__(1024)
k_conv_temp_iter(hx::ConvTempArgs a) { hx::conv_temp_step(a, threadIdx.x, blockDim.x); }

}  // namespace

extern "C" {

int hx_fband_iso(hx_context* ctx, double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                 const double* planckband_lay, const double* w_0, const double* M_term,
                 const double* N_term, const double* P_term, const double* G_plus,
                 const double* G_minus, const double* surf_albedo, const double* g_0_tot_lay,
                 double g_0, int singlewalk, double Rstar, double a, int numinterfaces, int nbin,
                 double f_factor, double mu_star, int ny, double epsi, int dir_beam, int clouds,
                 int scat_corr, int debug, double i2s_transition) {
    (void)singlewalk;
    k_fband_iso<<<hx_cdiv((long long)ny * nbin, 256), 256, 0, ctx->stream>>>(
        F_down_wg, F_up_wg, F_dir_wg, planckband_lay, w_0, M_term, N_term, P_term, G_plus, G_minus,
        surf_albedo, g_0_tot_lay, g_0, Rstar, a, numinterfaces, nbin, f_factor, mu_star, ny, epsi,
        dir_beam, clouds, scat_corr, i2s_transition);
    HX_LAUNCH_CHECK(ctx);
    if (debug == 1) {  // the negative-flux warnings (kernels.cu:1458, :1512) as counts
        const size_t n = (size_t)ny * nbin * numinterfaces;
        int rc = hx_internal_count_negative(ctx, F_down_wg, n, HX_DIAG_NEG_DOWN);
        if (!rc) rc = hx_internal_count_negative(ctx, F_up_wg, n, HX_DIAG_NEG_UP);
        if (rc) return rc;
    }
    return 0;
}

int hx_fband_noniso(hx_context* ctx, double* F_down_wg, double* F_up_wg, double* Fc_down_wg,
                    double* Fc_up_wg, const double* F_dir_wg, const double* Fc_dir_wg,
                    const double* planckband_lay, const double* planckband_int,
                    const double* w_0_upper, const double* w_0_lower,
                    const double* delta_tau_wg_upper, const double* delta_tau_wg_lower,
                    const double* delta_tau_all_clouds_upper,
                    const double* delta_tau_all_clouds_lower, const double* M_upper,
                    const double* M_lower, const double* N_upper, const double* N_lower,
                    const double* P_upper, const double* P_lower, const double* G_plus_upper,
                    const double* G_plus_lower, const double* G_minus_upper,
                    const double* G_minus_lower, const double* surf_albedo,
                    const double* g_0_tot_lay, const double* g_0_tot_int, double g_0,
                    int singlewalk, double Rstar, double a, int numinterfaces, int nbin,
                    double f_factor, double mu_star, int ny, double epsi, double delta_tau_limit,
                    int dir_beam, int clouds, int scat_corr, int debug, double i2s_transition) {
    (void)singlewalk;
    NonisoCoef q = {w_0_upper, w_0_lower, delta_tau_wg_upper, delta_tau_wg_lower,
                    delta_tau_all_clouds_upper, delta_tau_all_clouds_lower, M_upper, M_lower,
                    N_upper, N_lower, P_upper, P_lower, G_plus_upper, G_plus_lower, G_minus_upper,
                    G_minus_lower};
    k_fband_noniso<<<hx_cdiv((long long)ny * nbin, 256), 256, 0, ctx->stream>>>(
        F_down_wg, F_up_wg, Fc_down_wg, Fc_up_wg, F_dir_wg, Fc_dir_wg, planckband_lay, planckband_int,
        q, surf_albedo, g_0_tot_lay, g_0_tot_int, g_0, Rstar, a, numinterfaces, nbin, f_factor, mu_star,
        ny, epsi, delta_tau_limit, dir_beam, clouds, scat_corr, i2s_transition);
    HX_LAUNCH_CHECK(ctx);
    if (debug == 1) {  // kernels.cu:1663, :1690, :1767, :1794
        const size_t nc = (size_t)ny * nbin;
        int rc = hx_internal_count_negative(ctx, F_down_wg, nc * numinterfaces, HX_DIAG_NEG_DOWN);
        if (!rc) rc = hx_internal_count_negative(ctx, Fc_down_wg, nc * (numinterfaces - 1), HX_DIAG_NEG_DOWN);
        if (!rc) rc = hx_internal_count_negative(ctx, F_up_wg, nc * numinterfaces, HX_DIAG_NEG_UP);
        if (!rc) rc = hx_internal_count_negative(ctx, Fc_up_wg, nc * (numinterfaces - 1), HX_DIAG_NEG_UP);
        if (rc) return rc;
    }
    return 0;
}

int hx_integrate_flux(hx_context* ctx, const double* deltalambda, double* F_down_tot,
                      double* F_up_tot, double* F_net, const double* F_down_wg,
                      const double* F_up_wg, const double* F_dir_wg, double* F_down_band,
                      double* F_up_band, double* F_dir_band, const double* gauss_weight, int nbin,
                      int numinterfaces, int ny) {
    HX_REQUIRE(ctx, ny <= 64, HX_E_UNSUPPORTED, "more than 64 Gauss points per bin");
    k_band_quadrature<<<dim3(hx_cdiv(nbin, QBINS), numinterfaces), 256, 3 * QBINS * (ny + 1) * sizeof(double), ctx->stream>>>(
        F_down_wg, F_up_wg, F_dir_wg, F_down_band, F_up_band, F_dir_band, gauss_weight, nbin,
        numinterfaces, ny);
    HX_LAUNCH_CHECK(ctx);
    k_band_totals<<<numinterfaces, 1024, 0, ctx->stream>>>(deltalambda, F_down_band, F_up_band,
                                                          F_dir_band, F_down_tot, F_up_tot, F_net, nbin);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_rad_temp_iter(hx_context* ctx, const double* F_down_tot, const double* F_up_tot,
                     const double* F_net, double* F_net_diff, double* tlay, const double* play,
                     const double* tint, const double* pint, int* abrt, double* T_store,
                     double* deltat_prefactor, const double* F_add_heat_lay,
                     const double* F_add_heat_sum, double* F_smooth, double* F_smooth_sum,
                     const double* c_p_lay, const double* meanmolmass_lay, int itervalue,
                     double f_factor, int foreplay, double g, int numlayers, double physical_tstep,
                     double local_limit, int adapt_interval, int smooth, int dim, int step,
                     double F_intern, int no_atmo) {
    (void)F_up_tot;
    (void)tint;
    (void)f_factor;
    HX_REQUIRE(ctx, numlayers + 1 <= 1024 * 64, HX_E_ARG, "too many layers");
    hx::RadTempArgs a = {F_down_tot, F_net, F_net_diff, tlay, play, pint, abrt, T_store,
                         deltat_prefactor, F_add_heat_lay, F_add_heat_sum, F_smooth, F_smooth_sum,
                         c_p_lay, meanmolmass_lay, nullptr, itervalue, foreplay, g, numlayers,
                         physical_tstep, local_limit, adapt_interval, smooth, dim, step, F_intern,
                         no_atmo};
    k_rad_temp_iter<<<1, 1024, 0, ctx->stream>>>(a);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_conv_temp_iter(hx_context* ctx, const double* F_down_tot, const double* F_up_tot,
                      const double* F_net, double* F_net_diff, double* tlay, const double* play,
                      const double* pint, double* T_store, double* deltat_prefactor,
                      const int* marked_red, const double* F_add_heat_lay, double* F_smooth,
                      double* F_smooth_sum, int numlayers, int itervalue, int adapt_interval,
                      int smooth, double F_intern) {
    (void)F_down_tot;
    (void)F_up_tot;
    hx::ConvTempArgs a = {F_net, F_net_diff, tlay, play, pint, T_store, deltat_prefactor, marked_red,
                          F_add_heat_lay, F_smooth, F_smooth_sum, numlayers, itervalue, adapt_interval,
                          smooth, F_intern};
    k_conv_temp_iter<<<1, 1024, 0, ctx->stream>>>(a);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
