// Per-stage entry points, part 2: transmission coefficients, layer thickness, direct beam.
#include "two_stream.h"
#include <initializer_list>

using namespace hx;

namespace {

// calc_trans_iso (kernels.cu:1015-1104): thread per (c = y + ny*x, layer i), c fastest
__global__ void __launch_bounds__(256)
k_calc_trans_iso(double* __restrict__ trans_wg, double* __restrict__ delta_tau_wg,
                 double* __restrict__ M_term, double* __restrict__ N_term, double* __restrict__ P_term,
                 double* __restrict__ G_plus, double* __restrict__ G_minus,
                 const double* __restrict__ delta_colmass, const double* __restrict__ opac_wg_lay,
                 const double* __restrict__ meanmolmass_lay, const double* __restrict__ scat_cross_lay,
                 const double* __restrict__ abs_cl_lay, const double* __restrict__ scat_cl_lay,
                 double* __restrict__ delta_tau_all_clouds, double* __restrict__ w_0,
                 const double* __restrict__ g_0_tot_lay, int* __restrict__ scat_trigger, double g_0,
                 double epsi, double epsi2, double mu_star, double w_0_limit, double w_0_scat_limit,
                 int scat, int nbin, int ny, int nlayer, int clouds, int scat_corr, double i2s) {
    const int i = blockIdx.y;
    const size_t nc = (size_t)ny * nbin;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const int x = (int)(c / ny), y = (int)(c - (size_t)x * ny);
    const size_t b = x + (size_t)nbin * i, k = c + nc * i;
    const double g0 = clouds == 1 ? g_0_tot_lay[b] : g_0;
    const double ray = scat == 1 ? scat_cross_lay[b] : 0.0;
    const double csc = scat == 1 ? scat_cl_lay[b] : 0.0;
    const double cab = abs_cl_lay[b];
    const double mu = meanmolmass_lay[i];
    const double kap = opac_wg_lay[k];
    const double dtau_cl = delta_colmass[i] * (cab + csc) / mu;
    if (y == 0) delta_tau_all_clouds[b] = dtau_cl;
    const double w0 = single_scat_albedo(ray + csc, kap * mu + cab, w_0_limit);
    const double dtau = delta_colmass[i] * (kap + ray / mu);
    const Slab s = slab_coeffs(w0, dtau + dtau_cl, g0, epsi, epsi2, mu_star, scat_corr, i2s);
    w_0[k] = w0;
    delta_tau_wg[k] = dtau;
    trans_wg[k] = s.trans;
    M_term[k] = s.M;
    N_term[k] = s.N;
    P_term[k] = s.P;
    G_plus[k] = s.Gp;
    G_minus[k] = s.Gm;
    if (w0 > w_0_scat_limit) scat_trigger[c] = 1;
}

struct NonisoOut {
    double *trans_u, *trans_l, *dtau_u, *dtau_l, *M_u, *M_l, *N_u, *N_l, *P_u, *P_l, *Gp_u, *Gp_l,
        *Gm_u, *Gm_l, *dtc_u, *dtc_l, *w0_u, *w0_l;
    int* scat_trigger;
};
struct NonisoIn {
    const double *dcol_u, *dcol_l, *opac_lay, *opac_int, *mmm_lay, *mmm_int, *sc_lay, *sc_int,
        *cab_lay, *cab_int, *csc_lay, *csc_int, *g0_lay, *g0_int;
};

// calc_trans_noniso (kernels.cu:1107-1243)
__global__ void __launch_bounds__(256)
k_calc_trans_noniso(NonisoOut o, NonisoIn in, double g_0, double epsi, double epsi2, double mu_star,
                    double w_0_limit, double w_0_scat_limit, int scat, int nbin, int ny, int nlayer,
                    int clouds, int scat_corr, double i2s) {
    const int i = blockIdx.y;
    const size_t nc = (size_t)ny * nbin;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const int x = (int)(c / ny), y = (int)(c - (size_t)x * ny);
    const size_t b = x + (size_t)nbin * i, bu = b + nbin, k = c + nc * i, ku = k + nc;
    double g0_up = g_0, g0_low = g_0;
    if (clouds == 1) {
        g0_up = (in.g0_lay[b] + in.g0_int[bu]) / 2.0;
        g0_low = (in.g0_int[b] + in.g0_lay[b]) / 2.0;
    }
    double ray_up = 0, ray_low = 0, csc_up = 0, csc_low = 0;
    if (scat == 1) {
        ray_up = (in.sc_lay[b] + in.sc_int[bu]) / 2.0;
        ray_low = (in.sc_int[b] + in.sc_lay[b]) / 2.0;
        csc_up = (in.csc_lay[b] + in.csc_int[bu]) / 2.0;
        csc_low = (in.csc_int[b] + in.csc_lay[b]) / 2.0;
    }
    const double cab_up = (in.cab_lay[b] + in.cab_int[bu]) / 2.0;
    const double cab_low = (in.cab_int[b] + in.cab_lay[b]) / 2.0;
    const double mu_up = (in.mmm_lay[i] + in.mmm_int[i + 1]) / 2.0;
    const double mu_low = (in.mmm_int[i] + in.mmm_lay[i]) / 2.0;
    const double kap_up = (in.opac_lay[k] + in.opac_int[ku]) / 2.0;
    const double kap_low = (in.opac_int[k] + in.opac_lay[k]) / 2.0;
    const double dtc_up = in.dcol_u[i] * (cab_up + csc_up) / mu_up;
    const double dtc_low = in.dcol_l[i] * (cab_low + csc_low) / mu_low;
    if (y == 0) {
        o.dtc_u[b] = dtc_up;
        o.dtc_l[b] = dtc_low;
    }
    const double w_up = single_scat_albedo(ray_up + csc_up, kap_up * mu_up + cab_up, w_0_limit);
    const double w_low = single_scat_albedo(ray_low + csc_low, kap_low * mu_low + cab_low, w_0_limit);
    const double dt_up = in.dcol_u[i] * (kap_up + ray_up / mu_up);
    const double dt_low = in.dcol_l[i] * (kap_low + ray_low / mu_low);
    const Slab su = slab_coeffs(w_up, dt_up + dtc_up, g0_up, epsi, epsi2, mu_star, scat_corr, i2s);
    const Slab sl = slab_coeffs(w_low, dt_low + dtc_low, g0_low, epsi, epsi2, mu_star, scat_corr, i2s);
    o.w0_u[k] = w_up;
    o.w0_l[k] = w_low;
    o.dtau_u[k] = dt_up;
    o.dtau_l[k] = dt_low;
    o.trans_u[k] = su.trans;
    o.trans_l[k] = sl.trans;
    o.M_u[k] = su.M;
    o.M_l[k] = sl.M;
    o.N_u[k] = su.N;
    o.N_l[k] = sl.N;
    o.P_u[k] = su.P;
    o.P_l[k] = sl.P;
    o.Gp_u[k] = su.Gp;
    o.Gp_l[k] = sl.Gp;
    o.Gm_u[k] = su.Gm;
    o.Gm_l[k] = sl.Gm;
    if (w_up > w_0_scat_limit || w_low > w_0_scat_limit) o.scat_trigger[c] = 1;
}

__global__ void k_calc_delta_z(const double* __restrict__ tlay, const double* __restrict__ pint,
                               const double* __restrict__ mmm, double* __restrict__ dz, double g,
                               int nlayer) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nlayer) dz[i] = HX_KBOLTZMANN * tlay[i] / (mmm[i] * g) * log(pint[i] / pint[i + 1]);
}

// a batch of columns in one launch (the fused refresh): column blockIdx.z, per-column scalars from colpar, columns whose
// loop has ended skipped; colpar == nullptr: the single column of the per-stage entry points
struct BeamBatch {
    const hx_rt_column* colpar;
    const int* done;
};

__device__ __forceinline__ double slant_mu(double mu_star, double R_planet, const double* z_lay,
                                           int i, int j) {
    const double q = (R_planet + z_lay[i]) / (R_planet + z_lay[j]);
    return -sqrt(1.0 - (q * q) * (1.0 - mu_star * mu_star));
}

// Direct beam without the spherical correction (kernels.cu:1265-1362 with geom_zenith_corr == 0):
// F_dir[i] = F_toa * prod_{j=L-1..i} exp(dtau_j/mu*).  The reference recomputes the product for
// every interface (O(L^2)); the factors are the same and are applied in the same order, so one
// top-down running product per (x,y) gives bit-identical values in O(L).
template <bool NONISO>
__global__ void __launch_bounds__(256)
k_fdir_plane(double* __restrict__ F_dir, double* __restrict__ Fc_dir,
             const double* __restrict__ star, int star_stride, const double* __restrict__ dtau_u,
             const double* __restrict__ dtau_l, double mu_star, double R_star, double a, int dir_beam,
             int ninterface, int nbin, int ny, BeamBatch bb) {
    const size_t nc = (size_t)ny * nbin;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    if (bb.colpar) {  // column blockIdx.z of a batch: its own star, orbit and arrays
        const int col = blockIdx.z;
        if (bb.done[col]) return;
        mu_star = bb.colpar[col].mu_star; R_star = bb.colpar[col].R_star; a = bb.colpar[col].a;
        F_dir += col * nc * ninterface;
        dtau_u += col * nc * (ninterface - 1);
        if (NONISO) {
            Fc_dir += col * nc * ninterface;
            dtau_l += col * nc * (ninterface - 1);
        }
        star += (size_t)col * nbin * star_stride;
    }
    const int x = (int)(c / ny);
    const double I_dir = ((R_star / a) * (R_star / a)) * HX_PI * star[(size_t)x * star_stride];
    double F = -dir_beam * mu_star * I_dir;
    F_dir[c + nc * (ninterface - 1)] = F;
    for (int i = ninterface - 2; i >= 0; i--) {
        const size_t k = c + nc * i;
        if (NONISO) {
            Fc_dir[k] = F * exp(dtau_u[k] / mu_star);
            F *= exp((dtau_u[k] + dtau_l[k]) / mu_star);
        } else {
            F *= exp(dtau_u[k] / mu_star);
        }
        F_dir[k] = F;
    }
}

// with the spherical correction mu depends on (i, j): thread per (c, interface i), O(L) each
template <bool NONISO>
__global__ void __launch_bounds__(256)
k_fdir_sphere(double* __restrict__ F_dir, double* __restrict__ Fc_dir,
              const double* __restrict__ star, int star_stride, const double* __restrict__ dtau_u,
              const double* __restrict__ dtau_l, const double* __restrict__ z_lay, double mu_star,
              double R_planet, double R_star, double a, int dir_beam, int ninterface, int nbin, int ny,
              BeamBatch bb) {
    const size_t nc = (size_t)ny * nbin;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (c >= nc) return;
    if (bb.colpar) {
        const int col = blockIdx.z;
        if (bb.done[col]) return;
        mu_star = bb.colpar[col].mu_star; R_star = bb.colpar[col].R_star; a = bb.colpar[col].a;
        R_planet = bb.colpar[col].R_planet;
        F_dir += col * nc * ninterface;
        dtau_u += col * nc * (ninterface - 1);
        if (NONISO) {
            Fc_dir += col * nc * ninterface;
            dtau_l += col * nc * (ninterface - 1);
        }
        star += (size_t)col * nbin * star_stride;
        z_lay += (size_t)col * (ninterface - 1);
    }
    const int x = (int)(c / ny);
    const double I_dir = ((R_star / a) * (R_star / a)) * HX_PI * star[(size_t)x * star_stride];
    double F = -dir_beam * mu_star * I_dir;
    for (int j = ninterface - 2; j >= i; j--) {
        const double mu_j = slant_mu(mu_star, R_planet, z_lay, i, j);
        const size_t k = c + nc * j;
        if (NONISO) {
            if (j == i) Fc_dir[c + nc * i] = F * exp(dtau_u[k] / mu_j);
            F *= exp((dtau_u[k] + dtau_l[k]) / mu_j);
        } else {
            F *= exp(dtau_u[k] / mu_j);
        }
    }
    F_dir[c + nc * i] = F;
}

}  // namespace

extern "C" {

int hx_calc_trans_iso(hx_context* ctx, double* trans_wg, double* delta_tau_wg, double* M_term,
                      double* N_term, double* P_term, double* G_plus, double* G_minus,
                      const double* delta_colmass, const double* opac_wg_lay,
                      const double* meanmolmass_lay, const double* scat_cross_lay,
                      const double* abs_cross_all_clouds_lay,
                      const double* scat_cross_all_clouds_lay, double* delta_tau_all_clouds,
                      double* w_0, const double* g_0_tot_lay, int* scat_trigger, double g_0,
                      double epsi, double epsi2, double mu_star, double w_0_limit,
                      double w_0_scat_limit, int scat, int nbin, int ny, int nlayer, int clouds,
                      int scat_corr, int debug, double i2s_transition) {
    dim3 grid(hx_cdiv((long long)ny * nbin, 256), nlayer);
    k_calc_trans_iso<<<grid, 256, 0, ctx->stream>>>(
        trans_wg, delta_tau_wg, M_term, N_term, P_term, G_plus, G_minus, delta_colmass, opac_wg_lay,
        meanmolmass_lay, scat_cross_lay, abs_cross_all_clouds_lay, scat_cross_all_clouds_lay,
        delta_tau_all_clouds, w_0, g_0_tot_lay, scat_trigger, g_0, epsi, epsi2, mu_star, w_0_limit,
        w_0_scat_limit, scat, nbin, ny, nlayer, clouds, scat_corr, i2s_transition);
    HX_LAUNCH_CHECK(ctx);
    if (debug == 1) {  // G_limiter's warning (kernels.cu:217-231) as a count: clipped values are exactly +-1e8
        const size_t n = (size_t)ny * nbin * nlayer;
        for (const double* G : {G_plus, G_minus}) {
            const int rc = hx_internal_count_abs_ge(ctx, G, n, 1e8, HX_DIAG_G_LIMITED);
            if (rc) return rc;
        }
    }
    return 0;
}

int hx_calc_trans_noniso(
    hx_context* ctx, double* trans_wg_upper, double* trans_wg_lower, double* delta_tau_wg_upper,
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
    int nlayer, int clouds, int scat_corr, int debug, double i2s_transition) {
    NonisoOut o = {trans_wg_upper, trans_wg_lower, delta_tau_wg_upper, delta_tau_wg_lower, M_upper,
                   M_lower, N_upper, N_lower, P_upper, P_lower, G_plus_upper, G_plus_lower,
                   G_minus_upper, G_minus_lower, delta_tau_all_clouds_upper,
                   delta_tau_all_clouds_lower, w_0_upper, w_0_lower, scat_trigger};
    NonisoIn in = {delta_col_upper, delta_col_lower, opac_wg_lay, opac_wg_int, meanmolmass_lay,
                   meanmolmass_int, scat_cross_lay, scat_cross_int, abs_cross_all_clouds_lay,
                   abs_cross_all_clouds_int, scat_cross_all_clouds_lay, scat_cross_all_clouds_int,
                   g_0_tot_lay, g_0_tot_int};
    dim3 grid(hx_cdiv((long long)ny * nbin, 256), nlayer);
    k_calc_trans_noniso<<<grid, 256, 0, ctx->stream>>>(o, in, g_0, epsi, epsi2, mu_star, w_0_limit,
                                                      w_0_scat_limit, scat, nbin, ny, nlayer, clouds,
                                                      scat_corr, i2s_transition);
    HX_LAUNCH_CHECK(ctx);
    if (debug == 1) {
        const size_t n = (size_t)ny * nbin * nlayer;
        for (const double* G : {G_plus_upper, G_plus_lower, G_minus_upper, G_minus_lower}) {
            const int rc = hx_internal_count_abs_ge(ctx, G, n, 1e8, HX_DIAG_G_LIMITED);
            if (rc) return rc;
        }
    }
    return 0;
}

int hx_calc_delta_z(hx_context* ctx, const double* tlay, const double* pint, const double* play,
                    const double* meanmolmass_lay, double* delta_z_lay, double g, int nlayer) {
    (void)play;
    k_calc_delta_z<<<hx_cdiv(nlayer, 64), 64, 0, ctx->stream>>>(tlay, pint, meanmolmass_lay,
                                                               delta_z_lay, g, nlayer);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_fdir_iso(hx_context* ctx, double* F_dir_wg, const double* planckband_lay,
                const double* delta_tau_wg, const double* z_lay, double mu_star, double R_planet,
                double R_star, double a, int dir_beam, int geom_zenith_corr, int ninterface,
                int nbin, int ny) {
    const int nb = hx_cdiv((long long)ny * nbin, 256);
    const double* star = planckband_lay + (ninterface - 1);  // stellar row, stride nlayer + 2
    if (geom_zenith_corr == 1)
        k_fdir_sphere<false><<<dim3(nb, ninterface), 256, 0, ctx->stream>>>(
            F_dir_wg, nullptr, star, ninterface + 1, delta_tau_wg, nullptr, z_lay, mu_star, R_planet,
            R_star, a, dir_beam, ninterface, nbin, ny, BeamBatch{nullptr, nullptr});
    else
        k_fdir_plane<false><<<nb, 256, 0, ctx->stream>>>(F_dir_wg, nullptr, star, ninterface + 1,
                                                        delta_tau_wg, nullptr, mu_star, R_star, a,
                                                        dir_beam, ninterface, nbin, ny, BeamBatch{nullptr, nullptr});
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

// internal: the stellar Planck value per bin is star[x * star_stride]
int hx_internal_fdir_noniso(hx_context* ctx, double* F_dir_wg, double* Fc_dir_wg, const double* star,
                            int star_stride, const double* delta_tau_wg_upper,
                            const double* delta_tau_wg_lower, const double* z_lay, double mu_star,
                            double R_planet, double R_star, double a, int dir_beam,
                            int geom_zenith_corr, int ninterface, int nbin, int ny) {
    const int nb = hx_cdiv((long long)ny * nbin, 256);
    if (geom_zenith_corr == 1)
        k_fdir_sphere<true><<<dim3(nb, ninterface), 256, 0, ctx->stream>>>(
            F_dir_wg, Fc_dir_wg, star, star_stride, delta_tau_wg_upper, delta_tau_wg_lower, z_lay,
            mu_star, R_planet, R_star, a, dir_beam, ninterface, nbin, ny, BeamBatch{nullptr, nullptr});
    else
        k_fdir_plane<true><<<nb, 256, 0, ctx->stream>>>(F_dir_wg, Fc_dir_wg, star, star_stride,
                                                       delta_tau_wg_upper, delta_tau_wg_lower,
                                                       mu_star, R_star, a, dir_beam, ninterface,
                                                       nbin, ny, BeamBatch{nullptr, nullptr});
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

// internal: every column of a fused batch in one launch.  Arrays are column-major with the strides of hx_rt
// (F_dir / Fc_dir: ny*nbin*ninterface, dtau: ny*nbin*nlayer, star: nbin, z_lay: nlayer); mu_star, R_planet, R_star
// and a come from colpar[col]; columns with done[col] != 0 are skipped.
int hx_internal_fdir_noniso_batch(hx_context* ctx, double* F_dir_wg, double* Fc_dir_wg, const double* star,
                                  const double* delta_tau_wg_upper, const double* delta_tau_wg_lower,
                                  const double* z_lay, const hx_rt_column* colpar, const int* done, int ncol,
                                  int dir_beam, int geom_zenith_corr, int ninterface, int nbin, int ny) {
    const int nb = hx_cdiv((long long)ny * nbin, 256);
    const BeamBatch bb{colpar, done};
    const bool noniso = Fc_dir_wg != nullptr;  // isothermal layers (fdir_iso): one optical depth per layer, no centres
    if (geom_zenith_corr == 1) {
        if (noniso)
            k_fdir_sphere<true><<<dim3(nb, ninterface, ncol), 256, 0, ctx->stream>>>(
                F_dir_wg, Fc_dir_wg, star, 1, delta_tau_wg_upper, delta_tau_wg_lower, z_lay, 0.0, 0.0, 0.0, 0.0,
                dir_beam, ninterface, nbin, ny, bb);
        else
            k_fdir_sphere<false><<<dim3(nb, ninterface, ncol), 256, 0, ctx->stream>>>(
                F_dir_wg, nullptr, star, 1, delta_tau_wg_upper, nullptr, z_lay, 0.0, 0.0, 0.0, 0.0, dir_beam,
                ninterface, nbin, ny, bb);
    } else {
        if (noniso)
            k_fdir_plane<true><<<dim3(nb, 1, ncol), 256, 0, ctx->stream>>>(
                F_dir_wg, Fc_dir_wg, star, 1, delta_tau_wg_upper, delta_tau_wg_lower, 0.0, 0.0, 0.0, dir_beam,
                ninterface, nbin, ny, bb);
        else
            k_fdir_plane<false><<<dim3(nb, 1, ncol), 256, 0, ctx->stream>>>(
                F_dir_wg, nullptr, star, 1, delta_tau_wg_upper, nullptr, 0.0, 0.0, 0.0, dir_beam, ninterface, nbin,
                ny, bb);
    }
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_fdir_noniso(hx_context* ctx, double* F_dir_wg, double* Fc_dir_wg,
                   const double* planckband_lay, const double* delta_tau_wg_upper,
                   const double* delta_tau_wg_lower, const double* z_lay, double mu_star,
                   double R_planet, double R_star, double a, int dir_beam, int geom_zenith_corr,
                   int ninterface, int nbin, int ny) {
    return hx_internal_fdir_noniso(ctx, F_dir_wg, Fc_dir_wg, planckband_lay + (ninterface - 1),
                                   ninterface + 1, delta_tau_wg_upper, delta_tau_wg_lower, z_lay,
                                   mu_star, R_planet, R_star, a, dir_beam, geom_zenith_corr,
                                   ninterface, nbin, ny);
}

}  // extern "C"
