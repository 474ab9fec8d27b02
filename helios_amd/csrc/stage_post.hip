// Per-stage entry points, part 5: post-loop spectral diagnostics (SURVEY.md 8(f) rank 1;
// reference source/kernels.cu:2888-3139).  They run once per model, after the iteration loop.
#include "two_stream.h"

using namespace hx;

namespace {

template <bool NONISO>
__global__ void __launch_bounds__(256)
k_optdepth_transmission(const double* __restrict__ trans_u, const double* __restrict__ trans_l,
                        double* __restrict__ trans_band, const double* __restrict__ dtau_u,
                        const double* __restrict__ dtau_l, double* __restrict__ dtau_band,
                        const double* __restrict__ gauss_weight, double* __restrict__ dtc,
                        const double* __restrict__ dtc_u, const double* __restrict__ dtc_l, int nbin,
                        int nlayer, int ny) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (x >= nbin) return;
    const size_t base = (size_t)ny * x + (size_t)ny * nbin * i;
    double dt = 0.0, tr = 0.0;
    for (int y = 0; y < ny; y++) {
        const double w = 0.5 * gauss_weight[y];
        if (NONISO) {
            dt += w * (dtau_u[base + y] + dtau_l[base + y]);
            tr += w * (trans_u[base + y] * trans_l[base + y]);
        } else {
            dt += w * dtau_u[base + y];
            tr += w * trans_u[base + y];
        }
    }
    const size_t b = x + (size_t)nbin * i;
    dtau_band[b] = dt;
    trans_band[b] = tr;
    if (NONISO) dtc[b] = dtc_l[b] + dtc_u[b];
}

// contribution function: trans_weight[x,i] += sum_y w_y (1 - T_i) prod_{j>i} T_j (kernels.cu:2951-3020).
// One thread per bin; the transmission product above layer i is rebuilt per (i, y) in the reference's
// multiplication order (run-once diagnostic, kept bit-compatible rather than fast).
// Note the reference ACCUMULATES into trans_weight_band without zeroing it; so does this.
template <bool NONISO>
__global__ void __launch_bounds__(256)
k_contr_func(const double* __restrict__ trans_u, const double* __restrict__ trans_l,
             double* __restrict__ trans_weight_band, double* __restrict__ contr_func_band,
             const double* __restrict__ gauss_weight, const double* __restrict__ planckband_lay,
             double epsi, int nbin, int nlayer, int ny) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nbin) return;
    // per (x, i): loop y innermost exactly as the reference (product re-built per y)
    for (int i = 0; i < nlayer; i++) {
        double acc = trans_weight_band[x + (size_t)nbin * i];
        for (int y = 0; y < ny; y++) {
            const size_t c = (size_t)y + (size_t)ny * x, sl = (size_t)ny * nbin;
            double to_top = 1.0;
            for (int j = i + 1; j < nlayer; j++)
                to_top = NONISO ? to_top * trans_u[c + sl * j] * trans_l[c + sl * j]
                                : to_top * trans_u[c + sl * j];
            const double Ti = NONISO ? trans_u[c + sl * i] * trans_l[c + sl * i] : trans_u[c + sl * i];
            acc += 0.5 * gauss_weight[y] * (1.0 - Ti) * to_top;
        }
        trans_weight_band[x + (size_t)nbin * i] = acc;
        contr_func_band[x + (size_t)nbin * i] =
            2.0 * HX_PI * epsi * planckband_lay[i + (size_t)x * (nlayer + 2)] * acc;
    }
}

__device__ __forceinline__ double dB_dT(double lambda, double T) {  // kernels.cu:294-308
    const double c3 = HX_CSPEED * HX_CSPEED * HX_CSPEED;
    const double l2 = lambda * lambda;
    const double D = 2.0 * HX_HCONST * c3 * HX_HCONST / ((l2 * l2 * l2) * HX_KBOLTZMANN * (T * T));
    const double e = exp(HX_HCONST * HX_CSPEED / (lambda * HX_KBOLTZMANN * T));
    return D * e / ((e - 1.0) * (e - 1.0));
}

__device__ __forceinline__ double integrated_dB_dT(const double* kw, const double* ky, int ny,
                                                   double lb, double lt, double T) {
    double r = 0.0;
    for (int y = 0; y < ny; y++) {
        const double xx = (ky[y] - 0.5) * 2.0;
        const double arg = (lt - lb) / 2.0 * xx + (lt + lb) / 2.0;
        r += (lt - lb) / 2.0 * kw[y] * dB_dT(arg, T);
    }
    return r;
}

// Planck / Rosseland means (kernels.cu:3024-3115): one block per layer, fixed-order reduction
// over the bins (the reference: one THREAD per layer looping over all bins and Gauss points).
__global__ void __launch_bounds__(256)
k_mean_opacities(double* __restrict__ planck_pl, double* __restrict__ ross_pl,
                 double* __restrict__ planck_st, double* __restrict__ ross_st,
                 const double* __restrict__ opac_wg_lay, const double* __restrict__ abs_cl_lay,
                 const double* __restrict__ mmm_lay, const double* __restrict__ planckband_lay,
                 const double* __restrict__ interwave, const double* __restrict__ deltawave,
                 const double* __restrict__ T_lay, const double* __restrict__ gauss_weight,
                 const double* __restrict__ gauss_y, double* __restrict__ opac_band_lay, int nlayer,
                 int nbin, int ny, double T_star) {
    __shared__ double red[8][256];
    const int i = blockIdx.x;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = threadIdx.x; x < nbin; x += blockDim.x) {
        double ob = 0.0;
        const size_t base = (size_t)ny * x + (size_t)ny * nbin * i;
        for (int y = 0; y < ny; y++) ob += 0.5 * gauss_weight[y] * opac_wg_lay[base + y];
        const size_t b = x + (size_t)nbin * i;
        opac_band_lay[b] = ob;
        const double ext = ob + abs_cl_lay[b] / mmm_lay[i];
        const double Bp = planckband_lay[i + (size_t)x * (nlayer + 2)];
        const double Bs = planckband_lay[nlayer + (size_t)x * (nlayer + 2)];
        const double dbp = integrated_dB_dT(gauss_weight, gauss_y, ny, interwave[x], interwave[x + 1], T_lay[i]);
        const double dbs = integrated_dB_dT(gauss_weight, gauss_y, ny, interwave[x], interwave[x + 1], T_star);
        acc[0] += ext * Bp * deltawave[x];
        acc[1] += Bp * deltawave[x];
        acc[2] += dbp;
        if (ext > 0) acc[3] += dbp / ext;
        acc[4] += ext * Bs * deltawave[x];
        acc[5] += Bs * deltawave[x];
        acc[6] += dbs;
        if (ext > 0) acc[7] += dbs / ext;
    }
    for (int k = 0; k < 8; k++) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 8; k++) red[k][threadIdx.x] += red[k][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        planck_pl[i] = red[0][0] / red[1][0];
        ross_pl[i] = T_lay[i] < 70 ? -3.0 : red[2][0] / red[3][0];
        planck_st[i] = T_star < 70 ? -3.0 : red[4][0] / red[5][0];
        ross_st[i] = T_star < 70 ? -3.0 : red[6][0] / red[7][0];
    }
}

__global__ void __launch_bounds__(1024)
k_integrate_beamflux(double* __restrict__ F_dir_tot, const double* __restrict__ F_dir_band,
                     const double* __restrict__ dlambda, int nbin) {
    __shared__ double s[1024];
    const int i = blockIdx.x;
    double a = 0.0;
    for (int x = threadIdx.x; x < nbin; x += blockDim.x) a += F_dir_band[x + (size_t)nbin * i] * dlambda[x];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) F_dir_tot[i] = s[0];
}

}  // namespace

extern "C" {

int hx_integrate_optdepth_transmission_iso(hx_context* ctx, const double* trans_wg,
                                           double* trans_band, const double* delta_tau_wg,
                                           double* delta_tau_band, const double* gauss_weight,
                                           int nbin, int nlayer, int ny) {
    k_optdepth_transmission<false><<<dim3(hx_cdiv(nbin, 256), nlayer), 256, 0, ctx->stream>>>(
        trans_wg, nullptr, trans_band, delta_tau_wg, nullptr, delta_tau_band, gauss_weight, nullptr,
        nullptr, nullptr, nbin, nlayer, ny);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_integrate_optdepth_transmission_noniso(
    hx_context* ctx, const double* trans_wg_upper, const double* trans_wg_lower, double* trans_band,
    const double* delta_tau_wg_upper, const double* delta_tau_wg_lower, double* delta_tau_band,
    const double* gauss_weight, double* delta_tau_all_clouds,
    const double* delta_tau_all_clouds_upper, const double* delta_tau_all_clouds_lower, int nbin,
    int nlayer, int ny) {
    k_optdepth_transmission<true><<<dim3(hx_cdiv(nbin, 256), nlayer), 256, 0, ctx->stream>>>(
        trans_wg_upper, trans_wg_lower, trans_band, delta_tau_wg_upper, delta_tau_wg_lower,
        delta_tau_band, gauss_weight, delta_tau_all_clouds, delta_tau_all_clouds_upper,
        delta_tau_all_clouds_lower, nbin, nlayer, ny);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_calc_contr_func_iso(hx_context* ctx, const double* trans_wg, double* trans_weight_band,
                           double* contr_func_band, const double* gauss_weight,
                           const double* planckband_lay, double epsi, int nbin, int nlayer, int ny) {
    k_contr_func<false><<<hx_cdiv(nbin, 256), 256, 0, ctx->stream>>>(
        trans_wg, nullptr, trans_weight_band, contr_func_band, gauss_weight, planckband_lay, epsi, nbin,
        nlayer, ny);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_calc_contr_func_noniso(hx_context* ctx, const double* trans_wg_upper,
                              const double* trans_wg_lower, double* trans_weight_band,
                              double* contr_func_band, const double* gauss_weight,
                              const double* planckband_lay, double epsi, int nbin, int nlayer,
                              int ny) {
    k_contr_func<true><<<hx_cdiv(nbin, 256), 256, 0, ctx->stream>>>(
        trans_wg_upper, trans_wg_lower, trans_weight_band, contr_func_band, gauss_weight,
        planckband_lay, epsi, nbin, nlayer, ny);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_calc_mean_opacities(hx_context* ctx, double* planck_opac_T_pl, double* ross_opac_T_pl,
                           double* planck_opac_T_star, double* ross_opac_T_star,
                           const double* opac_wg_lay, const double* abs_cross_all_clouds_lay,
                           const double* meanmolmass_lay, const double* planckband_lay,
                           const double* opac_interwave, const double* opac_deltawave,
                           const double* T_lay, const double* gauss_weight, const double* gauss_y,
                           double* opac_band_lay, int nlayer, int nbin, int ny, double T_star) {
    k_mean_opacities<<<nlayer, 256, 0, ctx->stream>>>(
        planck_opac_T_pl, ross_opac_T_pl, planck_opac_T_star, ross_opac_T_star, opac_wg_lay,
        abs_cross_all_clouds_lay, meanmolmass_lay, planckband_lay, opac_interwave, opac_deltawave,
        T_lay, gauss_weight, gauss_y, opac_band_lay, nlayer, nbin, ny, T_star);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_integrate_beamflux(hx_context* ctx, double* F_dir_tot, const double* F_dir_band,
                          const double* deltalambda, const double* gauss_weight, int nbin,
                          int numinterfaces) {
    (void)gauss_weight;
    k_integrate_beamflux<<<numinterfaces, 1024, 0, ctx->stream>>>(F_dir_tot, F_dir_band, deltalambda,
                                                                 nbin);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
