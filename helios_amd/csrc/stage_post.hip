// Per-stage entry points, part 5: post-loop spectral diagnostics (SURVEY.md 8(f) rank 1;
// reference source/kernels.cu:2888-3139).  They run once per model, after the iteration loop.
#include "two_stream.h"

using namespace hx;

namespace {

// Band means of optical depth and transmission (kernels.cu:2888-2948).  grid (ceil(nbin / PBINS), nlayer), 256 threads: a
// workgroup reads the ny * PBINS spectral points of its bins as one contiguous run, weights them into LDS, and one thread
// per bin adds its Gauss points in the reference's order (a thread per bin walking its own ny values read one 64-byte
// sector per double: 1.4 ms at 10 000 x 100 x 20).
constexpr int PBINS = 32;
template <bool NONISO>
__global__ void __launch_bounds__(256)
k_optdepth_transmission(const double* __restrict__ trans_u, const double* __restrict__ trans_l,
                        double* __restrict__ trans_band, const double* __restrict__ dtau_u,
                        const double* __restrict__ dtau_l, double* __restrict__ dtau_band,
                        const double* __restrict__ gauss_weight, double* __restrict__ dtc,
                        const double* __restrict__ dtc_u, const double* __restrict__ dtc_l, int nbin,
                        int nlayer, int ny) {
    extern __shared__ __align__(16) double smem[];
    const int x0 = blockIdx.x * PBINS, i = blockIdx.y;
    const int nb = min(PBINS, nbin - x0), pitch = ny + 1;
    double* s_dt = smem;
    double* s_tr = smem + PBINS * pitch;
    const size_t base = (size_t)ny * x0 + (size_t)ny * nbin * i;
    for (int t = threadIdx.x; t < nb * ny; t += blockDim.x) {
        const int xl = t / ny, y = t - xl * ny;
        const double w = 0.5 * gauss_weight[y];
        if (NONISO) {
            s_dt[xl * pitch + y] = w * (dtau_u[base + t] + dtau_l[base + t]);
            s_tr[xl * pitch + y] = w * (trans_u[base + t] * trans_l[base + t]);
        } else {
            s_dt[xl * pitch + y] = w * dtau_u[base + t];
            s_tr[xl * pitch + y] = w * trans_u[base + t];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x >= nb) return;
    const int xl = threadIdx.x;
    double dt = 0.0, tr = 0.0;
    for (int y = 0; y < ny; y++) {
        dt += s_dt[xl * pitch + y];
        tr += s_tr[xl * pitch + y];
    }
    const size_t b = x0 + xl + (size_t)nbin * i;
    dtau_band[b] = dt;
    trans_band[b] = tr;
    if (NONISO) dtc[b] = dtc_l[b] + dtc_u[b];
}

// contribution function: trans_weight[x,i] += sum_y w_y (1 - T_i) prod_{j>i} T_j (kernels.cu:2951-3020), the product
// rebuilt for every (i, y) in the reference's multiplication order -- ((1 t_u[i+1]) t_l[i+1]) t_u[i+2] ... -- and the Gauss
// points added in the reference's order, so the bits are the reference's.  Note the reference ACCUMULATES into
// trans_weight_band without zeroing it; so does this.
// One thread per spectral point (consecutive threads, consecutive addresses) keeps the running products of CONTR_CH layers
// in registers while it walks up the column once per chunk of layers; the weighted terms of a chunk go through LDS to one
// thread per bin, which adds them over the Gauss points.  (One thread per bin rebuilding every product from memory, as the
// reference does: 73 ms at 10 000 x 100 x 20 -- a sixth of a whole run to equilibrium.)
constexpr int CONTR_CH = 16;
template <bool NONISO>
__global__ void __launch_bounds__(256)
k_contr_func(const double* __restrict__ trans_u, const double* __restrict__ trans_l,
             double* __restrict__ trans_weight_band, double* __restrict__ contr_func_band,
             const double* __restrict__ gauss_weight, const double* __restrict__ planckband_lay,
             double epsi, int nbin, int nlayer, int ny) {
    __shared__ double term[CONTR_CH][256];
    const int nbb = 256 / ny;                       // bins per workgroup
    const int x0 = blockIdx.x * nbb, nb = min(nbb, nbin - x0);
    const int tid = threadIdx.x;
    const bool active = tid < nb * ny;
    const int y = tid % ny;
    const size_t sl = (size_t)ny * nbin;
    const size_t c = (size_t)ny * x0 + (size_t)(active ? tid : 0);
    const double w = gauss_weight[y];
    for (int i0 = 0; i0 < nlayer; i0 += CONTR_CH) {
        double P[CONTR_CH];
#pragma unroll
        for (int u = 0; u < CONTR_CH; u++) P[u] = 1.0;
        // layers inside the chunk: layer j multiplies the products of the layers below it
        for (int j = i0 + 1; j < min(i0 + CONTR_CH, nlayer); j++) {
            const double a = trans_u[c + sl * j], b = NONISO ? trans_l[c + sl * j] : 1.0;
#pragma unroll
            for (int u = 0; u < CONTR_CH; u++)
                if (i0 + u < j) {
                    P[u] = P[u] * a;
                    if (NONISO) P[u] = P[u] * b;
                }
        }
        // layers above the chunk multiply all of them
        for (int j = i0 + CONTR_CH; j < nlayer; j++) {
            const double a = trans_u[c + sl * j], b = NONISO ? trans_l[c + sl * j] : 1.0;
#pragma unroll
            for (int u = 0; u < CONTR_CH; u++) {
                P[u] = P[u] * a;
                if (NONISO) P[u] = P[u] * b;
            }
        }
#pragma unroll
        for (int u = 0; u < CONTR_CH; u++) {
            const int i = min(i0 + u, nlayer - 1);
            const double Ti = NONISO ? trans_u[c + sl * i] * trans_l[c + sl * i] : trans_u[c + sl * i];
            term[u][tid] = 0.5 * w * (1.0 - Ti) * P[u];
        }
        __syncthreads();
        if (tid < nb) {
            const int x = x0 + tid;
            for (int u = 0; u < CONTR_CH && i0 + u < nlayer; u++) {
                const int i = i0 + u;
                double acc = trans_weight_band[x + (size_t)nbin * i];
                for (int yy = 0; yy < ny; yy++) acc += term[u][tid * ny + yy];
                trans_weight_band[x + (size_t)nbin * i] = acc;
                contr_func_band[x + (size_t)nbin * i] =
                    2.0 * HX_PI * epsi * planckband_lay[i + (size_t)x * (nlayer + 2)] * acc;
            }
        }
        __syncthreads();
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
    HX_REQUIRE(ctx, ny >= 1 && ny <= 64, HX_E_UNSUPPORTED, "more than 64 Gauss points per bin");
    k_optdepth_transmission<false><<<dim3(hx_cdiv(nbin, PBINS), nlayer), 256, 2 * PBINS * (ny + 1) * sizeof(double), ctx->stream>>>(
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
    HX_REQUIRE(ctx, ny >= 1 && ny <= 64, HX_E_UNSUPPORTED, "more than 64 Gauss points per bin");
    k_optdepth_transmission<true><<<dim3(hx_cdiv(nbin, PBINS), nlayer), 256, 2 * PBINS * (ny + 1) * sizeof(double), ctx->stream>>>(
        trans_wg_upper, trans_wg_lower, trans_band, delta_tau_wg_upper, delta_tau_wg_lower,
        delta_tau_band, gauss_weight, delta_tau_all_clouds, delta_tau_all_clouds_upper,
        delta_tau_all_clouds_lower, nbin, nlayer, ny);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_calc_contr_func_iso(hx_context* ctx, const double* trans_wg, double* trans_weight_band,
                           double* contr_func_band, const double* gauss_weight,
                           const double* planckband_lay, double epsi, int nbin, int nlayer, int ny) {
    HX_REQUIRE(ctx, ny >= 1 && ny <= 256, HX_E_UNSUPPORTED, "more than 256 Gauss points per bin");
    k_contr_func<false><<<hx_cdiv(nbin, 256 / ny), 256, 0, ctx->stream>>>(
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
    HX_REQUIRE(ctx, ny >= 1 && ny <= 256, HX_E_UNSUPPORTED, "more than 256 Gauss points per bin");
    k_contr_func<true><<<hx_cdiv(nbin, 256 / ny), 256, 0, ctx->stream>>>(
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
