/* helios_hip.h -- C-ABI of libhelios_hip.so, the MI355X (gfx950) implementation of the HELIOS
 * wavelength x layer radiative-transfer hot path.
 *
 * What this boundary replaces.  The reference has no FFI: its `Compute` class JIT-compiles
 * source/kernels.cu with PyCUDA (source/computation.py:34-37) and launches kernels by name with
 * `f(args..., block=..., grid=...)` + `cuda.Context.synchronize()`; device arrays are PyCUDA
 * `gpuarray.to_gpu(np_array)` / `cuda.mem_alloc(nbytes)` objects kept on `Store`
 * (source/quantities.py:463-665).  A maintainer of the reference binds THIS library with ctypes
 * instead (INTEGRATION.md shows the stub):
 *
 *   (1) hx_create/hx_destroy/hx_sync/hx_last_error        <- pycuda.autoinit, Context.synchronize
 *   (2) hx_alloc/hx_free/hx_h2d/hx_d2h/hx_d2d/hx_memset0  <- gpuarray.to_gpu, .get(), mem_alloc and the
 *                                                           "upload zeros" idiom (host_functions.py:1050)
 *   (3) one `hx_<kernel>` per reference kernel: SAME argument order as the `__global__` function,
 *       device pointers + scalars, no block/grid (launch geometry is the library's business)
 *   (4) hx_rt_*: the fused fast path (one call per refresh, one per iteration) that the shipped
 *       `helios_amd.computation.Compute.radiation_loop` uses by default
 *
 * Conventions: every function returns 0 on success, <0 for a HIP runtime error (-hipError_t),
 * >0 for a domain error (HX_E_*); hx_last_error() gives the text.  All arrays are fp64 unless
 * declared `int*`.  Pointers named *_dev / passed to hx_<kernel> are DEVICE pointers obtained from
 * hx_alloc.  Calls are asynchronous on the context's stream and ordered; hx_d2h and hx_sync block.
 * One host thread per context.  Array layouts are the reference's.
 */
#ifndef HELIOS_HIP_H
#define HELIOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hx_context hx_context;
typedef struct hx_rt hx_rt;

#define HX_E_ARG 1         /* invalid argument / unsupported dimension */
#define HX_E_RO_NY 2       /* random-overlap mixing needs ny == 20 (source/kernels.cu:3315) */
#define HX_E_UNSUPPORTED 3 /* configuration outside the fused path; use the per-stage entry points */
#define HX_E_STATE 4       /* call order violated (e.g. hx_rt_step before hx_rt_refresh) */

/* ---- (1) lifecycle ------------------------------------------------------------------------- */
int hx_create(int device_id, hx_context** out_ctx);
int hx_destroy(hx_context* ctx);
int hx_sync(hx_context* ctx);
const char* hx_last_error(hx_context* ctx);
int hx_device_name(hx_context* ctx, char* buf, int buflen);
/* Conditions the reference reports through device-side printf (kernels.cu:227 G_limiter, :1458 ff. negative fluxes,
 * :3385 random-overlap re-binning, :455 energy-budget correction) are counted in a per-context device record
 * instead.  The flux and G counters only run in calls that are given debug = 1 (the reference prints under the same
 * flag); the re-binning counter and the correction factor are always kept.  hx_diag_read blocks. */
typedef struct hx_diag {
    uint64_t negative_down_flux; /* entries < 0 in F_down_wg / Fc_down_wg after a flux solve */
    uint64_t negative_up_flux;   /* entries < 0 in F_up_wg / Fc_up_wg */
    uint64_t g_limited;          /* G+ / G- values clipped to +-1e8 */
    uint64_t ro_rebin_skipped;   /* Gauss points that met an already used interval in add_to_mixed_opac */
    double energy_correction;    /* factor applied by the last hx_corr_inc_energy (0 = none yet) */
    uint64_t ro_fixup_passes;    /* exact odd-even passes add_to_mixed_opac ran after its quantised-key sort (a cost
                                    figure: how often two pair sums of different rows agree to 2^-18) */
    uint64_t reserved[2];
} hx_diag;
int hx_diag_read(hx_context* ctx, hx_diag* out);
int hx_diag_reset(hx_context* ctx);
int hx_abi_version(void);
/* raw hipStream_t of the context (for interop with another runtime's stream guards) */
void* hx_stream(hx_context* ctx);
/* HIP-event timing on the context's stream (bench.py / profiling): record two marks, read ms */
int hx_timer_start(hx_context* ctx);
int hx_timer_stop_ms(hx_context* ctx, double* out_ms);

/* ---- (2) memory ---------------------------------------------------------------------------- */
int hx_alloc(hx_context* ctx, size_t nbytes, void** out_dptr);
int hx_free(hx_context* ctx, void* dptr);
int hx_h2d(hx_context* ctx, void* dptr, const void* hptr, size_t nbytes);
int hx_d2h(hx_context* ctx, void* hptr, const void* dptr, size_t nbytes);
int hx_d2d(hx_context* ctx, void* dst, const void* src, size_t nbytes);
int hx_memset0(hx_context* ctx, void* dptr, size_t nbytes);
int hx_mem_info(hx_context* ctx, size_t* out_free, size_t* out_total);
/* Host utility (no context, no device call): nrows rows "\n%-8g%-18.9g%-21.9g%-19.9g" of prefix[4 * r ...] followed by
 * ncols cells `cell_format` ("%-<width>[.<precision>]e" or "...g") of values[ncols * r ...] -- the rows of the reference's
 * per-bin output tables (source/write.py:576-714), formatted by `nthreads` threads.  *out_text is released with
 * hx_host_free. */
int hx_host_format_rows(const double* prefix, const double* values, int nrows, int ncols, const char* cell_format,
                        int nthreads, char** out_text, size_t* out_len);
void hx_host_free(void* p);

/* ---- (3) per-stage entry points: one per reference kernel ------------------------------------
 * Each comment gives the kernel it replaces (source/kernels.cu) and its launcher
 * (source/computation.py).                                                                      */

/* plancktable, kernels.cu:362 / computation.py:39 -- all ten p_iter launches in one call */
int hx_plancktable(hx_context* ctx, double* planck_grid, const double* lambda_edge,
                   const double* deltalambda, int nwave, double Tstar, int dim, int step);
/* corr_inc_energy, kernels.cu:420 / computation.py:62 */
int hx_corr_inc_energy(hx_context* ctx, double* planck_grid, double* starflux,
                       const double* deltalambda, int realstar, int nwave, double Tstar, int dim);
/* calc_total_g_0_of_gas_and_clouds, kernels.cu:472 / computation.py:331 */
int hx_calc_total_g_0_of_gas_and_clouds(hx_context* ctx, const double* scat_cross,
                                        const double* g_0_all_clouds,
                                        const double* scat_cross_all_clouds, double* g_0_tot,
                                        double g_0, int nbin, int nlay_or_nint);
/* temp_inter, kernels.cu:496 / computation.py:104 */
int hx_temp_inter(hx_context* ctx, const double* tlay, double* tint, int numinterfaces,
                  int itervalue);
/* opac_interpol, kernels.cu:524 / computation.py:119 */
int hx_opac_interpol(hx_context* ctx, const double* temp, const double* opactemp,
                     const double* press, const double* opacpress, const double* ktable,
                     double* opac, const double* crosstable, double* scat_cross, int npress,
                     int ntemp, int ny, int nbin, int nlay_or_nint);
/* meanmolmass_interpol, kernels.cu:649 / computation.py:163 */
int hx_meanmolmass_interpol(hx_context* ctx, const double* temp, const double* opactemp,
                            double* meanmolmass, const double* opac_meanmass, const double* press,
                            const double* opacpress, int npress, int ntemp, int ninterface);
/* kappa_interpol, kernels.cu:703 and cp_interpol, kernels.cu:761 / computation.py:199 */
int hx_kappa_interpol(hx_context* ctx, const double* temp, const double* entr_temp,
                      const double* press, const double* entr_press, double* kappa,
                      const double* entr_kappa, int entr_npress, int entr_ntemp, int nlay_or_nint);
int hx_cp_interpol(hx_context* ctx, const double* temp, const double* entr_temp,
                   const double* press, const double* entr_press, double* cp_lay,
                   const double* entr_cp, int entr_npress, int entr_ntemp, int nlayer);
/* entropy_interpol, kernels.cu:815 (log10 T grid) and phase_number_interpol, kernels.cu:869 /
 * computation.py:252, :273 -- diagnostics of the kappa-file modes */
int hx_entropy_interpol(hx_context* ctx, const double* temp, const double* entr_temp,
                        const double* press, const double* entr_press, double* entropy,
                        const double* entr_entropy, int entr_npress, int entr_ntemp, int nlayer);
int hx_phase_number_interpol(hx_context* ctx, const double* temp, const double* entr_temp,
                             const double* press, const double* entr_press, double* state,
                             const double* entr_state, int entr_npress, int entr_ntemp, int nlayer);
/* planck_interpol_layer, kernels.cu:923 / computation.py:294 */
int hx_planck_interpol_layer(hx_context* ctx, const double* temp, double* planckband_lay,
                             const double* planck_grid, const double* starflux, int realstar,
                             int numlayers, int nwave, int dim, int step);
/* planck_interpol_interface, kernels.cu:981 / computation.py:315 */
int hx_planck_interpol_interface(hx_context* ctx, const double* temp, double* planckband_int,
                                 const double* planck_grid, int numinterfaces, int nwave, int dim,
                                 int step);
/* calc_trans_iso, kernels.cu:1015 / computation.py:370 */
int hx_calc_trans_iso(hx_context* ctx, double* trans_wg, double* delta_tau_wg, double* M_term,
                      double* N_term, double* P_term, double* G_plus, double* G_minus,
                      const double* delta_colmass, const double* opac_wg_lay,
                      const double* meanmolmass_lay, const double* scat_cross_lay,
                      const double* abs_cross_all_clouds_lay,
                      const double* scat_cross_all_clouds_lay, double* delta_tau_all_clouds,
                      double* w_0, const double* g_0_tot_lay, int* scat_trigger, double g_0,
                      double epsi, double epsi2, double mu_star, double w_0_limit,
                      double w_0_scat_limit, int scat, int nbin, int ny, int nlayer, int clouds,
                      int scat_corr, int debug, double i2s_transition);
/* calc_trans_noniso, kernels.cu:1107 / computation.py:408 */
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
    int nlayer, int clouds, int scat_corr, int debug, double i2s_transition);
/* calc_delta_z, kernels.cu:1247 / computation.py:464 */
int hx_calc_delta_z(hx_context* ctx, const double* tlay, const double* pint, const double* play,
                    const double* meanmolmass_lay, double* delta_z_lay, double g, int nlayer);
/* fdir_iso, kernels.cu:1265 / computation.py:484 */
int hx_fdir_iso(hx_context* ctx, double* F_dir_wg, const double* planckband_lay,
                const double* delta_tau_wg, const double* z_lay, double mu_star, double R_planet,
                double R_star, double a, int dir_beam, int geom_zenith_corr, int ninterface,
                int nbin, int ny);
/* fdir_noniso, kernels.cu:1313 / computation.py:504 */
int hx_fdir_noniso(hx_context* ctx, double* F_dir_wg, double* Fc_dir_wg,
                   const double* planckband_lay, const double* delta_tau_wg_upper,
                   const double* delta_tau_wg_lower, const double* z_lay, double mu_star,
                   double R_planet, double R_star, double a, int dir_beam, int geom_zenith_corr,
                   int ninterface, int nbin, int ny);
/* fband_iso, kernels.cu:1366 / computation.py:539 (one sweep) */
int hx_fband_iso(hx_context* ctx, double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                 const double* planckband_lay, const double* w_0, const double* M_term,
                 const double* N_term, const double* P_term, const double* G_plus,
                 const double* G_minus, const double* surf_albedo, const double* g_0_tot_lay,
                 double g_0, int singlewalk, double Rstar, double a, int numinterfaces, int nbin,
                 double f_factor, double mu_star, int ny, double epsi, int dir_beam, int clouds,
                 int scat_corr, int debug, double i2s_transition);
/* fband_noniso, kernels.cu:1521 / computation.py:573 (one sweep) */
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
                    int dir_beam, int clouds, int scat_corr, int debug, double i2s_transition);
/* fband_matrix_iso, kernels.cu:1803 / computation.py:625 -- the flux solve as one tridiagonal system
 * (Thomas algorithm) per spectral point; alpha .. d_prime are caller-provided work arrays
 * (ny*nbin*nlayer each for alpha, beta and the two sources; ny*nbin*2*ninterface for c_prime, d_prime).
 * The surface albedo must be > 0 (the reference's reader enforces >= 1e-8, read.py:1261). */
int hx_fband_matrix_iso(hx_context* ctx, double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                        const double* planckband_lay, const double* w_0, const double* M_term,
                        const double* N_term, const double* P_term, const double* G_plus,
                        const double* G_minus, const double* g_0_tot_lay, double* alpha, double* beta,
                        double* source_term_down, double* source_term_up, double* c_prime,
                        double* d_prime, const int* scat_trigger, const double* trans_wg,
                        const double* surf_albedo, double g_0, int singlewalk, double Rstar, double a,
                        int numinterfaces, int nbin, double f_factor, double mu_star, int ny,
                        double epsi, int dir_beam, int clouds, int scat_corr, int debug,
                        double i2s_transition);
/* fband_matrix_noniso, kernels.cu:2028 / computation.py:667 (work arrays: 2*nlayer planes for alpha,
 * beta, sources; 4*ninterface-2 planes for c_prime, d_prime) */
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
    int dir_beam, int clouds, int scat_corr, int debug, double i2s_transition);
/* integrate_flux_double, kernels.cu:2428 / computation.py:731 (deterministic summation order) */
int hx_integrate_flux(hx_context* ctx, const double* deltalambda, double* F_down_tot,
                      double* F_up_tot, double* F_net, const double* F_down_wg,
                      const double* F_up_wg, const double* F_dir_wg, double* F_down_band,
                      double* F_up_band, double* F_dir_band, const double* gauss_weight, int nbin,
                      int numinterfaces, int ny);
/* rad_temp_iter, kernels.cu:2606 / computation.py:759 */
int hx_rad_temp_iter(hx_context* ctx, const double* F_down_tot, const double* F_up_tot,
                     const double* F_net, double* F_net_diff, double* tlay, const double* play,
                     const double* tint, const double* pint, int* abrt, double* T_store,
                     double* deltat_prefactor, const double* F_add_heat_lay,
                     const double* F_add_heat_sum, double* F_smooth, double* F_smooth_sum,
                     const double* c_p_lay, const double* meanmolmass_lay, int itervalue,
                     double f_factor, int foreplay, double g, int numlayers, double physical_tstep,
                     double local_limit, int adapt_interval, int smooth, int dim, int step,
                     double F_intern, int no_atmo);
/* conv_temp_iter, kernels.cu:2768 / computation.py:799 */
int hx_conv_temp_iter(hx_context* ctx, const double* F_down_tot, const double* F_up_tot,
                      const double* F_net, double* F_net_diff, double* tlay, const double* play,
                      const double* pint, double* T_store, double* deltat_prefactor,
                      const int* marked_red, const double* F_add_heat_lay, double* F_smooth,
                      double* F_smooth_sum, int numlayers, int itervalue, int adapt_interval,
                      int smooth, double F_intern);
/* integrate_optdepth_transmission_{iso,noniso}, kernels.cu:2888/:2916 / computation.py:1176 */
int hx_integrate_optdepth_transmission_iso(hx_context* ctx, const double* trans_wg,
                                           double* trans_band, const double* delta_tau_wg,
                                           double* delta_tau_band, const double* gauss_weight,
                                           int nbin, int nlayer, int ny);
int hx_integrate_optdepth_transmission_noniso(
    hx_context* ctx, const double* trans_wg_upper, const double* trans_wg_lower, double* trans_band,
    const double* delta_tau_wg_upper, const double* delta_tau_wg_lower, double* delta_tau_band,
    const double* gauss_weight, double* delta_tau_all_clouds,
    const double* delta_tau_all_clouds_upper, const double* delta_tau_all_clouds_lower, int nbin,
    int nlayer, int ny);
/* calc_contr_func_{iso,noniso}, kernels.cu:2951/:2987 / computation.py:1216 */
int hx_calc_contr_func_iso(hx_context* ctx, const double* trans_wg, double* trans_weight_band,
                           double* contr_func_band, const double* gauss_weight,
                           const double* planckband_lay, double epsi, int nbin, int nlayer, int ny);
int hx_calc_contr_func_noniso(hx_context* ctx, const double* trans_wg_upper,
                              const double* trans_wg_lower, double* trans_weight_band,
                              double* contr_func_band, const double* gauss_weight,
                              const double* planckband_lay, double epsi, int nbin, int nlayer,
                              int ny);
/* calc_mean_opacities, kernels.cu:3024 / computation.py:1254 */
int hx_calc_mean_opacities(hx_context* ctx, double* planck_opac_T_pl, double* ross_opac_T_pl,
                           double* planck_opac_T_star, double* ross_opac_T_star,
                           const double* opac_wg_lay, const double* abs_cross_all_clouds_lay,
                           const double* meanmolmass_lay, const double* planckband_lay,
                           const double* opac_interwave, const double* opac_deltawave,
                           const double* T_lay, const double* gauss_weight, const double* gauss_y,
                           double* opac_band_lay, int nlayer, int nbin, int ny, double T_star);
/* integrate_beamflux, kernels.cu:3119 / computation.py:1283 */
int hx_integrate_beamflux(hx_context* ctx, double* F_dir_tot, const double* F_dir_band,
                          const double* deltalambda, const double* gauss_weight, int nbin,
                          int numinterfaces);
/* opac_species_interpol, kernels.cu:3209 / computation.py:1298 */
int hx_opac_species_interpol(hx_context* ctx, const double* temp, const double* opactemp,
                             const double* press, const double* opacpress,
                             const double* opac_opacity_pretab, double* opac_spec_wg, int npress,
                             int ntemp, int ny, int nbin, int nlay_or_nint);
/* add_to_mixed_opac, kernels.cu:3263 / computation.py:1338 */
int hx_add_to_mixed_opac(hx_context* ctx, const double* vmr, const double* opac_spec,
                         double* opac_wg, const double* meanmolmass, const double* gauss_weight,
                         const double* gauss_y, double mass_spec, int s, int ro_method, int ny,
                         int nbin, int nlay_or_nint);
/* calc_h2o_scat, kernels.cu:3404 / computation.py:1390 */
int hx_calc_h2o_scat(hx_context* ctx, const double* temp, const double* press, const double* wave,
                     double* scat_cross, const double* vmr, double mass_h2o, int nbin,
                     int nlay_or_nint);
/* add_to_mixed_scat, kernels.cu:3444 / computation.py:1425 */
int hx_add_to_mixed_scat(hx_context* ctx, const double* vmr, const double* scat_cross_spec,
                         double* scat_cross, int nbin, int nlay_or_nint);

/* ---- (4) fused fast path --------------------------------------------------------------------
 * One hx_rt object = one batch of `ncol` independent atmosphere columns (planets / T-P profiles of
 * a parameter sweep) that share wavelength grid, Gauss points and opacity tables.  Non-isothermal
 * layers, iterative flux solver (the reference's defaults for an iterative run, param.dat:28,:109).
 *
 *   hx_rt_create            allocate the device-resident state (tiles, node arrays, per-column vectors)
 *   hx_rt_set_*             hand over shared tables / per-column inputs (copied to the device once)
 *   hx_rt_refresh           the every-10th-iteration block of radiation_loop (computation.py:860-879):
 *                           opacity interpolation or on-the-fly mixing, transmission coefficients,
 *                           delta z / altitude, direct beam  ->  compact coefficient tiles
 *   hx_rt_step              one iteration (computation.py:856-857, :880-888, :926-932): T_int, Planck
 *                           interpolation, (3*scat+1) two-stream sweeps, quadrature, totals, temperature
 *                           step, per-column convergence count -- no host round trip
 *   hx_rt_get / hx_rt_export_* read results back in the REFERENCE's layouts (hx_rt_get also: "flux_launch_policy",
 *                           two doubles -- whether the flux kernel's launches walk the grid back and forth and how
 *                           many MiB of up-flux state a launch leaves in the Infinity Cache; a choice of the batch,
 *                           HELIOS_RT_SERPENTINE / HELIOS_RT_STATE_CACHE_MB override it, results do not depend on it)
 */
typedef struct hx_rt_dims {
    int32_t nbin, ny, nlayer, ncol;
    int32_t ntemp, npress;        /* opacity-table grid */
    int32_t plancktable_dim, plancktable_step;
    int32_t nspecies;             /* 0 = premixed table; >0 = on-the-fly mixing of that many species */
    int32_t reserved[7];
} hx_rt_dims;

typedef struct hx_rt_flags {
    int32_t scat, dir_beam, clouds, scat_corr, geom_zenith_corr, smooth, real_star, planet_type_gas;
    int32_t kcoeff_mixing_ro;     /* 1 = random overlap, 0 = correlated-k (param.dat:110) */
    int32_t debug;
    int32_t iso;                  /* 1 = isothermal layers (fband_iso / calc_trans_iso / fdir_iso), read.py:888-935 */
    int32_t singlewalk;           /* 1 = post-processing run type: 1000*scat+1 sweeps, no temperature iteration
                                     (computation.py:531-537) */
    int32_t matrix;               /* 1 = `flux calculation method = matrix`: one tridiagonal solve per spectral point and
                                     iteration (fband_matrix_*, computation.py:625-710) instead of the sweeps.  The batch
                                     then holds the reference's per-half-layer arrays (calc_trans_*' outputs, ~20 arrays of
                                     ny*nbin*nlayer doubles per column) in place of the compact coefficient tiles */
    int32_t reserved[3];
    double epsi, epsi2, g_0, i2s_transition, w_0_limit, w_0_scat_limit, delta_tau_limit;
    double reserved_d[9];
} hx_rt_flags;

/* per-column scalars, host array of ncol structs */
typedef struct hx_rt_column {
    double g, a, R_planet, R_star, T_star, f_factor, mu_star, F_intern;
    double rad_convergence_limit, physical_tstep;
    int32_t adapt_interval, foreplay, no_atmo, reserved_i;
    double reserved_d[4];
} hx_rt_column;

int hx_rt_struct_sizes(int* dims_size, int* flags_size, int* column_size);
int hx_rt_create(hx_context* ctx, const hx_rt_dims* dims, const hx_rt_flags* flags,
                 const hx_rt_column* columns, hx_rt** out_rt);
int hx_rt_destroy(hx_rt* rt);
/* shared (per batch) host inputs */
int hx_rt_set_grid(hx_rt* rt, const double* opac_interwave, const double* opac_deltawave,
                   const double* opac_wave, const double* gauss_y, const double* gauss_weight,
                   const double* ktemp, const double* kpress);
int hx_rt_set_premixed_tables(hx_rt* rt, const double* opac_k, const double* opac_scat_cross,
                              const double* opac_meanmass);
/* one absorbing/scattering species of the on-the-fly mix (index s in species-file order, the first
 * absorber at s = 0, read.py:1373).  opacity_pretab may be NULL for a pure scatterer; scat_cross[nbin]
 * may be NULL for a non-scatterer; is_h2o selects calc_h2o_scat; is_cia forces correlated-k
 * (computation.py:1343); in_mu = contributes to the mean molecular mass (host_functions.py:940) */
int hx_rt_set_species(hx_rt* rt, int s, const double* opacity_pretab, const double* scat_cross,
                      double weight, int is_h2o, int is_cia, int in_mu);
/* Synthetic tables (bench.py, tests; helios_amd/synthetic.py:ktable): kappa[t][p][x][y] = kxy[y + ny*x] * ftp[p + npress*t],
 * formed on the device -- one fp64 product per entry, the same bits as the host array -- instead of a 0.96 GB host table and
 * its copy per species.  Otherwise as hx_rt_set_species / hx_rt_set_premixed_tables (the reference reads its tables from
 * HDF5 files, source/read.py:1041-1103; these two have no counterpart there). */
int hx_rt_set_species_separable(hx_rt* rt, int s, const double* kxy, const double* ftp, const double* scat_cross,
                                double weight, int is_h2o, int is_cia, int in_mu);
int hx_rt_set_premixed_separable(hx_rt* rt, const double* kxy, const double* ftp, const double* opac_scat_cross,
                                 const double* opac_meanmass);
/* A6 on the device: calculate_vmr_for_all_species + interpolate_grid_to_lay_or_int (source/host_functions.py:874-910).
 * vmr_pretab[p + npress * t] on the opacity tables' (T, P) grid, as source/read.py keeps it per FastChem species
 * (Species.vmr_pretab); the species' profile is then interpolated at every refresh from the device's temperatures,
 * bilinear in (T, log10 P), clamped at the table edges.  NULL returns the species to hx_rt_set_column_vmr's profiles. */
int hx_rt_set_species_vmr_table(hx_rt* rt, int s, const double* vmr_pretab);
/* the same per column: a parameter sweep gives every column its own chemistry (FastChem directory, metallicity, C/O --
 * source/read.py:577-606 reads one table per species and run), so the device keeps one table per (species, column);
 * col < 0 = all columns (what hx_rt_set_species_vmr_table does).  A column of a tabulated species that never received a
 * table reads zeros. */
int hx_rt_set_column_vmr_table(hx_rt* rt, int col, int s, const double* vmr_pretab);
/* per-column host inputs; col < 0 broadcasts to all columns.  vmr_* : [nspecies][nlayer] / [nspecies][ninterface] */
int hx_rt_set_column_profile(hx_rt* rt, int col, const double* p_lay, const double* p_int,
                             const double* T_lay, const double* surf_albedo,
                             const double* starflux);
int hx_rt_set_column_vmr(hx_rt* rt, int col, const double* vmr_lay, const double* vmr_int);
int hx_rt_set_column_clouds(hx_rt* rt, int col, const double* abs_cross_lay,
                            const double* abs_cross_int, const double* scat_cross_lay,
                            const double* scat_cross_int, const double* g_0_lay,
                            const double* g_0_int);
int hx_rt_set_column_heating(hx_rt* rt, int col, const double* F_add_heat_lay,
                             const double* F_add_heat_sum);
int hx_rt_set_temperatures(hx_rt* rt, int col, const double* T_lay);
int hx_rt_set_convergence_limit(hx_rt* rt, int col, double limit);
/* builds the Planck table (+ incident-energy correction) on the device */
int hx_rt_build_planck_table(hx_rt* rt, int energy_correction);
int hx_rt_refresh(hx_rt* rt);
/* itervalue is the reference's quant.iter_value; step_temperature = 0 skips C5 (post-processing) */
int hx_rt_step(hx_rt* rt, int itervalue, int step_temperature);
/* nsteps iterations starting at `itervalue`, refreshing whenever iter % 10 == 0; no host sync */
int hx_rt_run(hx_rt* rt, int itervalue, int nsteps);
/* Convection loop (computation.py:992-1174) without host round trips.  One iteration = hx_rt_conv_adjust (convective
 * adjustment of the profile: host_functions.py:337-635 run by one workgroup per column) + hx_rt_conv_advance (T_int,
 * Planck, [refresh], sweeps, totals, mark_convective_layers, check_for_radiative_eq, conv_temp_iter).  The two halves
 * are separate so that a host step (FastChem mixing ratios of the adjusted profile) can sit between them; hx_rt_conv_run
 * chains them.  When the reference's loop condition turns false the column is frozen ("done" = 1, "iters_done" = the
 * reference's final iter_value).  Inputs beyond the radiation loop's: hx_rt_set_state "kappa_lay", "kappa_int",
 * "conv_layer" (flags persist between iterations), "dampara" (<= 0: the reference's automatic choice). */
/* kappa (= delad) and c_p tables of `kappa value = file | water_atmo` (read.py:1105-1193), value[p + npress * t]; with
 * them kappa_lay / kappa_int / c_p_lay follow the profile on the device (kappa_interpol, cp_interpol) */
int hx_rt_set_kappa_table(hx_rt* rt, const double* entr_temp, int entr_ntemp, const double* entr_press,
                          int entr_npress, const double* entr_kappa, const double* entr_c_p);
/* kappa and c_p of every column at its current temperatures (one column: Compute.interpolate_kappa_and_cp,
 * computation.py:199-250); the convection loop refreshes them itself every 10th iteration */
int hx_rt_kappa_cp_refresh(hx_rt* rt);
int hx_rt_conv_adjust(hx_rt* rt, int itervalue);
int hx_rt_conv_advance(hx_rt* rt, int itervalue);
int hx_rt_conv_run(hx_rt* rt, int itervalue, int nsteps);
/* number of layers (+ghost layer) per column whose convergence flag is set: out[ncol] (blocks) */
int hx_rt_converged_layers(hx_rt* rt, int* out_counts);
/* named read-back in the reference's layout: "T_lay","T_int","F_up_band","F_down_band",
 * "F_dir_band","F_up_tot","F_down_tot","F_net","F_net_diff","planckband_lay","planckband_int",
 * "opac_wg_lay","opac_wg_int","scat_cross_lay","scat_cross_int","meanmolmass_lay",
 * "meanmolmass_int","F_up_wg","F_down_wg","Fc_up_wg","Fc_down_wg","F_dir_wg","Fc_dir_wg","abort",
 * "delta_z_lay","z_lay","g_0_tot_lay","g_0_tot_int","delta_t_prefactor","T_store","F_smooth_sum",
 * "conv_layer","conv_unstable","marked_red" (int32[nlayer+1]),"done","iters_done".
 * `out` is a HOST buffer of `out_bytes`; returns HX_E_ARG if the name is unknown or the size wrong. */
int hx_rt_get(hx_rt* rt, int col, const char* name, void* out, size_t out_bytes);
/* named write of host data into the batch: "T_lay", "c_p_lay", "delta_t_prefactor", "T_store", "done", "kappa_lay",
 * "kappa_int", "conv_layer", "conv_unstable", "add_heat_dens", "dampara" (per column, col = -1: all), "planck_grid",
 * "keep_down" (int32), and "restart" (int32): every column back to the state of a fresh batch -- the sweeps' flux state,
 * the time-step state and the convergence flags zeroed; temperatures stay the caller's. */
int hx_rt_set_state(hx_rt* rt, int col, const char* name, const void* in, size_t in_bytes);
/* device pointer of a named internal array (column `col`) for use with the per-stage entry points */
int hx_rt_device_ptr(hx_rt* rt, int col, const char* name, void** out_dptr);
/* the tiling hx_rt_create would choose for the flux kernel -- lanes per spectral point and half-layer rows per lane, i.e.
 * the instantiation k_rt_flux<rows, lanes> -- for a batch of `ncol` columns of `nlayer` layers (`iso`: one segment per
 * layer; `dir_beam`: the direct beam adds two planes per tile and favours fewer rows per lane).  Host only, no device
 * needed: tests hold the choice to the code objects' register notes. */
int hx_rt_flux_geometry(int nlayer, int iso, int dir_beam, int ny, int nbin, int ncol, int* out_lanes, int* out_rows);
/* algorithmic / actual HBM byte counts of the last refresh and step (for the roofline report) */
int hx_rt_traffic_model(hx_rt* rt, double* step_bytes_algorithmic, double* step_bytes_actual,
                        double* refresh_bytes_algorithmic, double* refresh_bytes_actual);
/* per-kernel HIP-event timing of the fused path: enable, then read the averages (ms) */
int hx_rt_profile(hx_rt* rt, int enable);
int hx_rt_profile_read(hx_rt* rt, const char* kernel, double* out_avg_ms, int* out_count);

#ifdef __cplusplus
}
#endif
#endif
