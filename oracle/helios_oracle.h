/* TEST INFRASTRUCTURE ONLY -- the CPU oracle for the HELIOS radiative-transfer hot path.
 *
 * Plain-C restatement of the algorithm of the reference's device code
 * (/root/reference/source/kernels.cu), one function per reference kernel, operating on the
 * reference's own flat fp64 array layouts:
 *     wg arrays   [y + ny*x + ny*nbin*i]        band arrays  [x + nbin*i]
 *     Planck      [i + x*(nlayer+2)] / [i + x*ninterface]     Planck table [x + t*nbin]
 *     k-tables    [y + ny*x + ny*nbin*p + ny*nbin*npress*t]
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (helios_amd/ + libhelios_hip.so) never does.
 *
 * Parity pin: every function here is held to the golden vectors committed under tests/golden/, which
 * the REFERENCE ITSELF produced on an MI355X: its source/kernels.cu compiled unmodified by hipcc for
 * gfx950 (oracle/_ref/libhelios_ref_gfx950.so, recipe in oracle/Makefile, launch geometry of
 * source/computation.py in oracle/ref_driver_gfx950.hip; generator tests/golden/make_golden.py
 * --backend gfx950) -- stage chains, mixing branches, the matrix solver, 64-bin x 100-layer columns and
 * whole radiation loops to convergence (tests/test_golden.py, tests/test_loop_golden.py).  On the GPU box
 * the same library runs next to the HIP kernels (tests/test_gpu_reference.py).  A host build of the same
 * file through oracle/ref_shim.h (oracle/_ref/libhelios_ref.so, tests/test_oracle_vs_ref.py) is a CPU-side
 * cross-check only.  The reference ships no tests or golden vectors of its own.
 */
#ifndef HELIOS_ORACLE_H
#define HELIOS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* --- set-up ---------------------------------------------------------------------------------- */
void orc_planck_table(double* planck_grid, const double* lambda_edge, const double* deltalambda,
                      int nbin, double T_star, int dim, int step);
void orc_corr_inc_energy(double* planck_grid, double* starflux, const double* deltalambda,
                         int realstar, int nbin, double T_star, int dim);

/* --- every iteration ------------------------------------------------------------------------- */
void orc_temp_inter(const double* T_lay, double* T_int, int ninterface);
void orc_planck_interpol_layer(const double* T_lay, double* planckband_lay, const double* planck_grid,
                               const double* starflux, int realstar, int nlayer, int nbin, int dim,
                               int step);
void orc_planck_interpol_interface(const double* T_int, double* planckband_int,
                                   const double* planck_grid, int ninterface, int nbin, int dim,
                                   int step);

/* --- opacity refresh (every 10th iteration) -------------------------------------------------- */
void orc_opac_interpol(const double* temp, const double* opactemp, const double* press,
                       const double* opacpress, const double* ktable, double* opac,
                       const double* crosstable, double* scat_cross, int npress, int ntemp, int ny,
                       int nbin, int nlev);
void orc_meanmolmass_interpol(const double* temp, const double* opactemp, double* meanmolmass,
                              const double* opac_meanmass, const double* press,
                              const double* opacpress, int npress, int ntemp, int nlev);
void orc_kappa_interpol(const double* temp, const double* entr_temp, const double* press,
                        const double* entr_press, double* kappa, const double* entr_kappa,
                        int entr_npress, int entr_ntemp, int nlev);
void orc_cp_interpol(const double* temp, const double* entr_temp, const double* press,
                     const double* entr_press, double* cp, const double* entr_cp, int entr_npress,
                     int entr_ntemp, int nlev);
void orc_entropy_interpol(const double* temp, const double* entr_temp, const double* press,
                          const double* entr_press, double* entropy, const double* entr_entropy,
                          int entr_npress, int entr_ntemp, int nlayer);
void orc_phase_number_interpol(const double* temp, const double* entr_temp, const double* press,
                               const double* entr_press, double* state, const double* entr_state,
                               int entr_npress, int entr_ntemp, int nlayer);
void orc_opac_species_interpol(const double* temp, const double* opactemp, const double* press,
                               const double* opacpress, const double* pretab, double* opac_spec,
                               int npress, int ntemp, int ny, int nbin, int nlev);
void orc_add_to_mixed_opac(const double* vmr, const double* opac_spec, double* opac_wg,
                           const double* meanmolmass, const double* gauss_weight,
                           const double* gauss_y, double mass_spec, int s, int ro_method, int ny,
                           int nbin, int nlev);
void orc_calc_h2o_scat(const double* temp, const double* press, const double* wave,
                       double* scat_cross, const double* vmr, double mass_h2o, int nbin, int nlev);
void orc_add_to_mixed_scat(const double* vmr, const double* scat_cross_spec, double* scat_cross,
                           int nbin, int nlev);
void orc_calc_total_g0(const double* scat_cross, const double* g_0_all_clouds,
                       const double* scat_cross_all_clouds, double* g_0_tot, double g_0, int nbin,
                       int nlev);

void orc_calc_trans_iso(double* trans_wg, double* delta_tau_wg, double* M_term, double* N_term,
                        double* P_term, double* G_plus, double* G_minus, const double* delta_colmass,
                        const double* opac_wg_lay, const double* meanmolmass_lay,
                        const double* scat_cross_lay, const double* abs_cross_all_clouds_lay,
                        const double* scat_cross_all_clouds_lay, double* delta_tau_all_clouds,
                        double* w_0, const double* g_0_tot_lay, int* scat_trigger, double g_0,
                        double epsi, double epsi2, double mu_star, double w_0_limit,
                        double w_0_scat_limit, int scat, int nbin, int ny, int nlayer, int clouds,
                        int scat_corr, double i2s_transition);
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
    int nlayer, int clouds, int scat_corr, double i2s_transition);
void orc_calc_delta_z(const double* T_lay, const double* p_int, const double* meanmolmass_lay,
                      double* delta_z_lay, double g, int nlayer);
void orc_fdir_iso(double* F_dir_wg, const double* planckband_lay, const double* delta_tau_wg,
                  const double* z_lay, double mu_star, double R_planet, double R_star, double a,
                  int dir_beam, int geom_zenith_corr, int ninterface, int nbin, int ny);
void orc_fdir_noniso(double* F_dir_wg, double* Fc_dir_wg, const double* planckband_lay,
                     const double* delta_tau_wg_upper, const double* delta_tau_wg_lower,
                     const double* z_lay, double mu_star, double R_planet, double R_star, double a,
                     int dir_beam, int geom_zenith_corr, int ninterface, int nbin, int ny);

/* --- flux solve (one sweep per call) --------------------------------------------------------- */
void orc_fband_iso(double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                   const double* planckband_lay, const double* w_0, const double* M_term,
                   const double* N_term, const double* P_term, const double* G_plus,
                   const double* G_minus, const double* surf_albedo, const double* g_0_tot_lay,
                   double g_0, double Rstar, double a, int ninterface, int nbin, double f_factor,
                   double mu_star, int ny, double epsi, int dir_beam, int clouds, int scat_corr,
                   double i2s_transition);
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
                      double i2s_transition);

/* matrix (Thomas) form of the flux solve, source/kernels.cu:1803-2424 */
void orc_fband_matrix_iso(double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
                          const double* planckband_lay, const double* w_0, const double* M_term,
                          const double* N_term, const double* P_term, const double* G_plus,
                          const double* G_minus, const double* g_0_tot_lay, double* alpha, double* beta,
                          double* source_term_down, double* source_term_up, double* c_prime,
                          double* d_prime, const int* scat_trigger, const double* trans_wg,
                          const double* surf_albedo, double g_0, double Rstar, double a, int ninterface,
                          int nbin, double f_factor, double mu_star, int ny, double epsi, int dir_beam,
                          int clouds, int scat_corr, double i2s_transition);
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
    int scat_corr, double i2s_transition);

void orc_integrate_flux(const double* deltalambda, double* F_down_tot, double* F_up_tot,
                        double* F_net, const double* F_down_wg, const double* F_up_wg,
                        const double* F_dir_wg, double* F_down_band, double* F_up_band,
                        double* F_dir_band, const double* gauss_weight, int nbin, int ninterface,
                        int ny);

/* --- temperature step ------------------------------------------------------------------------ */
void orc_rad_temp_iter(const double* F_down_tot, const double* F_up_tot, const double* F_net,
                       double* F_net_diff, double* T_lay, const double* p_lay, const double* p_int,
                       int* abrt, double* T_store, double* deltat_prefactor,
                       const double* F_add_heat_lay, const double* F_add_heat_sum, double* F_smooth,
                       double* F_smooth_sum, const double* c_p_lay, const double* meanmolmass_lay,
                       int itervalue, int foreplay, double g, int nlayer, double physical_tstep,
                       double local_limit, int adapt_interval, int smooth, int dim, int step,
                       double F_intern, int no_atmo);
void orc_conv_temp_iter(const double* F_net, double* F_net_diff, double* T_lay, const double* p_lay,
                        const double* p_int, double* T_store, double* deltat_prefactor,
                        const int* marked_red, const double* F_add_heat_lay, double* F_smooth,
                        double* F_smooth_sum, int nlayer, int itervalue, int adapt_interval,
                        int smooth, double F_intern);

/* --- post-loop diagnostics ("next" rows of SURVEY.md §8(f)) ----------------------------------- */
void orc_integrate_optdepth_transmission_iso(const double* trans_wg, double* trans_band,
                                             const double* delta_tau_wg, double* delta_tau_band,
                                             const double* gauss_weight, int nbin, int nlayer,
                                             int ny);
void orc_integrate_optdepth_transmission_noniso(
    const double* trans_wg_upper, const double* trans_wg_lower, double* trans_band,
    const double* delta_tau_wg_upper, const double* delta_tau_wg_lower, double* delta_tau_band,
    const double* gauss_weight, double* delta_tau_all_clouds,
    const double* delta_tau_all_clouds_upper, const double* delta_tau_all_clouds_lower, int nbin,
    int nlayer, int ny);
void orc_calc_contr_func_iso(const double* trans_wg, double* trans_weight_band,
                             double* contr_func_band, const double* gauss_weight,
                             const double* planckband_lay, double epsi, int nbin, int nlayer, int ny);
void orc_calc_contr_func_noniso(const double* trans_wg_upper, const double* trans_wg_lower,
                                double* trans_weight_band, double* contr_func_band,
                                const double* gauss_weight, const double* planckband_lay, double epsi,
                                int nbin, int nlayer, int ny);
void orc_calc_mean_opacities(double* planck_opac_T_pl, double* ross_opac_T_pl,
                             double* planck_opac_T_star, double* ross_opac_T_star,
                             const double* opac_wg_lay, const double* abs_cross_all_clouds_lay,
                             const double* meanmolmass_lay, const double* planckband_lay,
                             const double* opac_interwave, const double* opac_deltawave,
                             const double* T_lay, const double* gauss_weight, const double* gauss_y,
                             double* opac_band_lay, int nlayer, int nbin, int ny, double T_star);
void orc_integrate_beamflux(double* F_dir_tot, const double* F_dir_band, const double* deltalambda,
                            int nbin, int ninterface);

/* number of OpenMP threads the oracle will use (for the cpu_baseline report) */
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
