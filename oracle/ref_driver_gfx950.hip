// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// The reference's own kernels, built by the image's real GPU toolchain and run as real GPU threads.
//
// oracle/Makefile compiles THIS file with
//     hipcc -x hip --offload-arch=gfx950 -include hip/hip_runtime.h -fPIC -shared
// and this file includes the reference's single native source, source/kernels.cu, where it lies under
// /root/reference (path given on the command line).  Nothing stands in for a header, a library or a
// tool: hip/hip_runtime.h is ROCm's own header and kernels.cu includes only <stdio.h>.  No line of
// kernels.cu is copied into the repository or edited; its 35 __global__ functions are compiled
// unmodified for gfx950, with the hardware's atomicCAS / __syncthreads / blockIdx.
// Output: oracle/_ref/libhelios_ref_gfx950.so (git-ignored, travels to the GPU box with gpurun).
//
// Each `ref_<kernel>` entry launches the reference kernel of that name with the block and grid that
// the reference's own launcher passes (source/computation.py, lines cited per entry), then waits for
// it, as the reference does (`cuda.Context.synchronize()` after every launch).  Pointer arguments are
// DEVICE pointers obtained from refgpu_alloc; oracle/__init__.py (`oracle.refgpu`) moves numpy
// arrays in and out.  Same entry names and argument lists as the host build (oracle/ref_driver.cpp),
// so tests/impls.py::RefImpl and tests/cases.py drive either build through the same code.
#include HELIOS_REF_KERNELS  // = "/root/reference/source/kernels.cu", set by oracle/Makefile

#include <cstdio>
#include <cstdint>

namespace {

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

int g_status = 0;  // first HIP error seen since the last refgpu_status() call

inline void note(hipError_t e) {
    if (e != hipSuccess && g_status == 0) g_status = (int)e;
}

// launch + synchronise, recording (not hiding) any error
#define REF_LAUNCH(kernel, grid, block, ...)                 \
    do {                                                     \
        hipLaunchKernelGGL(kernel, grid, block, 0, 0, __VA_ARGS__); \
        note(hipGetLastError());                             \
        note(hipDeviceSynchronize());                        \
    } while (0)

}  // namespace

extern "C" {

// ---- memory + status (what gpuarray.to_gpu / .get() / mem_alloc are to the reference) ----------
int refgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
void* refgpu_alloc(size_t nbytes) {
    void* p = nullptr;
    note(hipMalloc(&p, nbytes ? nbytes : 8));
    return p;
}
void refgpu_free(void* p) { note(hipFree(p)); }
void refgpu_h2d(void* dst, const void* src, size_t nbytes) { note(hipMemcpy(dst, src, nbytes, hipMemcpyHostToDevice)); }
void refgpu_d2h(void* dst, const void* src, size_t nbytes) { note(hipMemcpy(dst, src, nbytes, hipMemcpyDeviceToHost)); }
void refgpu_memset0(void* p, size_t nbytes) { note(hipMemset(p, 0, nbytes)); }
// returns and clears the first HIP error recorded since the previous call (0 = none)
int refgpu_status(void) {
    int s = g_status;
    g_status = 0;
    return s;
}

// computation.py:39-60 (ten launches p_iter = 0..9)
void ref_plancktable(double* planck_grid, double* lambda_edge, double* deltalambda, int nwave,
                     double Tstar, int dim, int step) {
    for (int p_iter = 0; p_iter < 10; p_iter++)
        REF_LAUNCH(plancktable, dim3(cdiv(nwave, 16), cdiv(dim / 10 + 1, 16), 1), dim3(16, 16, 1),
                   planck_grid, lambda_edge, deltalambda, nwave, Tstar, p_iter, dim, step);
}

// computation.py:62-82.  Every thread sums ALL bins and then rescales its own bin in place
// (kernels.cu:434-466): whatever the hardware makes of that race is the reference's behaviour.
void ref_corr_inc_energy(double* planck_grid, double* starflux, double* deltalambda, int realstar,
                         int nwave, double Tstar, int dim) {
    REF_LAUNCH(corr_inc_energy, dim3(cdiv(nwave, 16), 1, 1), dim3(16, 1, 1), planck_grid, starflux,
               deltalambda, realstar, nwave, Tstar, dim);
}

// computation.py:331-362
void ref_calc_total_g_0_of_gas_and_clouds(double* scat_cross, double* g_0_all_clouds, double*
                                          scat_cross_all_clouds, double* g_0_tot, double g_0, int
                                          nbin, int nlay_or_nint) {
    REF_LAUNCH(calc_total_g_0_of_gas_and_clouds, dim3(cdiv(nbin, 16), cdiv(nlay_or_nint, 16), 1),
               dim3(16, 16, 1), scat_cross, g_0_all_clouds, scat_cross_all_clouds, g_0_tot, g_0,
               nbin, nlay_or_nint);
}

// computation.py:104-117
void ref_temp_inter(double* tlay, double* tint, int numinterfaces, int itervalue) {
    REF_LAUNCH(temp_inter, dim3(cdiv(numinterfaces, 16), 1, 1), dim3(16, 1, 1), tlay, tint,
               numinterfaces, itervalue);
}

// computation.py:119-161
void ref_opac_interpol(double* temp, double* opactemp, double* press, double* opacpress, double*
                       ktable, double* opac, double* crosstable, double* scat_cross, int npress, int
                       ntemp, int ny, int nbin, int nlay_or_nint) {
    REF_LAUNCH(opac_interpol, dim3(cdiv(nbin, 16), cdiv(nlay_or_nint, 16), 1), dim3(16, 16, 1),
               temp, opactemp, press, opacpress, ktable, opac, crosstable, scat_cross, npress,
               ntemp, ny, nbin, nlay_or_nint);
}

// computation.py:163-197
void ref_meanmolmass_interpol(double* temp, double* opactemp, double* meanmolmass, double*
                              opac_meanmass, double* press, double* opacpress, int npress, int
                              ntemp, int ninterface) {
    REF_LAUNCH(meanmolmass_interpol, dim3(cdiv(ninterface, 16), 1, 1), dim3(16, 1, 1), temp,
               opactemp, meanmolmass, opac_meanmass, press, opacpress, npress, ntemp, ninterface);
}

// computation.py:199-250 (kappa / c_p tables; only with "kappa value = file")
void ref_kappa_interpol(double* temp, double* entr_temp, double* press, double* entr_press, double*
                        kappa, double* entr_kappa, int entr_npress, int entr_ntemp, int
                        nlay_or_nint) {
    REF_LAUNCH(kappa_interpol, dim3(cdiv(nlay_or_nint, 16), 1, 1), dim3(16, 1, 1), temp, entr_temp,
               press, entr_press, kappa, entr_kappa, entr_npress, entr_ntemp, nlay_or_nint);
}

void ref_cp_interpol(double* temp, double* entr_temp, double* press, double* entr_press, double*
                     cp_lay, double* entr_cp, int entr_npress, int entr_ntemp, int nlayer) {
    REF_LAUNCH(cp_interpol, dim3(cdiv(nlayer, 16), 1, 1), dim3(16, 1, 1), temp, entr_temp, press,
               entr_press, cp_lay, entr_cp, entr_npress, entr_ntemp, nlayer);
}

// computation.py:252-292 (diagnostics of the kappa-file modes)
void ref_entropy_interpol(double* temp, double* entr_temp, double* press, double* entr_press,
                          double* entropy, double* entr_entropy, int entr_npress, int entr_ntemp,
                          int nlayer) {
    REF_LAUNCH(entropy_interpol, dim3(cdiv(nlayer, 16), 1, 1), dim3(16, 1, 1), temp, entr_temp,
               press, entr_press, entropy, entr_entropy, entr_npress, entr_ntemp, nlayer);
}

void ref_phase_number_interpol(double* temp, double* entr_temp, double* press, double* entr_press,
                               double* state, double* entr_state, int entr_npress, int entr_ntemp,
                               int nlayer) {
    REF_LAUNCH(phase_number_interpol, dim3(cdiv(nlayer, 16), 1, 1), dim3(16, 1, 1), temp, entr_temp,
               press, entr_press, state, entr_state, entr_npress, entr_ntemp, nlayer);
}

// computation.py:294-313
void ref_planck_interpol_layer(double* temp, double* planckband_lay, double* planck_grid, double*
                               starflux, int realstar, int numlayers, int nwave, int dim, int step)
                               {
    REF_LAUNCH(planck_interpol_layer, dim3(cdiv(nwave, 16), cdiv(numlayers + 2, 16), 1), dim3(16,
               16, 1), temp, planckband_lay, planck_grid, starflux, realstar, numlayers, nwave, dim,
               step);
}

// computation.py:315-329
void ref_planck_interpol_interface(double* temp, double* planckband_int, double* planck_grid, int
                                   numinterfaces, int nwave, int dim, int step) {
    REF_LAUNCH(planck_interpol_interface, dim3(cdiv(nwave, 16), cdiv(numinterfaces, 16), 1),
               dim3(16, 16, 1), temp, planckband_int, planck_grid, numinterfaces, nwave, dim, step);
}

// computation.py:370-406
void ref_calc_trans_iso(double* trans_wg, double* delta_tau_wg, double* M_term, double* N_term,
                        double* P_term, double* G_plus, double* G_minus, double* delta_colmass,
                        double* opac_wg_lay, double* meanmolmass_lay, double* scat_cross_lay,
                        double* abs_cross_all_clouds_lay, double* scat_cross_all_clouds_lay, double*
                        delta_tau_all_clouds, double* w_0, double* g_0_tot_lay, int* scat_trigger,
                        double g_0, double epsi, double epsi2, double mu_star, double w_0_limit,
                        double w_0_scat_limit, int scat, int nbin, int ny, int nlayer, int clouds,
                        int scat_corr, int debug, double i2s_transition) {
    REF_LAUNCH(calc_trans_iso, dim3(cdiv(nbin, 16), cdiv(ny, 4), cdiv(nlayer, 4)), dim3(16, 4, 4),
               trans_wg, delta_tau_wg, M_term, N_term, P_term, G_plus, G_minus, delta_colmass,
               opac_wg_lay, meanmolmass_lay, scat_cross_lay, abs_cross_all_clouds_lay,
               scat_cross_all_clouds_lay, delta_tau_all_clouds, w_0, g_0_tot_lay, scat_trigger, g_0,
               epsi, epsi2, mu_star, w_0_limit, w_0_scat_limit, scat, nbin, ny, nlayer, clouds,
               scat_corr, debug, i2s_transition);
}

// computation.py:408-460
void ref_calc_trans_noniso( double* trans_wg_upper, double* trans_wg_lower, double*
                           delta_tau_wg_upper, double* delta_tau_wg_lower, double* M_upper, double*
                           M_lower, double* N_upper, double* N_lower, double* P_upper, double*
                           P_lower, double* G_plus_upper, double* G_plus_lower, double*
                           G_minus_upper, double* G_minus_lower, double* delta_col_upper, double*
                           delta_col_lower, double* opac_wg_lay, double* opac_wg_int, double*
                           meanmolmass_lay, double* meanmolmass_int, double* scat_cross_lay, double*
                           scat_cross_int, double* abs_cross_all_clouds_lay, double*
                           abs_cross_all_clouds_int, double* scat_cross_all_clouds_lay, double*
                           scat_cross_all_clouds_int, double* delta_tau_all_clouds_upper, double*
                           delta_tau_all_clouds_lower, double* w_0_upper, double* w_0_lower, double*
                           g_0_tot_lay, double* g_0_tot_int, int* scat_trigger, double g_0, double
                           epsi, double epsi2, double mu_star, double w_0_limit, double
                           w_0_scat_limit, int scat, int nbin, int ny, int nlayer, int clouds, int
                           scat_corr, int debug, double i2s_transition) {
    REF_LAUNCH(calc_trans_noniso, dim3(cdiv(nbin, 16), cdiv(ny, 4), cdiv(nlayer, 4)), dim3(16, 4,
               4), trans_wg_upper, trans_wg_lower, delta_tau_wg_upper, delta_tau_wg_lower, M_upper,
               M_lower, N_upper, N_lower, P_upper, P_lower, G_plus_upper, G_plus_lower,
               G_minus_upper, G_minus_lower, delta_col_upper, delta_col_lower, opac_wg_lay,
               opac_wg_int, meanmolmass_lay, meanmolmass_int, scat_cross_lay, scat_cross_int,
               abs_cross_all_clouds_lay, abs_cross_all_clouds_int, scat_cross_all_clouds_lay,
               scat_cross_all_clouds_int, delta_tau_all_clouds_upper, delta_tau_all_clouds_lower,
               w_0_upper, w_0_lower, g_0_tot_lay, g_0_tot_int, scat_trigger, g_0, epsi, epsi2,
               mu_star, w_0_limit, w_0_scat_limit, scat, nbin, ny, nlayer, clouds, scat_corr, debug,
               i2s_transition);
}

// computation.py:464-479
void ref_calc_delta_z(double* tlay, double* pint, double* play, double* meanmolmass_lay, double*
                      delta_z_lay, double g, int nlayer) {
    REF_LAUNCH(calc_delta_z, dim3(cdiv(nlayer, 16), 1, 1), dim3(16, 1, 1), tlay, pint, play,
               meanmolmass_lay, delta_z_lay, g, nlayer);
}

// computation.py:484-502
void ref_fdir_iso(double* F_dir_wg, double* planckband_lay, double* delta_tau_wg, double* z_lay,
                  double mu_star, double R_planet, double R_star, double a, int dir_beam, int
                  geom_zenith_corr, int ninterface, int nbin, int ny) {
    REF_LAUNCH(fdir_iso, dim3(cdiv(ninterface, 4), cdiv(nbin, 32), cdiv(ny, 4)), dim3(4, 32, 4),
               F_dir_wg, planckband_lay, delta_tau_wg, z_lay, mu_star, R_planet, R_star, a,
               dir_beam, geom_zenith_corr, ninterface, nbin, ny);
}

// computation.py:504-524
void ref_fdir_noniso(double* F_dir_wg, double* Fc_dir_wg, double* planckband_lay, double*
                     delta_tau_wg_upper, double* delta_tau_wg_lower, double* z_lay, double mu_star,
                     double R_planet, double R_star, double a, int dir_beam, int geom_zenith_corr,
                     int ninterface, int nbin, int ny) {
    REF_LAUNCH(fdir_noniso, dim3(cdiv(ninterface, 4), cdiv(nbin, 32), cdiv(ny, 4)), dim3(4, 32, 4),
               F_dir_wg, Fc_dir_wg, planckband_lay, delta_tau_wg_upper, delta_tau_wg_lower, z_lay,
               mu_star, R_planet, R_star, a, dir_beam, geom_zenith_corr, ninterface, nbin, ny);
}

// computation.py:539-571 (one sweep; the caller repeats it 3*scat+1 times)
void ref_fband_iso(double* F_down_wg, double* F_up_wg, double* F_dir_wg, double* planckband_lay,
                   double* w_0, double* M_term, double* N_term, double* P_term, double* G_plus,
                   double* G_minus, double* surf_albedo, double* g_0_tot_lay, double g_0, int
                   singlewalk, double Rstar, double a, int numinterfaces, int nbin, double f_factor,
                   double mu_star, int ny, double epsi, int dir_beam, int clouds, int scat_corr, int
                   debug, double i2s_transition) {
    REF_LAUNCH(fband_iso, dim3(cdiv(nbin, 16), cdiv(ny, 16), 1), dim3(16, 16, 1), F_down_wg,
               F_up_wg, F_dir_wg, planckband_lay, w_0, M_term, N_term, P_term, G_plus, G_minus,
               surf_albedo, g_0_tot_lay, g_0, singlewalk, Rstar, a, numinterfaces, nbin, f_factor,
               mu_star, ny, epsi, dir_beam, clouds, scat_corr, debug, i2s_transition);
}

// computation.py:573-621 (one sweep)
void ref_fband_noniso(double* F_down_wg, double* F_up_wg, double* Fc_down_wg, double* Fc_up_wg,
                      double* F_dir_wg, double* Fc_dir_wg, double* planckband_lay, double*
                      planckband_int, double* w_0_upper, double* w_0_lower, double*
                      delta_tau_wg_upper, double* delta_tau_wg_lower, double*
                      delta_tau_all_clouds_upper, double* delta_tau_all_clouds_lower, double*
                      M_upper, double* M_lower, double* N_upper, double* N_lower, double* P_upper,
                      double* P_lower, double* G_plus_upper, double* G_plus_lower, double*
                      G_minus_upper, double* G_minus_lower, double* surf_albedo, double*
                      g_0_tot_lay, double* g_0_tot_int, double g_0, int singlewalk, double Rstar,
                      double a, int numinterfaces, int nbin, double f_factor, double mu_star, int
                      ny, double epsi, double delta_tau_limit, int dir_beam, int clouds, int
                      scat_corr, int debug, double i2s_transition) {
    REF_LAUNCH(fband_noniso, dim3(cdiv(nbin, 16), cdiv(ny, 16), 1), dim3(16, 16, 1), F_down_wg,
               F_up_wg, Fc_down_wg, Fc_up_wg, F_dir_wg, Fc_dir_wg, planckband_lay, planckband_int,
               w_0_upper, w_0_lower, delta_tau_wg_upper, delta_tau_wg_lower,
               delta_tau_all_clouds_upper, delta_tau_all_clouds_lower, M_upper, M_lower, N_upper,
               N_lower, P_upper, P_lower, G_plus_upper, G_plus_lower, G_minus_upper, G_minus_lower,
               surf_albedo, g_0_tot_lay, g_0_tot_int, g_0, singlewalk, Rstar, a, numinterfaces,
               nbin, f_factor, mu_star, ny, epsi, delta_tau_limit, dir_beam, clouds, scat_corr,
               debug, i2s_transition);
}

// computation.py:625-665
void ref_fband_matrix_iso(double* F_down_wg, double* F_up_wg, double* F_dir_wg, double*
                          planckband_lay, double* w_0, double* M_term, double* N_term, double*
                          P_term, double* G_plus, double* G_minus, double* g_0_tot_lay, double*
                          alpha, double* beta, double* source_term_down, double* source_term_up,
                          double* c_prime, double* d_prime, int* scat_trigger, double* trans_wg,
                          double* surf_albedo, double g_0, int singlewalk, double Rstar, double a,
                          int numinterfaces, int nbin, double f_factor, double mu_star, int ny,
                          double epsi, int dir_beam, int clouds, int scat_corr, int debug, double
                          i2s_transition) {
    REF_LAUNCH(fband_matrix_iso, dim3(cdiv(nbin, 16), cdiv(ny, 16), 1), dim3(16, 16, 1), F_down_wg,
               F_up_wg, F_dir_wg, planckband_lay, w_0, M_term, N_term, P_term, G_plus, G_minus,
               g_0_tot_lay, alpha, beta, source_term_down, source_term_up, c_prime, d_prime,
               scat_trigger, trans_wg, surf_albedo, g_0, singlewalk, Rstar, a, numinterfaces, nbin,
               f_factor, mu_star, ny, epsi, dir_beam, clouds, scat_corr, debug, i2s_transition);
}

// computation.py:667-727
void ref_fband_matrix_noniso( double* F_down_wg, double* F_up_wg, double* Fc_down_wg, double*
                             Fc_up_wg, double* F_dir_wg, double* Fc_dir_wg, double* planckband_lay,
                             double* planckband_int, double* w_0_upper, double* w_0_lower, double*
                             delta_tau_wg_upper, double* delta_tau_wg_lower, double*
                             delta_tau_all_clouds_upper, double* delta_tau_all_clouds_lower, double*
                             M_upper, double* M_lower, double* N_upper, double* N_lower, double*
                             P_upper, double* P_lower, double* G_plus_upper, double* G_plus_lower,
                             double* G_minus_upper, double* G_minus_lower, double* g_0_tot_lay,
                             double* g_0_tot_int, double* alpha, double* beta, double*
                             source_term_down, double* source_term_up, double* c_prime, double*
                             d_prime, int* scat_trigger, double* trans_wg_upper, double*
                             trans_wg_lower, double* surf_albedo, double g_0, int singlewalk, double
                             Rstar, double a, int numinterfaces, int nbin, double f_factor, double
                             mu_star, int ny, double epsi, double delta_tau_limit, int dir_beam, int
                             clouds, int scat_corr, int debug, double i2s_transition) {
    REF_LAUNCH(fband_matrix_noniso, dim3(cdiv(nbin, 16), cdiv(ny, 16), 1), dim3(16, 16, 1),
               F_down_wg, F_up_wg, Fc_down_wg, Fc_up_wg, F_dir_wg, Fc_dir_wg, planckband_lay,
               planckband_int, w_0_upper, w_0_lower, delta_tau_wg_upper, delta_tau_wg_lower,
               delta_tau_all_clouds_upper, delta_tau_all_clouds_lower, M_upper, M_lower, N_upper,
               N_lower, P_upper, P_lower, G_plus_upper, G_plus_lower, G_minus_upper, G_minus_lower,
               g_0_tot_lay, g_0_tot_int, alpha, beta, source_term_down, source_term_up, c_prime,
               d_prime, scat_trigger, trans_wg_upper, trans_wg_lower, surf_albedo, g_0, singlewalk,
               Rstar, a, numinterfaces, nbin, f_factor, mu_star, ny, epsi, delta_tau_limit,
               dir_beam, clouds, scat_corr, debug, i2s_transition);
}

// computation.py:731-757: ONE block of (32,4,8) threads with block barriers between phases.
void ref_integrate_flux_double(double* deltalambda, double* F_down_tot, double* F_up_tot, double*
                               F_net, double* F_down_wg, double* F_up_wg, double* F_dir_wg, double*
                               F_down_band, double* F_up_band, double* F_dir_band, double*
                               gauss_weight, int nbin, int numinterfaces, int ny) {
    REF_LAUNCH(integrate_flux_double, dim3(1, 1, 1), dim3(32, 4, 8), deltalambda, F_down_tot,
               F_up_tot, F_net, F_down_wg, F_up_wg, F_dir_wg, F_down_band, F_up_band, F_dir_band,
               gauss_weight, nbin, numinterfaces, ny);
}

// computation.py:759-797 (smooth must be 0 here: see SURVEY.md Q11 -- the smooth==1 path is racy)
void ref_rad_temp_iter(double* F_down_tot, double* F_up_tot, double* F_net, double* F_net_diff,
                       double* tlay, double* play, double* tint, double* pint, int* abrt, double*
                       T_store, double* deltat_prefactor, double* F_add_heat_lay, double*
                       F_add_heat_sum, double* F_smooth, double* F_smooth_sum, double* c_p_lay,
                       double* meanmolmass_lay, int itervalue, double f_factor, int foreplay, double
                       g, int numlayers, double physical_tstep, double local_limit, int
                       adapt_interval, int smooth, int dim, int step, double F_intern, int no_atmo)
                       {
    REF_LAUNCH(rad_temp_iter, dim3(cdiv(numlayers + 1, 16), 1, 1), dim3(16, 1, 1), F_down_tot,
               F_up_tot, F_net, F_net_diff, tlay, play, tint, pint, abrt, T_store, deltat_prefactor,
               F_add_heat_lay, F_add_heat_sum, F_smooth, F_smooth_sum, c_p_lay, meanmolmass_lay,
               itervalue, f_factor, foreplay, g, numlayers, physical_tstep, local_limit,
               adapt_interval, smooth, dim, step, F_intern, no_atmo);
}

// computation.py:799-825
void ref_conv_temp_iter(double* F_down_tot, double* F_up_tot, double* F_net, double* F_net_diff,
                        double* tlay, double* play, double* pint, double* T_store, double*
                        deltat_prefactor, int* marked_red, double* F_add_heat_lay, double* F_smooth,
                        double* F_smooth_sum, int numlayers, int itervalue, int adapt_interval, int
                        smooth, double F_intern) {
    REF_LAUNCH(conv_temp_iter, dim3(cdiv(numlayers + 1, 16), 1, 1), dim3(16, 1, 1), F_down_tot,
               F_up_tot, F_net, F_net_diff, tlay, play, pint, T_store, deltat_prefactor, marked_red,
               F_add_heat_lay, F_smooth, F_smooth_sum, numlayers, itervalue, adapt_interval, smooth,
               F_intern);
}

// computation.py:1176-1214
void ref_integrate_optdepth_transmission_iso(double* trans_wg, double* trans_band, double*
                                             delta_tau_wg, double* delta_tau_band, double*
                                             gauss_weight, int nbin, int nlayer, int ny) {
    REF_LAUNCH(integrate_optdepth_transmission_iso, dim3(cdiv(nbin, 16), cdiv(nlayer, 16), 1),
               dim3(16, 16, 1), trans_wg, trans_band, delta_tau_wg, delta_tau_band, gauss_weight,
               nbin, nlayer, ny);
}

void ref_integrate_optdepth_transmission_noniso( double* trans_wg_upper, double* trans_wg_lower,
                                                double* trans_band, double* delta_tau_wg_upper,
                                                double* delta_tau_wg_lower, double* delta_tau_band,
                                                double* gauss_weight, double* delta_tau_all_clouds,
                                                double* delta_tau_all_clouds_upper, double*
                                                delta_tau_all_clouds_lower, int nbin, int nlayer,
                                                int ny) {
    REF_LAUNCH(integrate_optdepth_transmission_noniso, dim3(cdiv(nbin, 16), cdiv(nlayer, 16), 1),
               dim3(16, 16, 1),  trans_wg_upper, trans_wg_lower, trans_band, delta_tau_wg_upper,
               delta_tau_wg_lower, delta_tau_band, gauss_weight, delta_tau_all_clouds,
               delta_tau_all_clouds_upper, delta_tau_all_clouds_lower, nbin, nlayer, ny);
}

// computation.py:1216-1252
void ref_calc_contr_func_iso(double* trans_wg, double* trans_weight_band, double* contr_func_band,
                             double* gauss_weight, double* planckband_lay, double epsi, int nbin,
                             int nlayer, int ny) {
    REF_LAUNCH(calc_contr_func_iso, dim3(cdiv(nbin, 16), cdiv(nlayer, 16), 1), dim3(16, 16, 1),
               trans_wg, trans_weight_band, contr_func_band, gauss_weight, planckband_lay, epsi,
               nbin, nlayer, ny);
}

void ref_calc_contr_func_noniso(double* trans_wg_upper, double* trans_wg_lower, double*
                                trans_weight_band, double* contr_func_band, double* gauss_weight,
                                double* planckband_lay, double epsi, int nbin, int nlayer, int ny) {
    REF_LAUNCH(calc_contr_func_noniso, dim3(cdiv(nbin, 16), cdiv(nlayer, 16), 1), dim3(16, 16, 1),
               trans_wg_upper, trans_wg_lower, trans_weight_band, contr_func_band, gauss_weight,
               planckband_lay, epsi, nbin, nlayer, ny);
}

// computation.py:1254-1281
void ref_calc_mean_opacities(double* planck_opac_T_pl, double* ross_opac_T_pl, double*
                             planck_opac_T_star, double* ross_opac_T_star, double* opac_wg_lay,
                             double* abs_cross_all_clouds_lay, double* meanmolmass_lay, double*
                             planckband_lay, double* opac_interwave, double* opac_deltawave, double*
                             T_lay, double* gauss_weight, double* gauss_y, double* opac_band_lay,
                             int nlayer, int nbin, int ny, double T_star) {
    REF_LAUNCH(calc_mean_opacities, dim3(cdiv(nlayer, 16), 1, 1), dim3(16, 1, 1), planck_opac_T_pl,
               ross_opac_T_pl, planck_opac_T_star, ross_opac_T_star, opac_wg_lay,
               abs_cross_all_clouds_lay, meanmolmass_lay, planckband_lay, opac_interwave,
               opac_deltawave, T_lay, gauss_weight, gauss_y, opac_band_lay, nlayer, nbin, ny,
               T_star);
}

// computation.py:1283-1296
void ref_integrate_beamflux(double* F_dir_tot, double* F_dir_band, double* deltalambda, double*
                            gauss_weight, int nbin, int numinterfaces) {
    REF_LAUNCH(integrate_beamflux, dim3(cdiv(numinterfaces, 16), 1, 1), dim3(16, 1, 1), F_dir_tot,
               F_dir_band, deltalambda, gauss_weight, nbin, numinterfaces);
}

// computation.py:1298-1336
void ref_opac_species_interpol(double* temp, double* opactemp, double* press, double* opacpress,
                               double* opac_opacity_pretab, double* opac_spec_wg, int npress, int
                               ntemp, int ny, int nbin, int nlay_or_nint) {
    REF_LAUNCH(opac_species_interpol, dim3(cdiv(nbin, 16), cdiv(nlay_or_nint, 16), 1), dim3(16, 16,
               1), temp, opactemp, press, opacpress, opac_opacity_pretab, opac_spec_wg, npress,
               ntemp, ny, nbin, nlay_or_nint);
}

// computation.py:1338-1388
void ref_add_to_mixed_opac(double* vmr, double* opac_spec, double* opac_wg, double* meanmolmass,
                           double* gauss_weight, double* gauss_y, double mass_spec, int s, int
                           ro_method, int ny, int nbin, int nlay_or_nint) {
    REF_LAUNCH(add_to_mixed_opac, dim3(cdiv(nbin, 32), cdiv(nlay_or_nint, 32), 1), dim3(32, 32, 1),
               vmr, opac_spec, opac_wg, meanmolmass, gauss_weight, gauss_y, mass_spec, s, ro_method,
               ny, nbin, nlay_or_nint);
}

// computation.py:1390-1423
void ref_calc_h2o_scat(double* temp, double* press, double* wave, double* scat_cross, double* vmr,
                       double mass_h2o, int nbin, int nlay_or_nint) {
    REF_LAUNCH(calc_h2o_scat, dim3(cdiv(nbin, 16), cdiv(nlay_or_nint, 16), 1), dim3(16, 16, 1),
               temp, press, wave, scat_cross, vmr, mass_h2o, nbin, nlay_or_nint);
}

// computation.py:1425-1452
void ref_add_to_mixed_scat(double* vmr, double* scat_cross_spec, double* scat_cross, int nbin, int
                           nlay_or_nint) {
    REF_LAUNCH(add_to_mixed_scat, dim3(cdiv(nbin, 16), cdiv(nlay_or_nint, 16), 1), dim3(16, 16, 1),
               vmr, scat_cross_spec, scat_cross, nbin, nlay_or_nint);
}

}  // extern "C"
