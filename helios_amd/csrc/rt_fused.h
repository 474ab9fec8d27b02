// Host-side state of the fused fast path (hx_rt).  See DESIGN.md "Data layout in HBM".
#pragma once
#include <string>
#include <vector>

#include "hx_common.h"

namespace hx {

// How the (bin x, Gauss point y, half-layer h) space is cut into wavefront tiles.
//
//   * k lanes cooperate on one spectral point (x,y); lane j owns the ROWS half-layers
//     [j*ROWS, (j+1)*ROWS), kept in registers for all sweeps;
//   * a wavefront holds S = 64/k spectral points; lane = s*k + j;
//   * a workgroup covers nxb bins x ypb Gauss points (G = nxb*ypb spectral points, NW wavefronts);
//     nparts = ny/ypb workgroups share one bin and each writes a partial Gauss sum.
struct TileGeom {
    int k, ROWS, S;
    int nxb, ypb, nparts, G, NW, threads;
    int nblk_x;       // workgroups along x per column = ceil(nbin / nxb)
    int nblk;         // workgroups per column = nblk_x * nparts
    int nplane;       // coefficient planes per tile: alpha, beta, u' [, v'] [, dd, du]
    int has_vp, pl_vp, pl_dd;
    size_t tile_rows; // ROWS
    size_t coef_elems_per_col, flux_elems_per_col;  // doubles
};

struct Species {
    double* pretab = nullptr;      // device, [t][p][x][y] flat (reference order), or null
    double* scat_cross = nullptr;  // device [nbin], or null
    double* vmr_tab = nullptr;     // device [column][ntemp][npress] mixing ratios on the opacity grid (hx_rt_set_column_vmr_table), or null
    bool vmr_from_tab = false;
    double weight = 0.0;
    int is_h2o = 0, is_cia = 0, in_mu = 1;
    bool absorbing = false, scattering = false;
};

// `flux calculation method = matrix` (hx_rt_flags.matrix): the reference's per-half-layer arrays in the reference's
// layouts, filled by the per-stage kernels of calc_trans_* (every refresh) and read by the per-stage matrix solver
// (every iteration), both launched from the device-resident loop.  Per column unless noted; with isothermal layers the
// `_u` members are the reference's single arrays.
struct MatrixArrays {
    double *trans_u = nullptr, *trans_l = nullptr, *M_u = nullptr, *M_l = nullptr, *N_u = nullptr, *N_l = nullptr,
           *P_u = nullptr, *P_l = nullptr, *Gp_u = nullptr, *Gp_l = nullptr, *Gm_u = nullptr, *Gm_l = nullptr,
           *w0_u = nullptr, *w0_l = nullptr;                                         // ny*nbin*nlayer
    double *dtc_u = nullptr, *dtc_l = nullptr;                                       // nbin*nlayer (cloud optical depths)
    double* dcol_iso = nullptr;                                                      // nlayer (delta_colmass, isothermal layers)
    int* trigger = nullptr;                                                          // ny*nbin
    double *F_down = nullptr, *F_up = nullptr, *Fc_down = nullptr, *Fc_up = nullptr;  // ny*nbin*ninterface
    double *pb_lay = nullptr, *pb_int = nullptr;                                     // nbin*(nlayer+2), nbin*ninterface
    // work arrays of the elimination, shared by the columns (solved one after the other on one stream)
    // (alpha, beta and the source terms, which nothing reads after the elimination, are not materialised here)
    double *c_prime = nullptr, *d_prime = nullptr;
};

struct ProfileEntry {
    std::string name;
    hipEvent_t e0, e1;
};

}  // namespace hx

struct hx_rt {
    hx_context* ctx;
    hx_rt_dims d;
    hx_rt_flags f;
    std::vector<hx_rt_column> cols;
    int X, Y, L, I, H, C, nsweep;
    hx::TileGeom g;
    bool have_grid = false, have_tables = false, have_planck = false, refreshed = false;
    bool keep_down = false;
    bool opac_stale = false;      // opac_wg_lay/int not materialised since the last (fused) refresh
    void* tp_lay = nullptr;       // TPIndex [C][I]
    void* tp_int = nullptr;
    double *T_lay_ref = nullptr, *T_int_ref = nullptr;  // temperatures of the last refresh
    int nchunk;  // x-chunks of the totals reduction
    int coef_tpb = 4;  // k_rt_coef: tiles (wavefronts) per workgroup
    hipError_t shmem_rc = hipSuccess;
    bool cloud_lds = true;       // k_rt_coef stages the clouds' half-layer terms in LDS when they fit (HELIOS_RT_CLOUD_LDS)
    bool serpentine = false;     // k_rt_flux walks its grid back and forth from launch to launch (HELIOS_RT_SERPENTINE)
    unsigned flux_launches = 0;
    double state_cache_mb = 0;   // MiB of up-flux state the tail of a k_rt_flux launch leaves in the Infinity Cache
    bool generic_scans = false;  // k_rt_flux with the runtime-k scans also where k = 16 or 32 (HELIOS_RT_GENERIC_SCANS)
    bool conv_shmem_raised = false;  // dynamic-LDS limit of the convection kernels lifted (deep atmospheres)
    bool coef_shmem_raised = false;  // k_rt_coef's dynamic-LDS limit lifted above 64 KiB (deep atmospheres)

    // shared device arrays
    double *interwave = nullptr, *deltawave = nullptr, *wave = nullptr, *gauss_y = nullptr,
           *gauss_w = nullptr, *ktemp = nullptr, *kpress = nullptr;
    double *opac_k = nullptr, *opac_scat_cross = nullptr, *opac_meanmass = nullptr;
    double* planck_grid = nullptr;  // [(dim+1) * X]; row dim = stellar row of column 0
    std::vector<hx::Species> species;
    void* species_dev = nullptr;   // SpeciesDev [nspecies]: what the batched mixing kernels read (rt_species.h)
    int* abs_list = nullptr;       // indices of the absorbing species
    int nabs = 0;
    bool species_dev_stale = true;
    double *fac_lay = nullptr, *fac_int = nullptr;        // vmr * mass / mu per (column, level, species)
    double *spec_lay = nullptr, *spec_int = nullptr;      // one species interpolated, [Y*X*I]
    double *sc_spec_lay = nullptr, *sc_spec_int = nullptr;  // one species' scattering cross-sections

    // per-column device arrays (column stride given in comments, in doubles)
    hx_rt_column* colpar = nullptr;  // [C]
    double *p_lay = nullptr, *p_int = nullptr, *dcol_u = nullptr, *dcol_l = nullptr;  // L, I, L, L
    double *T_lay = nullptr, *T_int = nullptr;                                        // L+1, I
    double *surf_albedo = nullptr, *starflux = nullptr, *Bstar = nullptr;             // X each
    double *opac_wg_lay = nullptr, *opac_wg_int = nullptr;                            // Y*X*I each
    double *scat_cross_lay = nullptr, *scat_cross_int = nullptr;                      // X*I each
    double *mmm_lay = nullptr, *mmm_int = nullptr;                                    // I each
    double *cl_abs_lay = nullptr, *cl_abs_int = nullptr, *cl_sc_lay = nullptr, *cl_sc_int = nullptr,
           *cl_g0_lay = nullptr, *cl_g0_int = nullptr, *g0_tot_lay = nullptr, *g0_tot_int = nullptr;  // X*I
    double *half_ray = nullptr, *half_g0 = nullptr, *half_cab = nullptr, *half_csc = nullptr;  // X*H, bin-major
    double *vmr_lay = nullptr, *vmr_int = nullptr;                                    // nspecies*I
    double *delta_z = nullptr, *z_lay = nullptr;                                      // L
    double *dtau_u = nullptr, *dtau_l = nullptr;                                      // Y*X*L (beam only)
    double *F_dir_wg = nullptr, *Fc_dir_wg = nullptr;                                 // Y*X*I (beam only)
    double* F_dir_band_n = nullptr;                                                   // [x][i], X*I
    double* Bn = nullptr;       // node Planck [x][H+3]
    double* coef = nullptr;     // coefficient tiles
    double* Utile = nullptr;    // up-flux state tiles
    double* Dtile = nullptr;    // down-flux tiles (only with keep_down)
    double *U0 = nullptr, *boaK = nullptr, *Fdir0 = nullptr;  // per spectral point, Y*X
    double *F_down_band_n = nullptr, *F_up_band_n = nullptr;  // [x][i]
    double* tot_part = nullptr;  // [chunk][dir][i]
    double *F_up_tot = nullptr, *F_down_tot = nullptr, *F_net = nullptr, *F_net_diff = nullptr;  // I
    double *T_store = nullptr, *prefactor = nullptr;                                            // L+1
    double *F_add_heat_lay = nullptr, *F_add_heat_sum = nullptr, *F_smooth = nullptr,
           *F_smooth_sum = nullptr, *c_p_lay = nullptr;                                         // L
    int *abort_flags = nullptr, *conv_count = nullptr, *done = nullptr, *iters_done = nullptr;  // L+1,1,1,1
    // convection loop (hx_rt_conv_*): adiabatic coefficients, layer flags, damping parameter of the flux fudging
    double *kappa_lay = nullptr, *kappa_int = nullptr, *dampara = nullptr;                     // L, I, 1
    int *conv_unstable = nullptr, *conv_layer = nullptr, *marked_red = nullptr;                  // L+1 each
    // kappa (= delad) / c_p table of `kappa value = file`: [p + npress * t] on (entr_temp, entr_press)
    double *entr_temp = nullptr, *entr_press = nullptr, *entr_kappa = nullptr, *entr_c_p = nullptr;
    int entr_ntemp = 0, entr_npress = 0;
    double* add_heat_dens = nullptr;   // L: additional heating density [erg cm^-3 s^-1]; flux = density * layer height
    bool has_heating = false;

    bool matrix = false;           // hx_rt_flags.matrix
    // the matrix method as three scans on the coefficient tiles (k_rt_flux<.., true>): the default.
    // HELIOS_RT_MATRIX=stage: the reference-shaped per-stage kernels (calc_trans_*, one Thomas elimination per thread with its
    // work arrays in HBM) inside the loop instead, as until round 4
    bool matrix_scan = false;
    bool matrix_keep_state = false;   // the direct solve stores its up-fluxes too (debug = 1: the negative-flux counts read them)
    int* zero_flags = nullptr;        // [C] zeros: the `done` flags of a launch that must cover every column
    // The iteration index lives on the device (iter_dev[0]: index of the next iteration; k_rt_nodes, the first kernel of an
    // iteration, moves it to iter_dev[1] and increments): the kernels of an iteration then have the SAME arguments every
    // time, and the nine refresh-free iterations between two opacity refreshes are replayed as one hipGraph where the
    // launches, not the GPU, bound the loop (small grids: hx_rt_run)
    int* iter_dev = nullptr;
    int iter_dev_expected = -1;    // what iter_dev[0] holds as far as the host knows (-1: unknown)
    hipGraphExec_t iter_graph = nullptr;
    hipGraphExec_t decade_graph = nullptr;   // refresh + ten iterations (hx_rt_run entered at a refresh boundary)
    long long iter_graph_replays = 0, decade_graph_replays = 0;
    long long iter_graph_builds = 0, decade_graph_builds = 0;
    // every setter that changes what a launch is given moves the generation on; a capture is good while it holds the
    // generation it was taken at (one number per capture: building one graph says nothing about the other)
    long long graph_gen = 1, iter_graph_gen = 0, decade_graph_gen = 0;
    // counts everything that can change the spectral fluxes (solves, refreshes, graph replays, setters); the direct solve's
    // tiles re-created for hx_rt_get carry the count they were made at
    long long solve_serial = 1, matrix_tiles_serial = 0;
    int use_graph = -1;            // -1: decide from the grid size (HELIOS_RT_GRAPH=0|1 overrides), 0 / 1
    std::vector<char> have_albedo; // per column: a surface albedo has been handed over (the matrix method divides by it)
    hx::MatrixArrays mx;

    // profiling
    bool profiling = false;
    std::vector<hx::ProfileEntry> prof;
    std::vector<std::pair<std::string, std::pair<double, int>>> prof_acc;
    std::vector<void*> allocs;
};
