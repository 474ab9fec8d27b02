"""Builds the product's `Store` (helios_amd.quantities) from a tests/cases.py problem, so that the SAME
seeded column can be run through `Compute.radiation_loop` -- the driver a user calls -- and through the
reference-kernel loop of tests/loop_driver.py."""
import numpy as np

from helios_amd import quantities as quant_mod


def store_from_case(ctx, c, crit_relaxation_numbers=(), max_nr_iterations=20000, name="case", convection=0,
                    kappa=2.0 / 7.0):
    q = quant_mod.Store(ctx)
    i32, f64 = np.int32, np.float64
    for k in ("nbin", "ny", "nlayer", "ntemp", "npress", "iso", "scat", "dir_beam", "clouds", "scat_corr",
              "geom_zenith_corr", "real_star", "plancktable_dim", "plancktable_step", "adapt_interval", "foreplay",
              "smooth"):
        setattr(q, k, i32(c[k]))
    for k in ("g_0", "epsi", "epsi2", "i2s_transition", "w_0_limit", "w_0_scat_limit", "delta_tau_limit", "f_factor",
              "mu_star", "g", "R_planet", "R_star", "a", "T_star", "F_intern", "rad_convergence_limit",
              "physical_tstep"):
        setattr(q, k, f64(c[k]))
    q.no_atmo_mode = i32(c.no_atmo)
    q.singlewalk = i32(0)
    q.convection = i32(convection)
    q.opacity_mixing = "premixed"
    q.kcoeff_mixing = "correlated-k"
    q.flux_calc_method = c.get("flux_calc_method", "iteration")
    q.planet_type = "gas"
    q.name = name
    q.debug = i32(0)
    q.coupling = i32(0)
    q.add_heating = i32(0)
    q.energy_correction = i32(1)
    q.input_kappa_value = np.float64(kappa)
    q.input_dampara = "automatic"
    q.runtime_limit = f64(0)
    q.max_nr_iterations = i32(max_nr_iterations)
    q.crit_relaxation_numbers = [i32(r) for r in crit_relaxation_numbers]
    q.species_list = []
    for k in ("p_lay", "p_int", "delta_colmass", "delta_col_upper", "delta_col_lower", "ktemp", "kpress", "opac_k",
              "gauss_y", "gauss_weight", "opac_wave", "opac_deltawave", "opac_interwave", "opac_scat_cross",
              "opac_meanmass", "c_p_lay", "starflux", "T_lay", "surf_albedo", "abs_cross_all_clouds_lay",
              "scat_cross_all_clouds_lay", "g_0_all_clouds_lay", "abs_cross_all_clouds_int",
              "scat_cross_all_clouds_int", "g_0_all_clouds_int"):
        setattr(q, k, np.array(c[k], f64).copy())
    q.kappa_lay = np.full(int(q.nlayer), float(kappa))
    q.kappa_int = np.full(int(q.nlayer) + 1, float(kappa))
    q.dimensions()
    q.create_zero_arrays()
    q.convert_input_list_to_array()
    q.copy_host_to_device()
    q.allocate_on_device()
    return q
