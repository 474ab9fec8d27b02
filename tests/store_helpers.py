"""Builds the product's `Store` (helios_amd.quantities) from a tests/cases.py problem, so that the SAME
seeded column can be run through `Compute.radiation_loop` -- the driver a user calls -- and through the
reference-kernel loop of tests/loop_driver.py."""
import numpy as np

from helios_amd import quantities as quant_mod


def species_list_from_case(c):
    """the case's synthetic absorbers and scatterers (bench.build_case / cases.add_species) as the reader's Species
    objects: constant mixing ratios; tables that the case only names by seed are generated here"""
    from helios_amd import synthetic as syn
    from helios_amd.read import Species
    out = []
    L = int(c.nlayer)
    for k, sp in enumerate(c.species):
        tab = sp["pretab"]
        if isinstance(tab, str):
            tab = syn.ktable(np.random.default_rng(sp["table_seed"]), c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y)
        o = Species(name=sp.get("name", "SPEC%02d" % k), absorbing="yes" if tab is not None else "no",
                    scattering="yes" if (sp["scat"] is not None or sp.get("is_h2o")) else "no", weight=float(sp["weight"]),
                    source_for_vmr="constant", mixing_ratio=float(sp["vmr"]))
        o.opacity_pretab = tab
        o.vmr_layer = np.full(L, float(sp["vmr"]))
        o.vmr_interface = np.full(L + 1, float(sp["vmr"]))
        if sp["scat"] is not None:
            o.scat_cross_sect_pretab = np.asarray(sp["scat"], float)
            o.scat_cross_sect_layer = np.tile(o.scat_cross_sect_pretab, L)
            o.scat_cross_sect_interface = np.tile(o.scat_cross_sect_pretab, L + 1)
        out.append(o)
    return out


def store_from_case(ctx, c, crit_relaxation_numbers=(), max_nr_iterations=20000, name="case", convection=0,
                    kappa=2.0 / 7.0, on_the_fly=False):
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
    q.opacity_mixing = "on-the-fly" if on_the_fly else "premixed"
    q.kcoeff_mixing = "RO" if on_the_fly else "correlated-k"
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
    q.species_list = species_list_from_case(c) if on_the_fly else []
    if "delta_colmass" not in c:        # bench.build_case leaves them to the library (host_functions.py:731-735)
        c = dict(c)
        p_lay, p_int = np.asarray(c["p_lay"], float), np.asarray(c["p_int"], float)
        c["delta_colmass"] = (p_int[:-1] - p_int[1:]) / c["g"]
        c["delta_col_upper"] = (p_lay - p_int[1:]) / c["g"]
        c["delta_col_lower"] = (p_int[:-1] - p_lay) / c["g"]
    for k in ("p_lay", "p_int", "delta_colmass", "delta_col_upper", "delta_col_lower", "ktemp", "kpress", "opac_k",
              "gauss_y", "gauss_weight", "opac_wave", "opac_deltawave", "opac_interwave", "opac_scat_cross",
              "opac_meanmass", "c_p_lay", "starflux", "T_lay", "surf_albedo", "abs_cross_all_clouds_lay",
              "scat_cross_all_clouds_lay", "g_0_all_clouds_lay", "abs_cross_all_clouds_int",
              "scat_cross_all_clouds_int", "g_0_all_clouds_int"):
        if k not in c and "all_clouds" in k:     # a case without clouds
            n = int(c["nbin"]) * (int(c["nlayer"]) + (1 if k.endswith("_int") else 0))
            setattr(q, k, np.zeros(n))
            continue
        setattr(q, k, np.array(c[k], f64).copy())
    q.kappa_lay = np.full(int(q.nlayer), float(kappa))
    q.kappa_int = np.full(int(q.nlayer) + 1, float(kappa))
    q.dimensions()
    q.create_zero_arrays()
    q.convert_input_list_to_array()
    q.copy_host_to_device()
    q.allocate_on_device()
    return q
