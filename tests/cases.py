"""Seeded small problems and the stage chain of one radiation_loop iteration, written once and run
through any implementation of the stage functions (oracle port, reference build, HIP library) so
that parity tests compare like with like.  Stage names/arguments are those of oracle/helios_oracle.h.
"""
import numpy as np

from helios_amd import phys_const as pc
from helios_amd import synthetic as syn


class Case(dict):
    """dict with attribute access"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def copy(self):
        c = Case()
        for k, v in self.items():
            c[k] = v.copy() if isinstance(v, np.ndarray) else v
        return c


def make_case(nbin=13, nlayer=9, ny=20, ntemp=7, npress=6, seed=20241, scat=1, dir_beam=0,
              clouds=0, scat_corr=0, g_0=0.0, iso=0, geom_zenith_corr=0, albedo=0.0,
              plancktable_dim=400, plancktable_step=10, thin_top=False, T_star=5000.0,
              zenith_deg=60.0, f_factor=0.5, T_intern=100.0, p_boa=1e9):
    rng = np.random.default_rng(seed)
    c = Case()
    c.nbin, c.nlayer, c.ninterface, c.ny = nbin, nlayer, nlayer + 1, ny
    c.ntemp, c.npress = ntemp, npress
    c.iso, c.scat, c.dir_beam, c.clouds, c.scat_corr = iso, scat, dir_beam, clouds, scat_corr
    c.geom_zenith_corr = geom_zenith_corr
    c.g_0 = g_0
    c.epsi, c.epsi2 = 0.5, 0.5
    c.i2s_transition = 0.1
    c.w_0_limit, c.w_0_scat_limit, c.delta_tau_limit = 1.0 - 1e-10, 1e-3, 1e-4
    c.f_factor = f_factor
    c.mu_star = float(np.cos(np.pi - zenith_deg * np.pi / 180.0)) if dir_beam else -0.5
    c.g = 1000.0
    c.R_planet = 1.0 * pc.R_JUP
    c.R_star = 1.0 * pc.R_SUN
    c.a = 0.05 * pc.AU
    c.T_star = T_star
    c.real_star = 0
    c.F_intern = pc.SIGMA_SB * T_intern ** 4
    c.plancktable_dim, c.plancktable_step = plancktable_dim, plancktable_step
    c.rad_convergence_limit = 1e-8
    c.adapt_interval = 20
    c.foreplay = 0
    c.smooth = 0
    c.physical_tstep = 0.0
    c.no_atmo = 0

    c.opac_interwave, c.opac_wave, c.opac_deltawave = syn.wavelength_grid(nbin)
    c.gauss_y, c.gauss_weight = syn.gauss_points(ny)
    c.ktemp, c.kpress = syn.tp_grid(ntemp, npress)
    c.opac_k = syn.ktable(rng, nbin, ny, c.ktemp, c.kpress, c.gauss_y)
    c.opac_scat_cross = syn.rayleigh_table(c.opac_wave, ntemp, npress)
    if scat:  # make scattering matter in the tiny cases: boost Rayleigh so that w0 spans (0, 1)
        c.opac_scat_cross = c.opac_scat_cross * 1e3
    c.opac_meanmass = syn.meanmass_table(ntemp, npress) * (1.0 + 0.1 * rng.uniform(-1, 1, ntemp * npress))
    p_toa = 1e-1 if not thin_top else 1e-6
    c.p_lay, c.p_int = syn.pressure_levels(p_boa, p_toa, nlayer)
    c.delta_colmass = (c.p_int[:-1] - c.p_int[1:]) / c.g
    c.delta_col_upper = (c.p_lay - c.p_int[1:]) / c.g
    c.delta_col_lower = (c.p_int[:-1] - c.p_lay) / c.g
    # a non-trivial temperature profile (hot deep, cool top) with some noise; index nlayer = surface
    T = 400.0 + 1400.0 * (np.log10(c.p_lay) + 1.0) / 10.0 + rng.uniform(-30, 30, nlayer)
    T = np.maximum(T, 150.0 + rng.uniform(0, 20, nlayer))
    c.T_lay = np.append(T, T[0] + 55.0)
    c.T_int = np.zeros(nlayer + 1)
    c.surf_albedo = np.full(nbin, float(albedo)) if np.isscalar(albedo) else np.asarray(albedo, float)
    c.starflux = np.zeros(nbin)
    c.z_lay = np.zeros(nlayer)
    c.F_add_heat_lay = np.zeros(nlayer)
    c.F_add_heat_sum = np.zeros(nlayer)
    c.F_smooth = np.zeros(nlayer)
    c.F_smooth_sum = np.zeros(nlayer)
    c.c_p_lay = np.full(nlayer, 3.5 * pc.R_UNIV)
    if clouds:
        cl = syn.cloud_arrays(nbin, nlayer, c.opac_wave, c.p_lay, c.p_int, rng,
                              decks=((1e7, 1.0), (1e4, 0.4)), sigma0=3e-26)
        c.update(cl)
    else:
        for nm, n in (("lay", nlayer), ("int", nlayer + 1)):
            c["abs_cross_all_clouds_" + nm] = np.zeros(n * nbin)
            c["scat_cross_all_clouds_" + nm] = np.zeros(n * nbin)
            c["g_0_all_clouds_" + nm] = np.zeros(n * nbin)
    return c


def alloc_state(c):
    """zero-initialised work/output arrays in the reference's sizes (source/quantities.py:411-461,
    :593-665; wg 'layer' arrays are over-allocated to ninterface there, Q2)."""
    X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
    s = Case()
    wg = Y * X * I
    for nm in ("F_up_wg", "F_down_wg", "F_dir_wg", "Fc_up_wg", "Fc_down_wg", "Fc_dir_wg",
               "opac_wg_lay", "opac_wg_int",
               "trans_wg", "delta_tau_wg", "w_0", "M_term", "N_term", "P_term", "G_plus", "G_minus",
               "trans_wg_upper", "trans_wg_lower", "delta_tau_wg_upper", "delta_tau_wg_lower",
               "M_upper", "M_lower", "N_upper", "N_lower", "P_upper", "P_lower",
               "G_plus_upper", "G_plus_lower", "G_minus_upper", "G_minus_lower",
               "w_0_upper", "w_0_lower"):
        s[nm] = np.zeros(wg)
    for nm in ("F_up_band", "F_down_band", "F_dir_band", "scat_cross_int", "g_0_tot_int",
               "planckband_int"):
        s[nm] = np.zeros(X * I)
    for nm in ("scat_cross_lay", "g_0_tot_lay", "delta_tau_all_clouds",
               "delta_tau_all_clouds_upper", "delta_tau_all_clouds_lower", "trans_band",
               "delta_tau_band", "trans_weight_band", "contr_func_band", "opac_band_lay"):
        s[nm] = np.zeros(X * L)
    s.planckband_lay = np.zeros(X * (L + 2))
    s.planck_grid = np.zeros((c.plancktable_dim + 1) * X)
    for nm in ("F_up_tot", "F_down_tot", "F_net", "F_dir_tot", "meanmolmass_int"):
        s[nm] = np.zeros(I)
    for nm in ("F_net_diff", "meanmolmass_lay", "delta_z_lay"):
        s[nm] = np.zeros(L)
    s.T_store = np.zeros(L + 1)
    s.deltat_prefactor = np.zeros(L + 1)
    s.abort = np.zeros(L + 1, np.int32)
    s.scat_trigger = np.zeros(Y * X, np.int32)
    return s


def setup_planck(impl, c, s):
    impl.planck_table(s.planck_grid, c.opac_interwave, c.opac_deltawave, c.nbin, c.T_star,
                      c.plancktable_dim, c.plancktable_step)
    if c.T_star > 10:
        impl.corr_inc_energy(s.planck_grid, c.starflux, c.opac_deltawave, c.real_star, c.nbin,
                             c.T_star, c.plancktable_dim)


def calculate_height_z(c, s, planet_type="gas"):
    """host step between calc_delta_z and fdir (source/host_functions.py:673-698)"""
    L = c.nlayer
    dz = s.delta_z_lay
    z = c.z_lay
    if planet_type == "gas":
        i0 = max(i for i in range(L) if c.p_lay[i] >= 1e7)
        z[i0] = 0
        for i in range(i0 + 1, L):
            z[i] = z[i - 1] + 0.5 * dz[i - 1] + 0.5 * dz[i]
        for i in range(i0 - 1, -1, -1):
            z[i] = z[i + 1] - 0.5 * dz[i + 1] - 0.5 * dz[i]
    else:
        z[0] = 0.5 * dz[0]
        for i in range(1, L):
            z[i] = z[i - 1] + 0.5 * dz[i - 1] + 0.5 * dz[i]


def refresh_premixed(impl, c, s):
    """the every-10th-iteration block of source/computation.py:860-879 (premixed opacities)"""
    X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
    impl.opac_interpol(c.T_lay, c.ktemp, c.p_lay, c.kpress, c.opac_k, s.opac_wg_lay,
                       c.opac_scat_cross, s.scat_cross_lay, c.npress, c.ntemp, Y, X, L)
    impl.meanmolmass_interpol(c.T_lay, c.ktemp, s.meanmolmass_lay, c.opac_meanmass, c.p_lay,
                              c.kpress, c.npress, c.ntemp, L)
    if c.iso == 0:
        impl.opac_interpol(c.T_int, c.ktemp, c.p_int, c.kpress, c.opac_k, s.opac_wg_int,
                           c.opac_scat_cross, s.scat_cross_int, c.npress, c.ntemp, Y, X, I)
        impl.meanmolmass_interpol(c.T_int, c.ktemp, s.meanmolmass_int, c.opac_meanmass, c.p_int,
                                  c.kpress, c.npress, c.ntemp, I)
    refresh_transmission(impl, c, s)


def refresh_transmission(impl, c, s):
    X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
    if c.clouds == 1:
        impl.calc_total_g0(s.scat_cross_lay, c.g_0_all_clouds_lay, c.scat_cross_all_clouds_lay,
                           s.g_0_tot_lay, c.g_0, X, L)
        if c.iso == 0:
            impl.calc_total_g0(s.scat_cross_int, c.g_0_all_clouds_int, c.scat_cross_all_clouds_int,
                               s.g_0_tot_int, c.g_0, X, I)
    s.scat_trigger[:] = 0
    if c.iso == 1:
        impl.calc_trans_iso(s.trans_wg, s.delta_tau_wg, s.M_term, s.N_term, s.P_term, s.G_plus,
                            s.G_minus, c.delta_colmass, s.opac_wg_lay, s.meanmolmass_lay,
                            s.scat_cross_lay, c.abs_cross_all_clouds_lay,
                            c.scat_cross_all_clouds_lay, s.delta_tau_all_clouds, s.w_0,
                            s.g_0_tot_lay, s.scat_trigger, c.g_0, c.epsi, c.epsi2, c.mu_star,
                            c.w_0_limit, c.w_0_scat_limit, c.scat, X, Y, L, c.clouds, c.scat_corr,
                            c.i2s_transition)
    else:
        impl.calc_trans_noniso(
            s.trans_wg_upper, s.trans_wg_lower, s.delta_tau_wg_upper, s.delta_tau_wg_lower,
            s.M_upper, s.M_lower, s.N_upper, s.N_lower, s.P_upper, s.P_lower, s.G_plus_upper,
            s.G_plus_lower, s.G_minus_upper, s.G_minus_lower, c.delta_col_upper, c.delta_col_lower,
            s.opac_wg_lay, s.opac_wg_int, s.meanmolmass_lay, s.meanmolmass_int, s.scat_cross_lay,
            s.scat_cross_int, c.abs_cross_all_clouds_lay, c.abs_cross_all_clouds_int,
            c.scat_cross_all_clouds_lay, c.scat_cross_all_clouds_int, s.delta_tau_all_clouds_upper,
            s.delta_tau_all_clouds_lower, s.w_0_upper, s.w_0_lower, s.g_0_tot_lay, s.g_0_tot_int,
            s.scat_trigger, c.g_0, c.epsi, c.epsi2, c.mu_star, c.w_0_limit, c.w_0_scat_limit,
            c.scat, X, Y, L, c.clouds, c.scat_corr, c.i2s_transition)
    impl.calc_delta_z(c.T_lay, c.p_int, s.meanmolmass_lay, s.delta_z_lay, c.g, L)
    calculate_height_z(c, s, c.get("planet_type", "gas"))
    if c.iso == 1:
        impl.fdir_iso(s.F_dir_wg, s.planckband_lay, s.delta_tau_wg, c.z_lay, c.mu_star, c.R_planet,
                      c.R_star, c.a, c.dir_beam, c.geom_zenith_corr, I, X, Y)
    else:
        impl.fdir_noniso(s.F_dir_wg, s.Fc_dir_wg, s.planckband_lay, s.delta_tau_wg_upper,
                         s.delta_tau_wg_lower, c.z_lay, c.mu_star, c.R_planet, c.R_star, c.a,
                         c.dir_beam, c.geom_zenith_corr, I, X, Y)


def interpolate_temperatures_and_planck(impl, c, s):
    X, L, I = c.nbin, c.nlayer, c.ninterface
    impl.temp_inter(c.T_lay, c.T_int, I)
    impl.planck_interpol_layer(c.T_lay, s.planckband_lay, s.planck_grid, c.starflux, c.real_star,
                               L, X, c.plancktable_dim, c.plancktable_step)
    if c.iso == 0:
        impl.planck_interpol_interface(c.T_int, s.planckband_int, s.planck_grid, I, X,
                                       c.plancktable_dim, c.plancktable_step)


def flux_sweeps(impl, c, s, nsweep=None):
    """source/computation.py:528-623: 3*scat+1 sweeps -- or, with `flux calculation method = matrix`
    (:882-883, :1088-1089), the tridiagonal solve"""
    X, Y, I = c.nbin, c.ny, c.ninterface
    if c.get("flux_calc_method", "iteration") == "matrix":
        s.matrix_work = flux_matrix(impl, c, s, s.get("matrix_work"))
        return
    if nsweep is None:
        nsweep = 3 * c.scat + 1
    for _ in range(nsweep):
        if c.iso == 1:
            impl.fband_iso(s.F_down_wg, s.F_up_wg, s.F_dir_wg, s.planckband_lay, s.w_0, s.M_term,
                           s.N_term, s.P_term, s.G_plus, s.G_minus, c.surf_albedo, s.g_0_tot_lay,
                           c.g_0, c.R_star, c.a, I, X, c.f_factor, c.mu_star, Y, c.epsi,
                           c.dir_beam, c.clouds, c.scat_corr, c.i2s_transition)
        else:
            impl.fband_noniso(
                s.F_down_wg, s.F_up_wg, s.Fc_down_wg, s.Fc_up_wg, s.F_dir_wg, s.Fc_dir_wg,
                s.planckband_lay, s.planckband_int, s.w_0_upper, s.w_0_lower, s.delta_tau_wg_upper,
                s.delta_tau_wg_lower, s.delta_tau_all_clouds_upper, s.delta_tau_all_clouds_lower,
                s.M_upper, s.M_lower, s.N_upper, s.N_lower, s.P_upper, s.P_lower, s.G_plus_upper,
                s.G_plus_lower, s.G_minus_upper, s.G_minus_lower, c.surf_albedo, s.g_0_tot_lay,
                s.g_0_tot_int, c.g_0, c.R_star, c.a, I, X, c.f_factor, c.mu_star, Y, c.epsi,
                c.delta_tau_limit, c.dir_beam, c.clouds, c.scat_corr, c.i2s_transition)


def matrix_scratch(c):
    """work arrays of the matrix method (source/quantities.py: dev_alpha ... dev_d_prime)"""
    n = c.ny * c.nbin
    H = c.nlayer if c.iso == 1 else 2 * c.nlayer
    rows = 2 * c.ninterface if c.iso == 1 else 4 * c.ninterface - 2
    return dict(alpha=np.zeros(n * H), beta=np.zeros(n * H), source_term_down=np.zeros(n * H),
                source_term_up=np.zeros(n * H), c_prime=np.zeros(n * rows), d_prime=np.zeros(n * rows))


def flux_matrix(impl, c, s, m=None):
    """source/computation.py:625-710: the tridiagonal form of the flux solve (one call, no sweeps)"""
    X, Y, I = c.nbin, c.ny, c.ninterface
    m = matrix_scratch(c) if m is None else m
    if c.iso == 1:
        impl.fband_matrix_iso(s.F_down_wg, s.F_up_wg, s.F_dir_wg, s.planckband_lay, s.w_0, s.M_term,
                              s.N_term, s.P_term, s.G_plus, s.G_minus, s.g_0_tot_lay, m["alpha"],
                              m["beta"], m["source_term_down"], m["source_term_up"], m["c_prime"],
                              m["d_prime"], s.scat_trigger, s.trans_wg, c.surf_albedo, c.g_0, c.R_star,
                              c.a, I, X, c.f_factor, c.mu_star, Y, c.epsi, c.dir_beam, c.clouds,
                              c.scat_corr, c.i2s_transition)
    else:
        impl.fband_matrix_noniso(
            s.F_down_wg, s.F_up_wg, s.Fc_down_wg, s.Fc_up_wg, s.F_dir_wg, s.Fc_dir_wg,
            s.planckband_lay, s.planckband_int, s.w_0_upper, s.w_0_lower, s.delta_tau_wg_upper,
            s.delta_tau_wg_lower, s.delta_tau_all_clouds_upper, s.delta_tau_all_clouds_lower,
            s.M_upper, s.M_lower, s.N_upper, s.N_lower, s.P_upper, s.P_lower, s.G_plus_upper,
            s.G_plus_lower, s.G_minus_upper, s.G_minus_lower, s.g_0_tot_lay, s.g_0_tot_int, m["alpha"],
            m["beta"], m["source_term_down"], m["source_term_up"], m["c_prime"], m["d_prime"],
            s.scat_trigger, s.trans_wg_upper, s.trans_wg_lower, c.surf_albedo, c.g_0, c.R_star, c.a, I,
            X, c.f_factor, c.mu_star, Y, c.epsi, c.delta_tau_limit, c.dir_beam, c.clouds, c.scat_corr,
            c.i2s_transition)
    return m


def integrate_and_step(impl, c, s, itervalue, step_temperature=True):
    X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
    impl.integrate_flux(c.opac_deltawave, s.F_down_tot, s.F_up_tot, s.F_net, s.F_down_wg, s.F_up_wg,
                        s.F_dir_wg, s.F_down_band, s.F_up_band, s.F_dir_band, c.gauss_weight, X, I, Y)
    if step_temperature and itervalue >= c.foreplay:
        impl.rad_temp_iter(s.F_down_tot, s.F_up_tot, s.F_net, s.F_net_diff, c.T_lay, c.p_lay,
                           c.p_int, s.abort, s.T_store, s.deltat_prefactor, c.F_add_heat_lay,
                           c.F_add_heat_sum, c.F_smooth, c.F_smooth_sum, c.c_p_lay,
                           s.meanmolmass_lay, itervalue, c.foreplay, c.g, L, c.physical_tstep,
                           c.rad_convergence_limit, c.adapt_interval, c.smooth,
                           c.plancktable_dim, c.plancktable_step, c.F_intern, c.no_atmo)


def radiation_iterations(impl, c, s, n_iter, start=0, refresh=refresh_premixed):
    """n_iter passes of the radiation_loop body (source/computation.py:851-984), premixed"""
    for it in range(start, start + n_iter):
        interpolate_temperatures_and_planck(impl, c, s)
        if it % 10 == 0:
            refresh(impl, c, s)
        flux_sweeps(impl, c, s)
        integrate_and_step(impl, c, s, it)
    return s


# ---- on-the-fly opacity mixing (source/computation.py:865-869, :1454-1501) ---------------------------
def add_species(c, nspecies=4, seed=3, with_h2o=True):
    """attach a synthetic species set to a case: absorbers with constant VMRs + H2/He-like scatterers"""
    rng = np.random.default_rng(seed)
    sp = []
    for s in range(nspecies):
        sp.append(dict(name="SPEC%02d" % s, absorbing=True, scattering=False, is_h2o=False, is_cia=False,
                       weight=float(rng.uniform(2.0, 64.0)), vmr=float(10.0 ** rng.uniform(-5.0, -2.0)),
                       pretab=syn.ktable(rng, c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y), scat=None))
    sp[0]["vmr"] = 0.8                       # the first absorber dominates mu (e.g. H2O-rich or CO2)
    if nspecies > 2:
        sp[2]["is_cia"] = True               # CIA pairs are always mixed correlated-k and skip mu
        sp[2]["name"] = "CIA_H2H2"
    if with_h2o:
        sp.append(dict(name="H2O", absorbing=False, scattering=True, is_h2o=True, is_cia=False, weight=18.0153,
                       vmr=1e-3, pretab=None, scat=None))
    sp.append(dict(name="H2", absorbing=False, scattering=True, is_h2o=False, is_cia=False, weight=2.016,
                   vmr=0.15, pretab=None, scat=1e-24 * (1e-4 / c.opac_wave) ** 4))
    c.species = sp
    return c


def species_vmr_arrays(c):
    S = len(c.species)
    vl = np.array([np.full(c.nlayer, sp["vmr"]) for sp in c.species]).reshape(S, c.nlayer)
    vi = np.array([np.full(c.ninterface, sp["vmr"]) for sp in c.species]).reshape(S, c.ninterface)
    return vl, vi


def refresh_onthefly(impl, c, s, ro=1):
    X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
    vl, vi = species_vmr_arrays(c)
    inmu = np.array([0.0 if sp["is_cia"] else 1.0 for sp in c.species])
    w = np.array([sp["weight"] for sp in c.species])
    s.meanmolmass_lay[:] = (vl * (w * inmu)[:, None]).sum(0) / (vl * inmu[:, None]).sum(0) * pc.AMU
    s.meanmolmass_int[:] = (vi * (w * inmu)[:, None]).sum(0) / (vi * inmu[:, None]).sum(0) * pc.AMU
    for a in (s.opac_wg_lay, s.opac_wg_int, s.scat_cross_lay, s.scat_cross_int):
        a[:] = 0
    spec_l, spec_i = np.zeros(Y * X * I), np.zeros(Y * X * I)
    sc_l, sc_i = np.zeros(X * L), np.zeros(X * I)
    for k, sp in enumerate(c.species):
        if sp["absorbing"]:
            impl.opac_species_interpol(c.T_lay, c.ktemp, c.p_lay, c.kpress, sp["pretab"], spec_l, c.npress,
                                       c.ntemp, Y, X, L)
            impl.opac_species_interpol(c.T_int, c.ktemp, c.p_int, c.kpress, sp["pretab"], spec_i, c.npress,
                                       c.ntemp, Y, X, I)
            ro_m = 0 if (sp["is_cia"] or not ro) else 1
            impl.add_to_mixed_opac(np.ascontiguousarray(vl[k]), spec_l, s.opac_wg_lay, s.meanmolmass_lay,
                                   c.gauss_weight, c.gauss_y, sp["weight"] * pc.AMU, k, ro_m, Y, X, L)
            impl.add_to_mixed_opac(np.ascontiguousarray(vi[k]), spec_i, s.opac_wg_int, s.meanmolmass_int,
                                   c.gauss_weight, c.gauss_y, sp["weight"] * pc.AMU, k, ro_m, Y, X, I)
        if sp["scattering"]:
            if sp["is_h2o"]:
                impl.calc_h2o_scat(c.T_lay, c.p_lay, c.opac_wave, sc_l, np.ascontiguousarray(vl[k]),
                                   sp["weight"] * pc.AMU, X, L)
                impl.calc_h2o_scat(c.T_int, c.p_int, c.opac_wave, sc_i, np.ascontiguousarray(vi[k]),
                                   sp["weight"] * pc.AMU, X, I)
            else:
                sc_l[:] = np.tile(sp["scat"], L)
                sc_i[:] = np.tile(sp["scat"], I)
            impl.add_to_mixed_scat(np.ascontiguousarray(vl[k]), sc_l, s.scat_cross_lay, X, L)
            impl.add_to_mixed_scat(np.ascontiguousarray(vi[k]), sc_i, s.scat_cross_int, X, I)
    refresh_transmission(impl, c, s)
