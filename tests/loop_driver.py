"""TEST INFRASTRUCTURE: the reference's radiation_loop control flow (source/computation.py:827-990)
restated around the stage functions of tests/cases.py, so that the SAME loop can be run through the
reference's own kernels (oracle.refgpu on the MI355X, oracle.ref on the host), through the C
restatement (oracle.port) and through the per-stage HIP entry points.  SURVEY.md section 8(c): "oracle
loop driver ... goldens {T_lay, F_net, F_up_band[TOA], F_down_band[BOA], iter_count} after N = 1, 10,
11, 50 iterations and at convergence".

Control flow followed line by line:
    :851-853  while condition1 and condition2 (condition3 belongs to physical time-stepping)
    :856-857  temperatures + Planck every iteration
    :860-879  opacities / transmission / heights / beam when iter % 10 == 0
    :880-883  3*scat+1 sweeps,  :888 quadrature
    :906      temperature step only once iter >= foreplay
    :927-932  abortsum = sum(abort);  :938 condition1 = abortsum < nlayer + 1
    :941-943  time-stepped runs: condition3 = (iter + 1) * physical_tstep < runtime_limit
    :946-952  every 100th iteration: condition2 = T_lay[nlayer] < dim*step - 2
    :954      iter += 1
    :974-975  rad_convergence_limit *= 10 when the NEW iter value is in crit_relaxation_numbers
    :978-981  abort when iter > max_nr_iterations
"""
import numpy as np

import cases

SNAP_KEYS = ("F_net", "F_up_tot", "F_down_tot")


def snapshot(c, s):
    X, L, I = c.nbin, c.nlayer, c.ninterface
    out = {k: s[k].copy() for k in SNAP_KEYS}
    out["T_lay"] = c.T_lay.copy()
    out["F_up_band_TOA"] = s.F_up_band[X * (I - 1):X * I].copy()
    out["F_down_band_BOA"] = s.F_down_band[:X].copy()
    out["F_dir_band_BOA"] = s.F_dir_band[:X].copy()
    out["abort"] = s.abort.copy()
    out["deltat_prefactor"] = s.deltat_prefactor.copy()
    return out


def radiation_loop(impl, c, s, snap_at=(), max_nr_iterations=20000, crit_relaxation_numbers=(),
                   refresh=cases.refresh_premixed, on_iteration=None, runtime_limit=None):
    """runs the loop until the reference's loop would leave it.  Returns (iter_count, snaps, reason)
    with snaps[n] = state after n completed iterations for n in snap_at, snaps['end'] = final state."""
    snaps = {}
    it = 0
    L = c.nlayer
    condition1 = condition2 = condition3 = True
    reason = "converged"
    while condition1 and condition2 and condition3:
        cases.interpolate_temperatures_and_planck(impl, c, s)
        if it % 10 == 0:
            refresh(impl, c, s)
        cases.flux_sweeps(impl, c, s)
        cases.integrate_and_step(impl, c, s, it)
        abortsum = int(s.abort.sum()) if it >= c.foreplay else 0
        condition1 = abortsum < L + 1
        if c.physical_tstep != 0 and runtime_limit is not None:      # :941-943, with the iteration index before the increment
            condition3 = (it + 1) * c.physical_tstep < runtime_limit
            if not condition3:
                reason = "runtime limit"
        if it % 100 == 0:
            condition2 = bool(c.T_lay[L] < c.plancktable_dim * c.plancktable_step - 2)
            if not condition2:
                reason = "surface temperature beyond the Planck table"
        it += 1
        if it in snap_at:
            snaps[it] = snapshot(c, s)
        if on_iteration is not None:
            on_iteration(it, c, s)
        if it in crit_relaxation_numbers:
            c.rad_convergence_limit *= 10.0
        if it > max_nr_iterations:
            reason = "iteration limit"
            break
    snaps["end"] = snapshot(c, s)
    return it, snaps, reason


def loop_case(name):
    """the small columns whose loops are committed as goldens (tests/golden/loop_<name>.npz)"""
    kw = dict(nbin=6, nlayer=8, ntemp=6, npress=5, plancktable_dim=400, plancktable_step=10)
    relax = ()
    if name == "default":
        pass
    elif name == "dirbeam_albedo":
        kw.update(dir_beam=1, albedo=0.3)
    elif name == "noscat_relax":
        # criterion relaxed at iteration 60 (computation.py:974-975): the loop must end because of it
        kw.update(scat=0)
        relax = (60,)
    elif name == "clouds_g0_i2s":
        kw.update(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2, T_star=3500.0)
    elif name == "onthefly":
        # opacities mixed on the fly at every refresh: 4 absorbers (one CIA pair), water-vapour and H2 scattering
        kw.update(nbin=5)
    elif name == "c5physics_onthefly":
        # BASELINE config 5's ingredients together in one small column: absorbers mixed on the fly with random overlap,
        # two cloud decks, g0 with the I2S correction, direct beam, reflecting surface
        kw.update(nbin=5, clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2, T_star=3500.0)
    elif name == "matrix":
        # `flux calculation method = matrix` (computation.py:882-883): one tridiagonal solve per spectral point instead
        # of the sweeps.  (The reader keeps the surface albedo >= 1e-8 "for matrix method to work", read.py:1261.)
        kw.update(albedo=0.1)
    elif name == "matrix_c5physics":
        kw.update(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2, T_star=3500.0)
    elif name == "matrix_noscat":
        # no scattering: every spectral point takes the solver's pure-absorption branch (kernels.cu:2286-2421)
        kw.update(scat=0, albedo=0.1)
    elif name == "matrix_iso":
        kw.update(iso=1, albedo=0.1, dir_beam=1)
    else:
        raise KeyError(name)
    c = cases.make_case(**kw)
    if name.startswith("matrix"):
        c.flux_calc_method = "matrix"
    if name == "noscat_relax":
        c.rad_convergence_limit = 1e-9
    if name in ("onthefly", "c5physics_onthefly"):
        cases.add_species(c, nspecies=4)
    return c, relax


def loop_refresh(c):
    return cases.refresh_onthefly if c.get("species") else cases.refresh_premixed


LOOP_NAMES = ("default", "dirbeam_albedo", "noscat_relax", "clouds_g0_i2s", "onthefly", "c5physics_onthefly",
              "matrix", "matrix_c5physics", "matrix_noscat", "matrix_iso")
SNAP_AT = (1, 10, 11, 50)


# ---- convection loop (source/computation.py:992-1174) ----------------------------------------------------------------
"""Control flow followed line by line (reference source/computation.py):
    :998-1009  kappa (a constant here: interpolate_kappa_and_cp launches nothing, :202), conv_check,
               mark_convective_layers(stitching=0); condition = sum(conv_unstable) > 0
    :1016      iter_value = 0 (the loop counts its own iterations)
    :1043      temp_inter
    :1046-1051 mean molecular mass when iter % 10 == 0 (premixed: meanmolmass_interpol for layers and interfaces)
    :1053-1061 D2H of kappa, c_p, mu, T, F_smooth_sum; host convective_adjustment; H2D of T_lay
    :1063-1064 temp_inter again, Planck interpolation
    :1067-1085 opacities / transmission / heights / beam when iter % 10 == 0
    :1086-1090 3*scat+1 sweeps (or the matrix solve), quadrature -- NO rad_temp_iter
    :1107      mark_convective_layers(stitching=1) with the adjusted profile
    :1110-1112 physical time step: one adjustment, no temperature iteration
    :1116      condition = not check_for_radiative_eq or iter < 400 or sum(conv_layer) == 0
    :1122-1149 if condition: conv_temp_iter with conv_layer / marked_red of this iteration, iter += 1
    :1158-1159 rad_convergence_limit *= 10 when iter is in crit_relaxation_numbers
    :1162-1165 abort when iter > max_nr_iterations
Host steps come from `hs`: helios_amd/host_functions.py (pinned function by function to the reference's Python,
tests/test_host_golden.py) on the GPU box, or the reference's own source/host_functions.py imported in the build
container (tests/golden/make_golden.py --hostref)."""
CONV_SNAP_AT = (1, 10, 11, 50, 400)
CONV_SNAP_KEYS = SNAP_KEYS + ("F_net_diff",)


def conv_quant(c, s, kappa, dampara="automatic"):
    """the attributes of the reference's Store that its host functions touch in the convection loop, sharing the
    arrays of the case `c` and the kernel state `s` (the kernels write F_net etc. in place)"""
    L = c.nlayer
    q = cases.Case()
    q.nlayer, q.ninterface = L, L + 1
    q.p_lay, q.p_int, q.T_lay = c.p_lay, c.p_int, c.T_lay
    q.kappa_lay, q.kappa_int = np.full(L, float(kappa)), np.full(L + 1, float(kappa))
    q.c_p_lay = c.c_p_lay
    q.meanmolmass_lay = s.meanmolmass_lay
    q.F_net, q.F_up_tot, q.F_down_tot, q.F_net_diff = s.F_net, s.F_up_tot, s.F_down_tot, s.F_net_diff
    q.F_add_heat_sum, q.F_smooth_sum = c.F_add_heat_sum, c.F_smooth_sum
    q.F_intern, q.T_star = c.F_intern, c.T_star
    q.rad_convergence_limit = c.rad_convergence_limit
    q.input_dampara = dampara
    q.conv_unstable = np.zeros(L + 1, np.int32)
    q.conv_layer = np.zeros(L + 1, np.int32)
    q.marked_red = np.zeros(L + 1, np.int32)
    q.converged = np.zeros(L + 1, np.int32)
    q.iter_value = 0
    q.delta_colmass = c.delta_colmass
    return q


def conv_snapshot(c, s, q):
    out = snapshot(c, s)
    out["F_net_diff"] = s.F_net_diff.copy()
    for k in ("conv_layer", "conv_unstable", "marked_red"):
        out[k] = np.asarray(q[k], np.int32).copy()
    return out


def mean_molecular_mass(impl, c, s):
    """:1046-1051 -- premixed: the table look-up on layers and interfaces; on the fly: constant mixing ratios in the
    test columns, so the value of the last refresh stands (calculate_meanmolecularmass gives the same numbers)"""
    if c.get("species"):
        return
    L, I = c.nlayer, c.ninterface
    impl.meanmolmass_interpol(c.T_lay, c.ktemp, s.meanmolmass_lay, c.opac_meanmass, c.p_lay, c.kpress, c.npress,
                              c.ntemp, L)
    if c.iso == 0:
        impl.meanmolmass_interpol(c.T_int, c.ktemp, s.meanmolmass_int, c.opac_meanmass, c.p_int, c.kpress, c.npress,
                                  c.ntemp, I)


def convection_loop(impl, hs, c, s, kappa, snap_at=(), max_nr_iterations=20000, crit_relaxation_numbers=(),
                    refresh=cases.refresh_premixed, dampara="automatic", on_iteration=None):
    """runs the loop until the reference's loop would leave it; the state (c, s) is the one the radiation loop left.
    Returns (iter_count, snaps, q): snaps[n] = state when the iteration counter has just become n."""
    L = c.nlayer
    q = conv_quant(c, s, kappa, dampara)
    hs.conv_check(q)
    hs.mark_convective_layers(q, stitching=0)
    condition = int(np.sum(q.conv_unstable)) > 0
    snaps = {"start": conv_snapshot(c, s, q)}
    while condition:
        it = int(q.iter_value)
        impl.temp_inter(c.T_lay, c.T_int, L + 1)
        if it % 10 == 0:
            mean_molecular_mass(impl, c, s)
        hs.convective_adjustment(q)
        impl.temp_inter(c.T_lay, c.T_int, L + 1)
        cases.interpolate_temperatures_and_planck(impl, c, s)
        if it % 10 == 0:
            refresh(impl, c, s)
        cases.flux_sweeps(impl, c, s)
        cases.integrate_and_step(impl, c, s, it, step_temperature=False)
        hs.mark_convective_layers(q, stitching=1)
        if c.physical_tstep != 0:
            break
        q.rad_convergence_limit = c.rad_convergence_limit
        condition = (not hs.check_for_radiative_eq(q)) or it < 400 or int(np.sum(q.conv_layer)) == 0
        if condition:
            marked = np.ascontiguousarray(q.marked_red, np.int32)
            impl.conv_temp_iter(s.F_net, s.F_net_diff, c.T_lay, c.p_lay, c.p_int, s.T_store, s.deltat_prefactor, marked,
                                c.F_add_heat_lay, c.F_smooth, c.F_smooth_sum, L, it, c.adapt_interval, c.smooth,
                                c.F_intern)
            q.iter_value = it + 1
            if q.iter_value in snap_at:
                snaps[int(q.iter_value)] = conv_snapshot(c, s, q)
            if on_iteration is not None:
                on_iteration(int(q.iter_value), c, s, q)
        if q.iter_value in crit_relaxation_numbers:
            c.rad_convergence_limit *= 10.0
        if q.iter_value > max_nr_iterations:
            break
    snaps["end"] = conv_snapshot(c, s, q)
    return int(q.iter_value), snaps, q


CONV_NAMES = ("deep", "deep_matrix", "beam_albedo", "clouds_g0_i2s", "surface_jump", "detached", "c5physics_onthefly")
CONV_SIZE = dict(nbin=6, nlayer=16, ntemp=6, npress=5, plancktable_dim=800, plancktable_step=10)
# Columns whose convection loop takes a marginal time-step decision (`|T - T_store| < adapt/2 |dT|`, kernels.cu:2869-2876)
# between iterations 50 and 400: last-bit differences decide it, the trajectories part for a few hundred iterations and
# meet again at the same equilibrium.  The REFERENCE ITSELF is not reproducible there: its gfx950 build leaves the loop of
# `c5physics_onthefly` after 1281 iterations, its host build (no FMA contraction) after 7861, this repo's oracle and library
# after 1241 -- end states within 6e-10 of each other.  For these the tests hold everything through iteration 50 and the end
# state, not the count.
CONV_COUNT_SENSITIVE = ("c5physics_onthefly",)


def conv_case(name):
    """the small columns whose rad.-conv. runs are committed as goldens (tests/golden/loopconv_<name>.npz):
    (case, kappa, run_radiation_loop_first).  `deep`: the radiation loop converges (1202 iterations), the equilibrium
    profile is super-adiabatic at the bottom, the convection loop (2921) ends with one zone at the bottom; `beam_albedo`:
    the same with direct beam and reflecting surface (rad_convergence_limit 1e-7); `clouds_g0_i2s`: config 5's physics
    (cloud decks, g0, I2S correction, beam, albedo) -- its radiation loop leaves through the surface-temperature check
    of iterations 0, 100, ... (computation.py:946-952: "jump directly to convective loop"), as does `surface_jump`
    (strong internal heat); `detached`: the convection loop entered with a profile that has two super-adiabatic regions
    separated by a radiative zone (two zones with their own fudge factors for ~200 iterations, kappa = 0.25, no
    radiation loop before it)"""
    from helios_amd import phys_const as pc
    kw = dict(CONV_SIZE)
    kappa, radiative_first = 2.0 / 7.0, True
    limit = 1e-8
    if name in ("deep", "deep_hostref"):
        kw.update(T_intern=250.0)
    elif name == "deep_matrix":       # both loops with `flux calculation method = matrix`
        kw.update(T_intern=250.0, albedo=0.1)
    elif name == "surface_jump":
        kw.update(T_intern=450.0)
    elif name == "beam_albedo":
        kw.update(dir_beam=1, albedo=0.3, T_intern=200.0)
        limit = 1e-7
    elif name == "clouds_g0_i2s":
        kw.update(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2, T_star=3500.0, T_intern=350.0)
    elif name == "c5physics_onthefly":   # config 5 in small: on-the-fly mixing + clouds + beam + albedo + I2S, then convection
        kw.update(nbin=5, clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2, T_star=3500.0, T_intern=350.0)
    elif name == "detached":
        kw.update(nlayer=20, dir_beam=1, albedo=0.2, T_intern=300.0)
        kappa, radiative_first = 0.25, False
    else:
        raise KeyError(name)
    c = cases.make_case(**kw)
    if name == "c5physics_onthefly":
        cases.add_species(c, nspecies=4)
    c.rad_convergence_limit = limit
    if name == "deep_matrix":
        c.flux_calc_method = "matrix"
    c.c_p_lay = np.full(c.nlayer, pc.R_UNIV / kappa)              # read.py:1178-1180
    if name == "detached":
        p = c.p_lay
        T = 1500.0 * (p / p[0]) ** 0.10
        T[2:6] = T[2] * (p[2:6] / p[2]) ** 0.5
        T[6:] = T[5] * (p[6:] / p[5]) ** 0.05
        T[9:12] = T[9] * (p[9:12] / p[9]) ** 0.45
        T[12:] = T[11] * (p[12:] / p[11]) ** 0.03
        c.T_lay = np.append(T, T[0] * 1.0001)
    return c, kappa, radiative_first


def rad_conv_run(impl, hs, c, s, kappa, radiative_first):
    """radiation loop (when the case has one), then the convection loop from the state it leaves"""
    n_rad, reason = 0, "not run"
    if radiative_first:
        n_rad, _snaps, reason = radiation_loop(impl, c, s, refresh=loop_refresh(c))
    n, snaps, q = convection_loop(impl, hs, c, s, kappa, snap_at=CONV_SNAP_AT, refresh=loop_refresh(c))
    q.rad_reason = reason
    return n_rad, n, snaps, q
