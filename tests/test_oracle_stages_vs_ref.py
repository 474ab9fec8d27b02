"""Stage-level pins of the C restatement against the reference's kernels (oracle/_ref): the
on-the-fly mixing stages, temperature steps and post-loop diagnostics that the premixed chain test
does not reach."""
import numpy as np
import pytest

import cases
from helios_amd import phys_const as pc
from helios_amd import synthetic as syn


def _close(a, b, rtol=1e-13, atol=1e-300, name=""):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=name)


def mixing_inputs(seed=7, nbin=11, nlev=8, ny=20, ntemp=5, npress=4):
    rng = np.random.default_rng(seed)
    gy, gw = syn.gauss_points(ny)
    ktemp, kpress = syn.tp_grid(ntemp, npress)
    _, wave, _ = syn.wavelength_grid(nbin)
    tabs = [syn.ktable(rng, nbin, ny, ktemp, kpress, gy) for _ in range(4)]
    # temperatures/pressures that hit: below grid, exact nodes, interior, above grid
    temp = np.array([50.0, ktemp[0], ktemp[1], 0.5 * (ktemp[1] + ktemp[2]), ktemp[-1], 5000.0, 777.0, 1234.5])[:nlev]
    press = np.array([0.1, kpress[0], kpress[1], 3e4, kpress[-1], 1e11, kpress[2], 5e6])[:nlev]
    return dict(rng=rng, gy=gy, gw=gw, ktemp=ktemp, kpress=kpress, wave=wave, tabs=tabs, temp=temp,
                press=press, nbin=nbin, nlev=nlev, ny=ny, ntemp=ntemp, npress=npress)


def test_species_interpol_edges(port, ref):
    m = mixing_inputs()
    n = m["ny"] * m["nbin"] * m["nlev"]
    a, b = np.zeros(n), np.zeros(n)
    for impl, out in ((port, a), (ref, b)):
        impl.opac_species_interpol(m["temp"], m["ktemp"], m["press"], m["kpress"], m["tabs"][0], out,
                                   m["npress"], m["ntemp"], m["ny"], m["nbin"], m["nlev"])
    assert np.all(a > 0)
    _close(a, b)


def _mix(impl, m, spec_scale, mix0, ro_method, s):
    """one add_to_mixed_opac call on crafted inputs; returns the updated mix"""
    ny, nbin, nlev = m["ny"], m["nbin"], m["nlev"]
    mix = mix0.copy()
    vmr = np.full(nlev, 1e-3)
    mmm = np.full(nlev, 2.3 * pc.AMU)
    impl.add_to_mixed_opac(vmr, spec_scale, mix, mmm, m["gw"], m["gy"], 18.0 * pc.AMU, s, ro_method,
                           ny, nbin, nlev)
    return mix


@pytest.mark.parametrize("kind", ["s0", "corrk", "neg_new", "neg_mix", "ro_nocross", "ro_onecross",
                                  "ro_multicross", "ro_mix_bigger", "ro_ties"])
def test_add_to_mixed_opac_branches(port, ref, kind):
    m = mixing_inputs()
    ny, nbin, nlev = m["ny"], m["nbin"], m["nlev"]
    rng = np.random.default_rng(99)
    n = ny * nbin * nlev
    base = np.sort(10.0 ** rng.uniform(-4, 0, (nlev, nbin, ny)), axis=2)         # ascending in y
    other = np.sort(10.0 ** rng.uniform(-4, 0, (nlev, nbin, ny)), axis=2)
    fac = 1e-3 * 18.0 / 2.3                                                        # vmr*m_s/mu
    s, ro = 1, 1
    if kind == "s0":
        s = 0
    elif kind == "corrk":
        ro = 0
    elif kind == "neg_new":
        other = other * 1e-9
    elif kind == "neg_mix":
        base = base * 1e-9
    elif kind == "ro_nocross":
        other = base * 0.5 / fac * (1 + 0.1 * rng.uniform(size=base.shape))
        other = np.sort(other, axis=2)
    elif kind == "ro_onecross":
        y = np.arange(ny)
        base = np.broadcast_to(10.0 ** (-3 + 3 * y / (ny - 1.0)), base.shape).copy()
        other = np.broadcast_to(10.0 ** (-2 + 1 * y / (ny - 1.0)), base.shape).copy() / fac
    elif kind == "ro_multicross":
        other = other / fac
    elif kind == "ro_mix_bigger":
        other = base * 0.9 / fac
    elif kind == "ro_ties":
        base = np.round(base, 2) + 0.01
        other = (np.round(other, 2) + 0.01) / fac
    if kind.startswith("ro_") and kind != "ro_multicross":
        pass
    a = _mix(port, m, other.reshape(-1), base.reshape(-1), ro, s)
    b = _mix(ref, m, other.reshape(-1), base.reshape(-1), ro, s)
    assert np.all(np.isfinite(a))
    _close(a, b, rtol=1e-12, name=kind)
    if kind.startswith("ro_"):
        # the result must differ from plain correlated-k addition (i.e. the RO branch really ran)
        ck = _mix(port, m, other.reshape(-1), base.reshape(-1), 0, s)
        assert np.abs(a - ck).max() > 0
        # and stay monotone in y
        assert np.all(np.diff(a.reshape(nlev, nbin, ny), axis=2) >= -1e-14 * a.max())


def test_species_loop(port, ref):
    """4 species through interpol + mixing, the running mix feeding the next species"""
    m = mixing_inputs(seed=3)
    ny, nbin, nlev = m["ny"], m["nbin"], m["nlev"]
    n = ny * nbin * nlev
    res = []
    for impl in (port, ref):
        mix = np.zeros(n)
        spec = np.zeros(n)
        mmm = np.full(nlev, 2.3 * pc.AMU)
        for s, (w, v) in enumerate(((2.0, 0.85), (18.0, 1e-3), (44.0, 3e-4), (16.0, 1e-5))):
            impl.opac_species_interpol(m["temp"], m["ktemp"], m["press"], m["kpress"], m["tabs"][s],
                                       spec, m["npress"], m["ntemp"], ny, nbin, nlev)
            impl.add_to_mixed_opac(np.full(nlev, v), spec, mix, mmm, m["gw"], m["gy"], w * pc.AMU, s,
                                   1, ny, nbin, nlev)
        res.append(mix)
    _close(res[0], res[1], rtol=1e-12)


def test_scattering_helpers(port, ref):
    m = mixing_inputs()
    nbin, nlev = m["nbin"], m["nlev"]
    rng = np.random.default_rng(5)
    temp = rng.uniform(200, 2000, nlev)
    press = 10.0 ** rng.uniform(0, 9.5, nlev)
    vmr = 10.0 ** rng.uniform(-6, -1, nlev)
    a, b = np.zeros(nbin * nlev), np.zeros(nbin * nlev)
    port.calc_h2o_scat(temp, press, m["wave"], a, vmr, 18.0 * pc.AMU, nbin, nlev)
    ref.calc_h2o_scat(temp, press, m["wave"], b, vmr, 18.0 * pc.AMU, nbin, nlev)
    assert a.max() > 0
    _close(a, b)
    sa, sb = a.copy() * 0.3, a.copy() * 0.3
    port.add_to_mixed_scat(vmr, a, sa, nbin, nlev)
    ref.add_to_mixed_scat(vmr, b, sb, nbin, nlev)
    _close(sa, sb)
    g_cl = rng.uniform(0, 0.9, nbin * nlev)
    s_cl = rng.uniform(0, 1e-26, nbin * nlev)
    ga, gb = np.zeros(nbin * nlev), np.zeros(nbin * nlev)
    port.calc_total_g0(sa, g_cl, s_cl, ga, 0.2, nbin, nlev)
    ref.calc_total_g0(sb, g_cl, s_cl, gb, 0.2, nbin, nlev)
    _close(ga, gb)


def test_kappa_cp_interpol(port, ref):
    rng = np.random.default_rng(11)
    nt, npr, nlev = 9, 7, 12
    et = np.linspace(100, 4000, nt)
    ep = np.logspace(0, 9, npr)
    tab = rng.uniform(0.1, 0.4, nt * npr)
    temp = rng.uniform(50, 4500, nlev)
    press = 10.0 ** rng.uniform(-1, 10, nlev)
    for fn in ("kappa_interpol", "cp_interpol", "entropy_interpol", "phase_number_interpol"):
        a, b = np.zeros(nlev), np.zeros(nlev)
        getattr(port, fn)(temp, et, press, ep, a, tab, npr, nt, nlev)
        getattr(ref, fn)(temp, et, press, ep, b, tab, npr, nt, nlev)
        _close(a, b, name=fn)


@pytest.mark.parametrize("itervalue", [0, 19, 20, 39, 6000])
def test_conv_temp_iter(port, ref, itervalue):
    rng = np.random.default_rng(itervalue + 1)
    L = 11
    c = cases.make_case(nlayer=L)
    outs = []
    for impl in (port, ref):
        T = c.T_lay.copy()
        F_net = rng.uniform(1e5, 1e6, L + 1) if impl is port else outs[0][3]
        F_net_diff = np.zeros(L)
        T_store = T + (rng.uniform(-3, 3, L + 1) if impl is port else 0)
        if impl is not port:
            T_store = outs[0][4]
        pref = np.full(L + 1, 0.5)
        marked = np.zeros(L + 1, np.int32)
        marked[3] = 1
        T_store0 = T_store.copy()
        impl.conv_temp_iter(F_net, F_net_diff, T, c.p_lay, c.p_int, T_store, pref, marked,
                            np.zeros(L), np.zeros(L), np.zeros(L), L, itervalue, 20, 0, c.F_intern)
        outs.append((T, pref, F_net_diff, F_net, T_store0, T_store))
    for k in (0, 1, 2, 5):
        _close(outs[0][k], outs[1][k], name=str(k))


def test_post_loop_diagnostics(port, ref):
    c0 = cases.make_case(clouds=1, g_0=0.2)
    res = []
    for impl in (port, ref):
        c = c0.copy()
        s = cases.alloc_state(c)
        cases.setup_planck(impl, c, s)
        cases.radiation_iterations(impl, c, s, 1)
        X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
        impl.integrate_optdepth_transmission_noniso(
            s.trans_wg_upper, s.trans_wg_lower, s.trans_band, s.delta_tau_wg_upper,
            s.delta_tau_wg_lower, s.delta_tau_band, c.gauss_weight, s.delta_tau_all_clouds,
            s.delta_tau_all_clouds_upper, s.delta_tau_all_clouds_lower, X, L, Y)
        impl.calc_contr_func_noniso(s.trans_wg_upper, s.trans_wg_lower, s.trans_weight_band,
                                    s.contr_func_band, c.gauss_weight, s.planckband_lay, c.epsi, X, L, Y)
        pm = [np.zeros(L) for _ in range(4)]
        impl.calc_mean_opacities(pm[0], pm[1], pm[2], pm[3], s.opac_wg_lay, c.abs_cross_all_clouds_lay,
                                 s.meanmolmass_lay, s.planckband_lay, c.opac_interwave,
                                 c.opac_deltawave, c.T_lay, c.gauss_weight, c.gauss_y, s.opac_band_lay,
                                 L, X, Y, c.T_star)
        impl.integrate_beamflux(s.F_dir_tot, s.F_dir_band, c.opac_deltawave, X, I)
        res.append((s, pm))
    (sp, pp), (sr, pr) = res
    for k in ("trans_band", "delta_tau_band", "delta_tau_all_clouds", "trans_weight_band",
              "contr_func_band", "opac_band_lay", "F_dir_tot"):
        _close(sp[k], sr[k], rtol=1e-12, name=k)
    for a, b in zip(pp, pr):
        _close(a, b, rtol=1e-12)


def test_post_loop_diagnostics_iso(port, ref):
    c0 = cases.make_case(iso=1)
    res = []
    for impl in (port, ref):
        c = c0.copy()
        s = cases.alloc_state(c)
        cases.setup_planck(impl, c, s)
        cases.radiation_iterations(impl, c, s, 1)
        X, Y, L = c.nbin, c.ny, c.nlayer
        impl.integrate_optdepth_transmission_iso(s.trans_wg, s.trans_band, s.delta_tau_wg,
                                                 s.delta_tau_band, c.gauss_weight, X, L, Y)
        impl.calc_contr_func_iso(s.trans_wg, s.trans_weight_band, s.contr_func_band, c.gauss_weight,
                                 s.planckband_lay, c.epsi, X, L, Y)
        res.append(s)
    for k in ("trans_band", "delta_tau_band", "trans_weight_band", "contr_func_band"):
        _close(res[0][k], res[1][k], rtol=1e-12, name=k)
