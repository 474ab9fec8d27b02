"""BASELINE.json's configurations at FULL size on the GPU, held to the CPU oracle where the domain allows it: spectral
bins are independent up to the sums over wavelength, so the oracle runs blocks of contiguous bins cut out of the
full-size column (same tables, same Planck table, same profile) and the HIP results of exactly those bins are
compared -- opacities of every Gauss point and level after the mixing, Rayleigh cross-sections, band fluxes of the
first flux solve (before the temperature step, which needs all bins) -- plus invariants over all bins.

  config 3   20 absorbers + H2/He, random overlap, 10 000 bins x 100 layers, one column
  config 4   8 columns of the parameter sweep with on-the-fly mixing in ONE batch == 8 single-column runs
  config 5   30 000 bins x 200 layers, two cloud decks, g0, I2S correction, surface albedo, direct beam, and the
             convection loop on top (radiative quantities against the oracle, the loop through invariants)
"""
import os
import sys

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def ctx():
    from helios_amd.device import Context
    return Context(0)


def _blocks(nbin, nb=16):
    return [0, (nbin // 3) // nb * nb, (2 * nbin // 3) // nb * nb, nbin - nb], nb


def _slice_table(tab, c, x0, nb):
    """[t][p][x][y] flat -> the same layout for bins x0 .. x0+nb"""
    t = np.asarray(tab).reshape(c.ntemp, c.npress, c.nbin, c.ny)
    return np.ascontiguousarray(t[:, :, x0:x0 + nb, :]).reshape(-1)


def _block_case(c, x0, nb, species_slices=None):
    """the sub-problem of bins [x0, x0 + nb) as a tests/cases.py Case"""
    b = cases.Case(c)
    b.nbin = nb
    b.opac_interwave = np.ascontiguousarray(c.opac_interwave[x0:x0 + nb + 1])
    for k in ("opac_wave", "opac_deltawave", "surf_albedo", "starflux"):
        b[k] = np.ascontiguousarray(np.asarray(c[k])[x0:x0 + nb])
    if not c.get("species"):       # with species the premixed table is never read (bench.build_case keeps a stub)
        b.opac_k = _slice_table(c.opac_k, c, x0, nb)
    b.opac_scat_cross = np.ascontiguousarray(
        np.asarray(c.opac_scat_cross).reshape(c.ntemp, c.npress, c.nbin)[:, :, x0:x0 + nb]).reshape(-1)
    for nm, n in (("lay", c.nlayer), ("int", c.nlayer + 1)):
        for p in ("abs_cross_all_clouds_", "scat_cross_all_clouds_", "g_0_all_clouds_"):
            if c.clouds:
                b[p + nm] = np.ascontiguousarray(np.asarray(c[p + nm]).reshape(n, c.nbin)[:, x0:x0 + nb]).reshape(-1)
            else:
                b[p + nm] = np.zeros(n * nb)
    b.z_lay = np.zeros(c.nlayer)
    b.T_int = np.zeros(c.nlayer + 1)
    b.T_lay = np.asarray(c.T_lay, float).copy()
    for k in ("F_add_heat_lay", "F_add_heat_sum", "F_smooth", "F_smooth_sum"):
        b[k] = np.zeros(c.nlayer)
    b.delta_colmass = (c.p_int[:-1] - c.p_int[1:]) / c.g
    b.delta_col_upper = (c.p_lay - c.p_int[1:]) / c.g
    b.delta_col_lower = (c.p_int[:-1] - c.p_lay) / c.g
    if species_slices is not None:
        b.species = [dict(sp, pretab=sl[x0], scat=None if sp["scat"] is None else np.ascontiguousarray(sp["scat"][x0:x0 + nb]))
                     for sp, sl in species_slices]
    return b


def _oracle_block(port, b, planck_grid_dev, c, x0, nb, refresh):
    s = cases.alloc_state(b)
    s.planck_grid[:] = np.asarray(planck_grid_dev).reshape(c.plancktable_dim + 1, c.nbin)[:, x0:x0 + nb].reshape(-1)
    cases.radiation_iterations(port, b, s, 1, refresh=refresh)
    return s


def _compare_block(got, s, c, x0, nb, keys_wg, keys_band, rtol=1e-9):
    """`got`: full-size arrays in the reference layouts"""
    Y, X, L, I = c.ny, c.nbin, c.nlayer, c.nlayer + 1
    for k in keys_wg:
        nlev = L if k.endswith("_lay") or k.startswith("Fc_") else I
        a = np.asarray(got[k])[:nlev * X * Y].reshape(nlev, X, Y)[:, x0:x0 + nb, :]
        w = np.asarray(s[k])[:nlev * nb * Y].reshape(nlev, nb, Y)
        # fluxes with the beam: G-weighted terms of O(G F_dir) cancel down to the diffuse flux inside the reference's
        # formula (kernels.cu:1652-1656), leaving ~1e-15 max(F) of absolute noise per half-layer; 2.4e-13 max(F) seen
        # after 400 half-layers
        atol = (1e-12 if k.startswith(("F_", "Fc_")) else 1e-13) * np.abs(w).max()
        np.testing.assert_allclose(a, w, rtol=rtol, atol=atol, err_msg="%s bins %d.." % (k, x0))
    for k in keys_band:
        nlev = L if k.endswith("_lay") else I
        a = np.asarray(got[k])[:nlev * X].reshape(nlev, X)[:, x0:x0 + nb]
        w = np.asarray(s[k])[:nlev * nb].reshape(nlev, nb)
        atol = (1e-12 if k.startswith("F_") else 1e-13) * np.abs(w).max()
        np.testing.assert_allclose(a, w, rtol=rtol, atol=atol, err_msg="%s bins %d.." % (k, x0))


def _upload_species(rt, c, xs, nb, separable=False):
    """upload every species table, keep the sampled bins of each for the oracle (`separable`: the device forms the synthetic
    tables from their two factors, hx_rt_set_species_separable; the oracle still gets slices of the host array)"""
    from helios_amd import synthetic as syn
    slices = []
    for k, sp in enumerate(c.species):
        tab = sp["pretab"]
        h2o, cia = bool(sp.get("is_h2o")), bool(sp.get("is_cia"))
        flags = dict(is_h2o=2 if h2o else 0, is_cia=1 if cia else 0, in_mu=0 if cia else 1)
        lazy = isinstance(tab, str)
        if lazy:
            tab = syn.ktable(np.random.default_rng(sp["table_seed"]), c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y)
        if lazy and separable:
            kxy, ftp = syn.ktable_factors(np.random.default_rng(sp["table_seed"]), c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y)
            rt.set_species_separable(k, kxy, ftp, sp["scat"], sp["weight"], **flags)
        else:
            rt.set_species(k, tab, sp["scat"], sp["weight"], **flags)
        sl = {x0: (_slice_table(tab, c, x0, nb) if tab is not None else None) for x0 in xs}
        slices.append((dict(name=sp.get("name", "S%02d" % k), absorbing=tab is not None, scattering=sp["scat"] is not None or h2o,
                            is_h2o=h2o, is_cia=cia, weight=sp["weight"], vmr=sp["vmr"], pretab=None, scat=sp["scat"]), sl))
        del tab
    vl = np.array([np.full(c.nlayer, sp["vmr"]) for sp in c.species])
    vi = np.array([np.full(c.nlayer + 1, sp["vmr"]) for sp in c.species])
    rt.set_column_vmr(-1, vl, vi)
    return slices


def test_config2_full_size_sampled_bins_vs_oracle(ctx, port):
    """the headline configuration (10 000 x 100 x 20, premixed, isotropic scattering): 64 bins of the first refresh and
    flux solve against the oracle, three iterations of temperature steps against the oracle's steps fed with the
    library's own totals"""
    import bench
    from helios_amd.rt import batch_from_case
    c = bench.build_case(bench.WORKLOADS["c2"], 20242)
    c.T_lay = 900.0 + 600.0 * (np.log10(np.append(c.p_lay, c.p_lay[0])) + 1.0) / 10.0
    xs, nb = _blocks(c.nbin)
    rt = batch_from_case(ctx, c, ncol=1)
    try:
        rt.keep_down_fluxes(True)
        rt.build_planck_table(1)
        rt.run(0, 1)
        got = {k: rt.get(k) for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "scat_cross_int", "F_up_band",
                                      "F_down_band", "F_up_wg", "F_down_wg", "Fc_up_wg")}
        grid = rt.get("planck_grid")
        tot = {k: rt.get(k) for k in ("F_up_tot", "F_down_tot", "F_net", "T_lay")}
    finally:
        rt.close()
    for x0 in xs:
        b = _block_case(c, x0, nb)
        s = _oracle_block(port, b, grid, c, x0, nb, cases.refresh_premixed)
        _compare_block(got, s, c, x0, nb, ("opac_wg_lay", "opac_wg_int", "F_up_wg", "F_down_wg", "Fc_up_wg"),
                       ("scat_cross_lay", "scat_cross_int", "F_up_band", "F_down_band"))
    # totals = sum over ALL bins of the band fluxes (checked per bin above), then the reference's temperature step
    dl = np.asarray(c.opac_deltawave)
    I = c.nlayer + 1
    np.testing.assert_allclose(tot["F_up_tot"], (got["F_up_band"].reshape(I, c.nbin) * dl).sum(1), rtol=1e-12)
    np.testing.assert_allclose(tot["F_down_tot"], (got["F_down_band"].reshape(I, c.nbin) * dl).sum(1), rtol=1e-12)
    b = _block_case(c, 0, nb)
    s = cases.alloc_state(b)
    s.F_up_tot[:], s.F_down_tot[:], s.F_net[:] = tot["F_up_tot"], tot["F_down_tot"], tot["F_net"]
    T0 = np.asarray(c.T_lay, float).copy()
    b.T_lay = T0.copy()
    s.meanmolmass_lay[:] = 2.3 * 1.6605390666e-24
    port.rad_temp_iter(s.F_down_tot, s.F_up_tot, s.F_net, s.F_net_diff, b.T_lay, b.p_lay, b.p_int, s.abort, s.T_store,
                       s.deltat_prefactor, b.F_add_heat_lay, b.F_add_heat_sum, b.F_smooth, b.F_smooth_sum, b.c_p_lay,
                       s.meanmolmass_lay, 0, b.foreplay, b.g, b.nlayer, b.physical_tstep, b.rad_convergence_limit,
                       b.adapt_interval, b.smooth, b.plancktable_dim, b.plancktable_step, b.F_intern, b.no_atmo)
    np.testing.assert_allclose(tot["T_lay"], b.T_lay, rtol=1e-12)
    assert np.abs(tot["T_lay"] - T0).max() > 1e-3


def test_config3_full_size_sampled_bins_vs_oracle(ctx, port):
    """20 species at 10 000 x 100: mixed opacities, scattering and the first flux solve of 64 bins against the oracle;
    k-distributions stay ascending and fluxes non-negative everywhere"""
    import bench
    from helios_amd.rt import batch_from_case
    c = bench.build_case(bench.WORKLOADS["c3"], 20242)
    c.T_lay = 900.0 + 600.0 * (np.log10(np.append(c.p_lay, c.p_lay[0])) + 1.0) / 10.0   # a profile with structure
    xs, nb = _blocks(c.nbin)
    rt = batch_from_case(ctx, c, ncol=1, nspecies=len(c.species))
    try:
        slices = _upload_species(rt, c, xs, nb)
        rt.build_planck_table(1)
        rt.run(0, 1)
        got = {k: rt.get(k) for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "scat_cross_int", "F_up_band",
                                      "F_down_band", "F_up_wg")}
        grid = rt.get("planck_grid")
        dg = ctx.diag()
    finally:
        rt.close()
    for x0 in xs:
        b = _block_case(c, x0, nb, slices)
        s = _oracle_block(port, b, grid, c, x0, nb, cases.refresh_onthefly)
        _compare_block(got, s, c, x0, nb, ("opac_wg_lay", "opac_wg_int", "F_up_wg"),
                       ("scat_cross_lay", "scat_cross_int", "F_up_band", "F_down_band"))
    Y, X, L = c.ny, c.nbin, c.nlayer
    for k, nlev in (("opac_wg_lay", L), ("opac_wg_int", L + 1)):
        a = got[k][:nlev * X * Y].reshape(nlev, X, Y)
        assert np.all(np.isfinite(a)) and a.min() > 0
        assert np.all(np.diff(a, axis=2) >= 0), "%s: a mixed k-distribution is not ascending" % k
    assert got["F_up_band"].min() >= 0 and got["F_down_band"].min() >= 0
    assert dg["ro_rebin_skipped"] == 0


def test_config4_batch_of_sweep_columns_equals_single_runs(ctx):
    """8 columns of the parameter sweep (different gravity, orbit, internal flux, start profile) with on-the-fly mixing
    of 20 species in one batch against 8 single-column runs: 12 iterations, two refreshes"""
    import bench
    w = dict(bench.WORKLOADS["c4"], nbin=1000, ntemp=10, npress=8)
    c = bench.build_case(w, 20242)
    picks = [0, 7, 63, 64, 200, 321, 448, 511]          # corners and interior of the 8 x 8 x 8 grid

    def run(columns):
        from helios_amd.rt import batch_from_case
        cols = [bench.sweep_column(c, gi) for gi in columns]
        rt = batch_from_case(ctx, c, ncol=len(cols), nspecies=len(c.species), columns=cols)
        try:
            _upload_species(rt, c, [], 1)
            for i, cp in enumerate(cols):
                rt.set_column_profile(i, c.p_lay, c.p_int, np.full(c.nlayer + 1, cp["T_start"]), c.surf_albedo, c.starflux)
            rt.build_planck_table(1)
            rt.run(0, 12)
            return [{k: rt.get(k, i) for k in ("T_lay", "F_up_band", "F_net", "opac_wg_int", "meanmolmass_lay")}
                    for i in range(len(cols))]
        finally:
            rt.close()
    batch = run(picks)
    assert np.abs(batch[0]["T_lay"] - batch[-1]["T_lay"]).max() > 1.0        # the columns really differ
    for i, gi in enumerate(picks):
        single = run([gi])[0]
        for k in single:
            np.testing.assert_allclose(batch[i][k], single[k], rtol=1e-12, atol=0, err_msg="column %d: %s" % (gi, k))


def test_config4_sixteen_full_size_columns_in_one_batch_vs_oracle(ctx, port):
    """config 4 at its size: 16 columns of the 8 x 8 x 8 sweep (different gravity, orbital distance, internal flux and
    start profile), 10 000 bins x 100 layers each, 20 species mixed on the fly, in ONE batch -- the first refresh and
    flux solve of three of the columns (first, one from the middle, last) against the CPU oracle on sampled bins
    (mixed opacities of every Gauss point and level, scattering, spectral and band fluxes)"""
    import bench
    from helios_amd.rt import batch_from_case
    c = bench.build_case(bench.WORKLOADS["c4"], 20242)
    picks = [0, 7, 63, 64, 100, 137, 200, 255, 256, 321, 383, 384, 448, 470, 504, 511]
    xs, nb = _blocks(c.nbin)
    cols = [bench.sweep_column(c, gi) for gi in picks]
    rt = batch_from_case(ctx, c, ncol=len(cols), nspecies=len(c.species), columns=cols)
    try:
        slices = _upload_species(rt, c, xs, nb)
        for i, cp in enumerate(cols):
            rt.set_column_profile(i, c.p_lay, c.p_int, np.full(c.nlayer + 1, cp["T_start"]) * (1.0 + 0.002 * i),
                                  c.surf_albedo, c.starflux)
        rt.build_planck_table(1)
        rt.run(0, 1)
        grid = rt.get("planck_grid")
        checked = [0, 9, 15]
        got = {i: {k: rt.get(k, i) for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "scat_cross_int",
                                             "F_up_band", "F_down_band", "F_up_wg")} for i in checked}
        spectra = [rt.get("F_up_band", i)[-c.nbin:] for i in range(len(cols))]
    finally:
        rt.close()
    for i in checked:
        cp = cols[i]
        ci = cases.Case(c)
        ci.g, ci.a, ci.F_intern = cp["g"], cp["a"], cp["F_intern"]
        ci.T_lay = np.full(c.nlayer + 1, cp["T_start"]) * (1.0 + 0.002 * i)
        for x0 in xs[:2] + xs[-1:]:
            b = _block_case(ci, x0, nb, slices)
            s = _oracle_block(port, b, grid, ci, x0, nb, cases.refresh_onthefly)
            _compare_block(got[i], s, ci, x0, nb, ("opac_wg_lay", "opac_wg_int", "F_up_wg"),
                           ("scat_cross_lay", "scat_cross_int", "F_up_band", "F_down_band"))
    for i in range(1, len(cols)):                     # sixteen different columns, none a copy of its neighbour
        assert np.abs(spectra[i] / spectra[i - 1] - 1.0).max() > 1e-6


def test_default_grid_batch_with_the_species_of_species_dat_vs_oracle(ctx, port):
    """the regime HELIOS users run (bench.py's `d64s`): the reference's default grid -- 386 bins x 20 Gauss points, 105
    layers, 120 x 28 (T, P) nodes -- with the 15 species of its input/species.dat (H2O first: correlated-k, ten absorbers by
    random overlap, two CIA pairs correlated-k; Rayleigh scattering by H2O (computed), CO2, CO, H2, He) as a batch of sweep
    columns; tables formed on the device from their factors.  First refresh and flux solve of three columns against the
    CPU oracle on sampled bins."""
    import bench
    from helios_amd.rt import batch_from_case
    c = bench.build_case(bench.WORKLOADS["d64s"], 20242)
    assert [sp["name"] for sp in c.species][:3] == ["H2O", "CO2", "CO"] and len(c.species) == 15
    picks = [0, 63, 137, 200, 321, 448, 470, 511]
    xs, nb = _blocks(c.nbin)
    cols = [bench.sweep_column(c, gi) for gi in picks]
    rt = batch_from_case(ctx, c, ncol=len(cols), nspecies=len(c.species), columns=cols)
    try:
        slices = _upload_species(rt, c, xs, nb, separable=True)
        for i, cp in enumerate(cols):
            rt.set_column_profile(i, c.p_lay, c.p_int, np.full(c.nlayer + 1, cp["T_start"]) * (1.0 + 0.004 * i),
                                  c.surf_albedo, c.starflux)
        rt.build_planck_table(1)
        rt.run(0, 1)
        grid = rt.get("planck_grid")
        checked = [0, 4, 7]
        got = {i: {k: rt.get(k, i) for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "scat_cross_int",
                                             "F_up_band", "F_down_band", "F_up_wg", "meanmolmass_lay")} for i in checked}
        dg = ctx.diag()
    finally:
        rt.close()
    for i in checked:
        cp = cols[i]
        ci = cases.Case(c)
        ci.g, ci.a, ci.F_intern = cp["g"], cp["a"], cp["F_intern"]
        ci.T_lay = np.full(c.nlayer + 1, cp["T_start"]) * (1.0 + 0.004 * i)
        for x0 in xs[:1] + xs[2:]:
            b = _block_case(ci, x0, nb, slices)
            s = _oracle_block(port, b, grid, ci, x0, nb, cases.refresh_onthefly)
            _compare_block(got[i], s, ci, x0, nb, ("opac_wg_lay", "opac_wg_int", "F_up_wg"),
                           ("scat_cross_lay", "scat_cross_int", "F_up_band", "F_down_band"))
            np.testing.assert_allclose(got[i]["meanmolmass_lay"][:c.nlayer], s.meanmolmass_lay[:c.nlayer], rtol=1e-13)
    assert dg["ro_rebin_skipped"] == 0


def test_default_grid_batch_premixed_table_from_its_factors_vs_oracle(ctx, port):
    """bench.py's `d64`: the default grid with a premixed table, 64 sweep columns in ONE batch; the table is formed on the
    device from its two factors (hx_rt_set_premixed_separable).  First refresh and flux solve of two columns against the
    oracle (which reads the host-built table) on sampled bins."""
    import bench
    from helios_amd.rt import batch_from_case
    w = bench.WORKLOADS["d64"]
    c = bench.build_case(w, 20242)                          # with the host table, for the oracle
    cf = bench.build_case(w, 20242, full_tables=False)      # by its factors, for the device
    assert cf.opac_k is None and cf.opac_k_factors is not None
    ncol = w["columns_per_gpu"]
    xs, nb = _blocks(c.nbin)
    cols = [bench.sweep_column(c, gi) for gi in range(ncol)]
    rt = batch_from_case(ctx, cf, ncol=ncol, columns=cols)
    try:
        for i, cp in enumerate(cols):
            rt.set_column_profile(i, c.p_lay, c.p_int, np.full(c.nlayer + 1, cp["T_start"]), c.surf_albedo, c.starflux)
        rt.build_planck_table(1)
        rt.run(0, 1)
        grid = rt.get("planck_grid")
        checked = [5, 63]
        got = {i: {k: rt.get(k, i) for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "F_up_band", "F_down_band",
                                             "F_up_wg")} for i in checked}
    finally:
        rt.close()
    for i in checked:
        cp = cols[i]
        ci = cases.Case(c)
        ci.g, ci.a, ci.F_intern = cp["g"], cp["a"], cp["F_intern"]
        ci.T_lay = np.full(c.nlayer + 1, cp["T_start"])
        for x0 in xs:
            b = _block_case(ci, x0, nb)
            s = _oracle_block(port, b, grid, ci, x0, nb, cases.refresh_premixed)
            _compare_block(got[i], s, ci, x0, nb, ("opac_wg_lay", "opac_wg_int", "F_up_wg"),
                           ("scat_cross_lay", "F_up_band", "F_down_band"))


def test_config5_on_the_fly_full_size_sampled_bins_vs_oracle(ctx, port):
    """BASELINE config 5 as named: 30 000 bins x 200 layers, 20 species mixed on the fly (random overlap), two cloud
    decks, g0 = 0.3 with the I2S correction, direct beam, surface albedo -- the first refresh and flux solve against the
    CPU oracle on sampled bins: mixed opacities of every Gauss point and level, scattering, total g0, spectral and band
    fluxes incl. the direct beam.  (The convection loop on this shape runs in the premixed test below.)"""
    import bench
    from helios_amd.rt import batch_from_case
    c = bench.build_case(bench.WORKLOADS["c5"], 20245)
    assert c.nbin == 30000 and c.nlayer == 200 and c.dir_beam == 1 and c.clouds == 1 and c.scat_corr == 1
    c.T_lay = 700.0 + 900.0 * (np.log10(np.append(c.p_lay, c.p_lay[0])) + 1.0) / 10.0
    xs, nb = _blocks(c.nbin)
    rt = batch_from_case(ctx, c, ncol=1, nspecies=len(c.species))
    try:
        slices = _upload_species(rt, c, xs, nb)
        rt.keep_down_fluxes(True)
        rt.build_planck_table(1)
        rt.run(0, 1)
        got = {k: rt.get(k) for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "scat_cross_int", "F_up_band",
                                      "F_down_band", "F_dir_band", "F_up_wg", "F_down_wg", "g_0_tot_lay", "g_0_tot_int")}
        grid = rt.get("planck_grid")
        dg = ctx.diag()
    finally:
        rt.close()
    for x0 in xs:
        b = _block_case(c, x0, nb, slices)
        s = _oracle_block(port, b, grid, c, x0, nb, cases.refresh_onthefly)
        _compare_block(got, s, c, x0, nb, ("opac_wg_lay", "opac_wg_int", "F_up_wg", "F_down_wg"),
                       ("scat_cross_lay", "scat_cross_int", "g_0_tot_lay", "g_0_tot_int", "F_up_band", "F_down_band",
                        "F_dir_band"))
    Y, X, L = c.ny, c.nbin, c.nlayer
    for k, nlev in (("opac_wg_lay", L), ("opac_wg_int", L + 1)):
        a = got[k][:nlev * X * Y].reshape(nlev, X, Y)
        assert np.all(np.isfinite(a)) and a.min() > 0 and np.all(np.diff(a, axis=2) >= 0)
    assert got["F_up_band"].min() >= 0 and got["F_dir_band"].min() >= 0 and got["F_dir_band"].max() > 0
    assert dg["ro_rebin_skipped"] == 0


def test_config5_full_size_clouds_beam_i2s_albedo_and_convection(ctx, port):
    """30 000 bins x 200 layers, two cloud decks, g0, I2S correction, surface albedo, direct beam: the first refresh and
    flux solve of 64 bins against the oracle; then the convection loop (adjustment + sweeps + temperature step on the
    device) runs on the same batch"""
    import bench
    from helios_amd import phys_const as pc
    from helios_amd.rt import batch_from_case
    c = bench.build_case(bench.WORKLOADS["c5premixed"], 20245)
    c.dir_beam = 1
    c.T_lay = 700.0 + 900.0 * (np.log10(np.append(c.p_lay, c.p_lay[0])) + 1.0) / 10.0
    xs, nb = _blocks(c.nbin)
    rt = batch_from_case(ctx, c, ncol=1)
    try:
        rt.keep_down_fluxes(True)
        rt.build_planck_table(1)
        rt.run(0, 1)
        got = {k: rt.get(k) for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "F_up_band", "F_down_band",
                                      "F_dir_band", "F_up_wg", "F_down_wg", "g_0_tot_lay", "g_0_tot_int")}
        grid = rt.get("planck_grid")
        for x0 in xs:
            b = _block_case(c, x0, nb)
            s = _oracle_block(port, b, grid, c, x0, nb, cases.refresh_premixed)
            _compare_block(got, s, c, x0, nb, ("opac_wg_lay", "opac_wg_int", "F_up_wg", "F_down_wg"),
                           ("scat_cross_lay", "g_0_tot_lay", "g_0_tot_int", "F_up_band", "F_down_band", "F_dir_band"))
        assert got["F_up_band"].min() >= 0 and got["F_dir_band"].min() >= 0 and got["F_dir_band"].max() > 0
        # the convection loop on this column: a super-adiabatic interior is adjusted, fluxes stay finite, and the run
        # ends with every convective layer on the adiabat
        L = c.nlayer
        kap = 2.0 / 7.0
        T = np.maximum(2500.0 * (np.asarray(c.p_lay) / c.p_lay[0]) ** 0.4, 600.0)
        rt.set_temperatures(0, np.append(T, 2600.0))
        for name, v in (("kappa_lay", np.full(L, kap)), ("kappa_int", np.full(L + 1, kap)),
                        ("c_p_lay", np.full(L, pc.R_UNIV / kap)), ("conv_layer", np.zeros(L + 1, np.int32)),
                        ("conv_unstable", np.zeros(L + 1, np.int32)), ("dampara", np.array([-1.0]))):
            rt.set_state(-1, name, v)
        rt.conv_run(0, 12)
        assert np.all(np.isfinite(rt.get("T_lay"))) and np.all(np.isfinite(rt.get("F_up_band")))
        rt.conv_adjust(12)                       # the adjustment half-step of the next iteration
        T2, F = rt.get("T_lay"), rt.get("F_net")
        conv = rt.get("conv_layer")
        assert np.all(np.isfinite(T2)) and np.all(np.isfinite(F)) and conv.sum() > 5
        lay = np.where(conv[:L - 1] * conv[1:L] == 1)[0]              # neighbouring convective layers: dry adiabat
        grad = np.log(T2[lay + 1] / T2[lay]) / np.log(c.p_lay[lay + 1] / c.p_lay[lay])
        np.testing.assert_allclose(grad, kap, rtol=2e-2)
    finally:
        rt.close()


def test_config1_run_to_equilibrium_vs_the_reference_on_the_gpu(ctx):
    """BASELINE config 1 as BASELINE.json names it -- single column, premixed opacity table, 300 bins x 50 layers, no
    scattering -- as a WHOLE run: the reference's kernels (gfx950 build) under the reference's control flow
    (tests/loop_driver.py) against hx_rt_run with its device-side convergence latch: same number of iterations, T-P
    profile, net flux and emission spectrum at the end and after 1, 10, 11, 50 iterations"""
    import bench
    import loop_driver as ld
    import oracle
    from impls import RefImpl
    from helios_amd.rt import batch_from_case
    if oracle.refgpu is None:
        pytest.skip("oracle/_ref/libhelios_ref_gfx950.so not present")
    refgpu = RefImpl(oracle.refgpu)
    c0 = bench.build_case(bench.WORKLOADS["c1"], 20241)
    assert (c0.nbin, c0.nlayer, c0.scat) == (300, 50, 0)
    cr = _block_case(c0, 0, c0.nbin)
    s = cases.alloc_state(cr)
    cases.setup_planck(refgpu, cr, s)
    n_ref, snaps, reason = ld.radiation_loop(refgpu, cr, s, ld.SNAP_AT)
    assert reason == "converged" and n_ref > 100
    X, I = c0.nbin, c0.nlayer + 1
    rt = batch_from_case(ctx, c0)
    try:
        rt.build_planck_table(1)
        rt.set_state(-1, "planck_grid", s.planck_grid)
        it = 0
        stops = sorted(ld.SNAP_AT)
        while True:
            nxt = min([p for p in stops if p > it] + [it + 10 - it % 10])
            rt.run(it, nxt - it)
            done = int(rt.get("done")[0])
            it = int(rt.get("iters_done")[0]) if done else nxt
            if it in ld.SNAP_AT or done:
                want = snaps["end" if done else it]
                fs = np.abs(want["F_up_tot"]).max()
                np.testing.assert_allclose(rt.get("T_lay"), want["T_lay"], rtol=1e-6, err_msg="T_lay after %d" % it)
                np.testing.assert_allclose(rt.get("F_net"), want["F_net"], rtol=1e-6, atol=1e-9 * fs)
                np.testing.assert_allclose(rt.get("F_up_band")[X * (I - 1):], want["F_up_band_TOA"], rtol=1e-6,
                                           atol=1e-13 * want["F_up_band_TOA"].max())
            if done:
                break
            assert it < 20000
        assert it == n_ref, "iterations: %d, reference %d" % (it, n_ref)
    finally:
        rt.close()


def _premixed_column_vs_reference(ctx, c0):
    """first iteration of a premixed column through the fused path and through the reference's kernels on this GPU
    (every array above 1 MB stays on the device between the reference's launches); all bins compared"""
    import oracle
    from impls import RefImpl
    from helios_amd.rt import batch_from_case
    lib = oracle.refgpu
    if lib is None:
        pytest.skip("oracle/_ref/libhelios_ref_gfx950.so not present")
    ref = RefImpl(lib)
    keys = ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "F_up_band", "F_down_band", "F_dir_band", "planckband_lay")
    rt = batch_from_case(ctx, c0, ncol=1)
    try:
        rt.build_planck_table(1)
        grid = rt.get("planck_grid")
        rt.run(0, 1)
        got = {k: rt.get(k) for k in keys + ("F_up_tot", "F_down_tot", "F_net", "T_lay", "delta_z_lay")}
    finally:
        rt.close()
    c = _block_case(c0, 0, c0.nbin)
    s = cases.alloc_state(c)
    s.planck_grid[:] = grid
    held = []
    for d in (c, s):
        for k, v in list(d.items()):
            if isinstance(v, np.ndarray) and v.nbytes > (1 << 20) and v.dtype == np.float64:   # (the host zeroes scat_trigger)
                d[k] = lib.buf(v)
                held.append(d[k])
    try:
        cases.radiation_iterations(ref, c, s, 1)
        want = {k: s[k].get() if hasattr(s[k], "get") else s[k] for k in keys}
        want.update(F_up_tot=s.F_up_tot, F_down_tot=s.F_down_tot, F_net=s.F_net, T_lay=c.T_lay,
                    delta_z_lay=s.delta_z_lay)
    finally:
        for b in held:
            b.free()
    nwg = c0.ny * c0.nbin * c0.nlayer
    np.testing.assert_allclose(got["opac_wg_lay"][:nwg], want["opac_wg_lay"][:nwg], rtol=1e-12)
    np.testing.assert_allclose(got["opac_wg_int"], want["opac_wg_int"], rtol=1e-12)
    np.testing.assert_allclose(got["scat_cross_lay"], want["scat_cross_lay"], rtol=1e-12)
    np.testing.assert_allclose(got["delta_z_lay"], want["delta_z_lay"], rtol=1e-12)
    np.testing.assert_allclose(got["planckband_lay"], want["planckband_lay"], rtol=1e-9,
                               atol=1e-13 * want["planckband_lay"].max())
    fscale = max(want[k].max() for k in ("F_up_band", "F_down_band", "F_dir_band"))
    for k in ("F_up_band", "F_down_band", "F_dir_band"):
        np.testing.assert_allclose(got[k], want[k], rtol=1e-9, atol=1e-12 * fscale, err_msg=k)
    for k in ("F_up_tot", "F_down_tot"):
        np.testing.assert_allclose(got[k], want[k], rtol=1e-10, err_msg=k)
    np.testing.assert_allclose(got["F_net"], want["F_net"], rtol=1e-9, atol=1e-12 * np.abs(want["F_up_tot"]).max())
    np.testing.assert_allclose(got["T_lay"], want["T_lay"], rtol=1e-7)
    assert np.abs(got["T_lay"] - c0.T_lay).max() > 1e-3            # the step moved the profile


def test_config2_full_size_all_bins_vs_the_reference_on_the_gpu(ctx):
    """the headline configuration against THE REFERENCE ITSELF at full size: its kernels.cu (hipcc build, oracle/_ref)
    runs the first iteration of the 10 000 x 100 x 20 column on this GPU and every bin's opacities, band fluxes, the
    totals and the temperature step are compared"""
    import bench
    c0 = bench.build_case(bench.WORKLOADS["c2"], 20242)
    c0.T_lay = 900.0 + 600.0 * (np.log10(np.append(c0.p_lay, c0.p_lay[0])) + 1.0) / 10.0
    _premixed_column_vs_reference(ctx, c0)


def test_config2_full_size_cached_state_and_launch_order_do_not_change_the_run(ctx, monkeypatch):
    """the headline batch chooses the back-and-forth launch order with 240 MiB of up-flux state left in the Infinity
    Cache by itself; 23 iterations -- launches in both directions, three refreshes -- are the run
    with front-to-back launches and non-temporal stores bit for bit"""
    import bench
    from helios_amd.rt import batch_from_case
    c0 = bench.build_case(bench.WORKLOADS["c2"], 20242)
    c0.T_lay = 900.0 + 600.0 * (np.log10(np.append(c0.p_lay, c0.p_lay[0])) + 1.0) / 10.0

    def run(force_off):
        if force_off:
            monkeypatch.setenv("HELIOS_RT_SERPENTINE", "0")
        else:
            monkeypatch.delenv("HELIOS_RT_SERPENTINE", raising=False)
        monkeypatch.delenv("HELIOS_RT_STATE_CACHE_MB", raising=False)
        rt = batch_from_case(ctx, c0, ncol=1)
        try:
            policy = rt.get("flux_launch_policy")
            rt.build_planck_table(1)
            rt.run(0, 23)
            return policy, {k: rt.get(k) for k in ("T_lay", "F_net", "F_up_band", "F_down_band", "delta_t_prefactor")}
        finally:
            rt.close()
    p_on, on = run(False)
    p_off, off = run(True)
    assert list(p_on) == [1.0, 240.0] and list(p_off) == [0.0, 0.0]
    for k in off:
        np.testing.assert_array_equal(on[k], off[k], err_msg=k)
    assert np.abs(on["T_lay"] - c0.T_lay).max() > 1.0


def test_config5_full_size_all_bins_vs_the_reference_on_the_gpu(ctx):
    """30 000 bins x 200 layers, two cloud decks, g0, I2S correction, surface albedo and the direct beam: the same
    comparison with the reference's kernels on this GPU, every bin"""
    import bench
    c0 = bench.build_case(bench.WORKLOADS["c5premixed"], 20245)
    c0.dir_beam = 1
    c0.T_lay = 700.0 + 900.0 * (np.log10(np.append(c0.p_lay, c0.p_lay[0])) + 1.0) / 10.0
    _premixed_column_vs_reference(ctx, c0)


def test_config3_full_size_all_bins_vs_the_reference_on_the_gpu(ctx):
    """config 3 against THE REFERENCE ITSELF at full size: its opac_species_interpol + add_to_mixed_opac (one thread per
    point, 9.6 KB of scratch, adjacent-swap sort) fold the 20 absorbers into 2.01 million (bin, level) points on this GPU,
    then its transmission / sweep / quadrature kernels run the first iteration; every bin's mixed opacities, scattering
    cross-sections and band fluxes are compared with the one-launch species loop of the fused path"""
    import bench
    import oracle
    from impls import RefImpl
    from helios_amd import phys_const as pc
    from helios_amd import synthetic as syn
    from helios_amd.rt import batch_from_case
    lib = oracle.refgpu
    if lib is None:
        pytest.skip("oracle/_ref/libhelios_ref_gfx950.so not present")
    ref = RefImpl(lib)
    c0 = bench.build_case(bench.WORKLOADS["c3"], 20242)
    c0.T_lay = 900.0 + 600.0 * (np.log10(np.append(c0.p_lay, c0.p_lay[0])) + 1.0) / 10.0
    X, Y, L, I = c0.nbin, c0.ny, c0.nlayer, c0.nlayer + 1
    c = _block_case(c0, 0, X)
    s = cases.alloc_state(c)
    held = []
    for d in (c, s):
        for k, v in list(d.items()):
            if isinstance(v, np.ndarray) and v.nbytes > (1 << 20):
                d[k] = lib.buf(v)
                held.append(d[k])
    rt = batch_from_case(ctx, c0, ncol=1, nspecies=len(c0.species))
    try:
        # the species loop of computation.py:1454-1501 through the reference's kernels, one table on the host at a time
        ref.temp_inter(c.T_lay, c.T_int, I)
        vl = np.array([np.full(L, sp["vmr"]) for sp in c0.species])
        vi = np.array([np.full(I, sp["vmr"]) for sp in c0.species])
        w = np.array([sp["weight"] for sp in c0.species])
        s.meanmolmass_lay[:] = (vl * w[:, None]).sum(0) / vl.sum(0) * pc.AMU
        s.meanmolmass_int[:] = (vi * w[:, None]).sum(0) / vi.sum(0) * pc.AMU
        spec_l, spec_i = lib.buf(np.zeros(Y * X * I)), lib.buf(np.zeros(Y * X * I))
        held += [spec_l, spec_i]
        sc_l, sc_i = np.zeros(X * L), np.zeros(X * I)
        scat_l, scat_i = np.zeros(X * L), np.zeros(X * I)
        for k, sp in enumerate(c0.species):
            tab = sp["pretab"]
            if isinstance(tab, str):
                tab = syn.ktable(np.random.default_rng(sp["table_seed"]), X, Y, c0.ktemp, c0.kpress, c0.gauss_y)
            rt.set_species(k, tab, sp["scat"], sp["weight"], is_h2o=0, is_cia=0, in_mu=1)
            if tab is not None:
                d_tab = lib.buf(tab)
                del tab
                ref.opac_species_interpol(c.T_lay, c0.ktemp, c0.p_lay, c0.kpress, d_tab, spec_l, c0.npress, c0.ntemp, Y, X, L)
                ref.opac_species_interpol(c.T_int, c0.ktemp, c0.p_int, c0.kpress, d_tab, spec_i, c0.npress, c0.ntemp, Y, X, I)
                d_tab.free()
                ref.add_to_mixed_opac(np.ascontiguousarray(vl[k]), spec_l, s.opac_wg_lay, s.meanmolmass_lay, c0.gauss_weight,
                                      c0.gauss_y, sp["weight"] * pc.AMU, k, 1, Y, X, L)
                ref.add_to_mixed_opac(np.ascontiguousarray(vi[k]), spec_i, s.opac_wg_int, s.meanmolmass_int, c0.gauss_weight,
                                      c0.gauss_y, sp["weight"] * pc.AMU, k, 1, Y, X, I)
            if sp["scat"] is not None:
                sc_l[:], sc_i[:] = np.tile(sp["scat"], L), np.tile(sp["scat"], I)
                ref.add_to_mixed_scat(np.ascontiguousarray(vl[k]), sc_l, scat_l, X, L)
                ref.add_to_mixed_scat(np.ascontiguousarray(vi[k]), sc_i, scat_i, X, I)
        s.scat_cross_lay.set(scat_l) if hasattr(s.scat_cross_lay, "set") else s.scat_cross_lay.__setitem__(slice(None), scat_l)
        s.scat_cross_int.set(scat_i) if hasattr(s.scat_cross_int, "set") else s.scat_cross_int.__setitem__(slice(None), scat_i)
        rt.set_column_vmr(-1, vl, vi)
        rt.build_planck_table(1)
        grid = rt.get("planck_grid")
        rt.run(0, 1)
        got = {k: rt.get(k) for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "scat_cross_int", "F_up_band",
                                      "F_down_band", "F_net", "T_lay")}
        # the rest of the first iteration through the reference: Planck, transmission, sweeps, quadrature, step
        s.planck_grid.set(grid) if hasattr(s.planck_grid, "set") else s.planck_grid.__setitem__(slice(None), grid)
        ref.planck_interpol_layer(c.T_lay, s.planckband_lay, s.planck_grid, c.starflux, c.real_star, L, X,
                                  c.plancktable_dim, c.plancktable_step)
        ref.planck_interpol_interface(c.T_int, s.planckband_int, s.planck_grid, I, X, c.plancktable_dim, c.plancktable_step)
        cases.refresh_transmission(ref, c, s)
        cases.flux_sweeps(ref, c, s)
        cases.integrate_and_step(ref, c, s, 0)
        want = {k: (s[k].get() if hasattr(s[k], "get") else s[k]) for k in ("opac_wg_lay", "opac_wg_int", "F_up_band",
                                                                          "F_down_band", "F_net")}
        want.update(scat_cross_lay=scat_l, scat_cross_int=scat_i, T_lay=c.T_lay)
    finally:
        rt.close()
        for b in held:
            if b.ptr:
                b.free()
    nwg = Y * X * L
    np.testing.assert_allclose(got["opac_wg_lay"][:nwg], want["opac_wg_lay"][:nwg], rtol=5e-11)
    np.testing.assert_allclose(got["opac_wg_int"], want["opac_wg_int"], rtol=5e-11)
    np.testing.assert_allclose(got["scat_cross_lay"], want["scat_cross_lay"], rtol=1e-12)
    np.testing.assert_allclose(got["scat_cross_int"], want["scat_cross_int"], rtol=1e-12)
    for k in ("F_up_band", "F_down_band"):
        np.testing.assert_allclose(got[k], want[k], rtol=1e-9, atol=1e-13 * want[k].max(), err_msg=k)
    np.testing.assert_allclose(got["F_net"], want["F_net"], rtol=1e-9, atol=1e-12 * np.abs(want["F_up_band"]).max() * 1e3)
    # the step divides F_net[i] - F_net[i+1], a difference of sums over 10 000 bins: 100 x the flux tolerance (1.1e-9 seen)
    np.testing.assert_allclose(got["T_lay"], want["T_lay"], rtol=1e-7)
