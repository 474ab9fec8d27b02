"""GPU parity of the fused fast path (hx_rt_*: coefficient tiles + register-resident sweeps) against
the CPU oracle chain, the golden vectors, and -- at BASELINE.json's full size -- against the per-stage
HIP kernels and physical invariants."""
import numpy as np
import pytest

import cases
import golden_checks as gc
import fused_helpers as fh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from helios_amd.device import Context
    c = Context(0)
    yield c


FUSED_CONFIGS = {
    "default": dict(),
    "noscat": dict(scat=0),
    "dirbeam": dict(dir_beam=1, albedo=0.3),
    "dirbeam_zenith": dict(dir_beam=1, geom_zenith_corr=1, zenith_deg=80.0),
    "clouds_g0": dict(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2),
    "thin_top": dict(thin_top=True, nlayer=12),
    "ny1": dict(ny=1, nbin=17),
    "L100": dict(nbin=24, nlayer=100),          # k=16, 13 rows per lane: the BASELINE shape
    "L105": dict(nbin=12, nlayer=105),          # the reference's default layer count
    "L50": dict(nbin=21, nlayer=50, clouds=1),  # k=16, 7 rows per lane (k=8 before the compile-time scans were preferred)
    "L200": dict(nbin=7, nlayer=200, dir_beam=1),  # k=32, two Gauss-point partials per bin
    "L33_ny16": dict(nbin=9, nlayer=33, ny=16),
    "L400": dict(nbin=5, nlayer=400, clouds=1, g_0=0.1),   # k=64: one spectral point per wavefront, >64 KiB LDS in k_rt_coef
    "L200_i2s": dict(nbin=6, nlayer=200, clouds=1, scat_corr=1, g_0=0.2, albedo=0.1),  # config-5 flags: 4 planes + clouds
    "L512": dict(nbin=3, nlayer=512, dir_beam=1),   # the largest column whose rows are all in registers: k=64, 16 rows per lane
    # beyond it (round 6): 20, 24, 28, 32 rows on 64 lanes, part of the register image in scratch -- up to 1024 layers
    "L600": dict(nbin=3, nlayer=600, dir_beam=1, albedo=0.2),            # 1200 half-layers: 20 rows (1280 slots)
    "L1024_clouds": dict(nbin=2, nlayer=1024, clouds=1, g_0=0.1),        # 2048 half-layers: 32 rows, every slot in use
    "iso_L1500": dict(iso=1, nbin=2, nlayer=1500, albedo=0.1),           # isothermal: 1500 segments, 24 rows
    "smallest": dict(nbin=2, nlayer=4),             # two bins, four layers (the height integration needs a layer below 10 bar)
    # fewer half-layers than the eight lanes of the smallest tiling (H = 6, 4 < k = 8: lanes without a row); a deeper bottom
    # boundary, because calculate_height_z (host_functions.py) starts from the lowest layer centre at >= 10 bar and the
    # three layers of the standard grid have none -- that, not the kernels, is why round 5's "L3" case became "L4"
    "L3": dict(nbin=5, nlayer=3, p_boa=1e11),
    "L2_beam": dict(nbin=3, nlayer=2, p_boa=1e11, dir_beam=1, albedo=0.2),
    # isothermal layers (fband_iso / calc_trans_iso / fdir_iso): one segment per layer
    "iso": dict(iso=1, nbin=11, nlayer=20),
    "iso_clouds_beam": dict(iso=1, nbin=9, nlayer=37, clouds=1, scat_corr=1, g_0=0.2, dir_beam=1, albedo=0.15),
    "iso_zenith_L100": dict(iso=1, nbin=6, nlayer=100, dir_beam=1, geom_zenith_corr=1, zenith_deg=70.0),
    "iso_noscat": dict(iso=1, scat=0, nbin=7, nlayer=16),
}


@pytest.mark.parametrize("name", sorted(FUSED_CONFIGS))
def test_fused_vs_oracle(ctx, port, name):
    c0 = cases.make_case(**FUSED_CONFIGS[name])
    for n_iter, rtol in ((1, 1e-9), (12, 1e-7)):
        # the oracle runs on the device-built Planck table: the table's Rayleigh-Jeans tail carries
        # ~1e-8 cancellation noise that differs between libm's (tests/golden_checks.py)
        f, grid = fh.run_fused(ctx, c0, n_iter, with_planck_grid=True)
        o = fh.run_oracle(port, c0, n_iter, planck_grid=grid)
        fh.compare(f, o, c0, rtol=rtol)


@pytest.mark.parametrize("option", ["real_star", "foreplay", "no_atmo", "smooth", "species_correlated_k"])
def test_fused_run_options_vs_oracle(ctx, port, option):
    """options of the iteration that the other cases leave at their defaults: a stellar spectrum read from file
    (`real_star`: planck_interpol_layer takes the star's row from `starflux`, kernels.cu:945-951), temperature steps only
    from iteration `foreplay` on (computation.py:906), the "no atmosphere" mode of rad_temp_iter (:2739-2745), the
    temperature smoothing flux (:2656-2670), and on-the-fly absorbers mixed correlated-k throughout
    (`kcoeff_mixing = correlated-k`, computation.py:1343-1348)"""
    c0 = cases.make_case(nbin=10, nlayer=14, dir_beam=1, albedo=0.1)
    refresh = None
    if option == "real_star":
        c0.real_star = 1
        c0.starflux = 3e5 * (1.0 + 0.5 * np.sin(np.arange(c0.nbin)))
    elif option == "foreplay":
        c0.foreplay = 5
    elif option == "no_atmo":
        c0.no_atmo = 1
    elif option == "smooth":
        c0.smooth = 1
    else:
        cases.add_species(c0, nspecies=4)
        c0.kcoeff_mixing = "correlated-k"
        refresh = lambda i, c, s: cases.refresh_onthefly(i, c, s, ro=0)        # noqa: E731
    for n_iter, rtol in ((1, 1e-9), (12, 1e-7)):
        f, grid = fh.run_fused(ctx, c0, n_iter, with_planck_grid=True)
        oc = c0.copy()
        if option == "real_star":    # the oracle is handed the device's Planck table; the energy correction of the star's
            # own spectrum (corr_inc_energy scales `starflux` in place, kernels.cu:434-466) is still its to do
            port.corr_inc_energy(np.array(grid), oc.starflux, oc.opac_deltawave, 1, oc.nbin, oc.T_star, oc.plancktable_dim)
            assert np.abs(oc.starflux / c0.starflux - 1.0).min() > 0.1
        o = fh.run_oracle(port, oc, n_iter, planck_grid=grid, refresh=refresh)
        fh.compare(f, o, c0, rtol=rtol)
    if option == "foreplay":         # five iterations without a temperature step
        np.testing.assert_array_equal(fh.run_fused(ctx, c0, 5)["T_lay"], c0.T_lay)
    if option == "no_atmo":
        assert np.all(f["T_lay"][:-1] == 1.001)


def test_fused_rocky_planet_heights_and_zenith_correction(ctx, port):
    """planet type "rocky": the layer heights are integrated from the surface instead of the 10-bar level
    (host_functions.py:673-698), and with the geometric zenith-angle correction they enter the direct beam
    (kernels.cu:1349) -- the fused refresh (k_rt_height on the device) against the oracle with the host's integration"""
    c0 = cases.make_case(nbin=11, nlayer=30, dir_beam=1, geom_zenith_corr=1, zenith_deg=82.0, albedo=0.1)
    c0.planet_type = "rocky"
    c0.R_planet = 0.1 * c0.R_planet          # a small planet: the correction is strong
    gas = c0.copy()
    gas.planet_type = "gas"
    for n_iter, rtol in ((1, 1e-9), (12, 1e-7)):
        f, grid = fh.run_fused(ctx, c0, n_iter, with_planck_grid=True)
        o = fh.run_oracle(port, c0, n_iter, planck_grid=grid)
        fh.compare(f, o, c0, rtol=rtol)
    fr, fg = fh.run_fused(ctx, c0, 1), fh.run_fused(ctx, gas, 1)
    assert fr["z_lay"][0] > 0 and fg["z_lay"][0] < 0           # heights above the surface / around the 10-bar level
    top = fg["F_dir_band"] > 1e-30 * fg["F_dir_band"].max()
    assert np.abs(fr["F_dir_band"][top] / fg["F_dir_band"][top] - 1.0).max() > 1e-6       # and the beam sees the difference


@pytest.mark.parametrize("name", ["default", "dirbeam", "clouds_g0", "L100", "L50", "L200", "L400", "L200_i2s"])
def test_fused_vs_oracle_single_wavefront_workgroups(ctx, port, name, monkeypatch):
    """small spectral grids default to 5-wavefront workgroups (all Gauss points of a bin at once); the shape large
    grids get -- one wavefront walking through the Gauss-point groups -- is pinned here on the same cases"""
    monkeypatch.setenv("HELIOS_RT_MAXTHREADS", "64")
    c0 = cases.make_case(**FUSED_CONFIGS[name])
    f, grid = fh.run_fused(ctx, c0, 12, with_planck_grid=True)
    o = fh.run_oracle(port, c0, 12, planck_grid=grid)
    fh.compare(f, o, c0, rtol=1e-7)


@pytest.mark.parametrize("name,k", [("L100", 16), ("L200", 32), ("L50", 16), ("clouds_g0", 16), ("iso_zenith_L100", 16),
                                    ("L200_i2s", 32), ("L100", 32), ("L50", 8), ("L400", 64), ("L512", 64), ("L200", 64)])
def test_compile_time_scans_equal_the_generic_kernel_bit_for_bit(ctx, name, k, monkeypatch):
    """k_rt_flux<ROWS, K> (straight-line DPP scans for K = 16 / 32 / 64, identity fill instead of selects, row_newbcast)
    against k_rt_flux<ROWS, 0> (runtime k, the kernel of rounds 1-2 and still the one for k = 8): the same bits in
    every flux and temperature after 12 iterations, in both workgroup shapes (k = 64: to rounding, see below).  (k = 8 has no compile-time variant: the
    pair is then the same kernel, which pins the knob itself.)"""
    c0 = cases.make_case(**FUSED_CONFIGS[name])
    monkeypatch.setenv("HELIOS_RT_K", str(k))
    for threads in ("64", "320"):
        monkeypatch.setenv("HELIOS_RT_MAXTHREADS", threads)
        # k = 64: the generic kernel composes over distances 1 .. 32, the compile-time one inside the rows first and then
        # the row totals -- another association of the same products (k = 32 does the same in both): equal to rounding
        # after one iteration; after twelve the temperature iteration has amplified that as it does any rounding
        # difference (cf. the tolerances of test_fused_vs_oracle)
        for n_iter, rtol in ((1, 1e-12), (12, 1e-7)) if k == 64 else ((12, 0.0),):
            monkeypatch.setenv("HELIOS_RT_GENERIC_SCANS", "0")
            a = fh.run_fused(ctx, c0, n_iter)
            monkeypatch.setenv("HELIOS_RT_GENERIC_SCANS", "1")
            b = fh.run_fused(ctx, c0, n_iter)
            for key in fh.keys_for(c0):
                msg = "%s, %s threads, %d iterations" % (key, threads, n_iter)
                if rtol == 0.0:
                    np.testing.assert_array_equal(a[key], b[key], err_msg=msg)
                else:
                    scale = np.abs(b[key]).max() if np.size(b[key]) else 0.0
                    # the temperature step divides differences of nearly equal fluxes: its output carries their
                    # rounding amplified (1e-10 observed at 512 layers after one step)
                    loose = key in ("T_lay", "T_int", "delta_t_prefactor")
                    np.testing.assert_allclose(a[key], b[key], rtol=max(rtol, 1e-8) if loose else rtol,
                                               atol=1e-13 * scale, err_msg=msg)


@pytest.mark.parametrize("name", ["clouds_g0", "L50", "L200_i2s", "iso_clouds_beam", "L400"])
def test_cloud_terms_from_lds_or_from_the_bin_major_rows(ctx, name, monkeypatch):
    """k_rt_coef takes the clouds' half-layer terms from its LDS image when that fits and straight from the rows
    k_rt_half_bands writes when it does not: the same bits either way"""
    c0 = cases.make_case(**FUSED_CONFIGS[name])
    monkeypatch.setenv("HELIOS_RT_CLOUD_LDS", "1")
    a = fh.run_fused(ctx, c0, 12)
    monkeypatch.setenv("HELIOS_RT_CLOUD_LDS", "0")
    b = fh.run_fused(ctx, c0, 12)
    for key in fh.keys_for(c0):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)


@pytest.mark.parametrize("k", [16, 32, 64])
def test_every_row_count_of_the_compile_time_kernels(ctx, k, monkeypatch):
    """k_rt_flux<ROWS, K> for ROWS = 1 .. 16 at K = 16, 32, 64 (columns of k/2, k, 3k/2, ... layers, and one layer
    less: a partly filled last lane) against the runtime-k kernel after three iterations: the same bits for K = 16 and
    32, fluxes to 1e-12 for K = 64 (another association of the scan, see above)"""
    monkeypatch.setenv("HELIOS_RT_K", str(k))
    keys = ["F_up_wg", "F_down_wg", "Fc_up_wg", "Fc_down_wg", "F_up_band", "F_down_band", "F_net"]
    for rows in range(1, 17):
        for nlayer in sorted({max(2, rows * k // 2), max(2, rows * k // 2 - 1)}):
            if (2 * nlayer + k - 1) // k != rows:
                continue
            c0 = cases.make_case(nbin=3, nlayer=nlayer, dir_beam=rows % 2, clouds=1 if rows % 3 == 0 else 0)
            monkeypatch.setenv("HELIOS_RT_GENERIC_SCANS", "0")
            a = fh.run_fused(ctx, c0, 3, keys=keys)
            monkeypatch.setenv("HELIOS_RT_GENERIC_SCANS", "1")
            b = fh.run_fused(ctx, c0, 3, keys=keys)
            for key in keys:
                msg = "%s, k = %d, ROWS = %d, %d layers" % (key, k, rows, nlayer)
                if k == 64:
                    np.testing.assert_allclose(a[key], b[key], rtol=1e-12, atol=1e-13 * np.abs(b[key]).max(), err_msg=msg)
                else:
                    np.testing.assert_array_equal(a[key], b[key], err_msg=msg)


def test_batches_give_their_device_memory_back(ctx):
    """batches (premixed, with clouds, deep column) created, stepped and closed over and over: after a few warm-up
    rounds (the HIP allocator keeps some blocks for itself) the free device memory no longer moves, and a batch that
    fails to build (1025 layers) leaves nothing behind either"""
    from helios_amd._lib import HeliosHipError

    def round_():
        for cfg in (dict(nbin=24, nlayer=100), dict(nbin=9, nlayer=50, clouds=1, scat_corr=1, g_0=0.2),
                    dict(nbin=5, nlayer=400, dir_beam=1)):
            fh.run_fused(ctx, cases.make_case(**cfg), 2, keys=["T_lay"])
        with pytest.raises(HeliosHipError):
            fh.run_fused(ctx, cases.make_case(nbin=3, nlayer=1025), 1, keys=["T_lay"])

    for _ in range(3):
        round_()
    ctx.synchronize()
    free0 = ctx.mem_info()[0]
    for _ in range(12):
        round_()
    ctx.synchronize()
    assert abs(ctx.mem_info()[0] - free0) <= 1 << 20, (free0, ctx.mem_info())


@pytest.mark.parametrize("name", gc.CHAIN_NAMES)
def test_fused_golden(ctx, name):
    """the fused path against the reference-generated golden vectors (isothermal and non-isothermal layers)"""
    c, z = gc.load_chain(name)
    for n_iter, prefix, rtol in ((1, "it1.", 1e-9), (12, "it12.", 1e-7)):
        f = fh.run_fused(ctx, c, n_iter)
        want = {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}
        want["delta_t_prefactor"] = want["deltat_prefactor"]
        L = c.nlayer
        want["opac_wg_lay"] = want["opac_wg_lay"][:c.ny * c.nbin * L]
        keys = [k for k in fh.FUSED_KEYS if k in want]
        fh.compare(f, want, c, rtol=rtol, keys=keys)


def test_fused_batch_of_columns(ctx, port):
    """three columns with different temperature profiles in one batch == three single runs"""
    c0 = cases.make_case(nbin=11, nlayer=14)
    Ts = [c0.T_lay, c0.T_lay * 1.1, c0.T_lay * 0.8 + 50.0]
    outs, grid = fh.run_fused(ctx, c0, 11, ncol=3, col=[0, 1, 2], T_per_col=Ts, with_planck_grid=True)
    for T, f in zip(Ts, outs):
        c = c0.copy()
        c.T_lay = T.copy()
        o = fh.run_oracle(port, c, 11, planck_grid=grid)
        fh.compare(f, o, c0, rtol=1e-7)


# `flux calculation method = matrix` inside the device-resident loop (hx_rt_flags.matrix): calc_trans_* per refresh and
# one tridiagonal solve per spectral point and iteration (kernels.cu:1803-2424) instead of coefficient tiles and sweeps
MATRIX_CONFIGS = {
    "default": dict(albedo=0.1),
    "noscat": dict(scat=0, albedo=0.1),                 # every point takes the solver's pure-absorption branch
    "dirbeam": dict(dir_beam=1, albedo=0.3),
    "clouds_g0": dict(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2),
    "L100": dict(nbin=24, nlayer=100, albedo=0.1),
    "iso": dict(iso=1, nbin=11, nlayer=20, albedo=0.1),
    "iso_clouds_beam": dict(iso=1, nbin=9, nlayer=37, clouds=1, scat_corr=1, g_0=0.2, dir_beam=1, albedo=0.15),
    # the direct solve on every kind of tiling (round 5): 32 and 64 lanes per spectral point (scans across DPP rows), the
    # runtime-k scans of short columns (k = 8), one Gauss point per bin, the reference's default layer count
    "L200_beam": dict(nbin=7, nlayer=200, dir_beam=1, albedo=0.2),
    "L400_clouds": dict(nbin=5, nlayer=400, clouds=1, g_0=0.1, albedo=0.1),
    "L4": dict(nbin=10, nlayer=4, albedo=0.3),
    # H = 6 and 4 half-layers on k = 8 lanes (see FUSED_CONFIGS["L3"] for the bottom pressure), isothermal: H = 5 segments
    "L3": dict(nbin=10, nlayer=3, albedo=0.3, p_boa=1e11),
    "L2_beam": dict(nbin=6, nlayer=2, albedo=0.2, dir_beam=1, p_boa=1e11),
    "iso_L5": dict(iso=1, nbin=7, nlayer=5, albedo=0.1),
    # a grazing beam into strongly forward-scattering clouds: the direct-beam sources min(0, ...) are at their largest
    # against the thermal ones (the regime in which a flux of the direct solve can come out negative, kernels.cu:2268)
    "beam_forward": dict(nbin=16, nlayer=30, dir_beam=1, clouds=1, g_0=0.85, scat_corr=1, albedo=0.3, zenith_deg=85.0, T_star=9000.0),
    "ny1": dict(ny=1, nbin=17, albedo=0.1),
    "L105_i2s": dict(nbin=12, nlayer=105, clouds=1, scat_corr=1, g_0=0.2, albedo=0.1),
    "L700_beam": dict(nbin=3, nlayer=700, dir_beam=1, albedo=0.2),      # the direct solve beyond 16 rows per lane: 24 rows (round 6)
}


@pytest.mark.parametrize("name", sorted(MATRIX_CONFIGS))
def test_fused_matrix_method_vs_oracle(ctx, port, name):
    c0 = cases.make_case(**MATRIX_CONFIGS[name])
    c0.flux_calc_method = "matrix"
    # 100 layers: the elimination over 402 unknowns amplifies last-bit differences of the temperatures it is fed -- the
    # same kernels agree with the oracle to 1e-12 on identical inputs (test_matrix_flux_solve_vs_oracle) and to 1e-9 after
    # one iteration here, 2e-7 after twelve (the reference's documentation calls the method unstable, parameters.rst:326)
    # Round 5: the device-resident loop solves the same equations as three scans (k_rt_flux<.., true>) that
    # carry the reflectivity of the atmosphere below a node where the reference's elimination carries its reciprocal.  The
    # reference's down-fluxes have rounding noise of their own (tests/matrix_referee.py); after ONE iteration the spectral
    # fluxes are therefore held, at the same 1e-9, to the extended-precision solution of the reference's system, with the
    # reference's distance from it asserted next to it.  Everything else, and the twelve-iteration run, against the oracle.
    import matrix_referee
    deep = name in ("L100", "L200_beam", "L400_clouds", "L105_i2s", "L700_beam")     # (the elimination over >= 400 unknowns, see above)
    # (700 layers, 2 804 unknowns: 5.6e-6 observed after twelve iterations on band fluxes 1e-6 of the largest -- the reference's
    # noise, which grows with the depth of the column, fed back through the temperatures)
    for n_iter, rtol in ((1, 1e-9), (12, 2e-5 if name == "L700_beam" else 5e-6 if deep else 1e-7)):
        f, grid = fh.run_fused(ctx, c0, n_iter, with_planck_grid=True)
        o = fh.run_oracle(port, c0, n_iter, planck_grid=grid)
        nscat = int(o["scat_trigger"].sum())      # both branches of the solver: "default" mixes them
        assert nscat == 0 if name == "noscat" else nscat > 0 and (name != "default" or nscat < o["scat_trigger"].size)
        if n_iter == 1:
            # (700 layers: three temperatures at 1.6e-7 -- the same amplification of the reference's down-flux noise, deeper)
            matrix_referee.compare_first_solve(fh, f, o, c0, rtol, rtol_T={"L200_beam": 1e-7, "L400_clouds": 1e-7, "L700_beam": 5e-7}.get(name))
        elif name == "L700_beam":
            # twelve iterations of a 700-layer column: temperatures and totals; the spectral down-fluxes of the reference's
            # own elimination are noise below 1e-9 of the largest flux at this depth (85 of 42 060 entries off by up to 3e-3 of
            # themselves, 1.4 against 2e12) -- held above, after the first solve, to the extended-precision solution
            fh.compare(f, o, c0, rtol=rtol, keys=["T_lay", "F_up_tot", "F_down_tot", "F_up_band"])
        else:
            fh.compare(f, o, c0, rtol=rtol)


def test_fused_matrix_method_where_negative_fluxes_would_show(ctx, port):
    """the one place where the three scans and the reference's elimination can part is a genuinely negative flux: the reference
    replaces it by its absolute value row by row during the back-substitution (kernels.cu:2268) and carries the flipped value
    on, the scans flip each value once it is produced.  The regime that could produce one -- a grazing beam into strongly
    forward-scattering clouds over a dark surface, under a hot star, the column hot and five times colder -- yields none
    (debug = 1: no negative flux counted after the solve) and spectral fluxes on the oracle's to 1e-11.  (A counter inside
    the solve was built and measured: +1.5 % on the memory-bound kernel for an event never seen; not kept.)"""
    for cold in (1.0, 0.2):
        c0 = cases.make_case(nbin=16, nlayer=60, dir_beam=1, clouds=1, g_0=0.99, scat_corr=1, albedo=0.01, zenith_deg=89.5,
                             T_star=40000.0, f_factor=1.0)
        c0.flux_calc_method, c0.debug = "matrix", 1
        c0.T_lay = c0.T_lay * cold
        ctx.diag_reset()
        f, grid = fh.run_fused(ctx, c0, 1, with_planck_grid=True)
        d = ctx.diag()
        assert d["negative_down_flux"] == 0 and d["negative_up_flux"] == 0, d
        o = fh.run_oracle(port, c0, 1, planck_grid=grid)
        for k in ("F_down_wg", "F_up_wg"):
            assert o[k].min() >= 0.0
            np.testing.assert_allclose(f[k], o[k], rtol=1e-9, atol=1e-11 * np.abs(o[k]).max(), err_msg=k)
    ctx.diag_reset()


def test_fused_matrix_method_batch_of_columns(ctx, port):
    """the matrix solve column by column inside one batch: three start profiles == three single runs, and a column whose
    loop has ended (its own, loose criterion) keeps the fluxes of its last iteration while the others go on"""
    from helios_amd.rt import batch_from_case
    c0 = cases.make_case(nbin=11, nlayer=14, dir_beam=1, albedo=0.1)
    c0.flux_calc_method = "matrix"
    Ts = [c0.T_lay, c0.T_lay * 1.1, c0.T_lay * 0.8 + 50.0]
    rt = batch_from_case(ctx, c0, ncol=3)
    try:
        for i, T in enumerate(Ts):
            rt.set_temperatures(i, T)
        rt.set_convergence_limit(1, 0.9)          # column 1 converges on its first step
        rt.build_planck_table(1)
        grid = rt.get("planck_grid")
        rt.run(0, 12)
        assert [int(rt.get("done", i)[0]) for i in range(3)] == [0, 1, 0]
        assert int(rt.get("iters_done", 1)[0]) == 1
        outs = [{k: rt.get(k, i) for k in fh.FUSED_KEYS} for i in range(3)]
    finally:
        rt.close()
    for i, (T, f) in enumerate(zip(Ts, outs)):
        c = c0.copy()
        c.T_lay = T.copy()
        if i == 1:
            c.rad_convergence_limit = 0.9
        o = fh.run_oracle(port, c, 1 if i == 1 else 12, planck_grid=grid)
        if i == 1:   # the loop left after the step of iteration 0: fluxes of that iteration, temperatures after its step
            assert o["abort"].sum() == c0.nlayer + 1
        fh.compare(f, o, c0, rtol=1e-7)


def test_fused_matrix_method_refuses_a_black_surface(ctx):
    """row 0 of the tridiagonal system divides by the surface albedo (kernels.cu:2203-2215; the reference's reader keeps it
    >= 1e-8, read.py:1261): a zero is refused when the column is handed over instead of ending in NaNs"""
    from helios_amd import _lib
    from helios_amd.rt import batch_from_case
    c0 = cases.make_case(nbin=5, nlayer=6, albedo=0.0)
    c0.flux_calc_method = "matrix"
    with pytest.raises(_lib.HeliosHipError, match="albedo"):
        batch_from_case(ctx, c0)


def test_fused_stops_at_convergence(ctx, port):
    """a column whose every layer satisfies the criterion is frozen on the device exactly where the
    reference's loop would exit (computation.py:938), however late the host looks"""
    c0 = cases.make_case(nbin=9, nlayer=8)
    c0.rad_convergence_limit = 0.9    # absurdly loose: converges on the first step
    from helios_amd.rt import batch_from_case
    rt = batch_from_case(ctx, c0)
    rt.build_planck_table(1)
    rt.run(0, 7)
    assert rt.get("done")[0] == 1 and rt.get("iters_done")[0] == 1
    T7 = rt.get("T_lay")
    rt.close()
    o = fh.run_oracle(port, c0, 1)
    np.testing.assert_allclose(T7, o["T_lay"], rtol=1e-9)
    assert o["abort"].sum() == c0.nlayer + 1


def test_fused_full_size_vs_stage_kernels(ctx):
    """BASELINE config 2 (10 000 bins x 100 layers x 20 Gauss points): the fused path against the
    per-stage HIP kernels on the same device-resident inputs, plus invariants"""
    from helios_amd.rt import batch_from_case
    from impls import hip_impl
    c0 = cases.make_case(nbin=10000, nlayer=100, ntemp=6, npress=5, plancktable_dim=800,
                         plancktable_step=5)
    n_iter = 3
    rt = batch_from_case(ctx, c0)
    rt.build_planck_table(1)
    rt.run(0, n_iter)
    f = {k: rt.get(k) for k in ("T_lay", "F_up_band", "F_down_band", "F_net", "F_up_tot", "F_down_tot",
                                "planckband_lay", "opac_wg_int", "F_up_wg")}
    rt.close()
    hip = hip_impl(ctx)
    c = c0.copy()
    s = cases.alloc_state(c)
    cases.setup_planck(hip, c, s)
    cases.radiation_iterations(hip, c, s, n_iter)
    scale = np.abs(s.F_down_wg).max()
    np.testing.assert_allclose(f["opac_wg_int"], s.opac_wg_int, rtol=1e-12)
    np.testing.assert_allclose(f["planckband_lay"], s.planckband_lay, rtol=1e-9, atol=1e-13 * s.planckband_lay.max())
    np.testing.assert_allclose(f["F_up_wg"], s.F_up_wg, rtol=1e-8, atol=1e-13 * scale)
    for k in ("F_up_band", "F_down_band", "F_up_tot", "F_down_tot"):
        np.testing.assert_allclose(f[k], s[k], rtol=1e-9, atol=1e-13 * scale, err_msg=k)
    np.testing.assert_allclose(f["T_lay"], c.T_lay, rtol=1e-8)
    # invariants: non-negative fluxes; TOA down-flux = f (R*/a)^2 sigma T*^4 after the energy correction
    assert f["F_up_band"].min() >= 0 and f["F_down_band"].min() >= 0
    from helios_amd import phys_const as pc
    toa = c0.f_factor * (c0.R_star / c0.a) ** 2 * 5.6703669999999995e-5 * c0.T_star ** 4
    np.testing.assert_allclose(f["F_down_tot"][-1], toa, rtol=1e-10)


def test_fused_refuses_more_than_1024_layers_and_the_driver_takes_the_stage_kernels(ctx, capsys):
    """the sweeps' tiles hold at most 32 half-layers per lane x 64 lanes (16 of them in registers; round 6: 20-32 with part of
    the image in scratch): beyond 1024 layers hx_rt_create reports HX_E_UNSUPPORTED and Compute runs the per-stage kernels
    instead (still on the GPU -- there is no CPU path) -- and says so, once per run, with the reason (the 5-10x slower path
    is not taken silently)"""
    from helios_amd._lib import HeliosHipError
    from helios_amd.computation import Compute
    c = cases.make_case(nbin=2, nlayer=1025)
    with pytest.raises(HeliosHipError, match="1024"):
        fh.run_fused(ctx, c, 1)

    class Q(object):
        iso, singlewalk, flux_calc_method, nlayer = 0, 0, "iteration", 1025
    comp = Compute(ctx)
    assert not comp._fused_supported(Q())
    assert not comp._fused_supported(Q())
    out = capsys.readouterr().out
    assert out.count("helios_amd: 1025 layers") == 1 and "per-stage kernels" in out and "slower" in out
    Q.nlayer = 1024
    assert comp._fused_supported(Q())
    assert capsys.readouterr().out == ""
    import ctypes
    from helios_amd import _lib
    k, r = ctypes.c_int(), ctypes.c_int()
    for L, iso, want in ((513, 0, (64, 20)), (640, 0, (64, 20)), (641, 0, (64, 24)), (1024, 0, (64, 32)), (2048, 1, (64, 32)), (1100, 1, (64, 20))):
        assert _lib.lib().hx_rt_flux_geometry(L, iso, 0, 20, 100, 1, ctypes.byref(k), ctypes.byref(r)) == 0
        assert (k.value, r.value) == want, (L, iso, k.value, r.value)
    assert _lib.lib().hx_rt_flux_geometry(1025, 0, 0, 20, 100, 1, ctypes.byref(k), ctypes.byref(r)) != 0


@pytest.mark.parametrize("name,ncol", [("L50", 1), ("clouds_g0", 3), ("L100", 2), ("L50+matrix", 1), ("clouds_g0+matrix", 2),
                                       ("dirbeam+species", 2)])
def test_graph_replay_of_refresh_free_iterations_equals_launch_by_launch(ctx, name, ncol, monkeypatch):
    """hx_rt_run replays the nine iterations between two opacity refreshes as ONE hipGraph where launches bound the loop
    (small grids; HELIOS_RT_GRAPH forces it either way) -- and, entered at a refresh boundary, the whole decade, refresh
    included (round 5): the iteration index comes from the device, so the captured kernels carry the same arguments every
    time.  Same kernels, same order, same arguments: the same bits -- after 47
    iterations in one call (refreshes, four graph replays, a tail of single steps), after a second call that continues
    from there, and for a batch whose columns differ."""
    matrix = name.endswith("+matrix")           # the direct solve of `flux calculation method = matrix` replays like the sweeps
    species = name.endswith("+species")         # ... and so does a refresh that mixes absorbers on the fly (the decade graph)
    c0 = cases.make_case(**dict(FUSED_CONFIGS[name.split("+")[0]], **({"albedo": 0.1} if matrix else {})))
    if matrix:
        c0.flux_calc_method = "matrix"
    if species:
        c0 = cases.add_species(c0, nspecies=4)
    T = [c0.T_lay * (1.0 + 0.02 * k) for k in range(ncol)] if ncol > 1 else None
    from helios_amd.rt import batch_from_case

    def run(graph):
        monkeypatch.setenv("HELIOS_RT_GRAPH", graph)
        rt = batch_from_case(ctx, c0, ncol=ncol, nspecies=len(c0.species) if species else 0)
        try:
            if species:
                for k, sp in enumerate(c0.species):
                    rt.set_species(k, sp["pretab"], sp["scat"], sp["weight"], is_h2o=2 if sp["is_h2o"] else 0,
                                   is_cia=1 if sp["is_cia"] else 0, in_mu=0 if sp["is_cia"] else 1)
                rt.set_column_vmr(-1, *cases.species_vmr_arrays(c0))
            if T is not None:
                for k, Tk in enumerate(T):
                    rt.set_temperatures(k, Tk)
            rt.build_planck_table(1 if c0.T_star > 10 else 0)
            rt.run(0, 47)
            first = [{k: rt.get(k, col) for k in ("T_lay", "F_net", "F_up_band", "delta_t_prefactor", "abort")} for col in range(ncol)]
            rt.run(47, 24)              # starts mid-decade: single steps up to the refresh, then a replay
            rt.set_convergence_limit(-1, 1e-3)   # an argument changes: the graph is captured again
            rt.run(71, 30)
            second = [{k: rt.get(k, col) for k in ("T_lay", "F_net", "F_up_band", "delta_t_prefactor", "abort", "iters_done")}
                      for col in range(ncol)]
            nine, decades, on = rt.get("graph_replays")
            # the replays really happened: 101 iterations = the first refresh launched, then whole decades and runs of nine
            assert (on == 1 and decades >= 5 and nine >= 2) if graph == "1" else (nine == 0 and decades == 0)
            # ... from few captures: each graph once, and once more after the setter -- a capture vouches for itself only
            # (a run entered at decade boundaries alone must not re-capture the decade graph on every call)
            builds_nine, builds_decade = rt.get("graph_builds")
            assert (1 <= builds_decade <= 2 and 1 <= builds_nine <= 2) if graph == "1" else (builds_nine == builds_decade == 0)
            if graph == "1":
                for it in range(110, 160, 10):          # five more calls, each one whole decade, no setter in between
                    rt.run(it, 10)
                assert tuple(rt.get("graph_builds")) == (builds_nine, builds_decade)
                assert rt.get("graph_replays")[1] == decades + 5
            return first, second
        finally:
            rt.close()
    a1, a2 = run("1")
    b1, b2 = run("0")
    for got, want in ((a1, b1), (a2, b2)):
        for col in range(ncol):
            for k in want[col]:
                np.testing.assert_array_equal(got[col][k], want[col][k], err_msg="%s column %d" % (k, col))
    if ncol > 1:
        assert np.abs(a1[0]["T_lay"] - a1[ncol - 1]["T_lay"]).max() > 1.0


@pytest.mark.parametrize("name,ncol,mb", [("L50", 1, "0.05"), ("clouds_g0", 3, "0.2"), ("L100", 2, "1000")])
def test_back_and_forth_launch_order_with_cached_state_is_the_same_run(ctx, name, ncol, mb, monkeypatch):
    """k_rt_flux walking its grid from the far end on every other launch, the workgroups dispatched last keeping their
    up-flux state in the Infinity Cache through write-through stores (what a batch of BASELINE config 2's size does by
    itself; HELIOS_RT_SERPENTINE / HELIOS_RT_STATE_CACHE_MB force it here, with a cached share of part of the grid, of
    a few workgroups and of everything): workgroups are independent, so the run is the same bit for bit -- launch by
    launch and through the graph replay, whose captured launches carry alternating directions."""
    c0 = cases.make_case(**FUSED_CONFIGS[name])
    T = [c0.T_lay * (1.0 + 0.02 * k) for k in range(ncol)] if ncol > 1 else None
    from helios_amd.rt import batch_from_case

    def run(serpentine, graph):
        monkeypatch.setenv("HELIOS_RT_SERPENTINE", serpentine)
        monkeypatch.setenv("HELIOS_RT_STATE_CACHE_MB", mb if serpentine == "1" else "0")
        monkeypatch.setenv("HELIOS_RT_GRAPH", graph)
        rt = batch_from_case(ctx, c0, ncol=ncol)
        try:
            if T is not None:
                for k, Tk in enumerate(T):
                    rt.set_temperatures(k, Tk)
            rt.build_planck_table(1 if c0.T_star > 10 else 0)
            rt.run(0, 33)
            return [{k: rt.get(k, col) for k in ("T_lay", "F_net", "F_up_band", "F_down_band", "delta_t_prefactor", "abort")}
                    for col in range(ncol)]
        finally:
            rt.close()
    want = run("0", "0")
    for graph in ("0", "1"):
        got = run("1", graph)
        for col in range(ncol):
            for k in want[col]:
                np.testing.assert_array_equal(got[col][k], want[col][k], err_msg="%s column %d graph %s" % (k, col, graph))

