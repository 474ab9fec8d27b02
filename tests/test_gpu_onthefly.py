"""GPU parity of on-the-fly opacity mixing (species loop with random overlap) through the fused path,
and of the Python driver surface (Store / Compute / run_helios) in both its fused and per-stage forms."""
import os

import numpy as np
import pytest

import cases
import fused_helpers as fh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from helios_amd.device import Context
    return Context(0)


@pytest.mark.parametrize("cfg", [dict(nbin=9, nlayer=8), dict(nbin=14, nlayer=21, dir_beam=1, albedo=0.1),
                                 dict(nbin=11, nlayer=8, ny=1)])
def test_fused_onthefly_vs_oracle(ctx, port, cfg):
    """the species loop of the fused refresh against the oracle's; ny = 1 (opacity sampling) with `kcoeff_mixing = RO`:
    the reference mixes correlated-k then (condition_for_correlated_k includes ny == 1, kernels.cu:3302), and so does
    the fused path instead of refusing the table"""
    c0 = cases.add_species(cases.make_case(**cfg), nspecies=4)
    for n_iter, rtol in ((1, 1e-9), (11, 1e-7)):
        f, grid = fh.run_fused(ctx, c0, n_iter, with_planck_grid=True)
        o = fh.run_oracle(port, c0, n_iter, planck_grid=grid, refresh=cases.refresh_onthefly)
        fh.compare(f, o, c0, rtol=rtol)
    if c0.ny == 1:
        return
    # the mix really went through random overlap: differs from the correlated-k mix
    o_ck = fh.run_oracle(port, c0, 1, planck_grid=grid,
                         refresh=lambda i, c, s: cases.refresh_onthefly(i, c, s, ro=0))
    assert np.abs(o_ck["opac_wg_lay"] - f["opac_wg_lay"]).max() > 1e-3 * np.abs(f["opac_wg_lay"]).max()


@pytest.mark.parametrize("nspecies", [48, 49, 101])
def test_fused_onthefly_with_more_absorbers_than_the_species_list_on_chip_holds(ctx, port, nspecies):
    """the species loop keeps 48 absorbers' table pointers and factors in LDS; a longer list is folded in block by block, every
    launch but the first starting from the mix the one before it wrote (round 6: the limit of rounds 2-5 -- and the fall-back
    to the per-stage kernels behind it -- is gone).  48 (one full block), 49 (a block of one), 101 (three launches) absorbers
    mixed by random overlap, against the oracle's species loop: the same order of absorbers, so the same sums"""
    c0 = cases.add_species(cases.make_case(nbin=6, nlayer=5), nspecies=nspecies)
    rng = np.random.default_rng(5)
    for sp in c0.species[1:nspecies]:          # comparable abundances: most problems take the network, not the 1 % short cut
        sp["vmr"] = float(10.0 ** rng.uniform(-3.0, -2.0))
    f, grid = fh.run_fused(ctx, c0, 1, with_planck_grid=True)
    o = fh.run_oracle(port, c0, 1, planck_grid=grid, refresh=cases.refresh_onthefly)
    fh.compare(f, o, c0, rtol=1e-9)
    o_ck = fh.run_oracle(port, c0, 1, planck_grid=grid, refresh=lambda i, c, s: cases.refresh_onthefly(i, c, s, ro=0))
    assert np.abs(o_ck["opac_wg_lay"] - f["opac_wg_lay"]).max() > 1e-3 * np.abs(f["opac_wg_lay"]).max()
    from helios_amd.computation import Compute

    class Sp(object):
        absorbing = "yes"

    class Q(object):
        iso, singlewalk, flux_calc_method, nlayer, species_list = 0, 0, "iteration", 50, [Sp()] * nspecies
    assert Compute(ctx)._fused_supported(Q())


@pytest.mark.parametrize("cfg", [dict(nbin=9, nlayer=8, albedo=0.1), dict(nbin=14, nlayer=21, dir_beam=1, albedo=0.1, clouds=1,
                                                                           g_0=0.2, scat_corr=1)])
def test_fused_onthefly_with_the_matrix_method_vs_oracle(ctx, port, cfg):
    """absorbers mixed on the fly at every refresh AND `flux calculation method = matrix`: the species loop of the fused
    refresh feeds calc_trans_* and the tridiagonal solve inside the device-resident loop"""
    c0 = cases.add_species(cases.make_case(**cfg), nspecies=4)
    c0.flux_calc_method = "matrix"
    # eleven iterations: the back-substitution x_i = d'_i - c'_i x_(i+1) cancels where the down-flux is small (thin bins
    # near the top: entries 1e-7 of the largest), and what the species loop's last bits start there is fed back through the
    # temperatures -- observed 2.6e-6 on four such entries of 6160 (3e-10 of their neighbours), 1e-7 elsewhere
    # (after one iteration the spectral fluxes are held to the extended-precision solution of the reference's system, the
    # reference's own distance from it asserted next to it: tests/matrix_referee.py, tests/test_gpu_fused.py)
    import matrix_referee
    for n_iter, rtol in ((1, 1e-9), (11, 1e-5)):
        f, grid = fh.run_fused(ctx, c0, n_iter, with_planck_grid=True)
        o = fh.run_oracle(port, c0, n_iter, planck_grid=grid, refresh=cases.refresh_onthefly)
        if n_iter == 1:
            matrix_referee.compare_first_solve(fh, f, o, c0, rtol)
        else:
            fh.compare(f, o, c0, rtol=rtol)


def _run_driver(argv, use_fused):
    import helios
    from helios_amd import computation
    orig = computation.Compute.__init__

    def patched(self, ctx=None):
        orig(self, ctx)
        self.use_fused = use_fused
    computation.Compute.__init__ = patched
    try:
        return helios.run_helios(argv)
    finally:
        computation.Compute.__init__ = orig


def test_run_helios_fused_equals_stagewise(tmp_path):
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "40 6 5 7",
            "-number_of_layers", "24", "-maximum_number_of_iterations", "20000", "-name", "drv",
            "-output_directory", str(tmp_path) + "/", "-radiative_equilibrium_criterion", "1e-3",
            "-convective_adjustment", "no"]
    a = _run_driver(argv, True)
    b = _run_driver(argv, False)
    assert int(a.iter_value) == int(b.iter_value) and int(a.iter_value) > 3
    np.testing.assert_allclose(a.T_lay, b.T_lay, rtol=1e-8)
    np.testing.assert_allclose(a.F_up_band, b.F_up_band, rtol=1e-7, atol=1e-12 * b.F_up_band.max())
    np.testing.assert_allclose(a.F_net, b.F_net, rtol=1e-7, atol=1e-10 * np.abs(b.F_up_tot).max())
    np.testing.assert_allclose(a.contr_func_band, b.contr_func_band, rtol=1e-6, atol=1e-12 * b.contr_func_band.max())
    np.testing.assert_allclose(a.planck_opac_T_pl, b.planck_opac_T_pl, rtol=1e-7)
    # converged: every layer satisfies the criterion and the global energy balance holds to the same level
    from helios_amd import host_functions as hs
    assert abs(hs.global_energy_imbalance(a)) < 1e-3
    import os
    for suffix in ("_tp.dat", "_integrated_flux.dat", "_spec_upflux.dat", "_spec_downflux.dat", "_TOA_flux_eclipse.dat"):
        assert os.path.getsize(os.path.join(str(tmp_path), "drv", "drv" + suffix)) > 100


@pytest.mark.parametrize("fused", [True, False])
def test_run_helios_with_debugging_feedback(tmp_path, capfd, fused):
    """`debugging_feedback = yes`: the run is the same as without it (the counters of hx_diag are read and reported
    between iterations instead of device-side printf), and the energy-budget message appears as in the reference"""
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 7",
            "-number_of_layers", "16", "-maximum_number_of_iterations", "20000", "-name", "dbg",
            "-output_directory", str(tmp_path) + "/", "-radiative_equilibrium_criterion", "1e-3",
            "-convective_adjustment", "no", "-direct_irradiation_beam", "yes"]
    a = _run_driver(argv, fused)
    capfd.readouterr()
    b = _run_driver(argv + ["-debugging_feedback", "yes"], fused)
    out = capfd.readouterr().out
    assert "Energy budget corrected" in out
    assert int(a.iter_value) == int(b.iter_value)
    np.testing.assert_array_equal(a.T_lay, b.T_lay)
    np.testing.assert_array_equal(a.F_up_band, b.F_up_band)
    # a healthy run has nothing to report
    assert "negative flux" not in out and "malfunctioning" not in out


def test_run_helios_with_more_layers_than_the_fused_path_holds(tmp_path, capfd):
    """1030 layers: hx_rt_* would refuse (limit 1024 since round 6, 512 before), the driver says so, runs the per-stage kernels
    and converges; 520 layers -- beyond the old limit -- stay in the device-resident loop (k = 64, 20 rows per lane)"""
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "20 6 5 7",
            "-number_of_layers", "1030", "-maximum_number_of_iterations", "2000", "-name", "big",
            "-output_directory", str(tmp_path) + "/", "-radiative_equilibrium_criterion", "1e-2",
            "-convective_adjustment", "no"]
    q = _run_driver(argv, True)
    assert q.rt is None and 3 < int(q.iter_value) < 2000
    assert "helios_amd: 1030 layers" in capfd.readouterr().out
    from helios_amd import host_functions as hs
    assert abs(hs.global_energy_imbalance(q)) < 2e-2
    argv[argv.index("1030")] = "520"
    q2 = _run_driver(argv, True)
    assert q2.rt is not None and 3 < int(q2.iter_value) < 2000
    assert "helios_amd:" not in capfd.readouterr().out
    assert abs(hs.global_energy_imbalance(q2)) < 2e-2
    # (to convergence the per-stage loop takes another path to the same criterion -- a 520-layer column's pseudo-time step
    # amplifies rounding differences, 8e-8 after 45 iterations against the oracle, the same as a 500-layer column on 16 rows;
    # the kernels of 20-32 rows are held to the oracle in tests/test_gpu_fused.py: L600, L1024_clouds, iso_L1500, L700_beam)


def test_run_helios_with_convection(tmp_path):
    """hot interior -> super-adiabatic deep layers -> the convection loop engages and ends stable"""
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 11",
            "-number_of_layers", "20", "-maximum_number_of_iterations", "20000", "-name", "conv",
            "-output_directory", str(tmp_path) + "/", "-radiative_equilibrium_criterion", "1e-4",
            "-internal_temperature", "800", "-kappa_value", "0.285714"]
    q = _run_driver(argv, True)
    from helios_amd import host_functions as hs
    q.kappa_int = np.full(int(q.ninterface), 0.285714)
    hs.conv_check(q)
    assert q.conv_unstable.sum() == 0
    assert np.all(np.isfinite(q.T_lay)) and q.F_up_band.min() >= 0


def test_run_helios_matrix_method(tmp_path):
    """`flux calculation method = matrix` through run_helios (the device-resident loop with hx_rt_flags.matrix): same
    equilibrium as the iterative sweeps, whose persistent up-flux state converges to the solution of the same linear
    system"""
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 7",
            "-number_of_layers", "16", "-maximum_number_of_iterations", "20000", "-name", "mat",
            "-output_directory", str(tmp_path) + "/", "-radiative_equilibrium_criterion", "1e-4",
            "-convective_adjustment", "no", "-surface_albedo", "0.1"]
    a = _run_driver(argv + ["-flux_calculation_method", "matrix"], True)
    b = _run_driver(argv, True)
    assert np.all(np.isfinite(a.T_lay)) and int(a.iter_value) > 3 and a.rt is not None
    # both stop at the same flux criterion from different sides; the thin top layer and the surface are the
    # loosest-constrained temperatures (observed: 0.5 % there, 1e-5 in the bulk)
    np.testing.assert_allclose(a.T_lay, b.T_lay, rtol=1e-2)
    np.testing.assert_allclose(a.T_lay[3:-1], b.T_lay[3:-1], rtol=1e-4)
    np.testing.assert_allclose(a.F_up_band[-int(a.nbin):].sum(), b.F_up_band[-int(b.nbin):].sum(), rtol=5e-3)


def test_run_helios_with_mie_cloud_deck(tmp_path):
    """parameter reader -> Mie tables -> size distribution -> deck -> fused path with clouds: the deck changes the
    emission spectrum and the run still balances its energy"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_host_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_host_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    mie = os.path.join(str(tmp_path), "mie") + "/"
    mk.write_mie_files(mie, 5)
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "40 6 5 7",
            "-number_of_layers", "20", "-maximum_number_of_iterations", "20000", "-name", "cl",
            "-output_directory", str(tmp_path) + "/", "-radiative_equilibrium_criterion", "1e-4",
            "-convective_adjustment", "no"]
    cloudy = _run_driver(argv + ["-number_of_cloud_decks", "1", "-path_to_mie_files", mie, "-aerosol_radius_mode", "1.0",
                                 "-aerosol_radius_geometric_std_dev", "1.8", "-cloud_bottom_pressure", "1e6",
                                 "-cloud_bottom_mixing_ratio", "1e-13", "-cloud_to_gas_scale_height_ratio", "0.5"], True)
    clear = _run_driver(argv, True)
    assert int(cloudy.clouds) == 1 and cloudy.abs_cross_all_clouds_lay.max() > 0
    assert np.all(np.isfinite(cloudy.T_lay)) and cloudy.F_up_band.min() >= 0
    X = int(clear.nbin)
    toa_c, toa_0 = cloudy.F_up_band[-X:], clear.F_up_band[-X:]
    assert np.abs(toa_c - toa_0).max() > 1e-3 * toa_0.max()
    from helios_amd import host_functions as hs
    assert abs(hs.global_energy_imbalance(cloudy)) < 1e-3
    assert os.path.getsize(os.path.join(str(tmp_path), "cl", "cl_cloud_opacities.dat")) > 100


def test_run_helios_with_kappa_table(tmp_path):
    """`kappa value = file`: the per-stage loop interpolates kappa and c_p from the table every refresh; a table
    holding the constant 2/7 everywhere must reproduce the constant-kappa run"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_host_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_host_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    path = os.path.join(str(tmp_path), "delad.dat")
    kap = 0.1                                            # small adiabatic gradient: the deep layers must convect
    mk.write_kappa_file(path, False, const_kappa=kap)
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 11",
            "-number_of_layers", "20", "-maximum_number_of_iterations", "20000", "-name", "kap",
            "-output_directory", str(tmp_path) + "/", "-radiative_equilibrium_criterion", "1e-4",
            "-internal_temperature", "1500"]
    a = _run_driver(argv + ["-kappa_value", "file", "-kappa_file_path", path], True)     # fused: table on the device
    b = _run_driver(argv + ["-kappa_value", repr(kap)], True)                            # fused: constant kappa
    c = _run_driver(argv + ["-kappa_value", "file", "-kappa_file_path", path], False)    # per-stage loop, table
    assert c.conv_layer.sum() > 0
    np.testing.assert_allclose(c.kappa_lay, kap, rtol=1e-12)
    np.testing.assert_array_equal(c.conv_layer, a.conv_layer)
    np.testing.assert_allclose(c.T_lay, a.T_lay, rtol=2e-4)
    assert a.conv_layer.sum() > 0                      # the convection loop engaged
    np.testing.assert_allclose(a.kappa_lay, kap, rtol=1e-12)
    np.testing.assert_array_equal(a.conv_layer, b.conv_layer)
    np.testing.assert_allclose(a.T_lay, b.T_lay, rtol=1e-9)      # same loop, same numbers: the table is constant
    assert int(a.iter_value) == int(b.iter_value)
    assert a.entropy_lay.min() > 0                     # entropy diagnostic interpolated from the table


def _host_golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_host_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_host_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk


def test_run_helios_on_the_fly_from_files(tmp_path):
    """`opacity mixing = on-the-fly` end to end: species file -> FastChem / profile / constant mixing ratios ->
    per-species containers -> species loop with random overlap on the device; fused and per-stage drivers agree"""
    wd = str(tmp_path)
    _host_golden_module().write_species_inputs(wd, nbin=14, ny=20, sorted_k=True)
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "on-the-fly",
            "-path_to_species_file", os.path.join(wd, "species.dat"),
            "-file_with_vertical_mixing_ratios", os.path.join(wd, "vmr.txt"),
            "-directory_with_fastchem_files", os.path.join(wd, "chem") + "/",
            "-directory_with_opacity_files", os.path.join(wd, "opac") + "/",
            "-number_of_layers", "18", "-maximum_number_of_iterations", "20000", "-name", "otf",
            "-output_directory", wd + "/", "-radiative_equilibrium_criterion", "1e-4",
            "-convective_adjustment", "no", "-toa_pressure", "1e0", "-boa_pressure", "1e8"]
    a = _run_driver(argv, True)
    b = _run_driver(argv, False)
    names = [sp.name for sp in a.species_list]
    assert names[0] == "H2O" and "H-_ff" in names and len(names) == 10
    assert int(a.iter_value) == int(b.iter_value) and int(a.iter_value) > 3
    np.testing.assert_allclose(a.T_lay, b.T_lay, rtol=1e-7)
    np.testing.assert_allclose(a.F_up_band, b.F_up_band, rtol=1e-6, atol=1e-12 * b.F_up_band.max())
    np.testing.assert_allclose(a.meanmolmass_lay, b.meanmolmass_lay, rtol=1e-12)
    from helios_amd import host_functions as hs
    assert abs(hs.global_energy_imbalance(a)) < 1e-3


def test_run_helios_on_the_fly_from_hdf5_files(tmp_path):
    """the same run with every opacity container as a real HDF5 file (gzip-compressed, written through the HDF5 C
    library) and the stellar spectrum from a nested data set of an HDF5 file: the readers go through libhdf5 (no h5py
    in the image) and the run is the .npz run bit for bit"""
    from helios_amd import hdf5_lite
    if not hdf5_lite.available():
        pytest.skip("no libhdf5 on this host")
    wd = str(tmp_path)
    _host_golden_module().write_species_inputs(wd, nbin=14, ny=20, sorted_k=True)
    star = 10.0 ** np.random.default_rng(4).uniform(5.0, 7.0, 14)
    np.savez(os.path.join(wd, "star.npz"), **{"r50_kdistr/phoenix/teststar": star})
    hdf5_lite.write(os.path.join(wd, "star.h5"), {"r50_kdistr/phoenix/teststar": star, "r50_kdistr/lambda": np.arange(14.0)})

    def argv(star_file, out):
        return ["-parameter_file", "/nonexistent", "-opacity_mixing", "on-the-fly",
                "-path_to_species_file", os.path.join(wd, "species.dat"),
                "-file_with_vertical_mixing_ratios", os.path.join(wd, "vmr.txt"),
                "-directory_with_fastchem_files", os.path.join(wd, "chem") + "/",
                "-directory_with_opacity_files", os.path.join(wd, "opac") + "/",
                "-stellar_spectral_model", "file", "-path_to_stellar_spectrum_file", os.path.join(wd, star_file),
                "-dataset_in_stellar_spectrum_file", "/r50_kdistr/phoenix/teststar",
                "-number_of_layers", "18", "-maximum_number_of_iterations", "20000", "-name", out,
                "-output_directory", wd + "/", "-radiative_equilibrium_criterion", "1e-4",
                "-convective_adjustment", "no", "-toa_pressure", "1e0", "-boa_pressure", "1e8"]

    a = _run_driver(argv("star.npz", "npz"), True)
    opac = os.path.join(wd, "opac")
    for f in sorted(os.listdir(opac)):
        if f.endswith(".npz"):
            hdf5_lite.write(os.path.join(opac, f[:-4] + ".h5"), dict(np.load(os.path.join(opac, f))), compression=4)
            os.remove(os.path.join(opac, f))
    assert all(f.endswith(".h5") for f in os.listdir(opac)) and open(os.path.join(opac, "scat_cross_sections.h5"), "rb").read(4) == b"\x89HDF"
    b = _run_driver(argv("star.h5", "h5"), True)
    assert int(a.real_star) == 1 and int(b.real_star) == 1
    assert int(a.iter_value) == int(b.iter_value) and int(a.iter_value) > 3
    np.testing.assert_array_equal(a.starflux, b.starflux)
    np.testing.assert_array_equal(a.T_lay, b.T_lay)
    np.testing.assert_array_equal(a.F_up_band, b.F_up_band)


def test_run_helios_premixed_from_a_real_h5py_file(tmp_path):
    """`opacity mixing = premixed` with the k-table container of tests/golden/reader/hdf5/ -- written by h5py 3.3.0 as the
    reference's k-table tool writes it (variable-length string datasets and all) -- through helios.py: the same run as
    from an .npz archive of the same datasets, bit for bit"""
    from helios_amd import hdf5_lite
    if not hdf5_lite.available():
        pytest.skip("no libhdf5 on this host")
    h5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reader", "hdf5", "premixed_opac_kdistr.h5")
    with hdf5_lite.File(h5) as f:
        names = [k for k in f.keys() if k not in ("included molecules", "FastChem path", "units")]
        np.savez(str(tmp_path / "premixed.npz"), **{k: f[k][()] for k in names})

    def run(table, name):
        argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "premixed", "-path_to_opacity_file", table,
                "-number_of_layers", "20", "-maximum_number_of_iterations", "20000", "-name", name,
                "-output_directory", str(tmp_path) + "/", "-convective_adjustment", "no", "-toa_pressure", "1e1",
                "-boa_pressure", "1e7", "-radiative_equilibrium_criterion", "1e-3"]
        return _run_driver(argv, True)
    a = run(h5, "h5")
    b = run(str(tmp_path / "premixed.npz"), "npz")
    assert int(a.nbin) == 6 and int(a.ny) == 20 and int(a.ntemp) == 3 and int(a.npress) == 4
    assert int(a.iter_value) == int(b.iter_value) and int(a.iter_value) >= 10
    np.testing.assert_array_equal(a.opac_k, b.opac_k)
    np.testing.assert_array_equal(a.T_lay, b.T_lay)
    np.testing.assert_array_equal(a.F_up_band, b.F_up_band)
    assert np.all(np.isfinite(a.T_lay)) and np.all(np.isfinite(a.F_up_band))


def test_run_helios_with_additional_heating(tmp_path):
    """a heating-density file (e.g. UV heating of the upper atmosphere): per-stage loop, flux refreshed from the layer
    heights every 10th iteration; in equilibrium the atmosphere radiates the extra energy away"""
    wd = str(tmp_path)
    with open(os.path.join(wd, "heating.txt"), "w") as f:      # default format: 1 header line, cgs pressure, x 1e7
        f.write("heating density in J m^-3 s^-1... times 1e7\nPressure Heating\n")
        for p, h in ((1e-1, 3e-12), (1e1, 1e-11), (1e3, 3e-12), (1e5, 1e-13), (1e9, 1e-16)):
            f.write("%g %g\n" % (p, h))
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 7",
            "-number_of_layers", "16", "-maximum_number_of_iterations", "20000", "-name", "heat",
            "-output_directory", wd + "/", "-radiative_equilibrium_criterion", "1e-5", "-convective_adjustment", "no"]
    heat = ["-include_additional_heating", "yes", "-path_to_heating_file", os.path.join(wd, "heating.txt")]
    hot = _run_driver(argv + heat, True)                 # heating flux refreshed on the device (fused path)
    hot_stagewise = _run_driver(argv + heat, False)      # ... and by the host in the per-stage loop
    cold = _run_driver(argv, True)
    assert int(hot.iter_value) == int(hot_stagewise.iter_value)
    np.testing.assert_allclose(hot.T_lay, hot_stagewise.T_lay, rtol=1e-7)
    np.testing.assert_allclose(hot.F_add_heat_sum, hot_stagewise.F_add_heat_sum, rtol=1e-9)
    extra = hot.F_add_heat_sum[-1]
    assert extra > 0 and np.all(hot.add_heat_dens > 0)
    L = int(hot.nlayer)
    # net flux leaving the top = internal flux + everything deposited below
    scale = hot.F_down_tot[L] + hot.F_intern
    assert abs(hot.F_net[L] - (hot.F_intern + extra)) < 1e-4 * scale
    assert abs(cold.F_net[L] - cold.F_intern) < 1e-4 * scale
    assert hot.T_lay[:L].mean() > cold.T_lay[:L].mean()


def test_run_helios_post_processing_of_a_converged_profile(tmp_path):
    """`run type = post-processing`: the T-P profile of a finished run is read back from its `_tp.dat`, layers are
    isothermal, one pass with 1000*scat+1 sweeps.  The converged multiple-scattering solution leaves the emission spectrum
    of the iterative run (non-isothermal layers, persistent flux state) within a few per cent."""
    wd = str(tmp_path)
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 7",
            "-number_of_layers", "40", "-maximum_number_of_iterations", "20000", "-output_directory", wd + "/",
            "-radiative_equilibrium_criterion", "1e-5", "-convective_adjustment", "no"]
    it = _run_driver(base + ["-name", "run"], True)
    pp = _run_driver(base + ["-name", "run_post", "-run_type", "post-processing", "-path_to_temperature_file",
                             os.path.join(wd, "run", "run_tp.dat")], True)
    assert int(pp.singlewalk) == 1 and int(pp.iso) == 1
    np.testing.assert_allclose(pp.T_lay, it.T_lay, rtol=1e-5)          # the text file carries 6 significant digits of T and P
    X, L = int(it.nbin), int(it.nlayer)
    a, b = pp.F_up_band[L * X:], it.F_up_band[L * X:]
    assert np.all(np.isfinite(a)) and a.min() >= 0
    assert abs(a @ it.opac_deltawave - b @ it.opac_deltawave) < 0.05 * (b @ it.opac_deltawave)
    assert os.path.getsize(os.path.join(wd, "run_post", "run_post_contribution.dat")) > 100
    # the same pass through the per-stage entry points (1001 launches of hx_fband_iso): the fused kernel runs the 1001
    # sweeps in registers inside one launch
    ps = _run_driver(base + ["-name", "run_post2", "-run_type", "post-processing", "-path_to_temperature_file",
                             os.path.join(wd, "run", "run_tp.dat")], False)
    np.testing.assert_allclose(pp.F_up_band, ps.F_up_band, rtol=1e-9, atol=1e-13 * ps.F_up_band.max())
    np.testing.assert_allclose(pp.F_down_band, ps.F_down_band, rtol=1e-9, atol=1e-13 * ps.F_down_band.max())
    np.testing.assert_allclose(pp.contr_func_band, ps.contr_func_band, rtol=1e-9, atol=1e-13 * ps.contr_func_band.max())
    np.testing.assert_allclose(pp.F_net, ps.F_net, rtol=1e-8, atol=1e-11 * np.abs(ps.F_up_tot).max())


def test_run_helios_post_processing_with_the_matrix_method(tmp_path):
    """the use the reference's documentation recommends the matrix method for (`docs/sections/parameters.rst:326`: "for
    post-processing only"): one pass over a given profile, isothermal layers, one tridiagonal solve per spectral point.
    The device-resident pass equals the per-stage one, and the direct solve is the limit of the 1001 sweeps"""
    wd = str(tmp_path)
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 7",
            "-number_of_layers", "40", "-maximum_number_of_iterations", "20000", "-output_directory", wd + "/",
            "-radiative_equilibrium_criterion", "1e-5", "-convective_adjustment", "no", "-surface_albedo", "0.1"]
    _run_driver(base + ["-name", "run"], True)
    pp = base + ["-run_type", "post-processing", "-path_to_temperature_file", os.path.join(wd, "run", "run_tp.dat")]
    mx = _run_driver(pp + ["-name", "mx", "-flux_calculation_method", "matrix"], True)
    assert int(mx.singlewalk) == 1 and int(mx.iso) == 1 and mx.rt is not None and mx.flux_calc_method == "matrix"
    ps = _run_driver(pp + ["-name", "mx2", "-flux_calculation_method", "matrix"], False)
    assert ps.rt is None
    for k in ("F_up_band", "F_down_band", "contr_func_band"):
        np.testing.assert_allclose(getattr(mx, k), getattr(ps, k), rtol=1e-9, atol=1e-13 * getattr(ps, k).max(), err_msg=k)
    np.testing.assert_allclose(mx.F_net, ps.F_net, rtol=1e-8, atol=1e-11 * np.abs(ps.F_up_tot).max())
    sw = _run_driver(pp + ["-name", "sw"], True)                    # the 1000*scat+1 sweeps of the iterative method
    X, L = int(mx.nbin), int(mx.nlayer)
    # (observed: the 1001 sweeps have come within 2e-6 of the direct solution)
    np.testing.assert_allclose(mx.F_up_band[L * X:], sw.F_up_band[L * X:], rtol=1e-4, atol=1e-10 * sw.F_up_band.max())


CONV_ARGV = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 11",
             "-number_of_layers", "25", "-maximum_number_of_iterations", "20000", "-name", "cv",
             "-radiative_equilibrium_criterion", "1e-4", "-internal_temperature", "1500", "-kappa_value", "0.2"]


def test_isothermal_layers_iterate_on_the_fused_path_and_refuse_convection(tmp_path):
    """`isothermal layers = yes` in an iterative run: the radiation loop on the fused path ends where the per-stage loop
    ends; with convective adjustment the reference fails with a TypeError (computation.py:1004-1009) -- here a clear
    message"""
    argv = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "30 6 5 7",
            "-number_of_layers", "24", "-maximum_number_of_iterations", "20000", "-name", "isoit",
            "-output_directory", str(tmp_path) + "/", "-radiative_equilibrium_criterion", "1e-4",
            "-isothermal_layers", "yes"]
    a = _run_driver(argv + ["-convective_adjustment", "no"], True)
    b = _run_driver(argv + ["-convective_adjustment", "no"], False)
    assert int(a.iso) == 1 and int(a.iter_value) == int(b.iter_value) > 3
    np.testing.assert_allclose(a.T_lay, b.T_lay, rtol=1e-8)
    np.testing.assert_allclose(a.F_up_band, b.F_up_band, rtol=1e-7, atol=1e-12 * b.F_up_band.max())
    with pytest.raises(IOError, match="non-isothermal"):
        _run_driver(argv + ["-convective_adjustment", "yes"], True)


@pytest.mark.parametrize("extra", [[], ["-direct_irradiation_beam", "yes", "-surface_albedo", "0.2"]])
def test_convection_loop_on_device_reaches_the_host_driven_equilibrium(tmp_path, extra):
    """the convection loop with the convective adjustment on the GPU (hx_rt_conv_*) against the per-stage loop whose
    adjustment runs in helios_amd/host_functions.py, both started by run_helios: same number of iterations, same
    convective layers, T-P profile and emission spectrum to 1e-7 (measured 8e-10 / 1e-15; with the criterion at 1e-8 the
    exit test itself is at rounding level and the two runs leave 0-60 iterations apart, 4e-10 from each other).  Both
    loops are held to the reference's by tests/test_gpu_reference.py::test_compute_convection_loop_golden."""
    argv = list(CONV_ARGV)
    argv[argv.index("-radiative_equilibrium_criterion") + 1] = "1e-6"
    argv += ["-output_directory", str(tmp_path) + "/"] + extra
    a = _run_driver(argv, True)
    b = _run_driver(argv, False)
    assert a.conv_layer.sum() > 2 and int(a.iter_value) >= 400
    assert int(a.iter_value) == int(b.iter_value)
    np.testing.assert_array_equal(a.conv_layer, b.conv_layer)
    np.testing.assert_allclose(a.T_lay, b.T_lay, rtol=1e-7)
    np.testing.assert_allclose(a.F_up_band, b.F_up_band, rtol=1e-7, atol=1e-12 * b.F_up_band.max())
    from helios_amd import host_functions as hs
    for q in (a, b):
        assert abs(hs.global_energy_imbalance(q)) < 1e-3
        q.kappa_int = np.full(int(q.ninterface), 0.2)
        hs.conv_check(q)
        assert q.conv_unstable.sum() == 0


@pytest.mark.parametrize("extra", [[], ["-tp_profile_smoothing", "yes"], ["-convective_damping_parameter", "2.5"]])
def test_convection_steps_on_device_match_host_functions(tmp_path, extra):
    """every half-step of the device loop against helios_amd/host_functions.py (pinned to the reference's Python) on
    IDENTICAL inputs: the adjusted profile to round-off, unstable / convective / not-converged flags exactly; 25
    iterations cross two refresh boundaries"""
    import helios
    from helios_amd import computation, host_functions as hs
    captured = {}
    orig = computation.Compute.convection_loop

    def stop(self, quant, write=None, read=None, rt_plot=None):
        captured["q"] = quant
        raise KeyboardInterrupt
    computation.Compute.convection_loop = stop
    try:
        helios.run_helios(CONV_ARGV + ["-output_directory", str(tmp_path) + "/"] + extra)
    except KeyboardInterrupt:
        pass
    finally:
        computation.Compute.convection_loop = orig
    q = captured["q"]
    rt = q.rt
    for n in ("T_lay", "F_net", "F_up_tot", "F_down_tot", "F_smooth_sum", "meanmolmass_lay"):
        setattr(q, n, rt.get(n))
    q.p_lay, q.p_int = np.asarray(q.p_lay, float), np.asarray(q.p_int, float)
    hs.conv_check(q)
    hs.mark_convective_layers(q, stitching=0)
    assert q.conv_unstable.sum() > 0
    for name, v in (("kappa_lay", q.kappa_lay), ("kappa_int", q.kappa_int), ("c_p_lay", q.c_p_lay)):
        rt.set_state(0, name, np.asarray(v, np.float64))
    rt.set_state(0, "conv_layer", np.asarray(q.conv_layer, np.int32))
    rt.set_state(0, "conv_unstable", np.asarray(q.conv_unstable, np.int32))
    rt.set_state(0, "dampara", np.array([-1.0 if q.input_dampara == "automatic" else float(q.input_dampara)]))
    rt.set_state(0, "done", np.zeros(1, np.int32))
    import oracle
    L = int(q.nlayer)
    for it in list(range(25)) + [5999, 6000, 6001]:       # 6000: the hard-coded time-step reset (kernels.cu:2844-2846)
        for n in ("T_lay", "F_net", "F_up_tot", "F_down_tot", "F_smooth_sum"):
            setattr(q, n, rt.get(n))
        q.conv_layer = rt.get("conv_layer").copy()
        q.iter_value = it
        rt.conv_adjust(it)
        q.meanmolmass_lay = rt.get("meanmolmass_lay")        # what the adjustment saw on the device
        hs.convective_adjustment(q)
        T_dev = rt.get("T_lay")
        np.testing.assert_allclose(T_dev, q.T_lay, rtol=1e-13, err_msg="adjusted profile, iteration %d" % it)
        np.testing.assert_array_equal(rt.get("conv_layer"), q.conv_layer)
        np.testing.assert_array_equal(rt.get("conv_unstable"), q.conv_unstable)
        pref0, store0 = rt.get("delta_t_prefactor"), rt.get("T_store")
        rt.conv_advance(it)
        for n in ("F_net", "F_up_tot", "F_down_tot"):
            setattr(q, n, rt.get(n))
        # the temperature step of this half (conv_temp_iter, kernels.cu:2768-2884) against the CPU oracle, fed with what the
        # device step saw: adjusted profile, this iteration's net flux, the layers marked by the equilibrium test
        T_o, pref_o, store_o = T_dev.copy(), pref0.copy(), store0.copy()
        oracle.port.conv_temp_iter(np.asarray(q.F_net, np.float64), np.zeros(L), T_o, q.p_lay, q.p_int, store_o, pref_o,
                                   rt.get("marked_red"), np.asarray(rt.get("F_add_heat_lay"), np.float64), np.zeros(L),
                                   np.asarray(rt.get("F_smooth_sum"), np.float64) * 0.0, L, it, int(q.adapt_interval),
                                   int(q.smooth), float(q.F_intern))
        if int(q.smooth) == 0:
            np.testing.assert_allclose(rt.get("T_lay"), T_o, rtol=1e-12, err_msg="conv_temp_iter, iteration %d" % it)
            np.testing.assert_allclose(rt.get("delta_t_prefactor"), pref_o, rtol=1e-14, err_msg="prefactor, iteration %d" % it)
            np.testing.assert_allclose(rt.get("T_store"), store_o, rtol=1e-14)
        q.T_lay = T_dev.copy()
        hs.mark_convective_layers(q, stitching=1)
        crit = hs.check_for_radiative_eq(q)
        np.testing.assert_array_equal(rt.get("conv_layer"), q.conv_layer)
        np.testing.assert_array_equal(rt.get("marked_red"), q.marked_red)
        assert (int(rt.get("done")[0]) == 0 or it >= 400) and crit in (0, 1)      # iter < 400: the loop must go on
        if int(rt.get("done")[0]):
            rt.set_state(0, "done", np.zeros(1, np.int32))
        assert np.abs(rt.get("T_lay") - T_dev).max() > 0            # and the temperature step was taken


def test_sweep_batch_equals_individual_runs(tmp_path):
    """six columns (3 internal temperatures x 2 heat-redistribution factors) through one device batch: every column ends
    where its own single run ends, with the same files"""
    import sweep
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "24 6 5 13",
            "-number_of_layers", "14", "-maximum_number_of_iterations", "20000", "-name", "sw",
            "-radiative_equilibrium_criterion", "1e-4", "-convective_adjustment", "no"]
    out = str(tmp_path) + "/"
    cols, spectra = sweep.main(["-sweep", "internal_temperature=100,400,900;f_factor=0.25,0.6"] + base +
                               ["-output_directory", out + "batch/"])
    assert len(cols) == 6 and spectra.shape == (6, 24)
    z = np.load(os.path.join(out, "batch", "sw_sweep_spectra.npz"))
    assert list(z["param_internal_temperature"]) == ["100", "100", "400", "400", "900", "900"]
    k = 0
    for T_int in ("100", "400", "900"):
        for f in ("0.25", "0.6"):
            single = _run_driver(base + ["-output_directory", out + "single/", "-name", "s%d" % k,
                                         "-internal_temperature", T_int, "-f_factor", f], True)
            q = cols[k]
            assert int(q.iter_value) == int(single.iter_value), k
            np.testing.assert_allclose(q.T_lay, single.T_lay, rtol=1e-12, err_msg="column %d" % k)
            np.testing.assert_allclose(spectra[k], single.F_up_band[-24:], rtol=1e-12)
            a = open(os.path.join(out, "batch", "sw_%d" % k, "sw_%d_tp.dat" % k)).read()
            b = open(os.path.join(out, "single", "s%d" % k, "s%d_tp.dat" % k)).read()
            assert a == b
            k += 1


def test_sweep_over_fastchem_directories_equals_individual_runs(tmp_path):
    """a sweep over chemistry (`-sweep "directory_with_fastchem_files=dirA/,dirB/,dirC/"`: metallicity, C/O): three columns in
    ONE device batch, each with its own FastChem tables on the device (hx_rt_set_column_vmr_table), end where their own
    single runs end -- same iteration counts, profiles and spectra -- and differ from each other"""
    import shutil
    import sweep
    wd = str(tmp_path)
    mk = _host_golden_module()
    mk.write_species_inputs(wd, nbin=14, ny=20, sorted_k=True)
    dirs = [os.path.join(wd, "chem") + "/"]
    for k, seed in enumerate((21, 22)):             # two more chemistries: the same files with other abundances
        alt = os.path.join(wd, "alt%d" % k)
        mk.write_species_inputs(alt, seed=seed, nbin=14, ny=20, sorted_k=True)
        d = os.path.join(wd, "chem%d" % k)
        shutil.copytree(os.path.join(alt, "chem"), d)
        dirs.append(d + "/")
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "on-the-fly",
            "-path_to_species_file", os.path.join(wd, "species.dat"),
            "-file_with_vertical_mixing_ratios", os.path.join(wd, "vmr.txt"),
            "-directory_with_opacity_files", os.path.join(wd, "opac") + "/",
            "-number_of_layers", "18", "-maximum_number_of_iterations", "20000", "-name", "chem",
            "-radiative_equilibrium_criterion", "1e-4", "-convective_adjustment", "no", "-toa_pressure", "1e0",
            "-boa_pressure", "1e8"]
    cols, spectra = sweep.main(["-sweep", "directory_with_fastchem_files=" + ",".join(dirs)] + base +
                               ["-output_directory", wd + "/batch/"])
    assert len(cols) == 3
    for k, d in enumerate(dirs):
        single = _run_driver(base + ["-directory_with_fastchem_files", d, "-output_directory", wd + "/single/",
                                     "-name", "s%d" % k], True)
        assert int(cols[k].iter_value) == int(single.iter_value), k
        np.testing.assert_allclose(cols[k].T_lay, single.T_lay, rtol=1e-12, err_msg="column %d" % k)
        np.testing.assert_allclose(spectra[k], single.F_up_band[-14:], rtol=1e-12)
        for sb, ss in zip(cols[k].species_list, single.species_list):
            if getattr(sb, "source_for_vmr", "") == "FastChem":
                np.testing.assert_allclose(sb.vmr_layer, ss.vmr_layer, rtol=1e-12, err_msg="%s column %d" % (sb.name, k))
    assert np.abs(cols[0].T_lay / cols[1].T_lay - 1.0).max() > 1e-4 and np.abs(spectra[0] / spectra[2] - 1.0).max() > 1e-4


def test_sweep_with_the_matrix_method_equals_individual_runs(tmp_path):
    """`flux calculation method = matrix` in a sweep: four columns through one device batch (the solver runs column by
    column inside the loop, the work arrays of the elimination are shared) end where their own single runs end"""
    import sweep
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "24 6 5 13",
            "-number_of_layers", "14", "-maximum_number_of_iterations", "20000", "-name", "mx",
            "-radiative_equilibrium_criterion", "1e-4", "-convective_adjustment", "no", "-surface_albedo", "0.1",
            "-flux_calculation_method", "matrix"]
    out = str(tmp_path) + "/"
    cols, spectra = sweep.main(["-sweep", "internal_temperature=100,700;f_factor=0.25,0.6"] + base +
                               ["-output_directory", out + "batch/"])
    assert len(cols) == 4 and spectra.shape == (4, 24) and all(q.flux_calc_method == "matrix" for q in cols)
    k = 0
    for T_int in ("100", "700"):
        for f in ("0.25", "0.6"):
            single = _run_driver(base + ["-output_directory", out + "single/", "-name", "s%d" % k,
                                         "-internal_temperature", T_int, "-f_factor", f], True)
            assert single.rt is not None and int(cols[k].iter_value) == int(single.iter_value), k
            np.testing.assert_allclose(cols[k].T_lay, single.T_lay, rtol=1e-12, err_msg="column %d" % k)
            np.testing.assert_allclose(spectra[k], single.F_up_band[-24:], rtol=1e-12)
            k += 1


def test_sweep_of_post_processing_runs_is_one_pass_per_column(tmp_path):
    """`-run_type post-processing` in a sweep (singlewalk = 1, isothermal layers): the batch takes ONE pass without a
    temperature step -- every column keeps the T-P profile it was given and ends where its own single post-processing run
    ends (an earlier version iterated the fixed profile 'to convergence' with 1001 sweeps per iteration)"""
    import sweep
    wd = str(tmp_path)
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "24 6 5 7",
            "-number_of_layers", "30", "-maximum_number_of_iterations", "20000", "-radiative_equilibrium_criterion", "1e-5",
            "-convective_adjustment", "no"]
    it = _run_driver(base + ["-output_directory", wd + "/", "-name", "run"], True)
    pp_args = base + ["-run_type", "post-processing", "-path_to_temperature_file", os.path.join(wd, "run", "run_tp.dat")]
    cols, spectra = sweep.main(["-sweep", "f_factor=0.25,0.6"] + pp_args + ["-output_directory", wd + "/batch/", "-name", "pp"])
    assert len(cols) == 2 and spectra.shape == (2, 24)
    for k, f in enumerate(("0.25", "0.6")):
        single = _run_driver(pp_args + ["-output_directory", wd + "/single/", "-name", "s%d" % k, "-f_factor", f], True)
        q = cols[k]
        assert int(q.singlewalk) == 1 and int(q.iso) == 1 and int(q.iter_value) == 0 == int(single.iter_value)
        np.testing.assert_allclose(q.T_lay, it.T_lay, rtol=1e-5)            # the given profile, untouched (6 digits in the file)
        np.testing.assert_allclose(q.T_lay, single.T_lay, rtol=1e-14)
        np.testing.assert_allclose(spectra[k], single.F_up_band[-24:], rtol=1e-12)
    assert np.abs(spectra[0] / spectra[1] - 1.0).max() > 1e-3


def test_sweep_with_convection_and_two_ranks(tmp_path):
    """a sweep with convective columns, sharded over two ranks (gloo hook: both on GPU 0): spectra gathered in sweep
    order and equal to the single-process sweep"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "20 6 5 11",
            "-number_of_layers", "16", "-maximum_number_of_iterations", "20000", "-name", "cvs",
            "-radiative_equilibrium_criterion", "1e-4", "-kappa_value", "0.2"]
    spec = "internal_temperature=200,1500,1800"
    import sweep
    cols, spectra = sweep.main(["-sweep", spec] + base + ["-output_directory", str(tmp_path) + "/one/"])
    assert [int(c.conv_layer.sum() > 0) if c.conv_layer is not None else 0 for c in cols] == [0, 1, 1]
    env = dict(os.environ, HELIOS_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29523", "sweep.py", "-sweep", spec] + base +
                       ["-output_directory", str(tmp_path) + "/two/"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "Load over 2 ranks" in p.stdout          # columns need different iteration counts: the imbalance is reported
    z = np.load(os.path.join(str(tmp_path), "two", "cvs_sweep_spectra.npz"))
    np.testing.assert_allclose(z["F_up_TOA"], spectra, rtol=1e-12)
    for k in range(3):
        assert os.path.getsize(os.path.join(str(tmp_path), "two", "cvs_%d" % k, "cvs_%d_tp.dat" % k)) > 100


def test_sweep_from_a_shared_work_list_on_two_ranks(tmp_path):
    """HELIOS_SWEEP_PARTITION=dynamic:1 -- two ranks (gloo hook: both on GPU 0) claim the six columns one at a time from
    the shared list, each column runs in a batch of its own: same spectra, in sweep order, as the one-batch sweep; every
    column's files are there exactly once"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "20 6 5 11",
            "-number_of_layers", "16", "-maximum_number_of_iterations", "20000", "-name", "wl",
            "-radiative_equilibrium_criterion", "1e-4", "-convective_adjustment", "no"]
    spec = "internal_temperature=150,600,1100;f_factor=0.25,0.5"
    import sweep
    cols, spectra = sweep.main(["-sweep", spec] + base + ["-output_directory", str(tmp_path) + "/one/"])
    assert len({int(c.iter_value) for c in cols}) > 1           # the columns do need different numbers of iterations
    env = dict(os.environ, HELIOS_BENCH_BACKEND="gloo", HELIOS_SWEEP_PARTITION="dynamic:1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29527", "sweep.py", "-sweep", spec] + base +
                       ["-output_directory", str(tmp_path) + "/two/"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "Load over 2 ranks" in p.stdout
    z = np.load(os.path.join(str(tmp_path), "two", "wl_sweep_spectra.npz"))
    np.testing.assert_allclose(z["F_up_TOA"], spectra, rtol=1e-12)
    for k in range(6):
        a = open(os.path.join(str(tmp_path), "one", "wl_%d" % k, "wl_%d_tp.dat" % k)).read()
        b = open(os.path.join(str(tmp_path), "two", "wl_%d" % k, "wl_%d_tp.dat" % k)).read()
        assert a == b, k


def test_sweep_with_kappa_table_equals_single_runs(tmp_path):
    """`kappa value = file` inside a sweep: kappa and c_p are interpolated at each column's own profile before the
    convective-stability check (as a single run does), and the output files report them"""
    import sweep
    mk = _host_golden_module()
    path = os.path.join(str(tmp_path), "delad.dat")
    mk.write_kappa_file(path, False, const_kappa=0.1)
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "20 6 5 11",
            "-number_of_layers", "16", "-maximum_number_of_iterations", "20000", "-name", "ks",
            "-radiative_equilibrium_criterion", "1e-4", "-kappa_value", "file", "-kappa_file_path", path]
    out = str(tmp_path) + "/"
    cols, spectra = sweep.main(["-sweep", "internal_temperature=200,1500"] + base + ["-output_directory", out + "b/"])
    flags = [int(c.conv_layer.sum() > 0) if c.conv_layer is not None else 0 for c in cols]
    assert flags == [0, 1]                          # the cool column stays radiative: no spurious instability from kappa = 0
    for k, T_int in enumerate(("200", "1500")):
        single = _run_driver(base + ["-output_directory", out + "s/", "-name", "k%d" % k, "-internal_temperature", T_int],
                             True)
        assert int(cols[k].iter_value) == int(single.iter_value), k
        np.testing.assert_allclose(cols[k].T_lay, single.T_lay, rtol=1e-12)
        np.testing.assert_allclose(cols[k].kappa_lay, single.kappa_lay, rtol=1e-12)
