"""Host logic of the driver surface (no GPU needed): grid, altitudes, VMR interpolation, convective
adjustment, parameter parsing, output formats."""
import io
import os

import numpy as np
import pytest

from helios_amd import host_functions as hs
from helios_amd import phys_const as pc
from helios_amd.read import Read
from helios_amd.write import Write


class Q(object):
    fl_prec = np.float64


def test_pressure_grid():
    q = Q()
    q.p_boa, q.p_toa, q.nlayer, q.g = 1e9, 1e-1, 105, 1000.0
    hs.construct_grid(q)
    p_lay, p_int = np.array(q.p_lay), np.array(q.p_int)
    assert len(p_lay) == 105 and len(p_int) == 106
    assert p_int[0] == 1e9 and np.isclose(p_lay[-1], 1e-1)
    assert np.all(np.diff(np.log10(p_int)) < 0)
    # centres sit between their interfaces, uniformly in log p
    np.testing.assert_allclose(np.log10(p_lay), 0.5 * (np.log10(p_int[:-1]) + np.log10(p_int[1:])), rtol=1e-12)
    np.testing.assert_allclose(np.array(q.delta_col_upper) + np.array(q.delta_col_lower), q.delta_colmass, rtol=1e-12)


def test_automatic_layer_count_and_param_parsing(tmp_path):
    p = tmp_path / "param.dat"
    p.write_text("name = abc [x]\nnumber of layers = automatic  [automatic, number > 0] (CL: Y)\n"
                 "TOA pressure [10^-6 bar] = 1e-1 [number > 0] (CL: Y)\nBOA pressure [10^-6 bar] = 1e9 [n]\n"
                 "direct irradiation beam = yes [yes, no]\nyes --> stellar zenith angle [deg] = 80 [number]\n")
    r = Read()
    q = Q()
    q.no_atmo_mode = 0
    r.read_param_file_and_command_line(q, None, ["-parameter_file", str(p), "-scattering", "no"])
    assert q.name == "abc" and int(q.nlayer) == 105 and int(q.ninterface) == 106     # ceil(10.5*10)
    assert q.scat == 0 and q.dir_beam == 1 and q.geom_zenith_corr == 1               # zenith > 70 deg
    assert np.isclose(q.mu_star, np.cos(np.pi - 80 * np.pi / 180)) and q.mu_star < 0
    assert q.iso == 0 and q.singlewalk == 0 and q.energy_correction == 1 and np.isclose(q.epsi, 0.5)
    assert q.crit_relaxation_numbers == [10000, 20000] and q.plancktable_dim == 8000


def test_isothermal_layers_with_convective_adjustment_are_refused_when_the_input_is_read(tmp_path):
    """the reference fails with a TypeError after the whole radiative loop (computation.py:1004-1009); here the
    combination is refused at set-up (Read.check_run_configuration, called by run_helios and the sweep driver), before any GPU work -- but not for the post-processing run type, which is isothermal
    by definition and never enters the convection loop"""
    import pytest
    p = tmp_path / "param.dat"
    p.write_text("name = abc [x]\n")
    for flags, ok in ((["-isothermal_layers", "yes", "-convective_adjustment", "yes"], False),
                      (["-isothermal_layers", "yes", "-convective_adjustment", "no"], True),
                      (["-run_type", "post-processing", "-convective_adjustment", "yes"], True)):
        r, q = Read(), Q()
        q.no_atmo_mode = 0
        r.read_param_file_and_command_line(q, None, ["-parameter_file", str(p)] + flags)   # parses, as the reference's reader
        assert q.iso == 1
        if ok:
            r.check_run_configuration(q)
        else:
            with pytest.raises(IOError, match="non-isothermal"):
                r.check_run_configuration(q)


def test_height_z_gas_and_rocky():
    q = Q()
    q.nlayer = 6
    q.p_lay = np.array([5e8, 5e7, 2e7, 5e6, 1e5, 1e3])
    q.delta_z_lay = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    q.z_lay = np.zeros(6)
    q.planet_type = "gas"
    hs.calculate_height_z(q)
    assert q.z_lay[2] == 0                      # highest layer with p >= 1e7
    np.testing.assert_allclose(q.z_lay, [-4.0, -2.5, 0.0, 3.5, 8.0, 13.5])
    q.planet_type = "rocky"
    hs.calculate_height_z(q)
    np.testing.assert_allclose(q.z_lay, [0.5, 2.0, 4.5, 8.0, 12.5, 18.0])


def test_vmr_interpolation_matches_scipy_bilinear():
    from scipy import interpolate
    rng = np.random.default_rng(1)
    temp = np.linspace(100, 3000, 9)
    logp = np.linspace(0, 9, 7)
    tab = rng.uniform(1e-8, 1e-2, (9, 7))
    Tq = rng.uniform(20, 3500, 40)          # includes out-of-grid queries (clamped)
    pq = rng.uniform(-2, 11, 40)
    got = hs.interpolate_grid_to_lay_or_int(logp, temp, tab, pq, Tq)
    f = interpolate.RectBivariateSpline(temp, logp, tab, kx=1, ky=1)
    want = [f(Tq[i], pq[i])[0][0] for i in range(40)]
    np.testing.assert_allclose(got, want, rtol=1e-12)


def _conv_quant(L=30):
    q = Q()
    q.nlayer, q.ninterface = L, L + 1
    q.p_boa, q.p_toa, q.g = 1e9, 1e-1, 1000.0
    hs.construct_grid(q)
    q.p_lay, q.p_int = np.array(q.p_lay), np.array(q.p_int)
    kap = 2.0 / 7.0
    q.kappa_lay, q.kappa_int = np.full(L, kap), np.full(L + 1, kap)
    q.c_p_lay = np.full(L, pc.R_UNIV / kap)
    q.meanmolmass_lay = np.full(L, 2.3 * pc.AMU)
    # strongly super-adiabatic below 1 bar, isothermal above
    T = np.where(q.p_lay > 1e6, 1500.0 * (q.p_lay / 1e6) ** 0.45, 1500.0)
    q.T_lay = np.append(T, T[0] * 1.3)
    q.conv_layer = np.zeros(L + 1, np.int32)
    q.iter_value, q.input_dampara, q.T_star = 0, "automatic", 5000.0
    q.F_intern = 1e3
    q.F_add_heat_sum, q.F_smooth_sum = np.zeros(L), np.zeros(L)
    q.F_down_tot, q.F_up_tot = np.full(L + 1, 5e8), np.full(L + 1, 5e8)
    return q


def test_convective_adjustment_removes_instability_and_conserves_enthalpy():
    q = _conv_quant()
    L = q.nlayer
    dp = q.p_int[:-1] - q.p_int[1:]
    H0 = np.sum(q.c_p_lay / q.meanmolmass_lay * q.T_lay[:L] * dp)
    hs.conv_check(q)
    assert q.conv_unstable.sum() > 5
    hs.convective_adjustment(q)
    hs.conv_check(q)
    assert q.conv_unstable.sum() == 0
    assert q.conv_layer.sum() > 5 and q.conv_layer[L] == 1           # surface joins the bottom zone
    # layers inside a zone follow the dry adiabat: T p^-kappa constant (centre to centre)
    idx = [i for i in range(L - 1) if q.conv_layer[i] and q.conv_layer[i + 1]]
    theta = q.T_lay[:L] * q.p_lay ** (-2.0 / 7.0)
    np.testing.assert_allclose(theta[idx], theta[[i + 1 for i in idx]], rtol=1e-6)
    H1 = np.sum(q.c_p_lay / q.meanmolmass_lay * q.T_lay[:L] * dp)
    assert abs(H1 - H0) / H0 < 0.011                                   # fudge factor is clipped to 1 +- 1 %


def test_radiative_eq_check_and_relaxation():
    q = _conv_quant(8)
    q.rad_convergence_limit = 1e-8
    q.F_net = np.full(9, q.F_intern)
    q.iter_value = 5
    assert hs.check_for_radiative_eq(q) == 1
    q.F_net[4] += 100.0
    assert hs.check_for_radiative_eq(q) == 0 and q.marked_red[3] == 1
    hs.relax_radiative_convergence_criterion(q)
    assert q.rad_convergence_limit == 1e-7 and q.relaxed_criterion_trigger == 1


def test_output_formats(tmp_path):
    """the fixed-width layouts downstream tools parse (source/tools.py:297, :397)"""
    q = Q()
    q.name, q.nbin, q.nlayer, q.ninterface = "t", 3, 2, 3
    q.iso, q.convection, q.singlewalk, q.T_star = 0, 0, 0, 5000.0
    q.T_lay = np.array([1000.123456789, 900.0, 1100.0])
    q.p_lay, q.p_int = np.array([1e8, 1e6]), np.array([1e9, 1e7, 1e5])
    q.z_lay, q.delta_z_lay = np.array([1e6, 3e6]), np.array([2e6, 2e6])
    q.conv_unstable = q.conv_layer = None
    q.R_star, q.a, q.f_factor, q.dir_beam, q.mu_star = 7e10, 7.5e11, 0.5, 0, -0.5
    q.F_down_tot, q.F_up_tot, q.F_net = np.array([1.0, 2.0, 3.0]), np.array([4.0, 5.0, 6.0]), np.array([3.0, 3.0, 3.0])
    q.F_dir_tot, q.F_net_diff, q.F_net_conv = np.zeros(3), np.array([0.1, 0.2]), np.zeros(3)
    q.F_add_heat_lay, q.F_intern = np.zeros(2), 45.9
    q.opac_wave, q.opac_interwave, q.opac_deltawave = np.array([1e-4, 2e-4, 3e-4]), np.array([.5e-4, 1.5e-4, 2.5e-4, 3.5e-4]), np.full(3, 1e-4)
    q.F_up_band = np.arange(9) * 1.23456789e3
    q.F_down_band = np.arange(9) * 2.0
    q.F_ratio = [1e-3, 2e-3, 3e-3]
    r = Read()
    r.output_path = str(tmp_path) + "/"
    for w in (Write.write_tp, Write.write_upward_spectral_flux, Write.write_TOA_flux_eclipse_depth):
        w(q, r)          # the full set is compared byte for byte in tests/test_host_golden.py
    tp = open(os.path.join(str(tmp_path), "t", "t_tp.dat")).read().split("\n")
    assert tp[2].startswith("BOA     1100") and "not_calculated" in tp[2]
    assert tp[3].split()[:3] == ["0", "1000.12", "1e+08"]             # {:g}: 6 significant digits
    up = open(os.path.join(str(tmp_path), "t", "t_spec_upflux.dat")).read().split("\n")
    assert up[2].startswith("bin     cent_lambda[um]") and "F_up[2]" in up[2]
    assert up[4].split()[4] == "1.23456789e+03"                       # {:<16.8e}: 9 digits
    toa = open(os.path.join(str(tmp_path), "t", "t_TOA_flux_eclipse.dat")).read().split("\n")
    assert toa[3].split()[-1] == "0.001"
