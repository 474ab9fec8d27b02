"""GPU parity, per-stage entry points (hx_<kernel>) of libhelios_hip.so, called through the C-ABI:
against the committed golden vectors (from the reference's kernels) and against the CPU oracle."""
import numpy as np
import pytest

import cases
import golden_checks as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from impls import hip_impl
    return hip_impl()


@pytest.mark.parametrize("name", gc.CHAIN_NAMES)
def test_stage_chain_golden(hip, name):
    # libm differences (OCML vs glibc exp/log10/pow, <= 1-2 ulp) enter here: 1e-10 after one
    # iteration, 1e-7 after twelve (tolerance of the north star: 1e-6 on fluxes)
    gc.check_chain(hip, name, rtol1=1e-10, rtol12=1e-7, planck_rtol=1e-6)


def test_mixing_golden(hip):
    gc.check_mixing(hip, rtol=1e-11)


@pytest.mark.parametrize("cfg", [dict(nbin=37, nlayer=23), dict(nbin=70, nlayer=50, clouds=1, g_0=0.2, dir_beam=1),
                                 dict(nbin=33, nlayer=17, iso=1, dir_beam=1, geom_zenith_corr=1, zenith_deg=75.0)])
def test_stage_chain_vs_oracle(hip, port, cfg):
    c0 = cases.make_case(**cfg)
    out = []
    grid = None
    for impl in (hip, port):
        c = c0.copy()
        s = cases.alloc_state(c)
        cases.setup_planck(impl, c, s)
        if grid is None:
            grid = s.planck_grid.copy()
        else:
            # Phi_n(y_top) - Phi_n(y_bot) cancels catastrophically in the Rayleigh-Jeans tail of hot rows
            # (kernels.cu:103-104): different exp() implementations differ there by up to ~1e-7; compare
            # the tables at that level and continue both chains from the same table
            np.testing.assert_allclose(grid, s.planck_grid, rtol=1e-6, atol=1e-280)
            s.planck_grid[:] = grid
        cases.radiation_iterations(impl, c, s, 3)
        out.append((c, s))
    (ch, sh), (cp, sp) = out
    scale = max(np.abs(sp.F_down_wg).max(), np.abs(sp.F_dir_wg).max())
    for k in ("opac_wg_lay", "opac_wg_int", "scat_cross_lay", "meanmolmass_lay", "planckband_lay"):
        np.testing.assert_allclose(sh[k], sp[k], rtol=1e-11, atol=1e-13 * np.abs(sp[k]).max(), err_msg=k)
    for k in ("F_down_wg", "F_up_wg", "F_dir_wg", "F_down_band", "F_up_band"):
        np.testing.assert_allclose(sh[k], sp[k], rtol=1e-8, atol=1e-13 * scale, err_msg=k)
    np.testing.assert_allclose(sh.F_net, sp.F_net, rtol=1e-8, atol=1e-12 * np.abs(sp.F_up_tot).max())
    np.testing.assert_allclose(ch.T_lay, cp.T_lay, rtol=1e-8)


@pytest.mark.parametrize("name", gc.MATRIX_NAMES)
def test_matrix_flux_solve_golden(hip, name):
    # inputs are the golden coefficient planes, the solve is +,-,*,/ only: same bits expected
    gc.check_matrix(hip, name, rtol=1e-12)


@pytest.mark.parametrize("cfg", [dict(nbin=41, nlayer=30, albedo=0.1), dict(nbin=25, nlayer=60, clouds=1, g_0=0.2,
                                                                           scat_corr=1, dir_beam=1, albedo=0.3),
                                 dict(nbin=19, nlayer=21, iso=1, dir_beam=1, albedo=1e-8)])
def test_matrix_flux_solve_vs_oracle(hip, port, cfg):
    c0 = cases.make_case(**cfg)
    out = []
    for impl in (hip, port):
        c = c0.copy()
        s = cases.alloc_state(c)
        cases.setup_planck(port, c, s)
        cases.interpolate_temperatures_and_planck(port, c, s)
        cases.refresh_premixed(port, c, s)
        s.scat_trigger[::3] = 0
        m = cases.flux_matrix(impl, c, s)
        out.append((s, m))
    (sh, mh), (sp, mp) = out
    for k in mp:
        np.testing.assert_allclose(mh[k], mp[k], rtol=1e-12, atol=1e-300, err_msg=k)
    for k in ("F_down_wg", "F_up_wg") + (("Fc_down_wg", "Fc_up_wg") if c0.iso == 0 else ()):
        np.testing.assert_allclose(sh[k], sp[k], rtol=1e-12, atol=1e-14 * np.abs(sp[k]).max(), err_msg=k)


def test_scalar_table_interpolations_vs_oracle(hip, port):
    """kappa / c_p / entropy / phase-number tables (kernels.cu:703-919): bilinear in (T or log10 T, log10 P) with the
    0.001 clamp; profile points inside, on the grid nodes and outside the table"""
    rng = np.random.default_rng(11)
    nt, npr = 9, 7
    et, ep = np.linspace(100, 4000, nt), np.logspace(0, 9, npr)
    tab = rng.uniform(0.1, 0.4, nt * npr)
    temp = np.concatenate((rng.uniform(50, 4500, 20), et[[0, 3, -1]], [10.0, 9000.0]))
    press = np.concatenate((10.0 ** rng.uniform(-1, 10, 20), ep[[0, 2, -1]], [1e-3, 1e12]))
    n = len(temp)
    for fn in ("kappa_interpol", "cp_interpol", "entropy_interpol", "phase_number_interpol"):
        a, b = np.zeros(n), np.zeros(n)
        getattr(hip, fn)(temp, et, press, ep, a, tab, npr, nt, n)
        getattr(port, fn)(temp, et, press, ep, b, tab, npr, nt, n)
        np.testing.assert_allclose(a, b, rtol=1e-12, err_msg=fn)


@pytest.mark.parametrize("cfg", [dict(clouds=1, g_0=0.2, nbin=21, nlayer=11), dict(nbin=37, nlayer=45, ny=16, dir_beam=1),
                                 dict(nbin=9, nlayer=100), dict(nbin=300, nlayer=33, ny=1), dict(iso=1, nbin=14, nlayer=37)])
def test_post_loop_diagnostics_vs_oracle(hip, port, cfg):
    """band means, contribution function (columns shorter and longer than the kernel's chunk of 16 layers, 1 / 16 / 20
    Gauss points, isothermal layers), mean opacities, beam flux.  The contribution function multiplies and adds in the
    reference's order: 1e-13, where any other association of the 100-factor products would show"""
    c0 = cases.make_case(**cfg)
    res = []
    for impl in (hip, port):
        c = c0.copy()
        s = cases.alloc_state(c)
        cases.setup_planck(impl, c, s)
        cases.radiation_iterations(impl, c, s, 1)
        X, Y, L, I = c.nbin, c.ny, c.nlayer, c.ninterface
        if c.iso == 1:
            impl.integrate_optdepth_transmission_iso(s.trans_wg, s.trans_band, s.delta_tau_wg, s.delta_tau_band,
                                                     c.gauss_weight, X, L, Y)
            impl.calc_contr_func_iso(s.trans_wg, s.trans_weight_band, s.contr_func_band, c.gauss_weight,
                                     s.planckband_lay, c.epsi, X, L, Y)
        else:
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
    (sh, ph), (sp, pp) = res
    for k in ("trans_band", "delta_tau_band", "delta_tau_all_clouds", "trans_weight_band",
              "contr_func_band", "opac_band_lay", "F_dir_tot"):
        rtol = 1e-13 if k in ("trans_weight_band", "contr_func_band", "trans_band", "delta_tau_band") else 1e-10
        np.testing.assert_allclose(sh[k], sp[k], rtol=rtol, atol=1e-13 * np.abs(sp[k]).max(), err_msg=k)
    for a, b in zip(ph, pp):
        np.testing.assert_allclose(a, b, rtol=1e-9)


def test_random_overlap_requires_ny20(hip):
    from helios_amd._lib import HeliosHipError
    ny, nbin, nlev = 8, 4, 3
    n = ny * nbin * nlev
    with pytest.raises(HeliosHipError):
        hip.add_to_mixed_opac(np.ones(nlev), np.ones(n), np.ones(n), np.ones(nlev), np.ones(ny),
                              np.linspace(0.1, 0.9, ny), 1.0, 1, 1, ny, nbin, nlev)


def _ro_cases():
    """random-overlap problems that stress the ordering of the 400 pair sums: generic, one absorber dominant (whole
    runs of sums round to the same double), coarse values (many exact ties in both operands), constant curves"""
    ny, nbin, nlev = 20, 96, 5
    rng = np.random.default_rng(17)
    shape = (nlev, nbin, ny)
    fac = 1e-3 * 18.0 / 2.3
    generic = np.sort(10.0 ** rng.uniform(-4, 0, shape), axis=2)
    out = {}
    out["generic"] = (generic, np.sort(10.0 ** rng.uniform(-4, 0, shape), axis=2) / fac)
    out["wide"] = (np.sort(10.0 ** rng.uniform(-12, 6, shape), axis=2),
                   np.sort(10.0 ** rng.uniform(-12, 6, shape), axis=2) / fac)     # sums absorb the small partner
    out["coarse"] = (np.round(generic, 1) + 0.1, (np.round(generic[::-1], 1) + 0.1) / fac)
    out["flat_mix"] = (np.full(shape, 0.25), out["generic"][1])
    out["flat_both"] = (np.full(shape, 0.25), np.full(shape, 0.5) / fac)
    out["unsorted"] = (10.0 ** rng.uniform(-2, 0, shape), 10.0 ** rng.uniform(-2, 0, shape) / fac)
    # the quantised keys' hard cases: two curves of the same shape (every anti-diagonal of the tableau ties to the last
    # bits), a narrow range (all sums within 1e-5: the key shift goes below 32), a wide one with a tiny minimum
    # (denormal-sized sums next to O(1e6)), and an absorber just above the 1 % negligibility threshold
    shape_curve = np.sort(10.0 ** rng.uniform(-3, 1, shape), axis=2)
    out["same_shape"] = (shape_curve, shape_curve * (1.0 + 1e-9 * rng.uniform(-1, 1, shape)) / fac)
    out["narrow"] = (1.0 + 1e-5 * np.sort(rng.uniform(0, 1, shape), axis=2),
                     (1.0 + 1e-5 * np.sort(rng.uniform(0, 1, shape), axis=2)) / fac)
    out["tiny_min"] = (np.sort(10.0 ** rng.uniform(-300, 6, shape), axis=2),
                       np.sort(10.0 ** rng.uniform(-300, 6, shape), axis=2) / fac)
    out["dominated"] = (generic, np.sort(generic[..., ::-1] * 0.03, axis=2) / fac)
    # levels that differ little from their predecessor (the same order, a slowly drifting one, drifting with exact ties)
    lev = np.arange(nlev)[:, None, None]
    base_m, base_a = generic[:1], np.sort(10.0 ** rng.uniform(-4, 0, (1, nbin, ny)), axis=2) / fac
    out["levels_same_order"] = (base_m * 1.3 ** lev, base_a * 1.3 ** lev)
    out["levels_drifting"] = (base_m * (1.0 + 0.01 * lev), base_a * (1.0 - 0.012 * lev) * (1.0 + 0.004 * lev * gy_like(ny)))
    out["levels_drifting_ties"] = (np.round(base_m, 2) + 0.01 + 0.0 * lev, (np.round(base_a * fac, 2) + 0.01) / fac * (1.0 + 0.5 * (lev % 2)))
    return (ny, nbin, nlev), out


def gy_like(ny):
    return np.linspace(0.0, 1.0, ny)[None, None, :]



def _ro_run(impl):
    from helios_amd import phys_const as pc
    from helios_amd import synthetic as syn
    (ny, nbin, nlev), cs = _ro_cases()
    gy, gw = syn.gauss_points(ny)
    res = {}
    for name, (mix0, spec) in cs.items():
        mix = np.ascontiguousarray(mix0.reshape(-1)).copy()
        impl.add_to_mixed_opac(np.full(nlev, 1e-3), np.ascontiguousarray(spec.reshape(-1)), mix,
                               np.full(nlev, 2.3 * pc.AMU), gw, gy, 18.0 * pc.AMU, 1, 1, ny, nbin, nlev)
        res[name] = mix
    return res


def test_random_overlap_orderings_vs_oracle(hip, port, tmp_path):
    """the quantised-key network with exact finish (default: the lean kernel of round 6, whose keys carry their cell) against
    the oracle's adjacent-swap sort, incl. problems full of equal sums, and against the three other device variants -- the
    kernel of rounds 2-5 (HELIOS_RO_SORT=q32: fill positions as tie-break, sums and weights through LDS images), the fp64
    bitonic network (HELIOS_RO_SORT=bitonic) and the all-pairs ranking (HELIOS_RO_SORT=rank); the knob is read once per
    process -> child processes.  All four must agree bit for bit."""
    import os
    import subprocess
    import sys
    got = _ro_run(hip)
    want = _ro_run(port)
    for k in want:
        assert np.all(np.isfinite(got[k])), k
        # the interpolation weight (y - Y[w-1]) / (Y[w] - Y[w-1]) carries the rounding of the cumulative weights Y ~ 1 (a
        # parallel scan here, 399 sequential additions in the reference) over an interval that can be as short as the
        # smallest weight product, (0.0176 / 2)^2 = 7.7e-5: 2.2e-16 / 7.7e-5 = 3e-12.  'wide': neighbouring sums up to 18
        # decades apart multiply that by K[w] / K[w-1]
        np.testing.assert_allclose(got[k], want[k], rtol=2e-11 if k == "wide" else 5e-12, err_msg=k)
    here = os.path.dirname(os.path.abspath(__file__))
    for kind in ("q32", "rank", "bitonic"):
        out = tmp_path / (kind + ".npz")
        code = ("import sys, numpy as np; sys.path.insert(0, %r); import test_gpu_stages as t; from impls import hip_impl; "
                "np.savez(%r, **t._ro_run(hip_impl()))" % (here, str(out)))
        env = dict(os.environ, HELIOS_RO_SORT=kind)
        subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd=os.path.dirname(here), timeout=600)
        z = np.load(out)
        for k in want:
            np.testing.assert_array_equal(got[k], z[k], err_msg="%s vs %s" % (k, kind))


def test_random_overlap_every_crossing_index_vs_oracle(hip, port):
    """two ascending curves that cross exactly once, behind Gauss point yx - 1, for every yx = 1 ... 19, with either curve
    the stronger one at y = 0, over a narrow range (key shift < 32 bits) and over 14 decades (shift >= 32 bits), and curves
    that cross several times (the LAST crossing decides the fill order, kernels.cu:3321-3329): all four fill variants of
    the run layout against the oracle's adjacent-swap sort; plus tableaux whose rows just touch (presorted or not by one
    ulp)"""
    from helios_amd import phys_const as pc
    from helios_amd import synthetic as syn
    ny = 20
    gy, gw = syn.gauss_points(ny)
    rng = np.random.default_rng(77)
    rows = []
    for decades in (1.5, 14.0):
        for yx in range(1, ny):
            for first in (0, 1):
                a = np.sort(10.0 ** rng.uniform(-3.0, -3.0 + decades, ny))
                # b below a before yx, above it from yx on (or the other way round)
                b = np.where(np.arange(ny) < yx, a * rng.uniform(0.55, 0.95, ny), a * rng.uniform(1.05, 1.8, ny))
                b = np.maximum.accumulate(b)
                rows.append((a, b) if first else (b, a))
        for _ in range(24):          # several crossings
            a = np.sort(10.0 ** rng.uniform(-3.0, -3.0 + decades, ny))
            b = np.maximum.accumulate(a * np.where(rng.uniform(size=ny) < 0.5, 0.8, 1.25))
            rows.append((a, b))
    # rows that touch: outer[i] + inner[19] == outer[i + 1] + inner[0] exactly, and one ulp above
    inner = np.linspace(1.0, 2.0, ny)
    outer = 4.0 + np.arange(ny) * 1.0
    rows.append((outer, inner))
    rows.append((outer, inner * (1.0 + np.where(np.arange(ny) == ny - 1, 2.3e-16, 0.0))))
    rows.append((outer * 64.0, inner))       # far apart: presorted
    nbin, nlev = len(rows), 1
    mix0 = np.array([r[0] for r in rows])
    spec = np.array([r[1] for r in rows]) / (1e-3 * 18.0 / 2.3)       # add = vmr * mass / mu * spec
    out = []
    for impl in (hip, port):
        mix = np.ascontiguousarray(mix0.reshape(-1)).copy()
        impl.add_to_mixed_opac(np.full(nlev, 1e-3), np.ascontiguousarray(spec.reshape(-1)), mix, np.full(nlev, 2.3 * pc.AMU),
                               gw, gy, 18.0 * pc.AMU, 1, 1, ny, nbin, nlev)
        out.append(mix.reshape(nbin, ny))
    assert np.all(np.isfinite(out[0])) and np.abs(out[1] - mix0).max() > 0
    for k in range(nbin):
        np.testing.assert_allclose(out[0][k], out[1][k], rtol=2e-11, err_msg="problem %d" % k)


def test_diagnostics_record(hip, port):
    """what the reference's kernels report through device printf is counted in the context's hx_diag record:
    energy-budget factor (kernels.cu:455), negative fluxes and limited G functions under debug = 1 (:1458, :227),
    skipped Gauss points of the random-overlap re-binning (:3385)"""
    from helios_amd import phys_const as pc
    from helios_amd import synthetic as syn
    raw, ctx = hip.r, hip.r.ctx
    rng = np.random.default_rng(4)
    ctx.diag_reset()
    assert ctx.diag() == dict(negative_down_flux=0, negative_up_flux=0, g_limited=0, ro_rebin_skipped=0,
                              energy_correction=0.0, ro_fixup_passes=0)
    # energy correction
    nbin = 64
    star, dl = rng.uniform(1e5, 1e6, nbin), rng.uniform(1e-6, 1e-5, nbin)
    want = 5.6703669999999995e-5 * 5000.0 ** 4 / np.sum(dl * star)        # the kernels' own constant (kernels.cu:40)
    scaled = star.copy()
    raw.corr_inc_energy(np.zeros(nbin), scaled, dl, 1, nbin, 5000.0, 0)
    np.testing.assert_allclose(ctx.diag()["energy_correction"], want, rtol=1e-13)
    np.testing.assert_allclose(scaled, star * want, rtol=1e-13)
    # negative fluxes: bins with a negative source function
    ny, nb, ni = 4, 8, 6
    n = ny * nb * ni
    B = np.repeat(np.where(np.arange(nb) % 2 == 0, 1.0, -1.0), ni + 1) * rng.uniform(1.0, 2.0, nb * (ni + 1))
    Fd, Fu = np.zeros(n), np.zeros(n)
    args = [Fd, Fu, np.zeros(n), B, np.zeros(n), np.ones(n), np.zeros(n), np.full(n, 0.5), np.zeros(n), np.zeros(n),
            np.zeros(nb), np.zeros(nb * ni), 0.0, 0, pc.R_SUN, 0.05 * pc.AU, ni, nb, 0.25, -0.5, ny, 0.5, 0, 0, 0]
    raw.fband_iso(*(args + [0, 0.1]))
    d = ctx.diag()
    assert d["negative_down_flux"] == 0 and d["negative_up_flux"] == 0       # debug = 0: not counted
    raw.fband_iso(*(args + [1, 0.1]))
    d = ctx.diag()
    assert d["negative_down_flux"] == int(np.sum(Fd < 0)) > 0
    assert d["negative_up_flux"] == int(np.sum(Fu < 0)) > 0
    # limited G functions: w0 = 0.5 and mu_star = -1/sqrt(2) put the denominator of G+- at rounding level
    ny, nb, nl = 4, 8, 5
    n = ny * nb * nl
    ray = rng.uniform(1e-27, 1e-26, nb * nl)
    kap = np.repeat(ray.reshape(nl, nb), ny, axis=1).reshape(nl, nb * ny).copy()
    kap[:, ny * (nb // 2):] *= 10.0                                          # second half of the bins: harmless w0
    outs = [np.zeros(n) for _ in range(7)]
    dtc, w0 = np.zeros(nb * nl), np.zeros(n)
    raw.calc_trans_iso(*outs, rng.uniform(1.0, 2.0, nl), kap.reshape(-1), np.ones(nl), ray, np.zeros(nb * nl),
                       np.zeros(nb * nl), dtc, w0, np.zeros(nb * nl), np.zeros(ny * nb, np.int32), 0.0, 0.5, 0.5,
                       -1.0 / np.sqrt(2.0), 1.0 - 1e-10, 1e-3, 1, nb, ny, nl, 0, 0, 1, 0.1)
    Gp, Gm = outs[5], outs[6]
    nlim = int(np.sum(np.abs(Gp) >= 1e8) + np.sum(np.abs(Gm) >= 1e8))
    assert nlim > 0 and ctx.diag()["g_limited"] == nlim
    # random overlap: one quadrature weight holds 90 % of the measure -> several Gauss points per interval
    ny, nb, nlev = 20, 16, 3
    gy, _ = syn.gauss_points(ny)
    gw = np.full(ny, 0.2 / (ny - 1))
    gw[-1] = 1.8
    mix0 = np.sort(10.0 ** rng.uniform(-3, 0, (nlev, nb, ny)), axis=2)
    spec = np.sort(10.0 ** rng.uniform(-3, 0, (nlev, nb, ny)), axis=2) / (1e-3 * 18.0 / 2.3)
    res = []
    for impl in (hip, port):
        mix = mix0.reshape(-1).copy()
        impl.add_to_mixed_opac(np.full(nlev, 1e-3), spec.reshape(-1).copy(), mix, np.full(nlev, 2.3 * pc.AMU), gw, gy,
                               18.0 * pc.AMU, 1, 1, ny, nb, nlev)
        res.append(mix)
    np.testing.assert_allclose(res[0], res[1], rtol=1e-12)
    assert ctx.diag()["ro_rebin_skipped"] > 0
    ctx.diag_reset()
    assert ctx.diag()["ro_rebin_skipped"] == 0 and ctx.diag()["energy_correction"] == 0.0


def test_integration_stub_from_the_docs_runs():
    """the PyCUDA-replacement stub printed in INTEGRATION.md section 1, executed as written (library path aside): a
    reference-style launch `f(args..., block=..., grid=...)` of temp_inter and planck_interpol_interface"""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"## 1\. The stub.*?```python\n(.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libhelios_hip.so")',
                        'ctypes.CDLL(%r)' % os.path.join(root, "helios_amd", "libhelios_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    ns["_lib"].hx_last_error.restype = __import__("ctypes").c_char_p
    mod = ns["_Module"]()
    L = 7
    T_lay = np.linspace(900.0, 400.0, L + 1)
    d_T, d_Ti = ns["to_gpu"](T_lay), ns["mem_alloc"]((L + 1) * 8)
    d_Ti.dtype, d_Ti.shape = np.float64, (L + 1,)
    mod.get_function("temp_inter")(d_T, d_Ti, np.int32(L + 1), np.int32(0), block=(16, 1, 1), grid=(1, 1, 1))
    T_int = d_Ti.get()
    want = np.empty(L + 1)
    want[1:L] = T_lay[:L - 1] + 0.5 * (T_lay[1:L] - T_lay[:L - 1])
    want[0] = T_lay[0] - 0.5 * (T_lay[1] - T_lay[0])
    want[L] = T_lay[L - 1] + 0.5 * (T_lay[L - 1] - T_lay[L - 2])
    np.testing.assert_allclose(T_int, want, rtol=1e-15)


@pytest.mark.parametrize("nbin,dim,step", [(10000, 8000, 2), (386, 8000, 2), (63, 400, 10), (64, 400, 10), (1, 400, 10)])
def test_planck_table_is_the_plain_series_bit_for_bit(nbin, dim, step):
    """k_plancktable (round 5: Phi_n once per bin edge and shared between neighbouring lanes, divisions through correctly
    rounded reciprocals with one fma correction, the series left where no later term can change the sum) against the
    reference's formula evaluated as it is written -- one thread per entry, 199 terms, two exp and eight divisions each
    (kernels.cu:95-105, :362-416) -- on this GPU: every entry of the table the SAME BITS, at BASELINE's full size (8001 rows x
    10 000 bins), on the reference's default grid and on grids that end inside / at / after a wavefront's 63 bins"""
    import ctypes
    from helios_amd import _lib
    from helios_amd import synthetic as syn
    from helios_amd.device import Context
    ctx = Context(0)
    try:
        raw = ctypes.CDLL(_lib.LIB_PATH)
        plain = raw.hx_internal_plancktable_plain
        plain.restype = ctypes.c_int
        P = ctypes.POINTER(ctypes.c_double)
        plain.argtypes = [ctypes.c_void_p, P, P, P, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        edge, _, dl = syn.wavelength_grid(nbin)
        if nbin == 386:            # a grid whose wavelengths DEcrease: the reference swaps the two edges (kernels.cu:399-403)
            edge, dl = edge[::-1].copy(), np.abs(dl[::-1]).copy()
        d_edge, d_dl = ctx.to_gpu(edge), ctx.to_gpu(dl)
        a, b = ctx.zeros((dim + 1) * nbin), ctx.zeros((dim + 1) * nbin)
        ctx.check(_lib.lib().hx_plancktable(ctx.handle, a.d, d_edge.d, d_dl.d, nbin, 5777.0, dim, step), "hx_plancktable")
        ctx.check(plain(ctx.handle, b.d, d_edge.d, d_dl.d, nbin, 5777.0, dim, step), "hx_internal_plancktable_plain")
        got, want = a.get(), b.get()
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), \
            "%d of %d entries differ, largest relative difference %.3e" % (
                int((got != want).sum()), got.size, float(np.nanmax(np.abs(got - want) / np.maximum(np.abs(want), 1e-300))))
        assert np.isfinite(got).all() and (got[-nbin:] > 0).all()     # (the stellar row; cold rows underflow at short wavelengths)
    finally:
        ctx.close()
