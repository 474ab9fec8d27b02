"""A6 on the device: calculate_vmr_for_all_species / interpolate_grid_to_lay_or_int (source/host_functions.py:874-910) as
part of the fused refresh -- a species hands over its (T, P) mixing-ratio table once (hx_rt_set_species_vmr_table) and its
profile follows the temperatures on the device.  Held to the reference's own values (tests/golden/host_functions.npz,
generated from the reference's imported Python with scipy's RectBivariateSpline) and to the host-driven refresh it replaces."""
import importlib.util
import os

import numpy as np
import pytest

import cases
from helios_amd import host_functions as hs
from helios_amd import synthetic as syn

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx():
    from helios_amd.device import Context
    return Context(0)


def _mk():
    spec = importlib.util.spec_from_file_location("make_host_golden", os.path.join(HERE, "golden", "make_host_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk


def test_vmr_profiles_from_tables_match_the_reference_interpolation(ctx):
    """the six species of the host golden (profile beyond the table at both ends, nodes hit exactly) through
    k_rt_species_prep: layer profiles against the reference's scipy values, interface profiles and the mean molecular
    mass against host_functions.py (itself pinned to them) with the device's interface temperatures"""
    from helios_amd.rt import batch_from_case
    q = _mk().mixing_state()
    Z = np.load(os.path.join(HERE, "golden", "host_functions.npz"), allow_pickle=False)
    L = int(q.nlayer)
    c = cases.make_case(nbin=5, nlayer=L, ntemp=len(q.ktemp), npress=len(q.log_kpress))
    c.ktemp, c.kpress = q.ktemp.copy(), 10.0 ** q.log_kpress
    c.p_lay, c.p_int = 10.0 ** q.log_p_lay, 10.0 ** q.log_p_int
    c.T_lay = np.append(q.T_prof_lay, 1234.0)
    S = len(q.species_list)
    rt = batch_from_case(ctx, c, ncol=1, nspecies=S)
    try:
        rng = np.random.default_rng(4)
        for s, sp in enumerate(q.species_list):
            pretab = syn.ktable(rng, c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y) if s == 0 else None
            rt.set_species(s, pretab, None, sp.weight, is_cia=1 if "CIA" in sp.name else 0,
                           in_mu=1 if hs._counts_for_mu(sp) else 0)
            rt.set_species_vmr_table(s, sp.vmr_pretab.reshape(-1))
        rt.set_column_vmr(-1, np.full((S, L), -1.0), np.full((S, L + 1), -1.0))   # must all be overwritten
        rt.build_planck_table(0)
        rt.step(0, step_temperature=False)
        vl, vi = rt.get("vmr_lay").reshape(S, L + 1), rt.get("vmr_int").reshape(S, L + 1)
        T_int = rt.get("T_int")
        for n, sp in enumerate(q.species_list):
            # the reference itself (scipy, bilinear spline evaluated point by point)
            np.testing.assert_allclose(vl[n, :L], Z["mix.vmr_lay.%d" % n], rtol=1e-11)
            want = hs.interpolate_grid_to_lay_or_int(q.log_kpress, q.ktemp, sp.vmr_pretab, q.log_p_int, T_int)
            np.testing.assert_allclose(vi[n], want, rtol=1e-11)
            sp.vmr_layer, sp.vmr_interface = vl[n, :L], vi[n]
        np.testing.assert_allclose(rt.get("meanmolmass_lay"), hs.calc_meanmolmass(q, type="layer"), rtol=1e-13)
        np.testing.assert_allclose(rt.get("meanmolmass_int"), hs.calc_meanmolmass(q, type="interface"), rtol=1e-13)
        # back to host-given profiles
        for s in range(S):
            rt.set_species_vmr_table(s, None)
        rt.set_column_vmr(-1, np.full((S, L), 0.1), np.full((S, L + 1), 0.1))
        rt.step(0, step_temperature=False)
        assert np.all(rt.get("vmr_lay").reshape(S, L + 1)[:, :L] == 0.1)
    finally:
        rt.close()


def _host_profiles(c, tabs, T_lay):
    """what Compute._push_vmr used to send: numpy interpolation along the present profile"""
    L = c.nlayer
    T_int = np.empty(L + 1)
    T_int[1:L] = T_lay[:L - 1] + 0.5 * (T_lay[1:L] - T_lay[:L - 1])
    T_int[0] = T_lay[0] - 0.5 * (T_lay[1] - T_lay[0])
    T_int[L] = T_lay[L - 1] + 0.5 * (T_lay[L - 1] - T_lay[L - 2])
    lk = np.log10(c.kpress)
    vl = np.array([hs.interpolate_grid_to_lay_or_int(lk, c.ktemp, t, np.log10(c.p_lay), T_lay[:L]) for t in tabs])
    vi = np.array([hs.interpolate_grid_to_lay_or_int(lk, c.ktemp, t, np.log10(c.p_int), T_int) for t in tabs])
    return vl, vi


@pytest.mark.parametrize("loop", ["radiation", "convection"])
def test_a_sweep_of_columns_with_tabulated_chemistry_needs_no_host_step(ctx, loop):
    """eight columns whose absorbers' mixing ratios depend on (T, P) -- FastChem-style tables -- iterated (a) with the
    tables on the device, one call for 31 iterations, and (b) the way the host used to drive it: stop at every refresh,
    read the temperatures, interpolate with numpy, upload the profiles.  Same temperatures, fluxes and profiles."""
    from helios_amd.rt import batch_from_case
    ncol, n_iter = 8, 31
    c = cases.add_species(cases.make_case(nbin=12, nlayer=14), nspecies=4)
    S = len(c.species)
    rng = np.random.default_rng(8)
    lt = (c.ktemp[None, :, None] - c.ktemp[0]) / (c.ktemp[-1] - c.ktemp[0])
    lp = np.log10(c.kpress)[None, None, :] / 9.0
    # smooth tables with strong gradients: two decades over the temperature range, one over the pressure range
    tabs = 10.0 ** (np.array([np.log10(sp["vmr"]) for sp in c.species])[:, None, None]
                    + rng.uniform(-2, 2, (S, 1, 1)) * lt + rng.uniform(-1, 1, (S, 1, 1)) * lp)
    tabs[0] = 0.8
    T0 = [c.T_lay * (1.0 + 0.03 * k) for k in range(ncol)]
    kap = 2.0 / 7.0

    def make():
        rt = batch_from_case(ctx, c, ncol=ncol, nspecies=S)
        for k, sp in enumerate(c.species):
            rt.set_species(k, sp["pretab"], sp["scat"], sp["weight"], is_h2o=2 if sp["is_h2o"] else 0,
                           is_cia=1 if sp["is_cia"] else 0, in_mu=0 if sp["is_cia"] else 1)
        for k in range(ncol):
            rt.set_temperatures(k, T0[k])
        if loop == "convection":
            from helios_amd import phys_const as pc
            L = c.nlayer
            for name, v in (("kappa_lay", np.full(L, kap)), ("kappa_int", np.full(L + 1, kap)),
                            ("c_p_lay", np.full(L, pc.R_UNIV / kap)), ("conv_layer", np.zeros(L + 1, np.int32)),
                            ("conv_unstable", np.zeros(L + 1, np.int32)), ("dampara", np.array([-1.0]))):
                rt.set_state(-1, name, v)
        rt.build_planck_table(1)
        return rt

    keys = ("T_lay", "F_net", "F_up_band", "meanmolmass_lay", "vmr_lay", "vmr_int", "scat_cross_lay")
    a = make()
    try:
        for k in range(S):
            a.set_species_vmr_table(k, tabs[k].reshape(-1))
        a.set_column_vmr(-1, np.zeros((S, c.nlayer)), np.zeros((S, c.nlayer + 1)))
        (a.run if loop == "radiation" else a.conv_run)(0, n_iter)
        got = [{k: a.get(k, col) for k in keys} for col in range(ncol)]
    finally:
        a.close()
    b = make()
    try:
        for it in range(0, n_iter, 10):
            n = min(10, n_iter - it)
            if loop == "radiation":
                for col in range(ncol):
                    b.set_column_vmr(col, *_host_profiles(c, tabs, b.get("T_lay", col)))
                b.run(it, n)
            else:   # computation.py:1030-1036 and :1056-1061: before the adjustment, and for the adjusted profile
                for col in range(ncol):
                    b.set_column_vmr(col, *_host_profiles(c, tabs, b.get("T_lay", col)))
                b.conv_adjust(it)
                for col in range(ncol):
                    b.set_column_vmr(col, *_host_profiles(c, tabs, b.get("T_lay", col)))
                b.conv_advance(it)
                b.conv_run(it + 1, n - 1)
        want = [{k: b.get(k, col) for k in keys} for col in range(ncol)]
    finally:
        b.close()
    for col in range(ncol):
        for k in keys:
            scale = np.abs(want[col][k]).max()
            np.testing.assert_allclose(got[col][k], want[col][k], rtol=1e-9, atol=1e-12 * scale, err_msg="%s column %d" % (k, col))
    # the columns really differ, and the profiles really moved with the temperatures
    assert np.abs(got[0]["T_lay"] - got[7]["T_lay"]).max() > 10.0
    assert np.abs(got[0]["vmr_lay"] - got[7]["vmr_lay"]).max() > 1e-6


@pytest.mark.parametrize("loop", ["radiation", "convection"])
def test_columns_of_one_batch_keep_their_own_chemistry_tables(ctx, loop):
    """a sweep over FastChem directories (metallicity, C/O): every column of a batch comes with its OWN (T, P) mixing-ratio
    tables (hx_rt_set_column_vmr_table).  Three columns with the same start profile and different tables in one batch give,
    column by column, what each gives alone with its table -- and not what column 0's table gives."""
    from helios_amd.rt import batch_from_case
    ncol, n_iter = 3, 21
    c = cases.add_species(cases.make_case(nbin=12, nlayer=14), nspecies=3)
    S = len(c.species)
    rng = np.random.default_rng(11)
    lt = (c.ktemp[None, :, None] - c.ktemp[0]) / (c.ktemp[-1] - c.ktemp[0])
    lp = np.log10(c.kpress)[None, None, :] / 9.0
    base = np.array([np.log10(sp["vmr"]) for sp in c.species])[:, None, None]
    tabs = [10.0 ** (base + 0.7 * col + rng.uniform(-2, 2, (S, 1, 1)) * lt + rng.uniform(-1, 1, (S, 1, 1)) * lp)
            for col in range(ncol)]           # "metallicity" rises with the column
    for t in tabs:
        t[0] = 0.8

    def make(n):
        rt = batch_from_case(ctx, c, ncol=n, nspecies=S)
        for k, sp in enumerate(c.species):
            rt.set_species(k, sp["pretab"], sp["scat"], sp["weight"], is_h2o=2 if sp["is_h2o"] else 0,
                           is_cia=1 if sp["is_cia"] else 0, in_mu=0 if sp["is_cia"] else 1)
        rt.set_column_vmr(-1, np.zeros((S, c.nlayer)), np.zeros((S, c.nlayer + 1)))
        if loop == "convection":    # (k_rt_mmm_from_vmr reads the column's table ahead of the adjustment, computation.py:1030-1036)
            from helios_amd import phys_const as pc
            L, kap = c.nlayer, 2.0 / 7.0
            for name, v in (("kappa_lay", np.full(L, kap)), ("kappa_int", np.full(L + 1, kap)),
                            ("c_p_lay", np.full(L, pc.R_UNIV / kap)), ("conv_layer", np.zeros(L + 1, np.int32)),
                            ("conv_unstable", np.zeros(L + 1, np.int32)), ("dampara", np.array([-1.0]))):
                rt.set_state(-1, name, v)
        rt.build_planck_table(1)
        return rt

    def run(rt):
        (rt.run if loop == "radiation" else rt.conv_run)(0, n_iter)

    keys = ("T_lay", "F_net", "F_up_band", "meanmolmass_lay", "vmr_lay", "vmr_int")
    a = make(ncol)
    try:
        for col in range(ncol):
            for k in range(S):
                a.set_column_vmr_table(col, k, tabs[col][k].reshape(-1))
        run(a)
        got = [{k: a.get(k, col) for k in keys} for col in range(ncol)]
    finally:
        a.close()
    for col in range(ncol):
        b = make(1)
        try:
            for k in range(S):
                b.set_species_vmr_table(k, tabs[col][k].reshape(-1))
            run(b)
            for k in keys:
                np.testing.assert_array_equal(got[col][k], b.get(k, 0), err_msg="%s column %d" % (k, col))
        finally:
            b.close()
    assert np.abs(got[0]["vmr_lay"] / np.maximum(got[2]["vmr_lay"], 1e-300) - 1.0).max() > 0.5
    assert np.abs(got[0]["T_lay"] - got[2]["T_lay"]).max() > 1e-3
