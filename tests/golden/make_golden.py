"""Generates the golden vectors under tests/golden/ from the REFERENCE's own kernels
(oracle/_ref/libhelios_ref.so = /root/reference/source/kernels.cu compiled for the host).

Run in the build container only (the reference tree is not present on the GPU box):

    python tests/golden/make_golden.py

A fixture is data: the seeded inputs of one small problem and the arrays the reference produces from
them.  Host constants (astropy-unpinned in the reference) are recorded in `meta`.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from impls import RefImpl  # noqa: E402
from helios_amd import phys_const as pc  # noqa: E402
from helios_amd import synthetic as syn  # noqa: E402

CHAIN_CONFIGS = {
    "default": dict(),
    "noscat": dict(scat=0),
    "dirbeam_albedo": dict(dir_beam=1, albedo=0.3),
    "dirbeam_zenith": dict(dir_beam=1, geom_zenith_corr=1, zenith_deg=80.0),
    "clouds_g0_i2s": dict(clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2),
    "iso": dict(iso=1),
    "iso_clouds": dict(iso=1, clouds=1, scat_corr=1, dir_beam=1),
    "thin_top": dict(thin_top=True, nlayer=8),
    "ny1": dict(ny=1, nbin=9),
}
SIZE = dict(nbin=5, nlayer=5, ntemp=5, npress=4, plancktable_dim=300, plancktable_step=10)

COEF_KEYS_NONISO = [a + b for a in ("trans_wg_", "delta_tau_wg_", "M_", "N_", "P_", "G_plus_", "G_minus_",
                                    "w_0_", "delta_tau_all_clouds_") for b in ("upper", "lower")]
COEF_KEYS_ISO = ["trans_wg", "delta_tau_wg", "M_term", "N_term", "P_term", "G_plus", "G_minus", "w_0",
                 "delta_tau_all_clouds"]
STATE_KEYS = ["planckband_lay", "planckband_int", "opac_wg_lay", "opac_wg_int", "scat_cross_lay",
              "scat_cross_int", "meanmolmass_lay", "meanmolmass_int", "delta_z_lay", "g_0_tot_lay",
              "g_0_tot_int", "scat_trigger", "F_dir_wg", "Fc_dir_wg", "F_down_wg", "F_up_wg",
              "Fc_down_wg", "Fc_up_wg", "F_dir_band", "F_down_band", "F_up_band", "F_down_tot",
              "F_up_tot", "F_net", "F_net_diff", "abort", "T_store", "deltat_prefactor"]


def meta():
    return json.dumps(dict(
        generator="tests/golden/make_golden.py",
        source="oracle/_ref/libhelios_ref.so (reference source/kernels.cu compiled as host C++)",
        host_constants={k: getattr(pc, k) for k in ("AU", "R_SUN", "R_JUP", "SIGMA_SB", "AMU", "R_UNIV")},
        numpy=np.__version__))


def snapshot(c, s, keys):
    out = {}
    for k in keys:
        out[k] = s[k].copy()
    out["T_lay"] = c.T_lay.copy()
    out["T_int"] = c.T_int.copy()
    out["z_lay"] = c.z_lay.copy()
    return out


def chain_fixture(name, cfg):
    kw = dict(SIZE)
    kw.update(cfg)
    c0 = cases.make_case(**kw)
    ref = RefImpl()
    c = c0.copy()
    s = cases.alloc_state(c)
    cases.setup_planck(ref, c, s)
    planck_grid = s.planck_grid.copy()
    coef = COEF_KEYS_ISO if c.iso == 1 else COEF_KEYS_NONISO
    data = {"meta": meta(), "config": json.dumps(kw)}
    for k, v in c0.items():
        data["in." + k] = np.asarray(v)
    data["planck_grid"] = planck_grid
    cases.radiation_iterations(ref, c, s, 1)
    for k, v in snapshot(c, s, STATE_KEYS + coef).items():
        data["it1." + k] = v
    cases.radiation_iterations(ref, c, s, 11, start=1)
    for k, v in snapshot(c, s, STATE_KEYS).items():
        data["it12." + k] = v
    np.savez_compressed(os.path.join(HERE, "chain_%s.npz" % name), **data)


def mixing_fixture():
    """species interpolation + k-coefficient mixing: every branch of add_to_mixed_opac"""
    ref = RefImpl()
    rng = np.random.default_rng(20243)
    nbin, nlev, ny, ntemp, npress = 6, 8, 20, 5, 4
    gy, gw = syn.gauss_points(ny)
    ktemp, kpress = syn.tp_grid(ntemp, npress)
    _, wave, _ = syn.wavelength_grid(nbin)
    temp = np.array([50.0, ktemp[0], ktemp[1], 0.5 * (ktemp[1] + ktemp[2]), ktemp[-1], 5000.0, 777.0, 1234.5])
    press = np.array([0.1, kpress[0], kpress[1], 3e4, kpress[-1], 1e11, kpress[2], 5e6])
    data = dict(meta=meta(), gauss_y=gy, gauss_weight=gw, ktemp=ktemp, kpress=kpress, temp=temp,
                press=press, wave=wave, dims=np.array([nbin, nlev, ny, ntemp, npress]))
    n = ny * nbin * nlev
    mmm = np.full(nlev, 2.3 * pc.AMU)
    data["meanmolmass"] = mmm
    # (a) a 4-species loop, RO
    weights = np.array([2.0, 18.0, 44.0, 16.0])
    vmrs = np.array([0.85, 1e-3, 3e-4, 1e-5])
    mix, spec = np.zeros(n), np.zeros(n)
    for s in range(4):
        tab = syn.ktable(rng, nbin, ny, ktemp, kpress, gy)
        data["loop.tab%d" % s] = tab
        ref.opac_species_interpol(temp, ktemp, press, kpress, tab, spec, npress, ntemp, ny, nbin, nlev)
        data["loop.spec%d" % s] = spec.copy()
        ref.add_to_mixed_opac(np.full(nlev, vmrs[s]), spec, mix, mmm, gw, gy, weights[s] * pc.AMU, s, 1,
                              ny, nbin, nlev)
        data["loop.mix%d" % s] = mix.copy()
    data["loop.weights"], data["loop.vmrs"] = weights, vmrs
    # (b) crafted branch cases
    fac = 1e-3 * 18.0 / 2.3
    y = np.arange(ny)
    base = np.sort(10.0 ** rng.uniform(-4, 0, (nlev, nbin, ny)), axis=2)
    other = np.sort(10.0 ** rng.uniform(-4, 0, (nlev, nbin, ny)), axis=2)
    branch = {
        "s0": (base, other, 0, 1), "corrk": (base, other, 1, 0),
        "neg_new": (base, other * 1e-9, 1, 1), "neg_mix": (base * 1e-9, other, 1, 1),
        "ro_nocross": (base, np.sort(base * 0.5 / fac * (1 + 0.1 * rng.uniform(size=base.shape)), axis=2), 1, 1),
        "ro_onecross": (np.broadcast_to(10.0 ** (-3 + 3 * y / (ny - 1.0)), base.shape).copy(),
                        np.broadcast_to(10.0 ** (-2 + 1 * y / (ny - 1.0)), base.shape).copy() / fac, 1, 1),
        "ro_multicross": (base, other / fac, 1, 1),
        "ro_ties": (np.round(base, 2) + 0.01, (np.round(other, 2) + 0.01) / fac, 1, 1),
    }
    for k, (b, o, s, ro) in branch.items():
        mix = b.reshape(-1).copy()
        ref.add_to_mixed_opac(np.full(nlev, 1e-3), o.reshape(-1).copy(), mix, mmm, gw, gy, 18.0 * pc.AMU,
                              s, ro, ny, nbin, nlev)
        data["br.%s.mix_in" % k] = b.reshape(-1)
        data["br.%s.spec" % k] = o.reshape(-1)
        data["br.%s.s_ro" % k] = np.array([s, ro])
        data["br.%s.mix_out" % k] = mix
    # (c) scattering helpers
    t2 = rng.uniform(200, 2000, nlev)
    p2 = 10.0 ** rng.uniform(0, 9.5, nlev)
    v2 = 10.0 ** rng.uniform(-6, -1, nlev)
    h2o = np.zeros(nbin * nlev)
    ref.calc_h2o_scat(t2, p2, wave, h2o, v2, 18.0 * pc.AMU, nbin, nlev)
    tot = h2o * 0.3
    ref.add_to_mixed_scat(v2, h2o, tot, nbin, nlev)
    data.update({"sc.temp": t2, "sc.press": p2, "sc.vmr": v2, "sc.h2o": h2o, "sc.total": tot})
    np.savez_compressed(os.path.join(HERE, "mixing.npz"), **data)


MATRIX_NAMES = ["default", "dirbeam_albedo", "clouds_g0_i2s", "iso_clouds", "thin_top"]


def matrix_state(name):
    """inputs of the tridiagonal flux solve = the coefficient planes of chain_<name>.npz after iteration 1"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_checks as gc
    c, z = gc.load_chain(name)
    s = cases.alloc_state(c)
    for k in z.files:
        if k.startswith("it1.") and k[4:] in s and k[4:] not in ("F_down_wg", "F_up_wg", "Fc_down_wg", "Fc_up_wg"):
            s[k[4:]][...] = z[k]
    c.surf_albedo = np.maximum(c.surf_albedo, 1e-8)     # reader's lower bound for this method (read.py:1261)
    if name == "default":
        s.scat_trigger[::2] = 0                          # half of the points through the pure-absorption branch
    return c, s


def matrix_fixture():
    ref = RefImpl()
    data = {"meta": meta()}
    for name in MATRIX_NAMES:
        c, s = matrix_state(name)
        m = cases.flux_matrix(ref, c, s)
        for k in ("F_down_wg", "F_up_wg") + (("Fc_down_wg", "Fc_up_wg") if c.iso == 0 else ()):
            data["%s.%s" % (name, k)] = s[k].copy()
        for k, v in m.items():
            data["%s.%s" % (name, k)] = v.copy()
    np.savez_compressed(os.path.join(HERE, "matrix.npz"), **data)


if __name__ == "__main__":
    if sys.argv[1:] == ["matrix"]:
        matrix_fixture()
        sys.exit(0)
    for name, cfg in CHAIN_CONFIGS.items():
        chain_fixture(name, cfg)
    mixing_fixture()
    matrix_fixture()
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("wrote fixtures, %.1f KB total" % (tot / 1024.0))
