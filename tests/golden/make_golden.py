"""Generates the golden vectors under tests/golden/ from the REFERENCE's own kernels.

    --backend gfx950  (the pin)  oracle/_ref/libhelios_ref_gfx950.so = /root/reference/source/kernels.cu
                      compiled UNMODIFIED by hipcc for gfx950 (ROCm's own headers, no shim) and run
                      on the MI355X as real GPU threads with the block/grid of source/computation.py.
                      Run on the GPU box (the prebuilt library travels there, the reference tree
                      does not):
                          gpurun -- python tests/golden/make_golden.py --backend gfx950 --out gpurun_out/golden
                      then copy gpurun_out/golden/*.npz to tests/golden/ and commit.
    --backend host    oracle/_ref/libhelios_ref.so = the same file parsed by g++ through
                      oracle/ref_shim.h (CPU only; kept as a cross-check of the shim, not a pin)
    --compare DIR     compare the fixtures in DIR with those in tests/golden/ key by key and print
                      the largest relative difference per file (how far two builds of the reference
                      are from each other)

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


BACKEND = "gfx950"
OUT = HERE
SOURCES = {
    "gfx950": "oracle/_ref/libhelios_ref_gfx950.so: reference source/kernels.cu compiled unmodified by hipcc "
              "(ROCm 7.2, -x hip --offload-arch=gfx950 -O2 -include hip/hip_runtime.h) and run on an MI355X",
    "host": "oracle/_ref/libhelios_ref.so (reference source/kernels.cu compiled as host C++ through oracle/ref_shim.h)",
}


def reference():
    import oracle
    lib = oracle.refgpu if BACKEND == "gfx950" else oracle.ref
    if lib is None:
        raise SystemExit("the %s build of the reference is not available here" % BACKEND)
    return RefImpl(lib)


def meta():
    return json.dumps(dict(
        generator="tests/golden/make_golden.py --backend " + BACKEND,
        source=SOURCES[BACKEND],
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
    ref = reference()
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
    np.savez_compressed(os.path.join(OUT, "chain_%s.npz" % name), **data)


def mixing_fixture():
    """species interpolation + k-coefficient mixing: every branch of add_to_mixed_opac"""
    ref = reference()
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
    np.savez_compressed(os.path.join(OUT, "mixing.npz"), **data)


MATRIX_NAMES = ["default", "dirbeam_albedo", "clouds_g0_i2s", "iso_clouds", "thin_top"]


def matrix_state(name):
    """inputs of the tridiagonal flux solve = the coefficient planes of chain_<name>.npz after iteration 1"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_checks as gc
    c, z = gc.load_chain(name, OUT)
    s = cases.alloc_state(c)
    for k in z.files:
        if k.startswith("it1.") and k[4:] in s and k[4:] not in ("F_down_wg", "F_up_wg", "Fc_down_wg", "Fc_up_wg"):
            s[k[4:]][...] = z[k]
    c.surf_albedo = np.maximum(c.surf_albedo, 1e-8)     # reader's lower bound for this method (read.py:1261)
    if name == "default":
        s.scat_trigger[::2] = 0                          # half of the points through the pure-absorption branch
    return c, s


def matrix_fixture():
    ref = reference()
    data = {"meta": meta()}
    for name in MATRIX_NAMES:
        c, s = matrix_state(name)
        m = cases.flux_matrix(ref, c, s)
        for k in ("F_down_wg", "F_up_wg") + (("Fc_down_wg", "Fc_up_wg") if c.iso == 0 else ()):
            data["%s.%s" % (name, k)] = s[k].copy()
        for k, v in m.items():
            data["%s.%s" % (name, k)] = v.copy()
    np.savez_compressed(os.path.join(OUT, "matrix.npz"), **data)


# ---- columns large enough for the fused path's k = 16 / k = 32 tilings (64 bins x 100 / 200 layers) ------------
BIG_CONFIGS = {
    "default": dict(nbin=64, nlayer=100),
    "clouds_beam": dict(nbin=64, nlayer=100, clouds=1, g_0=0.3, scat_corr=1, dir_beam=1, albedo=0.2),
    "L200_beam": dict(nbin=32, nlayer=200, dir_beam=1, albedo=0.1),
}
BIG_SIZE = dict(ntemp=5, npress=4, plancktable_dim=300, plancktable_step=10)
BIG_KEYS = ["F_up_band", "F_down_band", "F_dir_band", "F_up_tot", "F_down_tot", "F_net", "planckband_lay",
            "planckband_int", "scat_cross_lay", "scat_cross_int", "meanmolmass_lay", "meanmolmass_int",
            "delta_z_lay", "abort", "T_store", "deltat_prefactor"]
BIG_WG_KEYS = ["F_up_wg", "F_down_wg", "Fc_up_wg", "Fc_down_wg", "F_dir_wg", "opac_wg_lay", "opac_wg_int"]


def big_bins(nbin):
    return np.array(sorted(set([0, 1, nbin // 3, nbin // 2, nbin - 2, nbin - 1])))


def wg_sample(c, a):
    """every Gauss point and level of a few bins of a [y + ny*x + ny*nbin*i] array"""
    a = a.reshape(-1, c.nbin, c.ny)
    return a[:, big_bins(c.nbin), :].copy()


def big_fixture(name, cfg):
    kw = dict(BIG_SIZE)
    kw.update(cfg)
    c0 = cases.make_case(**kw)
    ref = reference()
    c = c0.copy()
    s = cases.alloc_state(c)
    cases.setup_planck(ref, c, s)
    data = {"meta": meta(), "config": json.dumps(kw), "planck_grid": s.planck_grid.copy(),
            "sample_bins": big_bins(c.nbin)}
    for k, v in c0.items():
        data["in." + k] = np.asarray(v)

    def snap(prefix):
        for k in BIG_KEYS:
            data[prefix + k] = s[k].copy()
        for k in BIG_WG_KEYS:
            data[prefix + k + ".sample"] = wg_sample(c, s[k])
        data[prefix + "T_lay"], data[prefix + "T_int"], data[prefix + "z_lay"] = c.T_lay.copy(), c.T_int.copy(), c.z_lay.copy()

    cases.radiation_iterations(ref, c, s, 1)
    snap("it1.")
    cases.radiation_iterations(ref, c, s, 11, start=1)
    snap("it12.")
    np.savez_compressed(os.path.join(OUT, "big_%s.npz" % name), **data)


# ---- whole loops: SURVEY 8(c) "after N = 1, 10, 11, 50 iterations and at convergence" -------------------------------
def loop_fixture(name):
    import loop_driver as ld
    c0, relax = ld.loop_case(name)
    ref = reference()
    c = c0.copy()
    s = cases.alloc_state(c)
    cases.setup_planck(ref, c, s)
    data = {"meta": meta(), "planck_grid": s.planck_grid.copy(), "crit_relaxation_numbers": np.array(relax, np.int64)}
    for k, v in c0.items():
        if k != "species":
            data["in." + k] = np.asarray(v)
    for i, sp in enumerate(c0.get("species") or []):     # the species list as plain arrays
        for k, v in sp.items():
            if v is not None:
                data["species.%d.%s" % (i, k)] = np.asarray(v)
    n, snaps, reason = ld.radiation_loop(ref, c, s, ld.SNAP_AT, crit_relaxation_numbers=relax,
                                         refresh=ld.loop_refresh(c))
    data["iter_count"] = np.array(n)
    data["reason"] = np.array(reason)
    for at, sn in snaps.items():
        for k, v in sn.items():
            data["%s.%s" % (("it%d" % at) if at != "end" else "end", k)] = v
    np.savez_compressed(os.path.join(OUT, "loop_%s.npz" % name), **data)
    print("loop_%s: %d iterations (%s)" % (name, n, reason))


def loopconv_fixture(name, hostref=False):
    """radiation loop + convection loop (source/computation.py:827-1174) of a small column: iteration counts,
    conv_layer / conv_unstable / marked_red, T_lay, F_net, totals and TOA spectrum after 1, 10, 11, 50, 400 iterations
    of the convection loop and at its end.  Kernels: the reference's (this BACKEND).  Host steps between the kernels
    (conv_check, mark_convective_layers, conv_correct, check_for_radiative_eq, source/host_functions.py:251-635):
    helios_amd/host_functions.py, pinned function by function to the reference's Python -- or, with `hostref`, the
    reference's own source/host_functions.py imported in the build container (tests/golden/make_host_golden.py
    explains the stand-ins for pycuda / astropy, which none of these functions touch)."""
    import contextlib
    import io
    import loop_driver as ld
    if hostref:
        import make_host_golden as mh
        hs, _ = mh.import_reference()
        host = "the reference's own source/host_functions.py (imported in the build container)"
    else:
        from helios_amd import host_functions as hs
        host = "helios_amd/host_functions.py (pinned to the reference's Python by tests/test_host_golden.py)"
    c0, kappa, radiative_first = ld.conv_case(name)
    ref = reference()
    c = c0.copy()
    s = cases.alloc_state(c)
    cases.setup_planck(ref, c, s)
    m = json.loads(meta())
    m["host_steps"] = host
    data = {"meta": json.dumps(m), "planck_grid": s.planck_grid.copy(), "kappa": np.array(kappa),
            "radiative_first": np.array(int(radiative_first))}
    for k, v in c0.items():
        if k != "species":
            data["in." + k] = np.asarray(v)
    for i, sp in enumerate(c0.get("species") or []):     # the species list as plain arrays
        for k, v in sp.items():
            if v is not None:
                data["species.%d.%s" % (i, k)] = np.asarray(v)
    with contextlib.redirect_stdout(io.StringIO()):       # check_for_radiative_eq prints every 100th iteration
        n_rad, n, snaps, q = ld.rad_conv_run(ref, hs, c, s, kappa, radiative_first)
    data["rad_iter_count"] = np.array(n_rad)
    data["rad_reason"] = np.array(q.rad_reason)
    data["iter_count"] = np.array(n)
    for at, sn in snaps.items():
        for k, v in sn.items():
            data["%s.%s" % (("it%d" % at) if isinstance(at, int) else at, k)] = v
    fn = "loopconv_%s%s.npz" % (name, "_hostref" if hostref else "")
    np.savez_compressed(os.path.join(OUT, fn), **data)
    print("%s: radiation loop %d (%s), convection loop %d iterations, conv_layer %s"
          % (fn, n_rad, q.rad_reason, n, "".join(str(int(v)) for v in q.conv_layer)), flush=True)


def compare(other):
    """largest relative difference per file between the fixtures in `other` and those next to this script"""
    worst_all = 0.0
    for f in sorted(os.listdir(other)):
        if not f.endswith(".npz") or not os.path.exists(os.path.join(HERE, f)) or f == "host_functions.npz":
            continue
        a, b = np.load(os.path.join(other, f)), np.load(os.path.join(HERE, f))
        worst, where = 0.0, ""
        for k in a.files:
            if k not in b.files or a[k].dtype.kind not in "fi" or k.startswith("in.") or a[k].shape != b[k].shape:
                continue
            x, y = a[k].astype(float), b[k].astype(float)
            scale = np.abs(y).max()
            if scale == 0:
                continue
            with np.errstate(invalid="ignore"):
                d = np.nanmax(np.abs(x - y)) / scale       # relative to the array's largest entry
            if d > worst:
                worst, where = d, k
        worst_all = max(worst_all, worst)
        print("%-28s max |a-b| / max|b| = %.3e  (%s)" % (f, worst, where))
    print("largest over all files: %.3e" % worst_all)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["gfx950", "host"], default="gfx950")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default="")
    ap.add_argument("--compare", default="")
    ap.add_argument("--loop-names", default="", help="comma-separated subset of the loop fixtures")
    ap.add_argument("--hostref", default="", help="build container only: loopconv fixtures (comma-separated names) whose "
                                                  "host steps are the reference's own imported host_functions.py")
    a = ap.parse_args()
    if a.compare:
        compare(a.compare)
        sys.exit(0)
    BACKEND, OUT = a.backend, os.path.abspath(a.out)
    os.makedirs(OUT, exist_ok=True)
    only = set(a.only.split(",")) if a.only else None
    if not only or "chain" in only:
        for name, cfg in CHAIN_CONFIGS.items():
            chain_fixture(name, cfg)
    if not only or "mixing" in only:
        mixing_fixture()
    if not only or "matrix" in only:
        matrix_fixture()
    if not only or "big" in only:
        for name, cfg in BIG_CONFIGS.items():
            big_fixture(name, cfg)
    if not only or "loop" in only:
        import loop_driver as ld
        for name in (a.loop_names.split(",") if a.loop_names else ld.LOOP_NAMES):
            loop_fixture(name)
    if a.hostref:
        for name in a.hostref.split(","):
            loopconv_fixture(name, hostref=True)
    elif not only or "loopconv" in only:
        import loop_driver as ld
        for name in (a.loop_names.split(",") if a.loop_names else ld.CONV_NAMES):
            loopconv_fixture(name)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print("wrote fixtures to %s, %.1f KB total" % (OUT, tot / 1024.0))
