#!/usr/bin/env python3
"""TEST INFRASTRUCTURE / evidence at size: BASELINE config 5 AS NAMED -- one column of 30 000 bins x 200 layers x 20 Gauss
points, 20 absorbers mixed on the fly with random overlap at every refresh, non-isotropic scattering (g0, I2S correction),
two cloud decks, direct beam, surface albedo -- taken to radiative-convective equilibrium by the product's own driver:
`Compute.radiation_loop` + `Compute.convection_loop` on a `Store` (the calls of run_helios, helios.py:82-95), everything
on libhelios_hip.so's device-resident loops.  The run is held to the reference's own self-checks:

  * global energy balance (source/host_functions.py:1040-1042) below the radiative-equilibrium criterion,
  * every pair of neighbouring convective layers on the dry adiabat d ln T / d ln p = kappa,
  * spectral fluxes finite and >= 0, k-distributions ascending, no re-binning malfunction in the mixing.

Iteration counts and wall times go to the JSON (profiles/r04_c5_equilibrium.json).  Nothing in the product uses this file.

    python tests/c5_equilibrium_on_gpu.py [--nbin N] [--nlayer L] [--T-intern 600] [--criterion 1e-8] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import bench  # noqa: E402
from store_helpers import store_from_case  # noqa: E402

KAPPA = 2.0 / 7.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c5")
    ap.add_argument("--nbin", type=int, default=0)
    ap.add_argument("--nlayer", type=int, default=0)
    ap.add_argument("--T-intern", type=float, default=600.0, help="internal temperature [K]: a hot interior, so that the "
                                                                    "radiative equilibrium is super-adiabatic at depth")
    ap.add_argument("--criterion", type=float, default=1e-8)
    ap.add_argument("--max-iterations", type=int, default=100000)
    ap.add_argument("--relax", default="10000,20000", help="iteration counts at which the criterion is relaxed tenfold "
                                                           "(the reference's default, param.dat:116)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from helios_amd import host_functions as hs
    from helios_amd import phys_const as pc
    from helios_amd.computation import Compute
    from helios_amd.device import Context
    w = dict(bench.WORKLOADS[a.workload])
    if a.nbin:
        w["nbin"] = a.nbin
    if a.nlayer:
        w["nlayer"] = a.nlayer
    t_all = time.perf_counter()
    c = bench.build_case(w, 20245)
    c.F_intern = pc.SIGMA_SB * a.T_intern ** 4
    c.rad_convergence_limit = a.criterion
    c.c_p_lay = np.full(c.nlayer, pc.R_UNIV / KAPPA)                # read.py:1178-1180
    ctx = Context(0)
    relax = tuple(int(float(v)) for v in a.relax.split(",") if v)
    q = store_from_case(ctx, c, crit_relaxation_numbers=relax, max_nr_iterations=a.max_iterations, name="c5", convection=1,
                        kappa=KAPPA,
                        on_the_fly=bool(c.get("species")))
    q.T_intern = np.float64(a.T_intern)
    comp = Compute(ctx)
    q.rt = comp._make_rt(q)
    for sp in q.species_list:        # the tables live on the device now (11.5 GB at full size)
        sp.opacity_pretab = sp.opacity_pretab if sp.opacity_pretab is None else sp.opacity_pretab[:1]
    setup_s = time.perf_counter() - t_all
    ctx.diag_reset()
    t0 = time.perf_counter()
    comp.radiation_loop(q)
    ctx.synchronize()
    rad_s, n_rad = time.perf_counter() - t0, int(q.iter_value)
    T_rad = q.rt.get("T_lay")
    t0 = time.perf_counter()
    comp.convection_loop(q)
    ctx.synchronize()
    conv_s, n_conv = time.perf_counter() - t0, int(q.iter_value)
    rt = q.rt
    X, Y, L, I = int(q.nbin), int(q.ny), int(q.nlayer), int(q.ninterface)
    T = rt.get("T_lay")
    for n in ("F_net", "F_up_tot", "F_down_tot"):
        setattr(q, n, rt.get(n))
    q.F_smooth_sum = rt.get("F_smooth_sum")
    q.F_add_heat_sum = np.zeros(L)
    imbalance = float(hs.global_energy_imbalance(q))
    conv = rt.get("conv_layer")
    lay = np.where(conv[:L - 1] * conv[1:L] == 1)[0]
    grad = np.log(T[lay + 1] / T[lay]) / np.log(np.asarray(q.p_lay)[lay + 1] / np.asarray(q.p_lay)[lay])
    up, down, fdir = rt.get("F_up_band"), rt.get("F_down_band"), rt.get("F_dir_band")
    opl = rt.get("opac_wg_lay")[:L * X * Y].reshape(L, X, Y)
    dg = ctx.diag()
    # local radiative equilibrium of the layers above the convective zone (check_for_radiative_eq's criterion)
    F_net = np.asarray(q.F_net)
    rad_layers = np.where(conv[:L] == 0)[0]
    local = np.abs(F_net[rad_layers] - F_net[rad_layers + 1]) / np.maximum(np.asarray(q.F_down_tot)[rad_layers + 1], 1e-300)
    checks = {
        "global energy imbalance (host_functions.py:1040) [ppm]": imbalance * 1e6,
        "convective layers": int(conv[:L].sum()),
        "neighbouring convective pairs": int(len(lay)),
        "max |d ln T / d ln p - kappa| / kappa over those pairs": float(np.abs(grad / KAPPA - 1.0).max()) if len(lay) else None,
        "largest local flux divergence of a radiative layer / F_down (criterion at the end %g)" % float(q.rad_convergence_limit): float(local.max()) if len(local) else None,
        "all temperatures and band fluxes finite": bool(np.all(np.isfinite(T)) and np.all(np.isfinite(up)) and np.all(np.isfinite(down))),
        "smallest band flux (up, down, direct)": [float(up.min()), float(down.min()), float(fdir.min())],
        "k-distributions of the last refresh ascending and positive": bool(opl.min() > 0 and np.all(np.diff(opl, axis=2) >= 0)),
        "random-overlap re-binning malfunctions (kernels.cu:3385)": int(dg["ro_rebin_skipped"]),
        "surface temperature [K], top temperature [K]": [float(T[L]), float(T[L - 1])],
        "temperature change through the convection loop, max [K]": float(np.abs(T - T_rad).max()),
    }
    ok = (abs(imbalance) < 1e-3 and checks["all temperatures and band fluxes finite"]
          and up.min() >= 0 and down.min() >= 0 and fdir.min() >= 0
          and checks["k-distributions of the last refresh ascending and positive"] and dg["ro_rebin_skipped"] == 0
          and (len(lay) == 0 or np.abs(grad / KAPPA - 1.0).max() < 1e-6))
    out = {"workload": w["desc"], "nbin": X, "nlayer": L, "ny": Y, "species": len(q.species_list), "T_intern": a.T_intern,
           "rad_convergence_limit": a.criterion, "criterion_relaxed_tenfold_at": list(relax),
           "criterion_at_the_end": float(q.rad_convergence_limit), "driver": "Compute.radiation_loop + Compute.convection_loop on a Store",
           "radiation_loop": {"iterations": n_rad, "seconds": rad_s, "ms_per_iteration": rad_s / max(n_rad, 1) * 1e3},
           "convection_loop": {"iterations": n_conv, "seconds": conv_s, "ms_per_iteration": conv_s / max(n_conv, 1) * 1e3},
           "set_up_seconds (tables generated and uploaded, Store allocated)": setup_s,
           "self_checks": checks, "all_checks_hold": bool(ok)}
    text = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(json.dumps(out))
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
