#!/usr/bin/env python3
"""TEST INFRASTRUCTURE / parity evidence at size: whole runs of a BASELINE-shaped column on this MI355X, twice --

  * through the reference's own kernels (source/kernels.cu built unmodified by hipcc for gfx950,
    oracle/_ref/libhelios_ref_gfx950.so) under the reference's loop control as restated in tests/loop_driver.py
    (source/computation.py:827-990 radiation loop, :992-1174 convection loop; host steps of the latter from
    helios_amd/host_functions.py, pinned function by function to the reference's Python), all large arrays on the device;
  * through libhelios_hip.so's device-resident loops (hx_rt_run, hx_rt_conv_run), from the same start profile and the same
    Planck table --

with BOTH sides recorded after every iteration in their discrete state (the convergence flags `abort`, the time-step
prefactors -- products of the discrete factors 1.1 and 1/1.5, so equal decisions give equal bits -- and, in the convection
loop, the layer flags) and in full (T, totals, net flux, emission spectrum) after 1, 10, 11, 50, 400 iterations and every
100 thereafter.  Reported: both iteration counts, the first iteration at which any discrete state differs, the differences
at every snapshot and at the end.  The reference's quadrature kernel (one block, CAS atomics) makes its side slow: config 2
takes ten minutes, config 3 twenty.  Nothing in the product uses this file.

    python tests/loop_to_convergence_on_gpu.py [--workload c2] [--nbin N] [--nlayer L] [--convection --T-intern 600]
                                               [--skip-reference] [--out file.json]
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
import cases  # noqa: E402
import loop_driver  # noqa: E402
import oracle  # noqa: E402
from impls import RefImpl  # noqa: E402
from test_gpu_fullsize import _block_case  # noqa: E402

KAPPA = 2.0 / 7.0


def snap_points(n_max):
    return sorted({1, 10, 11, 50, 400} | set(range(500, n_max + 1, 100)))


def rel(x, y):
    x, y = np.asarray(x, float), np.asarray(y, float)
    return float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-300))


def compare_snap(a, b):
    # where the temperatures differ most: the flux divergence F_net[i] - F_net[i+1] that drove the layer's last step, on
    # both sides, in units of the largest net flux (a divergence at the rounding level of the totals is amplified by the
    # pseudo-time step, dT ~ |dF|^0.1, kernels.cu:2694-2698)
    Ta, Tb = np.asarray(a["T_lay"], float), np.asarray(b["T_lay"], float)
    L = len(Ta) - 1
    i = int(np.argmax(np.abs(Ta[:L] / Tb[:L] - 1.0)))
    fa, fb = np.asarray(a["F_net"], float), np.asarray(b["F_net"], float)
    scale = max(np.abs(fb).max(), 1e-300)
    worst = {"layer": i, "flux divergence there / max |F_net| (library, reference)": [float((fa[i] - fa[i + 1]) / scale),
                                                                                    float((fb[i] - fb[i + 1]) / scale)]}
    return {"T_lay": float(np.abs(Ta / Tb - 1.0).max()), "layer with the largest temperature difference": worst,
            "F_net (of max |F_net|)": rel(a["F_net"], b["F_net"]),
            "F_net (of max F_up_tot: the scale of the totals it is the difference of)":
                float(np.abs(fa - fb).max() / max(np.abs(np.asarray(b["F_up_tot"], float)).max(), 1e-300)),
            "F_up_tot": rel(a["F_up_tot"], b["F_up_tot"]), "F_down_tot": rel(a["F_down_tot"], b["F_down_tot"]),
            "emission spectrum (of its maximum)": rel(a["F_up_band_TOA"], b["F_up_band_TOA"]),
            "abort flags set": [int(np.sum(a["abort"])), int(np.sum(b["abort"]))],
            "abort flags that differ": int(np.sum(np.asarray(a["abort"]) != np.asarray(b["abort"]))),
            "time-step prefactors that differ": int(np.sum(np.asarray(a["deltat_prefactor"]) != np.asarray(b["deltat_prefactor"])))}


def first_difference(seq_a, seq_b):
    """first iteration (1-based count of completed iterations) at which the per-iteration records differ; None if the
    common part is identical"""
    for it, (x, y) in enumerate(zip(seq_a, seq_b), start=1):
        if not np.array_equal(x, y):
            return it
    return None


def library_snapshot(rt, c):
    X, I = c.nbin, c.nlayer + 1
    out = {k: rt.get(k) for k in ("T_lay", "F_net", "F_up_tot", "F_down_tot")}
    out["F_up_band_TOA"] = rt.get("F_up_band")[X * (I - 1):X * I]
    out["abort"] = rt.get("abort")
    out["deltat_prefactor"] = rt.get("delta_t_prefactor")
    return out


def run_library(rt, c0, max_iterations, convection, relax=()):
    """the device-resident loops, one iteration per call so that the discrete state of every iteration can be read"""
    from helios_amd import host_functions as hs
    L = c0.nlayer
    rt.build_planck_table(1)
    grid = rt.get("planck_grid")
    rec = dict(abort=[], prefactor=[])
    snaps, at = {}, set(snap_points(max_iterations))
    it, done, jump = 0, 0, False
    limit = float(c0.rad_convergence_limit)
    t0 = time.perf_counter()
    while not done and not jump and it < max_iterations + 1:
        rt.run(it, 1)
        it += 1
        if it in relax:                     # computation.py:974-975: the criterion is relaxed when the NEW count is listed
            limit *= 10.0
            rt.set_convergence_limit(0, limit)
        rec["abort"].append(rt.get("abort"))
        rec["prefactor"].append(rt.get("delta_t_prefactor"))
        done = int(rt.get("done", 0)[0])
        if it in at:
            snaps[it] = library_snapshot(rt, c0)
        if not done and it % 100 == 1:      # computation.py:946-952, looked at inside iterations 0, 100, ...
            jump = not rt.get("T_lay")[L] < c0.plancktable_dim * c0.plancktable_step - 2
    n_rad = int(rt.get("iters_done", 0)[0]) if done else it
    snaps["end"] = library_snapshot(rt, c0)
    out = dict(n_rad=n_rad, rad_seconds=time.perf_counter() - t0, rad_snaps=snaps, rad_rec=rec, grid=grid,
               rad_exit="converged" if done else ("surface temperature beyond the Planck table" if jump else "iteration limit"))
    if convection:
        cb = _block_case(c0, 0, min(8, c0.nbin))                      # only the small vectors are used here
        q = loop_driver.conv_quant(cb, cases.alloc_state(cb), KAPPA)
        q.T_lay = rt.get("T_lay")
        hs.conv_check(q)
        hs.mark_convective_layers(q, stitching=0)
        out["conv_unstable_at_entry"] = int(q.conv_unstable.sum())
        crec = dict(conv_layer=[], marked_red=[], prefactor=[])
        csnaps = {}
        n = 0
        t0 = time.perf_counter()
        if q.conv_unstable.sum() > 0:
            for nm, v in (("kappa_lay", q.kappa_lay), ("kappa_int", q.kappa_int), ("c_p_lay", np.asarray(c0.c_p_lay, float)),
                          ("conv_layer", q.conv_layer), ("conv_unstable", q.conv_unstable), ("dampara", np.array([-1.0])),
                          ("done", np.zeros(1, np.int32))):
                rt.set_state(0, nm, v)
            done = 0
            while not done and n < max_iterations + 1:
                rt.conv_run(n, 1)
                done = int(rt.get("done", 0)[0])
                if done:
                    n = int(rt.get("iters_done", 0)[0])
                    break
                n += 1
                if n in relax:              # computation.py:1158-1159
                    limit *= 10.0
                    rt.set_convergence_limit(0, limit)
                crec["conv_layer"].append(rt.get("conv_layer"))
                crec["marked_red"].append(rt.get("marked_red"))
                crec["prefactor"].append(rt.get("delta_t_prefactor"))
                if n in at:
                    csnaps[n] = dict(library_snapshot(rt, c0), conv_layer=rt.get("conv_layer"))
        csnaps["end"] = dict(library_snapshot(rt, c0), conv_layer=rt.get("conv_layer"))
        out.update(n_conv=n, conv_seconds=time.perf_counter() - t0, conv_snaps=csnaps, conv_rec=crec)
    rt.close()
    return out


def run_reference(lib, c0, species_tabs_dev, vl, vi, grid, max_iterations, convection, relax=()):
    from helios_amd import host_functions as hs
    from helios_amd import phys_const as pc
    X, Y, L, I = c0.nbin, c0.ny, c0.nlayer, c0.nlayer + 1
    species = c0.get("species")
    ref = RefImpl(lib)
    c = _block_case(c0, 0, c0.nbin)
    s = cases.alloc_state(c)
    s.planck_grid[:] = grid
    held = []
    for d in (c, s):
        for k, v in list(d.items()):
            if isinstance(v, np.ndarray) and v.nbytes > (1 << 20) and k not in ("F_up_band", "F_down_band", "F_dir_band"):
                d[k] = lib.buf(v)
                held.append(d[k])
    if c.get("flux_calc_method", "iteration") == "matrix":
        # the work arrays of the reference's elimination (alpha ... d_prime, quantities.py:652-665) stay on the device too
        s.matrix_work = {k: lib.buf(v) for k, v in cases.matrix_scratch(c).items()}
        held += list(s.matrix_work.values())
    refresh = cases.refresh_premixed
    if species:
        spec_l, spec_i = lib.buf(np.zeros(Y * X * I)), lib.buf(np.zeros(Y * X * I))
        held += [spec_l, spec_i]
        wgt = np.array([sp["weight"] for sp in species])
        zeros_wg = np.zeros(Y * X * I)

        def refresh(impl, c, s):
            """the species loop of computation.py:1454-1501 (host_functions.py:927-959, :1050-1056 around it) through the
            reference's kernels, everything large on the device"""
            s.meanmolmass_lay[:] = (vl * wgt[:, None]).sum(0) / vl.sum(0) * pc.AMU
            s.meanmolmass_int[:] = (vi * wgt[:, None]).sum(0) / vi.sum(0) * pc.AMU
            s.opac_wg_lay.set(zeros_wg[:s.opac_wg_lay.nbytes // 8])
            s.opac_wg_int.set(zeros_wg[:s.opac_wg_int.nbytes // 8])
            scat_l, scat_i = np.zeros(X * L), np.zeros(X * I)
            sc_l, sc_i = np.zeros(X * L), np.zeros(X * I)
            for k, sp in enumerate(species):
                if species_tabs_dev[k] is not None:
                    impl.opac_species_interpol(c.T_lay, c.ktemp, c.p_lay, c.kpress, species_tabs_dev[k], spec_l, c.npress, c.ntemp, Y, X, L)
                    impl.opac_species_interpol(c.T_int, c.ktemp, c.p_int, c.kpress, species_tabs_dev[k], spec_i, c.npress, c.ntemp, Y, X, I)
                    impl.add_to_mixed_opac(np.ascontiguousarray(vl[k]), spec_l, s.opac_wg_lay, s.meanmolmass_lay,
                                           c.gauss_weight, c.gauss_y, sp["weight"] * pc.AMU, k, 1, Y, X, L)
                    impl.add_to_mixed_opac(np.ascontiguousarray(vi[k]), spec_i, s.opac_wg_int, s.meanmolmass_int,
                                           c.gauss_weight, c.gauss_y, sp["weight"] * pc.AMU, k, 1, Y, X, I)
                if sp["scat"] is not None:
                    sc_l[:], sc_i[:] = np.tile(sp["scat"], L), np.tile(sp["scat"], I)
                    impl.add_to_mixed_scat(np.ascontiguousarray(vl[k]), sc_l, scat_l, X, L)
                    impl.add_to_mixed_scat(np.ascontiguousarray(vi[k]), sc_i, scat_i, X, I)
            for name, v in (("scat_cross_lay", scat_l), ("scat_cross_int", scat_i)):
                if hasattr(s[name], "set"):
                    s[name].set(v)
                else:
                    s[name][:] = v
            cases.refresh_transmission(impl, c, s)

    rec = dict(abort=[], prefactor=[])

    def each(it, c_, s_):
        rec["abort"].append(s_.abort.copy())
        rec["prefactor"].append(s_.deltat_prefactor.copy())

    t0 = time.perf_counter()
    n_ref, snaps, reason = loop_driver.radiation_loop(ref, c, s, snap_at=snap_points(max_iterations),
                                                      max_nr_iterations=max_iterations, refresh=refresh, on_iteration=each,
                                                      crit_relaxation_numbers=relax)
    out = dict(n_rad=int(n_ref), rad_seconds=time.perf_counter() - t0, rad_snaps=snaps, rad_rec=rec, rad_exit=reason)
    if convection:
        crec = dict(conv_layer=[], marked_red=[], prefactor=[])

        def ceach(it, c_, s_, q_):
            crec["conv_layer"].append(np.asarray(q_.conv_layer, np.int32).copy())
            crec["marked_red"].append(np.asarray(q_.marked_red, np.int32).copy())
            crec["prefactor"].append(s_.deltat_prefactor.copy())

        t0 = time.perf_counter()
        n, csnaps, q = loop_driver.convection_loop(ref, hs, c, s, KAPPA, snap_at=snap_points(max_iterations),
                                                   max_nr_iterations=max_iterations, refresh=refresh, on_iteration=ceach,
                                                   crit_relaxation_numbers=relax)
        out.update(n_conv=int(n), conv_seconds=time.perf_counter() - t0, conv_snaps=csnaps, conv_rec=crec)
    for b in held:
        b.free()
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--max-iterations", type=int, default=100000)
    ap.add_argument("--nbin", type=int, default=0, help="override the workload's number of bins")
    ap.add_argument("--nlayer", type=int, default=0, help="override the workload's number of layers")
    ap.add_argument("--convection", action="store_true", help="radiation loop, then the convection loop from its end state")
    ap.add_argument("--T-intern", type=float, default=0.0, help="internal temperature [K] (convection runs want a hot interior)")
    ap.add_argument("--criterion", type=float, default=0.0, help="rad_convergence_limit (default: the workload's 1e-8)")
    ap.add_argument("--relax", default="10000,20000", help="iteration counts at which the criterion is relaxed tenfold "
                                                           "(the reference's default, param.dat:116)")
    ap.add_argument("--skip-reference", action="store_true", help="library only (to look for a suitable column)")
    ap.add_argument("--reference-twice", action="store_true",
                    help="run the reference's loop a second time and report how far its two runs are apart (its flux "
                         "quadrature adds with atomics in arbitrary order)")
    ap.add_argument("--out", default="")
    a = ap.parse_args(argv)
    lib = oracle.refgpu
    if lib is None and not a.skip_reference:
        raise SystemExit("oracle/_ref/libhelios_ref_gfx950.so (or a GPU) is not available")
    w = dict(bench.WORKLOADS[a.workload])
    if a.nbin:
        w["nbin"] = a.nbin
    if a.nlayer:
        w["nlayer"] = a.nlayer
    c0 = bench.build_case(w, 20242)
    from helios_amd import phys_const as pc
    from helios_amd import synthetic as syn
    from helios_amd.device import Context
    if a.T_intern:
        c0.F_intern = pc.SIGMA_SB * a.T_intern ** 4
    if a.criterion:
        c0.rad_convergence_limit = a.criterion
    if a.convection:
        c0.c_p_lay = np.full(c0.nlayer, pc.R_UNIV / KAPPA)              # read.py:1178-1180
    X, Y, L, I = c0.nbin, c0.ny, c0.nlayer, c0.nlayer + 1
    ctx = Context(0)
    species = c0.get("species")
    from helios_amd.rt import batch_from_case
    rt = batch_from_case(ctx, c0, ncol=1, nspecies=len(species) if species else 0)
    d_tabs, vl, vi = [], None, None
    if species:   # on-the-fly mixing: every k-table once, to the library and -- device-resident -- to the reference
        vl = np.array([np.full(L, sp["vmr"]) for sp in species])
        vi = np.array([np.full(I, sp["vmr"]) for sp in species])
        for k, sp in enumerate(species):
            tab = sp["pretab"]
            if isinstance(tab, str):
                tab = syn.ktable(np.random.default_rng(sp["table_seed"]), X, Y, c0.ktemp, c0.kpress, c0.gauss_y)
            rt.set_species(k, tab, sp["scat"], sp["weight"], is_h2o=0, is_cia=0, in_mu=1)
            d_tabs.append(lib.buf(tab) if (tab is not None and not a.skip_reference) else None)
            del tab
        rt.set_column_vmr(-1, vl, vi)
    relax = tuple(int(float(v)) for v in a.relax.split(",") if v)
    ours = run_library(rt, c0, a.max_iterations, a.convection, relax)
    out = {"workload": w["desc"], "nbin": int(X), "nlayer": int(L), "T_intern": a.T_intern or 100.0,
           "rad_convergence_limit": float(c0.rad_convergence_limit), "criterion_relaxed_tenfold_at": list(relax),
           "libhelios_hip": {"radiation_loop_iterations": ours["n_rad"], "left_the_loop": ours["rad_exit"],
                             "seconds (one call and four read-backs per iteration)": ours["rad_seconds"]}}
    if a.convection:
        out["libhelios_hip"].update({"convection_loop_iterations": ours["n_conv"], "convection_seconds": ours["conv_seconds"],
                                     "unstable_layers_at_entry": ours["conv_unstable_at_entry"]})
    if not a.skip_reference:
        theirs = run_reference(lib, c0, d_tabs, vl, vi, ours["grid"], a.max_iterations, a.convection, relax)
        out["reference_kernels_on_this_gpu"] = {"radiation_loop_iterations": theirs["n_rad"], "left_the_loop": theirs["rad_exit"],
                                                "seconds": theirs["rad_seconds"]}
        out["same_iteration_count"] = ours["n_rad"] == theirs["n_rad"]
        out["radiation_loop"] = {
            "first_iteration_with_different_abort_flags": first_difference(ours["rad_rec"]["abort"], theirs["rad_rec"]["abort"]),
            "first_iteration_with_different_time_step_prefactors": first_difference(ours["rad_rec"]["prefactor"],
                                                                                    theirs["rad_rec"]["prefactor"]),
            "snapshots (library vs reference, maximum relative difference)": {
                str(k): compare_snap(ours["rad_snaps"][k], theirs["rad_snaps"][k])
                for k in sorted(set(ours["rad_snaps"]) & set(theirs["rad_snaps"]) - {"end"})},
            "end states (each side where it left the loop)": compare_snap(ours["rad_snaps"]["end"], theirs["rad_snaps"]["end"]),
        }
        if a.reference_twice:
            again = run_reference(lib, c0, d_tabs, vl, vi, ours["grid"], a.max_iterations, False, relax)
            out["reference_against_itself"] = {
                "radiation_loop_iterations": [theirs["n_rad"], again["n_rad"]],
                "first_iteration_with_different_abort_flags": first_difference(again["rad_rec"]["abort"], theirs["rad_rec"]["abort"]),
                "first_iteration_with_different_time_step_prefactors": first_difference(again["rad_rec"]["prefactor"],
                                                                                        theirs["rad_rec"]["prefactor"]),
                "snapshots (second run vs first run, maximum relative difference)": {
                    str(k): compare_snap(again["rad_snaps"][k], theirs["rad_snaps"][k])
                    for k in sorted(set(again["rad_snaps"]) & set(theirs["rad_snaps"]) - {"end"})},
                "end states": compare_snap(again["rad_snaps"]["end"], theirs["rad_snaps"]["end"])}
        if a.convection:
            out["reference_kernels_on_this_gpu"].update({"convection_loop_iterations": theirs["n_conv"],
                                                         "convection_seconds": theirs["conv_seconds"]})
            out["same_convection_iteration_count"] = ours["n_conv"] == theirs["n_conv"]
            both = sorted(set(ours["conv_snaps"]) & set(theirs["conv_snaps"]) - {"end"})

            def csnap(x, y):
                d = compare_snap(x, y)
                d["convective layers"] = [int(np.sum(x["conv_layer"])), int(np.sum(y["conv_layer"]))]
                d["layer flags that differ"] = int(np.sum(np.asarray(x["conv_layer"]) != np.asarray(y["conv_layer"])))
                return d
            out["convection_loop"] = {
                "first_iteration_with_different_conv_layer_flags": first_difference(ours["conv_rec"]["conv_layer"],
                                                                                    theirs["conv_rec"]["conv_layer"]),
                "first_iteration_with_different_marked_layers": first_difference(ours["conv_rec"]["marked_red"],
                                                                                 theirs["conv_rec"]["marked_red"]),
                "first_iteration_with_different_time_step_prefactors": first_difference(ours["conv_rec"]["prefactor"],
                                                                                        theirs["conv_rec"]["prefactor"]),
                "snapshots (library vs reference, maximum relative difference)": {str(k): csnap(ours["conv_snaps"][k],
                                                                                                  theirs["conv_snaps"][k]) for k in both},
                "end states (each side where it left the loop)": csnap(ours["conv_snaps"]["end"], theirs["conv_snaps"]["end"]),
            }
    for t in d_tabs:
        if t is not None:
            t.free()
    text = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")
    if argv is None:
        print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
