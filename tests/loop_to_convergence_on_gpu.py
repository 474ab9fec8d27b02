#!/usr/bin/env python3
"""TEST INFRASTRUCTURE / parity evidence at full size: the radiation loop of BASELINE config 2 (10 000 bins x 100 layers
x 20 Gauss points) run TO CONVERGENCE twice on this MI355X (--workload c3: config 3, the 20 absorbers mixed on the fly at every refresh) --

  * through the reference's own kernels (source/kernels.cu built unmodified by hipcc for gfx950,
    oracle/_ref/libhelios_ref_gfx950.so) under the reference's loop control as restated in tests/loop_driver.py
    (source/computation.py:827-990), all large arrays resident on the device;
  * through libhelios_hip.so's fused path, paced the way helios_amd/computation.py paces it (chunks that end where the
    reference looks at the state), from the same start profile and the same Planck table --

and compared: number of iterations, T-P profile, net flux, emission spectrum.  About ten minutes of GPU time, nearly all of
it the reference's single-block flux quadrature.  Nothing in the product uses this file.

    python tests/loop_to_convergence_on_gpu.py [--workload c2] [--max-iterations 5000]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--max-iterations", type=int, default=5000)
    ap.add_argument("--nbin", type=int, default=0, help="override the workload's number of bins")
    ap.add_argument("--nlayer", type=int, default=0, help="override the workload's number of layers")
    a = ap.parse_args()
    lib = oracle.refgpu
    if lib is None:
        raise SystemExit("oracle/_ref/libhelios_ref_gfx950.so (or a GPU) is not available")
    w = dict(bench.WORKLOADS[a.workload])
    if a.nbin:
        w["nbin"] = a.nbin
    if a.nlayer:
        w["nlayer"] = a.nlayer
    c0 = bench.build_case(w, 20242)
    X, Y, L, I = c0.nbin, c0.ny, c0.nlayer, c0.nlayer + 1
    from helios_amd import phys_const as pc
    from helios_amd import synthetic as syn
    from helios_amd.device import Context
    from helios_amd.rt import batch_from_case
    ctx = Context(0)
    species = c0.get("species")
    rt = batch_from_case(ctx, c0, ncol=1, nspecies=len(species) if species else 0)
    d_tabs, vl, vi = [], None, None
    if species:   # on-the-fly mixing (config 3): every k-table once, to the library and -- device-resident -- to the reference
        vl = np.array([np.full(L, sp["vmr"]) for sp in species])
        vi = np.array([np.full(I, sp["vmr"]) for sp in species])
        for k, sp in enumerate(species):
            tab = sp["pretab"]
            if isinstance(tab, str):
                tab = syn.ktable(np.random.default_rng(sp["table_seed"]), X, Y, c0.ktemp, c0.kpress, c0.gauss_y)
            rt.set_species(k, tab, sp["scat"], sp["weight"], is_h2o=0, is_cia=0, in_mu=1)
            d_tabs.append(lib.buf(tab) if tab is not None else None)
            del tab
        rt.set_column_vmr(-1, vl, vi)
    rt.build_planck_table(1)
    grid = rt.get("planck_grid")
    it, t0 = 0, time.perf_counter()
    while it < a.max_iterations + 1:                                          # the reference gives up after max + 1 iterations
        nxt = min(it + (10 - it % 10), it + 1 + (100 - it % 100) % 100,      # refresh boundaries and the 100-iteration check
                  a.max_iterations + 1)
        rt.run(it, nxt - it)
        it = nxt
        if int(rt.get("done", 0)[0]):
            break
    ctx.synchronize()
    ours_s = time.perf_counter() - t0
    ours_iters = int(rt.get("iters_done", 0)[0]) if int(rt.get("done", 0)[0]) else it
    ours = {k: rt.get(k) for k in ("T_lay", "F_net", "F_up_band", "F_up_tot", "F_down_tot")}
    rt.close()

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
    refresh = cases.refresh_premixed
    if species:
        spec_l, spec_i = lib.buf(np.zeros(Y * X * I)), lib.buf(np.zeros(Y * X * I))
        held += [spec_l, spec_i] + [t for t in d_tabs if t is not None]
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
                if d_tabs[k] is not None:
                    impl.opac_species_interpol(c.T_lay, c.ktemp, c.p_lay, c.kpress, d_tabs[k], spec_l, c.npress, c.ntemp, Y, X, L)
                    impl.opac_species_interpol(c.T_int, c.ktemp, c.p_int, c.kpress, d_tabs[k], spec_i, c.npress, c.ntemp, Y, X, I)
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

    t0 = time.perf_counter()
    n_ref, snaps, reason = loop_driver.radiation_loop(ref, c, s, max_nr_iterations=a.max_iterations, refresh=refresh)
    ref_s = time.perf_counter() - t0
    end = snaps["end"]
    for b in held:
        b.free()

    def rel(x, y):
        x, y = np.asarray(x, float), np.asarray(y, float)
        return float(np.abs(x - y).max() / np.abs(y).max())

    out = {
        "workload": w["desc"], "nbin": int(X), "nlayer": int(L),
        "reference_kernels_on_this_gpu": {"iterations": int(n_ref), "left_the_loop": reason, "seconds": ref_s},
        "libhelios_hip": {"iterations": ours_iters, "seconds": ours_s},
        "same_iteration_count": ours_iters == int(n_ref),
        "max_relative_difference": {
            "T_lay": float(np.abs(ours["T_lay"] / end["T_lay"] - 1.0).max()),
            "F_net (of max |F_net|)": rel(ours["F_net"], end["F_net"]),
            "F_up_tot": rel(ours["F_up_tot"], end["F_up_tot"]),
            "F_down_tot": rel(ours["F_down_tot"], end["F_down_tot"]),
            "F_up_band at the top of the atmosphere (emission spectrum)": rel(ours["F_up_band"][X * (I - 1):X * I],
                                                                              end["F_up_band_TOA"]),
        },
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
