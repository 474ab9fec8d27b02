#!/usr/bin/env python3
"""TEST INFRASTRUCTURE / reported baseline: the reference's own kernels (source/kernels.cu built unmodified by hipcc for
gfx950, oracle/_ref/libhelios_ref_gfx950.so) iterating BASELINE config 2 on this MI355X with the reference's launch
geometry and its synchronise-after-every-launch pattern (source/computation.py), next to libhelios_hip.so on the same
column.  This is the north star's comparator ("the reference's single-GPU iteration throughput on a 10k-wavelength x
100-layer grid"), measured on the same hardware; it is a baseline, not the product, and nothing in the product uses it.

    python tests/time_reference_on_gpu.py [--iterations 30] [--workload c2|c5]

One iteration = temp_inter, planck_interpol_layer/_interface, [every 10th: opac_interpol x2, meanmolmass_interpol x2,
calc_trans_noniso, calc_delta_z, fdir_noniso], 4 x fband_noniso, integrate_flux_double, rad_temp_iter.  All large arrays
stay on the device between launches (as gpuarrays do in the reference); the small host steps of the loop are included.
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
import oracle  # noqa: E402
from impls import RefImpl  # noqa: E402
from test_gpu_fullsize import _block_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--workload", default="c2")
    a = ap.parse_args()
    lib = oracle.refgpu
    if lib is None:
        raise SystemExit("oracle/_ref/libhelios_ref_gfx950.so (or a GPU) is not available")
    ref = RefImpl(lib)
    w = bench.WORKLOADS[a.workload]
    c0 = bench.build_case(w, 20242)
    from helios_amd.device import Context
    from helios_amd.rt import batch_from_case
    ctx = Context(0)
    rt = batch_from_case(ctx, c0, ncol=1)
    rt.build_planck_table(1)
    grid = rt.get("planck_grid")
    rt.run(0, 10)
    ctx.synchronize()
    t0 = time.perf_counter()
    rt.run(10, a.iterations)
    ctx.synchronize()
    ours = (time.perf_counter() - t0) / a.iterations
    T_ours = rt.get("T_lay")
    rt.close()

    c = _block_case(c0, 0, c0.nbin)
    s = cases.alloc_state(c)
    s.planck_grid[:] = grid
    held = []
    for d in (c, s):
        for k, v in list(d.items()):
            if isinstance(v, np.ndarray) and v.nbytes > (1 << 20):
                d[k] = lib.buf(v)
                held.append(d[k])
    cases.radiation_iterations(ref, c, s, 10)            # same warm-up: iterations 0..9
    spent = {}
    for name in ("temp_inter", "planck_interpol_layer", "planck_interpol_interface", "opac_interpol",
                 "meanmolmass_interpol", "calc_trans_noniso", "calc_delta_z", "fdir_noniso", "fband_noniso",
                 "integrate_flux", "rad_temp_iter"):
        fn = getattr(ref, name)

        def timed(*args, _fn=fn, _n=name):
            t = time.perf_counter()
            _fn(*args)
            spent[_n] = spent.get(_n, 0.0) + time.perf_counter() - t
        setattr(ref, name, timed)
    t0 = time.perf_counter()
    cases.radiation_iterations(ref, c, s, a.iterations, start=10)
    theirs = (time.perf_counter() - t0) / a.iterations
    T_ref = c.T_lay.copy()
    for b in held:
        b.free()
    X, L = c0.nbin, c0.nlayer
    print(json.dumps({
        "workload": w["desc"], "iterations": a.iterations, "from_iteration": 10,
        "reference_kernels_on_this_gpu": {"ms_per_iteration": theirs * 1e3, "value": X * L / theirs,
                                          "build": "source/kernels.cu, hipcc -O2 --offload-arch=gfx950, launch geometry and "
                                                   "per-launch synchronisation of source/computation.py",
                                          "ms_per_iteration_by_kernel": {k: v / a.iterations * 1e3 for k, v in spent.items()}},
        "libhelios_hip": {"ms_per_iteration": ours * 1e3, "value": X * L / ours},
        "unit": "bin*layer*iterations/s", "speedup": theirs / ours,
        "T_lay_max_relative_difference_after_the_run": float(np.abs(T_ours / T_ref - 1.0).max())}))


if __name__ == "__main__":
    main()
