#!/usr/bin/env python3
"""Micro-benchmark of hx_add_to_mixed_opac (random-overlap mixing) on problems that ALL take the random-overlap branch.

    python tools/ro_bench.py [--nbin 10000] [--nlev 101] [--kind generic|dominated|interleaved] [--reps 5]

Prints the HIP-event time per launch, problems/s and the exact-finish pass count (hx_diag.ro_fixup_passes).
HELIOS_RO_SORT=q32|bitonic|rank selects the older kernels for A/B runs on the same box."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helios_amd import _lib, phys_const as pc, synthetic as syn  # noqa: E402
from helios_amd.device import Context  # noqa: E402


def problem(kind, nbin, nlev, ny, rng):
    shape = (nlev, nbin, ny)
    fac = 1e-3 * 18.0 / 2.3
    mix = np.sort(10.0 ** rng.uniform(-4, 0, shape), axis=2)
    if kind == "generic":        # two unrelated curves over four decades: the rows of the tableau interleave
        add = np.sort(10.0 ** rng.uniform(-4, 0, shape), axis=2)
    elif kind == "dominated":    # the new absorber is 3 % of the mix: nearly row-major order
        add = np.sort(mix[..., ::-1] * 0.03, axis=2)
    elif kind == "ktable":       # shapes of the synthetic k-tables of bench.py (3.5 decades over the 20 points)
        gy, _ = syn.gauss_points(ny)
        a1, a2 = rng.uniform(-6, -1, shape[:2] + (1,)), rng.uniform(-6, -1, shape[:2] + (1,))
        b1, b2 = rng.uniform(1, 4, shape[:2] + (1,)), rng.uniform(1, 4, shape[:2] + (1,))
        mix = 10.0 ** (a1 + 3.5 * gy ** b1)
        add = 10.0 ** (a2 + 3.5 * gy ** b2)
    else:
        raise SystemExit("unknown kind")
    return mix, add / fac


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbin", type=int, default=10000)
    ap.add_argument("--nlev", type=int, default=101)
    ap.add_argument("--kind", default="generic")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    ny = 20
    rng = np.random.default_rng(1)
    gy, gw = syn.gauss_points(ny)
    mix, add = problem(a.kind, a.nbin, a.nlev, ny, rng)
    ctx = Context(0)
    L = _lib.lib()
    d_mix0 = ctx.to_gpu(mix.reshape(-1))
    d_mix = ctx.to_gpu(mix.reshape(-1))
    d_add = ctx.to_gpu(add.reshape(-1))
    d_vmr = ctx.to_gpu(np.full(a.nlev, 1e-3))
    d_mmm = ctx.to_gpu(np.full(a.nlev, 2.3 * pc.AMU))
    d_gw, d_gy = ctx.to_gpu(gw), ctx.to_gpu(gy)
    P = ctypes.POINTER(ctypes.c_double)

    def launch():
        ctx.check(L.hx_add_to_mixed_opac(ctx.handle, ctypes.cast(d_vmr.ptr, P), ctypes.cast(d_add.ptr, P),
                                         ctypes.cast(d_mix.ptr, P), ctypes.cast(d_mmm.ptr, P),
                                         ctypes.cast(d_gw.ptr, P), ctypes.cast(d_gy.ptr, P), 18.0 * pc.AMU, 1, 1, ny,
                                         a.nbin, a.nlev))
    times = []
    ctx.diag_reset()
    for r in range(a.reps + 1):
        d_mix.copy_from_device(d_mix0.ptr, d_mix0.nbytes)
        ctx.synchronize()
        ctx.timer_start()
        launch()
        ms = ctx.timer_stop_ms()
        if r:
            times.append(ms)
    out = d_mix.get()
    dg = ctx.diag()
    n = a.nbin * a.nlev
    best = min(times)
    print("%s kind=%s  %d problems: %.3f ms per launch (min of %d; mean %.3f)  = %.1f M problems/s;  fix-up passes per "
          "launch %.0f, re-binning skips %d;  checksum %.17g"
          % (os.environ.get("HELIOS_RO_SORT", "lean"), a.kind, n, best, a.reps, np.mean(times), n / best / 1e3,
             dg["ro_fixup_passes"] / (a.reps + 1), dg["ro_rebin_skipped"], float(out.sum())))


if __name__ == "__main__":
    main()
