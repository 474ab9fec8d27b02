#!/usr/bin/env python3
"""A/B of the flux kernel's tilings where the cheapest one (fewest padded nodes) keeps registers in scratch: the same
column through the tiling hx_rt_create now chooses and through the one HELIOS_RT_K forces, on ONE box, alternating.

    python tools/geometry_ab.py [--nbin 10000] [--layers 120,128,240,256] [--reps 3]

Prints ms per iteration without a refresh (HIP events on the library's stream) for every (nlayer, k)."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %(root)r)
import numpy as np
import bench
from helios_amd.device import Context
w = dict(nbin=%(nbin)d, nlayer=%(L)d, ny=20, ntemp=8, npress=6, desc="")
c = bench.build_case(w, 20242)
ctx = Context(0)
rt = bench.make_batch(ctx, c, 1)
rt.build_planck_table(1)
rt.run(0, 21)
ctx.synchronize()
best = 1e9
for _ in range(%(reps)d):
    ctx.timer_start(); rt.run(21, 9); best = min(best, ctx.timer_stop_ms() / 9.0)
rt.profile(True); rt.run(31, 9); rt.profile(False)
print(json.dumps(dict(ms=best, flux_ms=rt.profile_read("rt_flux")[0])))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbin", type=int, default=10000)
    ap.add_argument("--layers", default="120,128,240,256")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import json
    print("nlayer  k(forced)   ms/iteration   k_rt_flux ms   (default = the library's choice)")
    for L in [int(x) for x in a.layers.split(",")]:
        for rnd in range(2):
            for k in ("default", "16", "32", "64"):
                if 2 * L > 16 * int(k if k != "default" else 64):
                    continue
                env = dict(os.environ)
                env.pop("HELIOS_RT_K", None)
                if k != "default":
                    env["HELIOS_RT_K"] = k
                p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, nbin=a.nbin, L=L, reps=a.reps)], env=env,
                                   capture_output=True, text=True)
                if p.returncode:
                    print(L, k, "failed:", p.stderr[-300:])
                    continue
                r = json.loads(p.stdout.strip().splitlines()[-1])
                print("%5d   %-8s    %8.4f       %8.4f     (round %d)" % (L, k, r["ms"], r["flux_ms"], rnd), flush=True)


if __name__ == "__main__":
    main()
