#!/usr/bin/env python3
"""Markdown table of the whole-run comparisons under profiles/ (tests/loop_to_convergence_on_gpu.py):
    python tools/trajectory_table.py profiles/r04_trajectory_*.json"""
import json
import sys


def row(path):
    z = json.load(open(path))
    ours, ref = z["libhelios_hip"], z["reference_kernels_on_this_gpu"]
    out = []
    for loop, key_n in (("radiation_loop", "radiation_loop_iterations"), ("convection_loop", "convection_loop_iterations")):
        if loop not in z:
            continue
        r = z[loop]
        sn = r["snapshots (library vs reference, maximum relative difference)"]
        early = max(v["T_lay"] for k, v in sn.items() if int(k) <= 400)
        worst_k = max(sn, key=lambda k: sn[k]["T_lay"])
        e = r["end states (each side where it left the loop)"]
        firsts = [v for k, v in r.items() if k.startswith("first_iteration") and v is not None]
        out.append("| %s, %d x %d%s | %d / %d | %s | %.1e | %.1e (after %s) | %.1e | %.1e / %.1e | %.1e |" % (
            z["workload"].split(":")[0].replace("BASELINE ", ""), z["nbin"], z["nlayer"],
            " — convection loop" if loop == "convection_loop" else "", ours[key_n], ref[key_n],
            min(firsts) if firsts else "none", early, sn[worst_k]["T_lay"], worst_k, e["T_lay"], e["F_up_tot"], e["F_down_tot"],
            e["emission spectrum (of its maximum)"]))
    return out


print("| run | iterations (library / reference) | first iteration with a different discrete state | largest T difference "
      "through iteration 400 | largest T difference at any snapshot | end: T | end: F_up_tot / F_down_tot | end: spectrum |")
print("|---|---|---|---|---|---|---|---|")
for p in sys.argv[1:]:
    for line in row(p):
        print(line)
