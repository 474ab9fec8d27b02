#!/usr/bin/env python3
"""Extracts HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
SEPARATE runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) and writes profiles/traffic.json,
which bench.py quotes as `roofline.traffic`.

Units and corrections (same guide, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a coalesced streaming read, so it is doubled.  The correction is checked on
this code base: k_rt_flux's doubled FETCH_SIZE (1.37 GB) equals the bytes of the planes it streams
(3 coefficient planes + the up-flux state, 1.35 GB), and WRITE_SIZE needs no correction (k_rt_coef:
978125 KiB counted = 3 planes x 332.8 MB + 3.2 MB written, exact).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/c2_results.db gpurun_out/pmc_write/c2_results.db c2
"""
import json
import os
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection "
                      "where counter_name = ? group by kernel_name", (counter,))
    return {name: (n, avg) for name, n, avg in rows}


def main(fetch_db, write_db, workload, valu_db=None):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    v = per_kernel(valu_db, "SQ_INSTS_VALU") if valu_db else {}
    out = {}
    for name in sorted(set(f) | set(w)):
        short = name.split("(")[0].replace("void ", "").strip()
        fk = f.get(name, (0, 0.0))[1]
        wk = w.get(name, (0, 0.0))[1]
        out[short] = dict(launches=f.get(name, w.get(name))[0], FETCH_SIZE_KiB_raw=fk, WRITE_SIZE_KiB_raw=wk,
                          hbm_bytes_per_launch=2.0 * fk * 1024.0 + wk * 1024.0)
        if name in v:
            out[short]["SQ_INSTS_VALU_per_launch"] = v[name][1]      # wavefront instructions, summed over the chip
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    flux = [v for k, v in out.items() if "k_rt_flux" in k]
    mix = [v_ for k, v_ in out.items() if "k_rt_mix_species" in k]
    data[workload] = dict(rt_flux_hbm_bytes_per_launch=flux[0]["hbm_bytes_per_launch"] if flux else None,
                          rt_mix_hbm_bytes_per_launch=mix[0]["hbm_bytes_per_launch"] if mix else None,
                          rt_mix_valu_instructions_per_launch=mix[0].get("SQ_INSTS_VALU_per_launch") if mix else None,
                          correction="hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half)",
                          kernels=out)
    json.dump(data, open(path, "w"), indent=1)
    for k, v in out.items():
        print("%-40s launches %4d  fetch(raw) %12.1f KiB  write %12.1f KiB  -> %8.1f MB/launch"
              % (k[:40], v["launches"], v["FETCH_SIZE_KiB_raw"], v["WRITE_SIZE_KiB_raw"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "c2", sys.argv[4] if len(sys.argv) > 4 else None)
