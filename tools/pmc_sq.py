#!/usr/bin/env python3
"""Per-kernel averages of the shader PMC counters collected by tools/pmc_sq.sh (one rocpd database per counter
group), plus a few ratios that say what bounds each kernel: VALU share of issued instructions, fp64 share of the
VALU work, the fraction of wave cycles spent waiting, LDS bank-conflict rate and mean resident waves.

    python tools/pmc_sq.py gpurun_out/r01/sq_0/.../c2_results.db gpurun_out/r01/sq_1/.../c2_results.db ...
"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"\bk_\w+", name)
    return m.group(0) if m else name.split("(")[0].strip()


def main(paths):
    vals = {}
    for p in paths:
        db = sqlite3.connect(p)
        for name, counter, n, avg in db.execute(
                "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                "group by kernel_name, counter_name"):
            vals.setdefault(short(name), {})[counter] = avg
            vals[short(name)]["launches"] = n
    keep = [k for k in vals if k.startswith("k_")]
    keep.sort(key=lambda k: -vals[k].get("SQ_BUSY_CYCLES", 0.0))
    counters = sorted({c for k in keep for c in vals[k] if c != "launches"})
    for k in keep:
        v = vals[k]
        print("== %s  (launches %d)" % (k, v["launches"]))
        for c in counters:
            if c in v:
                print("   %-28s %16.1f" % (c, v[c]))
        g = lambda c: v.get(c) or float("nan")
        insts = g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU") + g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR") + g("SQ_INSTS_LDS") + g("SQ_INSTS_SMEM")
        f64 = g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_TRANS_F64")
        print("   -- VALU / counted instructions     %.3f" % (g("SQ_INSTS_VALU") / insts))
        print("   -- fp64 arithmetic / VALU          %.3f" % (f64 / g("SQ_INSTS_VALU")))
        print("   -- VALU instructions per wave      %.1f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
        print("   -- wait cycles / wave cycles       %.3f" % (g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ_ACTIVE_* counters count quad-cycles summed over all SIMDs:
        # 1024 SIMDs * (GRBM / 8) / 4 quad-cycles are available per launch
        print("   -- VALU-active share of SIMD time  %.3f" % (g("SQ_ACTIVE_INST_VALU") / (g("GRBM_GUI_ACTIVE") * 32.0)))
        print("   -- LDS bank conflict / LDS active  %.3f" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
        print("   -- mean resident waves (LEVEL/CYC) %.2f" % (g("SQ_LEVEL_WAVES") / g("SQ_CYCLES")))


if __name__ == "__main__":
    main(sys.argv[1:])
