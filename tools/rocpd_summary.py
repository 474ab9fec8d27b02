#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.2) rocpd sqlite database into the plain-text per-kernel summary that is
committed under profiles/ (the `--stats` view: calls, total, average, share).

    python tools/rocpd_summary.py gpurun_out/prof_c2/c2_results.db > profiles/r01_c2_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats summary of %s (durations in microseconds)" % path)
    print("%-90s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "share%"))
    for name, calls, total, avg, pct in rows:
        short = name if len(name) <= 90 else name[:87] + "..."
        print("%-90s %8d %14.1f %12.2f %8.2f" % (short, calls, total, avg, pct))
    try:
        pmc = list(db.execute("select * from counters_collection limit 1"))
        if pmc:
            cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
            print("\n# counters_collection columns: %s" % ", ".join(cols))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
