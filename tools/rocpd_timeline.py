#!/usr/bin/env python3
"""Timeline of consecutive kernel dispatches from a rocprofv3 (ROCm 7.2) rocpd database: duration of every kernel and the
idle gap in front of it -- what a launch-bound iteration is made of.

    python tools/rocpd_timeline.py <results.db> [first] [count]
"""
import sqlite3
import sys


def main(path, first=2000, count=60):
    db = sqlite3.connect(path)
    rows = None
    for q in ("select name, start, end from kernels order by start",
              "select kernel_name, start, end from kernels order by start",
              "select name, start_timestamp, end_timestamp from kernels order by start_timestamp"):
        try:
            rows = list(db.execute(q))
            break
        except sqlite3.Error:
            continue
    if rows is None:
        print("no `kernels` view with (name, start, end); tables and views of this database:")
        for name, sql in db.execute("select name, sql from sqlite_master where type in ('table', 'view')"):
            print(" ", name, (sql or "")[:200].replace("\n", " "))
        return
    first = min(first, max(0, len(rows) - count))
    print("# %d dispatches; %d from number %d on: gap before [us], duration [us], kernel" % (len(rows), count, first))
    prev_end = None
    tot_gap = tot_dur = 0.0
    for name, s, e in rows[first:first + count]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        dur = (e - s) / 1e3
        tot_gap += gap
        tot_dur += dur
        print("%8.2f %8.2f  %s" % (gap, dur, name.split("(")[0][:70]))
        prev_end = e
    span = (rows[first + count - 1][2] - rows[first][1]) / 1e3 if len(rows) >= first + count else 0.0
    print("# span %.1f us: kernels %.1f us, idle between them %.1f us" % (span, tot_dur, tot_gap))


if __name__ == "__main__":
    main(sys.argv[1], *(int(v) for v in sys.argv[2:4]))
