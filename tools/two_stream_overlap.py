#!/usr/bin/env python3
"""Experiment: config 4's 64 columns per GPU as ONE batch on one stream against TWO half-batches on two
streams, the second five iterations ahead, so that the HBM-bound iterations of one half run under the vector-bound opacity
refresh of the other.     python tools/two_stream_overlap.py [--workload c4] [--columns 64] [--iterations 30]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from helios_amd.device import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c4")
ap.add_argument("--columns", type=int, default=64)
ap.add_argument("--iterations", type=int, default=30)
a = ap.parse_args()
w = bench.WORKLOADS[a.workload]
c = bench.build_case(w, 20242, full_tables=False)
sweep = bool(w.get("sweep"))
n = a.iterations


def batch(ctx, ncol, first):
    rt = bench.make_batch(ctx, c, ncol, first_column=first, sweep=sweep)
    rt.build_planck_table(1)
    return rt


ctxA, ctxB = Context(0), Context(0)
# one batch, one stream
rt = batch(ctxA, a.columns, 0)
rt.run(0, 10)
ctxA.synchronize()
t0 = time.perf_counter()
rt.run(10, n)
ctxA.synchronize()
t_one = time.perf_counter() - t0
chk_one = [float(rt.get("F_up_band", i)[-c.nbin:].sum()) for i in (0, a.columns // 2, a.columns - 1)]
rt.close()
# two half-batches, two streams, the second five iterations ahead
h = a.columns // 2
ra, rb = batch(ctxA, h, 0), batch(ctxB, a.columns - h, h)
ra.run(0, 10)
rb.run(0, 15)
ctxA.synchronize()
ctxB.synchronize()
t0 = time.perf_counter()
ia, ib = 10, 15
for k in range(0, n, 5):              # the host queues ahead of the device on both streams
    ra.run(ia, 5)
    rb.run(ib, 5)
    ia += 5
    ib += 5
ctxA.synchronize()
ctxB.synchronize()
t_two = time.perf_counter() - t0
rb2_extra = 0
chk_a = float(ra.get("F_up_band", 0)[-c.nbin:].sum())
ra.close()
rb.close()
print("TWO_STREAMS %s, %d columns, %d iterations: one batch on one stream %.3f s (%.2f ms per iteration); two half-batches on "
      "two streams, five iterations apart %.3f s (%.2f ms per iteration): %.1f %%" % (
          a.workload, a.columns, n, t_one, 1e3 * t_one / n, t_two, 1e3 * t_two / n, 100.0 * (t_two / t_one - 1.0)))
print("TWO_STREAMS spectrum checksum of column 0 after %d iterations: one batch %.17g, half-batch %.17g" % (10 + n, chk_one[0], chk_a))
