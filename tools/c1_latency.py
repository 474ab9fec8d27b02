import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from helios_amd.device import Context
ctx = Context(0)
c = bench.build_case(bench.WORKLOADS["c1"], 20241)
rt = bench.make_batch(ctx, c, 1)
rt.build_planck_table(1)
rt.run(0, 40); ctx.synchronize()
host, total = [], []
for it in range(40, 80):
    t0 = time.perf_counter(); rt.run(it, 1); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e6); total.append((t2 - t0) * 1e6)
print("per call, synchronised after each: host enqueue us", np.round(host, 1))
print("total us", np.round(total, 1))
# decades in one call, not synchronised in between
for n in (10, 100, 500):
    ctx.synchronize(); t0 = time.perf_counter(); rt.run(80, n); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
    print("run(80, %d): host %.1f us per iteration, total %.1f us per iteration" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# nine refresh-free iterations only
ctx.synchronize(); t0 = time.perf_counter(); rt.run(81, 9); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
print("run(81, 9): host %.1f us, total %.1f us" % ((t1 - t0) * 1e6, (t2 - t0) * 1e6))
ctx.synchronize(); t0 = time.perf_counter(); rt.run(90, 1); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
print("run(90, 1) (refresh): host %.1f us, total %.1f us" % ((t1 - t0) * 1e6, (t2 - t0) * 1e6))
rt.close()
