"""How many iterations config 2's radiation loop needs under the knobs that change nothing but rounding
(lane count of the flux kernel, bin chunks of the totals): the sensitivity of the exit iteration.

    python tools/convergence_count.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r)
import bench
from helios_amd.device import Context
from helios_amd.rt import batch_from_case
c0 = bench.build_case(bench.WORKLOADS["c2"], 20242)
ctx = Context(0)
rt = batch_from_case(ctx, c0, ncol=1)
rt.build_planck_table(1)
it = 0
while it < 5000:
    rt.run(it, 10); it += 10
    if int(rt.get("done", 0)[0]): break
print(int(rt.get("iters_done", 0)[0]))
''' % ROOT

for env in ({}, {"HELIOS_RT_K": "32"}, {"HELIOS_RT_GENERIC_SCANS": "1"}, {"HELIOS_RT_NCHUNK": "312"}, {"HELIOS_RT_NCHUNK": "100"},
            {"HELIOS_RT_MAXTHREADS": "320"}):
    p = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True)
    print("%-36s iterations to convergence: %s" % (env or "defaults", p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]))
