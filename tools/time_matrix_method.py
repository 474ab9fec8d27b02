"""`flux calculation method = matrix` at a given size: iterations of the device-resident loop (hx_rt_flags.matrix) against
the same iterations driven stage by stage from Python (Compute._radiation_loop_stagewise), and against the sweeps.
Each run goes to radiative equilibrium at the given criterion.
    python tools/time_matrix_method.py NBIN NLAYER [CRITERION] [--sweeps-stagewise]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import helios  # noqa: E402
from helios_amd import computation  # noqa: E402

_args = [a for a in sys.argv[1:] if not a.startswith("--")]
nbin, nlayer = _args[0], _args[1]
crit = _args[2] if len(_args) > 2 else "1e-4"
base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "%s 30 20 20242" % nbin,
        "-number_of_layers", nlayer, "-output_directory", "/tmp/mm_out/", "-radiative_equilibrium_criterion", crit,
        "-maximum_number_of_iterations", "100000", "-internal_temperature", "800", "-convective_adjustment", "no",
        "-surface_albedo", "0.1"]
res = {}
runs = [("matrix, device-resident loop", "matrix", True), ("matrix, stage by stage", "matrix", False),
        ("sweeps, device-resident loop", "iteration", True)]
if "--sweeps-stagewise" in sys.argv:      # the price of leaving the device-resident loop (deep columns beyond its tilings)
    runs.append(("sweeps, stage by stage", "iteration", False))
for label, method, fused in runs:
    orig = computation.Compute.__init__

    def patched(self, ctx=None, _o=orig, _f=fused):
        _o(self, ctx)
        self.use_fused = _f
    computation.Compute.__init__ = patched
    loop = computation.Compute.radiation_loop
    spent = {}

    def timed(self, *a, _l=loop, **k):
        self.ctx.synchronize()
        t0 = time.perf_counter()
        r = _l(self, *a, **k)
        self.ctx.synchronize()
        spent["loop"] = time.perf_counter() - t0
        return r
    computation.Compute.radiation_loop = timed
    try:
        q = helios.run_helios(base + ["-name", "mm", "-flux_calculation_method", method])
    finally:
        computation.Compute.__init__ = orig
        computation.Compute.radiation_loop = loop
    res[label] = (spent["loop"], int(q.iter_value), np.asarray(q.T_lay).copy())
for label, (t, n, _) in res.items():
    print("MATRIX_METHOD %s bins x %s layers, %-30s %d iterations in %.3f s = %.2f ms per iteration" %
          (nbin, nlayer, label + ":", n, t, 1e3 * t / max(n, 1)))
a, b = res["matrix, device-resident loop"][2], res["matrix, stage by stage"][2]
print("MATRIX_METHOD largest relative difference of T_lay between the two matrix runs: %.2e" % np.abs(a / b - 1).max())
