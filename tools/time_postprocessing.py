"""Post-processing run type (isothermal layers, 1000*scat+1 sweeps, one pass) at a given size: the fused path (all
sweeps inside one launch of the flux kernel) against the per-stage entry points (1001 launches of hx_fband_iso).
    python tools/time_postprocessing.py NBIN NLAYER"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import helios  # noqa: E402
from helios_amd import computation  # noqa: E402

nbin, nlayer = sys.argv[1], sys.argv[2]
base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "%s 30 20 20242" % nbin,
        "-number_of_layers", nlayer, "-output_directory", "/tmp/pp_out/", "-radiative_equilibrium_criterion", "1e-3",
        "-internal_temperature", "800", "-convective_adjustment", "no", "-surface_albedo", "0.1"]
helios.run_helios(base + ["-name", "it"])
res = {}
for fused, method in ((True, "iteration"), (False, "iteration"), (True, "matrix")):
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
        q = helios.run_helios(base + ["-name", "pp%d%s" % (fused, method), "-run_type", "post-processing",
                                      "-flux_calculation_method", method,
                                      "-path_to_temperature_file", "/tmp/pp_out/it/it_tp.dat"])
    finally:
        computation.Compute.__init__ = orig
        computation.Compute.radiation_loop = loop
    res[fused if method == "iteration" else "matrix"] = (spent["loop"], q.F_up_band.copy())
print("POSTPROCESSING %s bins x %s layers with the matrix method (one tridiagonal solve per spectral point, device-resident "
      "pass): %.4f s; emission spectrum within %.1e of the 1001 sweeps'" %
      (nbin, nlayer, res["matrix"][0], np.abs(res["matrix"][1][-int(nbin):] / res[True][1][-int(nbin):] - 1).max()))
print("POSTPROCESSING %s bins x %s layers: fused %.3f s, per-stage %.3f s (x%.1f); largest relative difference of the "
      "emission spectrum %.2e" % (nbin, nlayer, res[True][0], res[False][0], res[False][0] / res[True][0],
                                  np.abs(res[True][1] / res[False][1] - 1).max()))
