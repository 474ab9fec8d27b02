"""Wall-clock of a sweep vs the same columns run one by one (tuning aid): python tools/time_sweep.py NCOL NBIN NLAYER"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sweep, helios
ncol, nbin, nlayer = int(sys.argv[1]), sys.argv[2], sys.argv[3]
temps = ",".join(str(int(t)) for t in np.linspace(100, 900, ncol))
base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "%s 30 20 20242" % nbin,
        "-number_of_layers", nlayer, "-name", "ts", "-radiative_equilibrium_criterion", "1e-4", "-convective_adjustment", "no"]
t0 = time.perf_counter()
cols, spec = sweep.main(["-sweep", "internal_temperature=" + temps] + base + ["-output_directory", "/tmp/ts_batch/"])
t_batch = time.perf_counter() - t0
iters = [int(c.iter_value) for c in cols]
t0 = time.perf_counter()
for k, T in enumerate(temps.split(",")[:4]):
    helios.run_helios(base + ["-output_directory", "/tmp/ts_single/", "-name", "s%d" % k, "-internal_temperature", T])
t_single = (time.perf_counter() - t0) / 4
print("TIMING sweep of %d columns: %.2f s (%.3f s per column); single runs: %.3f s per column; iterations %d..%d"
      % (ncol, t_batch, t_batch / ncol, t_single, min(iters), max(iters)))
