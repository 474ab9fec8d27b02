"""Wall-clock of the phases of a full driver run at a given size (tuning aid):  python tools/time_run.py NBIN NLAYER"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helios
from helios_amd import computation, write
nbin, nlayer = sys.argv[1], sys.argv[2]
marks = []
def wrap(cls, name):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        self.ctx.synchronize() if hasattr(self, "ctx") else None
        marks.append((name, time.perf_counter() - t0))
        return r
    setattr(cls, name, f)
for n in ("radiation_loop", "sync_store_from_rt", "convection_loop", "integrate_optdepth_transmission",
          "calculate_contribution_function", "calculate_mean_opacities", "integrate_beamflux"):
    wrap(computation.Compute, n)
wrap(write.Write, "write_all")
t0 = time.perf_counter()
helios.run_helios(["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "%s 30 20 20242" % nbin,
                   "-number_of_layers", nlayer, "-name", "timing", "-output_directory", "/tmp/timing_out/",
                   "-radiative_equilibrium_criterion", "1e-3", "-internal_temperature", "800"])
print("TIMING total %.2f s" % (time.perf_counter() - t0))
for n, t in marks:
    print("TIMING %-36s %.3f s" % (n, t))
