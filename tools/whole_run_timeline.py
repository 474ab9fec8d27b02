#!/usr/bin/env python3
"""Where the wall clock of a WHOLE run goes (measurement only): helios.py to equilibrium at BASELINE config 2 and config 3,
seconds per phase in the order of the reference's run_helios (helios.py:35-137) -- read, host set-up, upload, Planck table,
radiation loop, convection loop, post-loop diagnostics, copy-back, write -- so that the ratio "iterations : everything else"
is on record next to the iteration rate.

    python tools/whole_run_timeline.py [--out profiles/r05_whole_run_timeline.json] [--configs c2,c3] [--criterion 1e-8]

Every method of Read / Store / Compute / Write, every function of host_functions and RTBatch.build_planck_table is wrapped by
a timer that books its EXCLUSIVE time (nested wrapped calls booked to themselves; the device is synchronised when a wrapped
call of the driver's top level returns) to one of the phases; nothing in the product is changed.
Config 3's twenty absorbers are read from .npz containers written here (reference dataset names); to keep 19 GB off the box's
disk they are tabulated on 6 x 5 (T, P) nodes instead of 30 x 20 -- the refresh does not depend on the table's node count,
the read and upload phases scale with it (stated in the output)."""
import argparse
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

PHASE_OF = {}        # qualified name -> phase
PHASES = ["read", "host set-up", "upload", "Planck table", "radiation loop", "convection loop", "post-loop diagnostics",
          "copy-back", "write", "other"]


class Clock(object):
    def __init__(self):
        self.stack, self.excl, self.calls, self.sync = [], {}, {}, None

    def wrap(self, owner, name, phase, qual):
        orig = getattr(owner, name)
        clock = self

        def timed(*a, **k):
            t0 = time.perf_counter()
            clock.stack.append(0.0)
            try:
                return orig(*a, **k)
            finally:
                if len(clock.stack) == 1 and clock.sync is not None:
                    clock.sync()
                dt = time.perf_counter() - t0
                inner = clock.stack.pop()
                clock.excl[qual] = clock.excl.get(qual, 0.0) + dt - inner
                clock.calls[qual] = clock.calls.get(qual, 0) + 1
                if clock.stack:
                    clock.stack[-1] += dt
        timed.__name__ = getattr(orig, "__name__", name)
        setattr(owner, name, timed)
        PHASE_OF[qual] = phase


def instrument(clock):
    from helios_amd import computation, host_functions, quantities, read, rt, write

    def methods(cls):
        return [n for n, v in vars(cls).items() if isinstance(v, types.FunctionType) and not n.startswith("__")]
    for n in methods(read.Read):
        clock.wrap(read.Read, n, "read", "Read." + n)
    for n in methods(quantities.Store):
        phase = "upload" if n in ("copy_host_to_device", "allocate_on_device") else "copy-back" if n == "copy_device_to_host" else "host set-up"
        clock.wrap(quantities.Store, n, phase, "Store." + n)
    post = ("integrate_optdepth_transmission", "calculate_contribution_function", "interpolate_entropy", "interpolate_phase_state",
            "calculate_mean_opacities", "integrate_beamflux")
    for n in methods(computation.Compute):
        phase = ("radiation loop" if n in ("radiation_loop", "_radiation_loop_stagewise") else
                 "convection loop" if n in ("convection_loop", "_convection_loop_stagewise") else
                 "post-loop diagnostics" if n in post else
                 "upload" if n in ("make_rt_batch", "_make_rt") else
                 "copy-back" if n in ("sync_store_from_rt", "_pull_vmr") else
                 "Planck table" if n in ("construct_planck_table", "correct_incident_energy") else None)
        if phase:
            clock.wrap(computation.Compute, n, phase, "Compute." + n)
    clock.wrap(rt.RTBatch, "build_planck_table", "Planck table", "RTBatch.build_planck_table")
    for n in methods(write.Write):
        clock.wrap(write.Write, n, "write", "Write." + n)
    for n, v in list(vars(host_functions).items()):
        if isinstance(v, types.FunctionType) and not n.startswith("_"):
            late = n in ("calculate_conv_flux", "calc_F_ratio", "calc_tau_lw_sw", "success_message", "calculate_coupling_convergence")
            clock.wrap(host_functions, n, "write" if late else "host set-up", "host_functions." + n)


def write_config3_inputs(wd, nbin=10000, ny=20, ntemp=6, npress=5, nspecies=20, seed=2024200):
    from helios_amd import synthetic as syn
    names = ["H2O", "CO2", "CO", "CH4", "NH3", "HCN", "PH3", "C2H2", "H2S", "SO2", "NO", "OH", "SiO", "TiO", "VO", "Na", "K", "O3",
             "N2O", "NO2"][:nspecies]
    os.makedirs(os.path.join(wd, "opac"), exist_ok=True)
    rng = np.random.default_rng(seed)
    with open(os.path.join(wd, "species.dat"), "w") as f:
        f.write("species      absorbing       scattering         mixing_ratio\n\n")
        for n in names:
            f.write("%s yes no %.6e\n\n" % (n, 10.0 ** rng.uniform(-8.0, -2.0)))
        f.write("H2 no yes 0.85\n\nHe no yes 0.15\n")
    _, wave, _ = syn.wavelength_grid(nbin)
    gy, _ = syn.gauss_points(ny)
    ktemp, kpress = syn.tp_grid(ntemp, npress)
    for k, n in enumerate(names):
        d = {"kpoints": syn.ktable(np.random.default_rng(seed + k), nbin, ny, ktemp, kpress, gy)}
        if k == 0:
            d.update({"center wavelengths": wave, "ypoints": gy, "temperatures": ktemp, "pressures": kpress,
                      "interface wavelengths": syn.wavelength_grid(nbin)[0], "wavelength width of bins": syn.wavelength_grid(nbin)[2]})
        np.savez(os.path.join(wd, "opac", n + "_opac_ip_kdistr.npz"), **d)
    np.savez(os.path.join(wd, "opac", "scat_cross_sections.npz"), rayleigh_H2=1e-27 * (1e-4 / wave) ** 4,
             rayleigh_He=1e-28 * (1e-4 / wave) ** 4)
    return dict(species=len(names) + 2, absorbers=len(names), table_TP_nodes=[ntemp, npress],
                bytes_of_tables=int(len(names) * nbin * ny * ntemp * npress * 8))


def run(config, criterion, wd):
    import helios
    from helios_amd import computation
    clock = Clock()
    instrument(clock)
    out = os.path.join(wd, "out") + "/"
    common = ["-parameter_file", "/nonexistent", "-number_of_layers", "100", "-name", config, "-output_directory", out,
              "-radiative_equilibrium_criterion", criterion, "-maximum_number_of_iterations", "100000",
              "-convective_adjustment", "no", "-internal_temperature", "100"]
    extra = {}
    if config == "c2":
        argv = common + ["-opacity_mixing", "synthetic", "-synthetic", "10000 30 20 20242"]
    else:
        t0 = time.perf_counter()
        extra = write_config3_inputs(wd)
        extra["seconds_writing_the_input_files_(not_part_of_the_run)"] = time.perf_counter() - t0
        argv = common + ["-opacity_mixing", "on-the-fly", "-path_to_species_file", os.path.join(wd, "species.dat"),
                         "-directory_with_opacity_files", os.path.join(wd, "opac") + "/"]
    orig_init = computation.Compute.__init__

    def init(self, ctx=None):
        orig_init(self, ctx)
        clock.sync = self.ctx.synchronize
    computation.Compute.__init__ = init
    t0 = time.perf_counter()
    q = helios.run_helios(argv)
    total = time.perf_counter() - t0
    per_phase = {p: 0.0 for p in PHASES}
    for qual, t in clock.excl.items():
        per_phase[PHASE_OF.get(qual, "other")] += t
    booked = sum(per_phase.values())
    per_phase["other"] += total - booked          # interpreter time between the wrapped calls, imports
    L, X = int(q.nlayer), int(q.nbin)
    iters = int(q.iter_value)
    loop = per_phase["radiation loop"]
    return dict(config=config, nbin=X, nlayer=L, ny=int(q.ny), criterion=float(criterion), iterations=iters,
                seconds_total=total, seconds_per_phase={p: round(per_phase[p], 4) for p in PHASES},
                share_of_the_radiation_loop=loop / total,
                ms_per_iteration_in_the_loop=1e3 * loop / max(iters, 1),
                bin_layer_iterations_per_s_in_the_loop=iters * X * L / loop if loop > 0 else None,
                bin_layer_iterations_per_s_whole_run=iters * X * L / total,
                largest_items={k: round(v, 4) for k, v in sorted(clock.excl.items(), key=lambda kv: -kv[1])[:12]}, **extra)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "whole_run_timeline.json"))
    ap.add_argument("--configs", default="c2,c3")
    ap.add_argument("--criterion", default="1e-8")
    a = ap.parse_args()
    res = []
    for cfg in a.configs.split(","):
        # one process per configuration: the wrappers are installed once per interpreter
        if len(a.configs.split(",")) > 1:
            import subprocess
            with tempfile.NamedTemporaryFile(suffix=".json") as tf:
                subprocess.run([sys.executable, os.path.abspath(__file__), "--configs", cfg, "--criterion", a.criterion, "--out", tf.name],
                               check=True, stdout=subprocess.DEVNULL)
                res += json.load(open(tf.name))["runs"]
            continue
        with tempfile.TemporaryDirectory(prefix="helios_timeline_") as wd:
            res.append(run(cfg, a.criterion, wd))
    doc = dict(what="helios.py (run_helios) to radiative equilibrium: wall-clock seconds per phase, exclusive times, device synchronised "
                    "at the end of every top-level call; order of the reference's helios.py:35-137", runs=res)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    for r in res:
        print("TIMELINE %s: %d iterations, %.2f s in total; %s" % (r["config"], r["iterations"], r["seconds_total"],
                                                                   ", ".join("%s %.2f" % kv for kv in r["seconds_per_phase"].items())))


if __name__ == "__main__":
    main()
