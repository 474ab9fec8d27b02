#!/usr/bin/env python3
"""Headline benchmark: rad-conv iterations/s x (lambda-bins x layers) on BASELINE.json's config 2
(single column per GPU, 10 000 bins x 100 layers x 20 Gauss points, premixed corr-k table,
isotropic scattering, fp64), synthetic tables per SURVEY.md section 8(d).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the radiation_loop body (reference source/computation.py:851-984): interface
temperatures, Planck interpolation, every-10th-iteration opacity/transmission refresh, 3*scat+1
two-stream sweeps, flux integration, temperature step -- all through libhelios_hip.so (hx_rt_step).
Inputs are resident in HBM before the timed region.

ONE workload for every N: `value` is config 2 on every GPU (one column per GPU, each with its own start profile; weak
scaling, no collective on the iteration path, the emission spectra gathered over RCCL once after the timed region), so
the N = 1 member of a 1/2/4/8 series IS the single-GPU headline.  `python bench.py --gpus N` without a launcher starts
the N ranks itself (a child `python -m torch.distributed.run ...`, before this process touches a GPU).

The same run also measures, live, into `secondary`: at N = 1 config 3 (20 species mixed on the fly), config 4 (one GPU's
share of the 512-column sweep: 64 on-the-fly columns in one batch -- the same shape at every N) and config 5 (30 000 x 200,
20 species on the fly, two cloud decks, beam, albedo, I2S); at N > 1 config 4 alone (`--secondary all` adds config 5 from
the convection loop and the default-grid batches).  Synthetic k-tables are formed on the device from their two factors
(hx_rt_set_*_separable): a rank's set-up (`setup_s`) does not depend on the host's core count.  `steady_state_200` is the
headline workload over 200 iterations from iteration 0 (20 refreshes), SURVEY.md 8(d)'s definition of the metric.

OUTPUT.  Rank 0 prints ONE compact JSON line (compact_line(): < 4 KB at every N -- the contract's fields, `config`,
`roofline`, `cpu_baseline`, `steady_state_200`, `process_group`, and per secondary workload value / ms_per_step / roofline
fraction / kernel) as the LAST thing on stdout.  Everything else the run measured (per-kernel times, per-column spectrum
checksums, byte models, the thread probe of the CPU baseline, notes) goes to the side file `--detail` names (default
bench_detail.json next to this script), never to stdout.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the cores this process may use, asked BEFORE any OpenMP runtime is loaded: with OMP_PROC_BIND set (below) the runtime pins the
# initial thread to its first place when it loads, and sched_getaffinity then answers "one core" on a 128-core host (rounds
# 1-5 probed the CPU baseline's thread count up to that answer: two threads)
HOST_CPUS = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

def _cpu_baseline_will_run(argv, env):
    """only the N = 1 process that times the CPU oracle at the end"""
    if int(env.get("WORLD_SIZE", "1")) != 1 or "--no-cpu-baseline" in argv:
        return False
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv) and argv[i + 1] != "1":
            return False
        if a.startswith("--gpus=") and a != "--gpus=1":
            return False
    return True


# the CPU baseline's OpenMP threads stay on the cores they start on, packed onto neighbouring cores (one NUMA domain for the
# thread counts that win): read by the OpenMP runtime when it is loaded, so set before anything loads it -- and ONLY in the
# process that runs the baseline: with binding on, the runtime pins the initial thread to place 0, i.e. the N ranks of a
# multi-GPU run (and every rocprofv3 child pass) would issue their launches from one and the same core
_OMP_SET_HERE = []
if __name__ == "__main__" and _cpu_baseline_will_run(sys.argv[1:], os.environ):     # (never on import: tests import this module)
    for _k, _v in (("OMP_PROC_BIND", "close"), ("OMP_PLACES", "cores")):
        if _k not in os.environ:
            os.environ[_k] = _v
            _OMP_SET_HERE.append(_k)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (nbin, nlayer, ny, ntemp, npress)
    "c2": dict(nbin=10000, nlayer=100, ny=20, ntemp=30, npress=20,
               desc="BASELINE config 2: single column, 10 000 bins x 100 layers, premixed corr-k, isotropic scattering"),
    "c2matrix": dict(nbin=10000, nlayer=100, ny=20, ntemp=30, npress=20, albedo=0.1, matrix=1,
                     desc="config 2's grid with `flux calculation method = matrix` (the direct solve of the two-stream equations as "
                          "three scans inside k_rt_flux<.., true>) and a surface albedo of 0.1"),
    "c1": dict(nbin=300, nlayer=50, ny=20, ntemp=30, npress=20, scat=0,
               desc="BASELINE config 1: single column, 300 bins x 50 layers, premixed, no scattering"),
    "c3": dict(nbin=10000, nlayer=100, ny=20, ntemp=30, npress=20, nspecies=20,
               desc="BASELINE config 3: on-the-fly mixing, 20 species random overlap, 10 000 bins x 100 layers"),
    "c5": dict(nbin=30000, nlayer=200, ny=20, ntemp=12, npress=10, nspecies=20, clouds=2, albedo="smoothed", scat_corr=1,
               g_0=0.3, dir_beam=1,
               desc="BASELINE config 5, one column: 30 000 bins x 200 layers, 20 species mixed on the fly (random overlap), "
                    "non-isotropic scattering (g0 = 0.3, I2S correction), two cloud decks, direct beam at 60 deg, surface "
                    "albedo U(0.05, 0.4); 12 x 10 (T, P) nodes per k-table (11.5 GB of tables)"),
    "c5premixed": dict(nbin=30000, nlayer=200, ny=20, ntemp=12, npress=10, clouds=2, albedo=0.1, scat_corr=1, g_0=0.1,
                       desc="config 5's shape with a premixed table: 30 000 bins x 200 layers, I2S correction, two cloud "
                            "decks, surface albedo, no beam (round 1-2's c5)"),
    "c2beam": dict(nbin=10000, nlayer=100, ny=20, ntemp=30, npress=20, dir_beam=1, albedo=0.1,
                   desc="config 2's grid with the direct stellar beam and a surface albedo (five coefficient planes, "
                        "13 rows per lane)"),
    "c4": dict(nbin=10000, nlayer=100, ny=20, ntemp=30, npress=20, nspecies=20, columns_per_gpu=64, sweep=True,
               desc="BASELINE config 4, one GPU's share of the 512-column sweep at 8 GPUs: 64 columns per GPU (g x a x "
                    "T_intern grid), on-the-fly mixing of 20 species (random overlap), 10 000 bins x 100 layers"),
    "c4small": dict(nbin=500, nlayer=40, ny=20, ntemp=8, npress=6, nspecies=5, columns_per_gpu=6, sweep=True,
                    desc="config 4 shape at 500 bins x 40 layers x 5 species, 6 columns per GPU (quick check)"),
    "c3small": dict(nbin=1000, nlayer=100, ny=20, ntemp=12, npress=10, nspecies=20,
                    desc="config 3 shape at 1000 bins (quick check)"),
    "c2small": dict(nbin=1000, nlayer=100, ny=20, ntemp=12, npress=10,
                    desc="config 2 shape at 1000 bins (quick check)"),
    "c5conv": dict(nbin=30000, nlayer=200, ny=20, ntemp=12, npress=10, nspecies=20, clouds=2, albedo="smoothed", scat_corr=1,
                   g_0=0.3, dir_beam=1, phase="convection",
                   desc="BASELINE config 5, one column, steps taken from the CONVECTION loop (convective adjustment of a deep "
                        "zone, sweeps, layer marking, equilibrium test and temperature step on the device): 30 000 bins x 200 "
                        "layers, 20 species mixed on the fly, I2S correction, two cloud decks, direct beam, surface albedo"),
    "d64": dict(nbin=386, nlayer=105, ny=20, ntemp=120, npress=28, columns_per_gpu=64, sweep=True,
                desc="the reference's DEFAULT grid (R = 50: 386 bins x 20 Gauss points, 105 layers, 120 x 28 (T, P) nodes; "
                     "BASELINE.md section 1) as a batch of 64 sweep columns on one GPU, premixed table, isotropic scattering"),
    "d64s": dict(nbin=386, nlayer=105, ny=20, ntemp=120, npress=28, columns_per_gpu=64, sweep=True, species_dat=True,
                 desc="the reference's default grid (386 bins x 20 Gauss points, 105 layers, 120 x 28 (T, P) nodes) as a batch of 64 "
                      "sweep columns, on-the-fly mixing of the 15 species of the reference's input/species.dat: 13 absorbers -- the "
                      "first and the two CIA pairs correlated-k, ten by random overlap --, Rayleigh scattering by H2O (computed), "
                      "CO2, CO, H2, He"),
}


def build_case(w, seed, full_tables=True):
    """synthetic column + premixed table; attribute names follow the reference's Store"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helios_amd import phys_const as pc
    from helios_amd import synthetic as syn

    class C(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    rng = np.random.default_rng(seed)
    c = C()
    c.nbin, c.nlayer, c.ny, c.ntemp, c.npress = w["nbin"], w["nlayer"], w["ny"], w["ntemp"], w["npress"]
    c.ninterface = c.nlayer + 1
    c.iso, c.scat, c.dir_beam, c.clouds, c.scat_corr, c.geom_zenith_corr = 0, int(w.get("scat", 1)), 0, 0, 0, 0
    c.g_0, c.epsi, c.epsi2, c.i2s_transition = 0.0, 0.5, 0.5, 0.1
    c.w_0_limit, c.w_0_scat_limit, c.delta_tau_limit = 1.0 - 1e-10, 1e-3, 1e-4
    c.f_factor, c.mu_star = 0.5, float(np.cos(np.pi - np.pi / 3.0))
    c.g, c.R_planet, c.R_star, c.a = 1000.0, pc.R_JUP, pc.R_SUN, 0.05 * pc.AU
    c.T_star, c.real_star = 5000.0, 0
    c.F_intern = pc.SIGMA_SB * 100.0 ** 4
    c.plancktable_dim, c.plancktable_step = 8000, 2
    c.rad_convergence_limit, c.adapt_interval, c.foreplay, c.smooth = 1e-8, 20, 0, 0
    c.physical_tstep, c.no_atmo = 0.0, 0
    c.opac_interwave, c.opac_wave, c.opac_deltawave = syn.wavelength_grid(c.nbin)
    c.gauss_y, c.gauss_weight = syn.gauss_points(c.ny)
    c.ktemp, c.kpress = syn.tp_grid(c.ntemp, c.npress)
    # the premixed table by its two factors kxy[x][y], ftp[t][p]: the device forms the 0.6-1 GB array itself
    # (hx_rt_set_premixed_separable; the same bits as synthetic.ktable) where the caller does not need the host array
    # (`full_tables`: tests and the CPU oracle do).
    # (with species the premixed table is never read: draw the same random numbers, skip the array)
    otf = bool(w.get("nspecies") or w.get("species_dat"))
    if otf or full_tables:
        c.opac_k = syn.ktable(rng, 8 if otf else c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y)
        c.opac_k_factors = None
    else:
        c.opac_k, c.opac_k_factors = None, syn.ktable_factors(rng, c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y)
    c.opac_scat_cross = syn.rayleigh_table(c.opac_wave, c.ntemp, c.npress)
    c.opac_meanmass = syn.meanmass_table(c.ntemp, c.npress)
    c.p_lay, c.p_int = syn.pressure_levels(1e9, 1e-1, c.nlayer)
    T_eff = c.f_factor ** 0.25 * (c.R_star / c.a) ** 0.5 * c.T_star
    c.T_lay = np.ones(c.nlayer + 1) * max(T_eff, 500.0)       # host_functions.py:164-176
    c.surf_albedo = np.zeros(c.nbin)
    c.starflux = np.zeros(c.nbin)
    c.c_p_lay = np.full(c.nlayer, 3.5 * pc.R_UNIV)
    if w.get("albedo") == "smoothed":       # U(0.05, 0.4) smoothed over 5 bins
        c.surf_albedo = syn._smooth5(rng.uniform(0.05, 0.4, c.nbin))
    elif w.get("albedo"):
        c.surf_albedo = np.full(c.nbin, float(w["albedo"]))
    if w.get("matrix"):
        c.flux_calc_method = "matrix"
    if w.get("scat_corr"):
        c.scat_corr, c.g_0 = 1, float(w.get("g_0", 0.1))
    if w.get("dir_beam"):
        c.dir_beam = 1                      # zenith angle 60 deg: mu_star above
    if w.get("clouds"):
        c.clouds = 1            # two synthetic decks (bases at 0.1 bar and 1 mbar), SURVEY.md 8(d) config 5
        c.update(syn.cloud_arrays(c.nbin, c.nlayer, c.opac_wave, np.asarray(c.p_lay), np.asarray(c.p_int), rng))
    c.species = None
    if w.get("nspecies") and not w.get("species_dat"):
        # absorbers with molar weights U(2,64) and constant VMRs log-uniform in
        # [1e-8,1e-2], plus H2/He filler (0.85/0.15) as Rayleigh scatterers; mixing method RO
        sp = []
        for s_ in range(w["nspecies"]):
            # the k-table (0.96 GB each at C3 size) is generated when it is uploaded, then dropped
            sp.append(dict(weight=float(rng.uniform(2.0, 64.0)), vmr=float(10.0 ** rng.uniform(-8.0, -2.0)),
                           pretab="lazy", table_seed=seed * 100 + s_, scat=None, is_h2o=False, is_cia=False))
        sp.append(dict(weight=2.016, vmr=0.85, pretab=None, scat=1e-27 * (1e-4 / c.opac_wave) ** 4,
                       is_h2o=False, is_cia=False))
        sp.append(dict(weight=4.0026, vmr=0.15, pretab=None, scat=1e-28 * (1e-4 / c.opac_wave) ** 4,
                       is_h2o=False, is_cia=False))
        c.species = sp
    if w.get("species_dat"):
        # the species list of the reference's input/species.dat (15 entries, in file order): eleven molecules and atoms
        # that absorb (H2O, CO2, CO scatter as well; H2O's Rayleigh cross-section is computed, kernels.cu:3404-3440), H2 and
        # He as scatterers only, two CIA pairs (absorbing, correlated-k by name, not part of the mean molecular mass)
        names = ["H2O", "CO2", "CO", "CH4", "NH3", "HCN", "PH3", "C2H2", "H2S", "Na", "K", "H2", "He", "CIA_H2H2", "CIA_H2He"]
        weights = [18.015, 44.01, 28.01, 16.04, 17.03, 27.03, 34.0, 26.04, 34.08, 22.99, 39.10, 2.016, 4.0026, 4.032, 6.019]
        sp = []
        for k, (nm, wt) in enumerate(zip(names, weights)):
            absorbing, cia = nm not in ("H2", "He"), nm.startswith("CIA")
            scat = None
            if nm in ("CO2", "CO", "H2", "He"):
                scat = (1e-27 if nm != "He" else 1e-28) * (1e-4 / c.opac_wave) ** 4
            vmr = {"H2": 0.85, "He": 0.15, "CIA_H2H2": 0.72, "CIA_H2He": 0.13}.get(nm)
            if vmr is None:
                vmr = float(10.0 ** rng.uniform(-8.0, -3.0))
            sp.append(dict(name=nm, weight=wt, vmr=vmr, pretab="lazy" if absorbing else None, table_seed=seed * 100 + k,
                           scat=scat, is_h2o=nm == "H2O", is_cia=cia))
        c.species = sp
    return c


def sweep_column(c, gi):
    """column `gi` of the 8 x 8 x 8 sweep of SURVEY.md 8(d): g in logspace(2.5, 3.5), a in logspace(-2, -0.5) AU,
    T_intern in linspace(50, 400); everything else as in the single-column case"""
    from helios_amd import phys_const as pc
    i, j, k = (gi // 64) % 8, (gi // 8) % 8, gi % 8
    g = float(np.logspace(2.5, 3.5, 8)[i])
    a = float(np.logspace(-2.0, -0.5, 8)[j]) * pc.AU
    T_intern = float(np.linspace(50.0, 400.0, 8)[k])
    T_eff = c.f_factor ** 0.25 * (c.R_star / a) ** 0.5 * c.T_star
    return dict(g=g, a=a, F_intern=pc.SIGMA_SB * T_intern ** 4, T_start=max(T_eff, 500.0))


def make_batch(ctx, c, ncol, first_column=0, sweep=False):
    from helios_amd.rt import batch_from_case
    sp = c.species
    cols = [sweep_column(c, first_column + i) for i in range(ncol)] if sweep else None
    rt = batch_from_case(ctx, c, ncol=ncol, nspecies=len(sp) if sp else 0, columns=cols)
    if sp:
        from helios_amd import synthetic as syn
        for k, s_ in enumerate(sp):
            if isinstance(s_["pretab"], str):
                # a synthetic k-table (0.96 GB each at C3 size) by its two factors (1.6 MB): the device forms it
                # (hx_rt_set_species_separable) -- 20 of them cost a rank 85 core-seconds of numpy before
                kxy, ftp = syn.ktable_factors(np.random.default_rng(s_["table_seed"]), c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y)
                rt.set_species_separable(k, kxy, ftp, s_["scat"], s_["weight"], is_h2o=2 if s_["is_h2o"] else 0,
                                         is_cia=1 if s_["is_cia"] else 0, in_mu=0 if s_["is_cia"] else 1)
            else:
                rt.set_species(k, s_["pretab"], s_["scat"], s_["weight"], is_h2o=2 if s_["is_h2o"] else 0,
                               is_cia=1 if s_["is_cia"] else 0, in_mu=0 if s_["is_cia"] else 1)
        vl = np.array([np.full(c.nlayer, s_["vmr"]) for s_ in sp])
        vi = np.array([np.full(c.nlayer + 1, s_["vmr"]) for s_ in sp])
        rt.set_column_vmr(-1, vl, vi)
    if cols:
        for i, cp in enumerate(cols):      # pressure grid and start profile follow the column's gravity / orbit
            rt.set_column_profile(i, c.p_lay, c.p_int, np.full(c.nlayer + 1, cp["T_start"]), c.surf_albedo, c.starflux)
    return rt


def native_oracle():
    """the CPU oracle compiled on THIS host with -O3 -march=native; the prebuilt -O2 library
    (which travels from the build container, whose CPU differs) if no compiler is at hand"""
    import subprocess
    import tempfile
    import oracle
    src = os.path.join(ROOT, "oracle", "helios_oracle.c")
    out = os.path.join(tempfile.gettempdir(), "libhelios_oracle_native_%d.so" % os.getuid())
    try:
        subprocess.run(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
                        "-o", out, src, "-lm"], check=True, capture_output=True, timeout=120)
        with open(os.path.join(ROOT, "oracle", "helios_oracle.h")) as f:
            return oracle._CLib(out, f.read(), "orc_"), "-O3 -march=native"
    except Exception:
        return oracle.port, "-O2 (prebuilt)"


def cpu_baseline(w, seed, budget_s=20.0):
    """the CPU oracle (oracle/helios_oracle.c, OpenMP over bins) on a bounded sample of the same workload, from
    iteration 0 (one opacity refresh per 10 iterations): whole decades of iterations, as many as fit about 20 s of CPU
    work at the rate the thread probe measured (at least one decade, at most ten).  Premixed: ALL bins.  On-the-fly
    mixing (the reference's bubble sort of 400 sums per point and species): 4000 of the bins.  The thread count is
    probed (8 ... all usable cores) and the winner stated."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    lib, flags = native_oracle()
    species = bool(w.get("nspecies"))
    nb = min(w["nbin"], 4000 if species else w["nbin"])
    ws = dict(w)
    ws["nbin"] = nb
    c = build_case(ws, seed)
    cc = cases.Case(c)
    cc.z_lay = np.zeros(c.nlayer)
    cc.T_int = np.zeros(c.nlayer + 1)
    for k in ("F_add_heat_lay", "F_add_heat_sum", "F_smooth", "F_smooth_sum"):
        cc[k] = np.zeros(c.nlayer)
    if not c.clouds:
        for nm, n in (("lay", c.nlayer), ("int", c.nlayer + 1)):
            for p in ("abs_cross_all_clouds_", "scat_cross_all_clouds_", "g_0_all_clouds_"):
                cc[p + nm] = np.zeros(n * nb)
    cc.delta_colmass = (c.p_int[:-1] - c.p_int[1:]) / c.g
    cc.delta_col_upper = (c.p_lay - c.p_int[1:]) / c.g
    cc.delta_col_lower = (c.p_int[:-1] - c.p_lay) / c.g
    # a coarse Planck table keeps the (untimed) set-up short; the timed part does not depend on it
    cc.plancktable_dim, cc.plancktable_step = 800, 20
    refresh = cases.refresh_premixed
    if species:
        from helios_amd import synthetic as syn
        sp = []
        for k, s_ in enumerate(c.species):
            tab = s_["pretab"]
            if isinstance(tab, str):
                tab = syn.ktable(np.random.default_rng(s_["table_seed"]), nb, c.ny, c.ktemp, c.kpress, c.gauss_y)
            sp.append(dict(name="S%02d" % k, absorbing=tab is not None, scattering=s_["scat"] is not None,
                           is_h2o=False, is_cia=False, weight=s_["weight"], vmr=s_["vmr"], pretab=tab,
                           scat=s_["scat"]))
        cc.species = sp
        refresh = cases.refresh_onthefly
    ncpu = HOST_CPUS
    T0 = np.array(c.T_lay, float).copy()
    s = cases.alloc_state(cc)
    cases.setup_planck(lib, cc, s)
    # thread-count probe (all cores or a shown optimum): the first refresh + iteration, then three
    # refresh-free iterations per candidate on the same state; the fastest count runs the sample
    lib.set_num_threads(min(ncpu, 64))
    t0 = time.perf_counter()
    cases.radiation_iterations(lib, cc, s, 1, refresh=refresh)
    t_first = time.perf_counter() - t0          # one refresh + one iteration
    probe = {}
    for th in sorted(set(t for t in (2, 4, 8, 16, 32, 64, 96, 128, 192, 256, ncpu) if t <= ncpu)):
        lib.set_num_threads(th)
        t0 = time.perf_counter()
        cases.radiation_iterations(lib, cc, s, 3, start=1, refresh=refresh)
        probe[th] = 3.0 / (time.perf_counter() - t0)
    threads = max(probe, key=probe.get)
    lib.set_num_threads(threads)
    # a decade = one refresh + ten iterations: (t_first - t_it) + 10 t_it with t_it from the probe; as many as fit the budget
    t_it = 1.0 / probe[threads]
    t_decade = max(t_first - t_it, 0.0) * (min(ncpu, 64) / float(threads) if threads < min(ncpu, 64) else 1.0) + 10.0 * t_it
    n_it = 10 * int(max(1, min(10, budget_s // t_decade)))
    cc2 = cases.Case(cc)                      # the sample starts from iteration 0 on a fresh state
    cc2.T_lay = T0
    cc2.T_int = np.zeros(c.nlayer + 1)
    cc2.z_lay = np.zeros(c.nlayer)
    s = cases.alloc_state(cc2)
    cases.setup_planck(lib, cc2, s)
    t0 = time.perf_counter()
    cases.radiation_iterations(lib, cc2, s, n_it, refresh=refresh)
    dt = time.perf_counter() - t0
    return dict(value=n_it / dt * nb * c.nlayer, unit="bin*layer*iterations/s", cores=threads, kind="port",
                host_cores=ncpu, seconds=dt,
                sample_short="%d of %d bins x %d layers x %d Gauss points%s, %d iterations from iteration 0 (refresh every 10), "
                             "oracle/helios_oracle.c %s, OpenMP %d threads, %.1f s"
                             % (nb, w["nbin"], c.nlayer, c.ny, ", %d species" % len(c.species) if species else "", n_it, flags,
                                threads, dt), thread_probe_iterations_per_s={str(k): round(v, 3) for k, v in probe.items()},
                sample="%d of %d bins x %d layers x %d Gauss points%s, %d iterations from iteration 0 (one opacity "
                       "refresh per 10), oracle/helios_oracle.c built %s with OpenMP over bins on %d threads (the fastest "
                       "of %s on %d usable cores, probed on refresh-free iterations; OMP_PROC_BIND=%s OMP_PLACES=%s), %.1f s; "
                       "the value is the rate of the bins that were run (bins are independent: nothing is extrapolated)"
                       % (nb, w["nbin"], c.nlayer, c.ny, ", %d species" % len(c.species) if species else "", n_it,
                          flags, threads, sorted(probe), ncpu, os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), dt))


def load_counters(workload):
    """per-launch counter figures: measured live by this run when it could (LIVE_COUNTERS, filled by live_counters()),
    else those committed under profiles/ from an earlier rocprofv3 --pmc run: (dict, source) or (None, None)"""
    if workload in LIVE_COUNTERS:
        return LIVE_COUNTERS[workload]
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None
    alias = {"c4": "c3"}      # same kernels on the same column shape: per-column figures scale with the column count
    with open(tpath) as f:
        d = json.load(f).get(alias.get(workload, workload))
    src = "profiles/traffic.json" + (" (per column, measured on %s)" % alias[workload] if workload in alias else "")
    return (d, src) if d else (None, None)


LIVE_COUNTERS = {}


def live_counters(workload, counters=("FETCH_SIZE", "WRITE_SIZE"), ncol=1):
    """HBM traffic per launch, measured NOW: this command again as a child under `rocprofv3 --pmc <counter>
    --kernel-trace`, one pass per counter (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in separate passes, both in KiB,
    FETCH_SIZE doubled on gfx950), a short run of the same workload without secondaries.  Returns (dict, source) in the
    form of profiles/traffic.json, or None when the profiler is not there or a pass fails (the committed figures are
    quoted then, and `traffic_source` says so)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None
    per = {}
    with tempfile.TemporaryDirectory(prefix="helios_pmc_") as tmp:
        for cname in counters:
            out = os.path.join(tmp, cname)
            cmd = [prof, "--pmc", cname, "--kernel-trace", "-d", out, "-o", "run", "--", sys.executable,
                   os.path.abspath(__file__), "--workload", workload, "--steps", "10", "--warmup", "10", "--no-cpu-baseline",
                   "--profile-steps", "0", "--secondary", "none", "--live-counters", "off", "--device-warmup-ms", "0",
                   "--columns-per-gpu", str(int(ncol))]
            env = dict(os.environ, TMPDIR=tmp)
            for k in _OMP_SET_HERE:       # the binding is the CPU baseline's, not the profiled children's
                env.pop(k, None)
            for k in [k for k in env if k.startswith(("ROCPROF_", "ROCP_")) or k == "ROCPROFILER_LIBRARY_CTOR"]:
                env.pop(k, None)          # (the child profiler sets its own)
            try:
                p = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True,
                                   timeout=900 if WORKLOADS[workload].get("nspecies") else 240)
            except Exception:
                return None
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if p.returncode != 0 or not dbs:
                return None
            try:
                db = sqlite3.connect(dbs[0])
                rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection "
                                  "where counter_name = ? group by kernel_name", (cname,)).fetchall()
            except Exception:
                return None
            per[cname] = {name: (n, avg) for name, n, avg in rows}

    def pick(key):
        f = [v for k, v in per.get("FETCH_SIZE", {}).items() if key in k]
        w = [v for k, v in per.get("WRITE_SIZE", {}).items() if key in k]
        if not f or not w:
            return None, None
        return (2.0 * f[0][1] + w[0][1]) * 1024.0, f[0][0]
    flux, nflux = pick("k_rt_flux")
    mix, nmix = pick("k_rt_mix_species")
    valu = [v for k, v in per.get("SQ_INSTS_VALU", {}).items() if "k_rt_mix_species" in k]
    d = dict(rt_flux_hbm_bytes_per_launch=flux, rt_mix_hbm_bytes_per_launch=mix,
             rt_mix_valu_instructions_per_launch=valu[0][1] if valu else None, launches_counted=nflux or nmix,
             columns_in_the_measured_batch=int(ncol))
    src = ("measured in this run: `rocprofv3 --pmc %s --kernel-trace -- python3 bench.py --workload %s --steps 10 ...`, one pass "
           "per counter, hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half)"
           % (" | ".join(counters), workload))
    return d, src


def under_a_profiler(env):
    """rocprofv3 (or rocprof) has launched this process: its tool library is preloaded and ROCPROF_* variables configure it"""
    return ("rocprofiler" in env.get("LD_PRELOAD", "") or "ROCPROFILER_LIBRARY_CTOR" in env
            or any(k.startswith(("ROCPROF_", "ROCP_")) for k in env))


def env_knobs(allow_debug):
    """every HELIOS_* tuning variable that is set goes into the line; the profiling knobs that change results
    (HELIOS_RT_DEBUG_*) are refused unless asked for"""
    knobs = {k: v for k, v in sorted(os.environ.items())
             if k.startswith(("HELIOS_RT_", "HELIOS_RO_", "HELIOS_SWEEP_", "HELIOS_BENCH_"))}
    bad = [k for k in knobs if k.startswith("HELIOS_RT_DEBUG_")]
    if bad and not allow_debug:
        raise SystemExit("bench.py: %s set -- these profiling knobs skip work inside the kernels (results are wrong, "
                         "times are not the product's); unset them or pass --allow-debug-knobs" % ", ".join(bad))
    return knobs


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (this process
    has not touched a GPU and never will), relay its output -- rank 0 prints the line -- and leave with its code"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, cwd=ROOT).returncode)


class Ranks(object):
    """the process group of this run (None at N = 1) and what the measurements need from it"""

    def __init__(self, dist, world, rank, coll_device):
        self.dist, self.world, self.rank, self.coll_device = dist, world, rank, coll_device

    def barrier(self, ctx):
        import torch
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def gather_scalars(self, v, dtype=None):
        import torch
        if self.dist is None:
            return [v]
        t = torch.tensor([v], dtype=dtype or torch.float64, device=self.coll_device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [x.item() for x in out]


def kernel_roofline(rt, c, wname, ncol, run, start, profile_steps, step_s):
    """the dominant kernel against the HBM roofline: algorithmic bytes per launch / HIP-event average of its launches on
    the library's stream, in separate short passes behind the timed region (E-only iterations, one refresh on its own)"""
    ctx = rt.ctx
    tm = rt.traffic_model()
    counters, counter_source = load_counters(wname)
    it0 = start + (-start) % 10 + 1                 # right behind a refresh boundary
    run(start, it0 - start)
    ctx.synchronize()
    ctx.timer_start()
    run(it0, 9)                                     # nine iterations without a refresh
    e_only_ms = ctx.timer_stop_ms() / 9.0
    rt.profile(True)
    run(it0 + 9, 1)                                 # iteration index = 0 mod 10: refresh + iteration
    run(it0 + 10, min(profile_steps, 9))            # E-iterations under the event profiler
    rt.profile(False)
    flux_ms, nflux = rt.profile_read("matrix_solve" if c.get("flux_calc_method") == "matrix" else "rt_flux")
    prof = {k: rt.profile_read(k)[0] for k in ("rt_flux", "matrix_solve", "refresh_total", "add_to_mixed_opac", "mixed_scat", "rt_coef",
                                               "opac_interpol", "direct_beam", "rt_nodes", "rt_totals_a", "rt_totals_b")}
    t_only_ms = prof["refresh_total"]
    B_E, B_T = tm["step_algorithmic"], tm["refresh_algorithmic"]      # bytes per launch, all columns
    B = B_E + B_T / 10.0
    mix_bound = bool(c.species) and prof["add_to_mixed_opac"] > 5.0 * flux_ms
    roofline = None
    if nflux and not mix_bound:
        achieved = B_E / (flux_ms * 1e-3) / 1e9
        traffic = counters.get("rt_flux_hbm_bytes_per_launch") if counters else None
        if traffic is not None:
            traffic *= ncol / float(counters.get("columns_in_the_measured_batch", 1))   # (the committed figures: one column)
        roofline = dict(bound="hbm", kernel="k_rt_flux<.., true> (matrix method)" if c.get("flux_calc_method") == "matrix" else "k_rt_flux",
                        achieved=achieved, peak=8000.0, unit="GB/s",
                        frac=achieved / 8000.0, traffic=traffic, traffic_source=counter_source if traffic else None,
                        algorithmic_bytes_per_launch=B_E, actual_bytes_model_per_launch=tm["step_actual"],
                        avg_launch_ms=flux_ms, launches_timed=nflux)
        # what the chip streams at best (MI355X_MICROARCH.md: 6.29 TB/s float4 copy) on the bytes this kernel has to move
        # with the reference's sweep semantics -- coefficient planes + up-flux state read and written
        moved = traffic if traffic else tm["step_actual"]
        roofline["frac_of_achievable"] = dict(
            achieved=moved / (flux_ms * 1e-3) / 1e9, achievable=6290.0, unit="GB/s",
            frac=moved / (flux_ms * 1e-3) / 1e9 / 6290.0, bytes_moved_per_launch=moved,
            note="bytes the kernel moves through the L2's memory side (Infinity-Cache hits included) / its launch time, against the 6.29 TB/s copy rate of the guide")
    elif mix_bound:
        # on-the-fly mixing: the refresh's species kernel is where the time goes.  Against HBM: B_T per launch;
        # against vector issue (what actually binds it): instructions per launch from the committed PMC run
        mix_ms = prof["add_to_mixed_opac"]
        achieved = B_T / (mix_ms * 1e-3) / 1e9
        traffic = counters.get("rt_mix_hbm_bytes_per_launch") if counters else None
        roofline = dict(bound="hbm", kernel="k_rt_mix_species", achieved=achieved, peak=8000.0, unit="GB/s",
                        frac=achieved / 8000.0,
                        traffic=traffic * ncol / float(counters.get("columns_in_the_measured_batch", 1)) if traffic else None,
                        traffic_source=counter_source if traffic else None,
                        algorithmic_bytes_per_launch=B_T, avg_launch_ms=mix_ms, launches_timed=1)
        valu = counters.get("rt_mix_valu_instructions_per_launch") if counters else None
        if valu:
            peak = 1024 * 2.4e9 / 4.0               # SIMDs x clock / 4 cycles per wave64 instruction
            valu = valu / float(counters.get("columns_in_the_measured_batch", 1))
            rate = valu * ncol / (mix_ms * 1e-3)
            roofline["valu_issue"] = dict(achieved=rate / 1e9, peak=peak / 1e9, unit="G wavefront-instructions/s",
                                          frac=rate / peak, instructions_per_launch=valu * ncol, source=counter_source)
    if roofline is not None:
        whole = B / step_s / 1e9                    # B * iterations/s over the whole step
        roofline["whole_step"] = dict(achieved=whole, peak=8000.0, unit="GB/s", frac=whole / 8000.0,
                                      algorithmic_bytes_per_iteration=B)
        roofline["e_only_ms_per_iteration"] = e_only_ms
        roofline["t_only_ms_per_refresh"] = t_only_ms
        roofline["kernels_ms"] = {k: v for k, v in prof.items() if v}
    return roofline


def measure(ctx, ranks, wname, ncol, steps, warmup, seed, phase="radiative", profile_steps=20, steady_state=False,
            device_warmup_ms=25.0):
    """one workload on this rank's GPU: W warm-up steps, K timed steps between barriers (max over ranks), roofline
    passes, the gather of the emission spectra.  Returns the fields of its line."""
    import torch
    w = WORKLOADS[wname]
    rank, world = ranks.rank, ranks.world
    sweep = bool(w.get("sweep"))
    err = None
    t_setup = time.perf_counter()
    try:   # everything that allocates: a batch that does not fit one rank's GPU must not leave the others in a collective
        c = build_case(w, seed, full_tables=False)
        rt = make_batch(ctx, c, ncol, first_column=rank * ncol, sweep=sweep)
        # every column of a run differs: the sweep's own parameters, or a few per cent in the start profile
        T_start = [c.T_lay * (1.0 + 0.01 * (rank * ncol + i)) for i in range(ncol)]
        if not sweep and (ncol > 1 or world > 1):
            for i in range(ncol):
                rt.set_temperatures(i, T_start[i])
        rt.build_planck_table(1)
        ctx.synchronize()
    except Exception as e:
        err = e
    setup_s = time.perf_counter() - t_setup      # synthetic inputs, tables formed on the device, uploads, the Planck table
    failed = [r for r, v in enumerate(ranks.gather_scalars(0.0 if err is None else 1.0)) if v]
    if failed:   # the same decision on every rank, before the first barrier of the measurement
        raise RuntimeError("workload %s: rank(s) %s could not set up the batch%s"
                           % (wname, failed, ": %s: %s" % (type(err).__name__, err) if err is not None else ""))
    run = rt.run
    if phase == "convection":
        # a profile with a super-adiabatic interior, so that every iteration adjusts a deep convective zone
        from helios_amd import phys_const as pc
        kap = 2.0 / 7.0
        T = np.maximum(2500.0 * (np.asarray(c.p_lay) / c.p_lay[0]) ** 0.4, 600.0)
        for i in range(ncol):
            rt.set_temperatures(i, np.append(T, 2600.0) * (1.0 + 0.01 * (rank * ncol + i)))
        L = c.nlayer
        for name, v in (("kappa_lay", np.full(L, kap)), ("kappa_int", np.full(L + 1, kap)),
                        ("c_p_lay", np.full(L, pc.R_UNIV / kap)), ("conv_layer", np.zeros(L + 1, np.int32)),
                        ("conv_unstable", np.zeros(L + 1, np.int32)), ("dampara", np.array([-1.0]))):
            rt.set_state(-1, name, v)
        run = rt.conv_run
    # warm-up: W untimed steps starting at iteration 0 (includes the first refresh) ...
    tw = time.perf_counter()
    run(0, warmup)
    ctx.synchronize()
    # ... and, when those took less than 25 ms, further untimed decades until the device has been busy that long: a GPU
    # that comes out of idle runs its first milliseconds slower (20 timed steps after 5 warm-up steps read 0.440 ms,
    # after 45 0.417 ms, after 205 0.417 ms at config 2).  Whole decades, so that the timed window sees the opacity
    # refreshes where it would have seen them; reported as device_warmup_iterations.
    # The device warm-up also brings the run to a refresh boundary and through one whole decade from there, so that the timed
    # region starts at an iteration index = 0 mod 10 (exactly one refresh per ten timed steps, as the metric is defined) and
    # replays the decade graph the warm-up has captured -- 20 timed steps at config 2 are 8 ms, a capture inside them is 4 ms.
    device_warmup = 0
    if device_warmup_ms > 0:
        device_warmup = (-warmup) % 10 + 10
        run(warmup, device_warmup)
        ctx.synchronize()
    while time.perf_counter() - tw < device_warmup_ms * 1e-3 and device_warmup < 200:
        run(warmup + device_warmup, 10)
        ctx.synchronize()
        device_warmup += 10
    most = int(max(ranks.gather_scalars(device_warmup, torch.int64)))   # every rank the same number of iterations
    if most > device_warmup:
        run(warmup + device_warmup, most - device_warmup)
        device_warmup = most
    start = warmup + device_warmup
    ranks.barrier(ctx)
    t0 = time.perf_counter()
    ctx.timer_start()
    run(start, steps)
    ev_ms = ctx.timer_stop_ms()
    ranks.barrier(ctx)
    dt_local = time.perf_counter() - t0
    all_dt = ranks.gather_scalars(dt_local)
    dt = max(all_dt)
    rank_ms = [t / steps * 1e3 for t in all_dt]
    done = [int(rt.get("done", i)[0]) for i in range(ncol)]

    roofline = None
    if profile_steps > 0:
        roofline = kernel_roofline(rt, c, wname, ncol, run, start + steps, profile_steps, dt / steps)

    # the path's only exchange: gather the emission spectra of all columns once, after the run
    # (from the library's band array on the device: strided view -> one [columns, bins] tensor -> all-gather; the host
    # sees the result only)
    from helios_amd.parallel import gather_spectra, emission_spectra_on_device
    ctx.synchronize()
    spec = emission_spectra_on_device(rt, ncol)
    gather_ms = None
    if ranks.dist is not None:
        torch.cuda.synchronize()
        tg = time.perf_counter()
        spec = gather_spectra(spec, ranks.dist, device=ranks.coll_device)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
    else:
        spec = spec.cpu().numpy()
    assert spec.shape == (ncol * world, c.nbin)

    total_cols = ncol * world
    flux_policy = rt.get("flux_launch_policy")
    graphs = rt.get("graph_replays")
    graph_builds = rt.get("graph_builds")
    out = {
        "value": steps / dt * c.nbin * c.nlayer * total_cols, "unit": "bin*layer*iterations/s",
        "steps": steps, "warmup": warmup, "device_warmup_iterations": device_warmup, "ms_per_step": dt / steps * 1e3,
        "config": {"workload": w["desc"], "name": wname, "nbin": c.nbin, "nlayer": c.nlayer, "ny": c.ny,
                   "species": len(c.species) if c.species else 0,
                   "opacity_table_TP_grid": [c.ntemp, c.npress], "columns_per_gpu": ncol,
                   "columns_total": total_cols, "refresh_every": 10, "sweeps_per_iteration": 3 * c.scat + 1,
                   "loop": phase, "parallelism": "columns sharded, %d per GPU" % ncol,
                   # what the batch chose for k_rt_flux: launches walking the grid back and forth,
                   # MiB of up-flux state the tail of a launch leaves in the Infinity Cache
                   "flux_launch_policy": {"back_and_forth": bool(flux_policy[0]), "state_cached_mib": float(flux_policy[1])},
                   # small grids: iterations replayed as hipGraphs (runs of nine refresh-free iterations, whole decades with
                   # their refresh), counted over the whole measurement
                   "graph_replays": {"nine_iterations": int(graphs[0]), "decades": int(graphs[1]), "in_use": bool(graphs[2]),
                                     "builds": int(graph_builds[0]) + int(graph_builds[1])}},
        "iterations_per_s_per_column": steps / dt,
        "setup_s": setup_s, "setup_s_per_rank": ranks.gather_scalars(setup_s),
        "stream_event_ms_per_step": ev_ms / steps,
        "per_rank_ms_per_step": rank_ms,
        "rank_imbalance": (max(rank_ms) - min(rank_ms)) / max(rank_ms),
        "columns_converged_during_run": int(sum(done)),
        "spectra_gather_ms": gather_ms,
        "spectrum_checksum": float(np.sum(spec)),
        # per column, in global column order (rank-major blocks): shows that the gather put every rank's rows where they belong
        "spectrum_checksum_per_column": [float(v) for v in spec.sum(axis=1)] if total_cols <= 64 else None,
        "roofline": roofline,
    }
    if steady_state and phase == "radiative":
        # the metric over >= 200 iterations from iteration 0, i.e. with exactly one refresh per 10
        # iterations -- the column back in its start state (temperatures, zeroed flux and time-step state), 200
        # iterations in one go between barriers
        n_it = 200
        for i in range(ncol):
            rt.set_temperatures(i, T_start[i] if not sweep else np.full(c.nlayer + 1, sweep_column(c, rank * ncol + i)["T_start"]))
        rt.set_state(-1, "restart", np.array([1], np.int32))
        ranks.barrier(ctx)
        t0 = time.perf_counter()
        run(0, n_it)
        ranks.barrier(ctx)
        dts = max(ranks.gather_scalars(time.perf_counter() - t0))
        conv = [int(rt.get("done", i)[0]) for i in range(ncol)]
        out["steady_state_200"] = dict(iterations=n_it, from_iteration=0, refreshes=n_it // 10,
                                       ms_per_iteration=dts / n_it * 1e3,
                                       value=n_it / dts * c.nbin * c.nlayer * total_cols,
                                       unit="bin*layer*iterations/s", columns_converged_during_run=int(sum(conv)))
    rt.close()
    return out


def _sig(v, digits=6):
    """floats at `digits` significant digits (the side file keeps full precision), containers recursively"""
    if isinstance(v, float):
        return float("%.*g" % (digits, v)) if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


LINE_LIMIT = 4000        # bytes of the ONE line on stdout; tests/test_bench_cli.py holds compact_line() to it at N = 1 and N = 8


def compact_line(full, detail_path=None):
    """the line the driver parses, from the full record of the run: the contract's fields as they are (value, ms_per_step at
    full precision), everything else reduced to what a reader of the line needs and rounded to six digits"""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    cfg = full["config"]
    line["config"] = {"workload": cfg["workload"][:160], "name": cfg.get("name"), "nbin": cfg["nbin"], "nlayer": cfg["nlayer"],
                      "ny": cfg["ny"], "species": cfg["species"], "columns_per_gpu": cfg["columns_per_gpu"],
                      "columns_total": cfg["columns_total"], "loop": cfg["loop"], "parallelism": cfg["parallelism"]}

    def roof(r):
        if not r:
            return None
        out = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                     "launches_timed", "algorithmic_bytes_per_launch")}
        if r.get("whole_step"):
            out["whole_step_frac"] = r["whole_step"]["frac"]
        if r.get("frac_of_achievable"):
            out["frac_of_copy_rate"] = r["frac_of_achievable"]["frac"]
        if r.get("valu_issue"):
            out["valu_issue_frac"] = r["valu_issue"]["frac"]
        out["traffic_live"] = bool(r.get("traffic_source") and r["traffic_source"].startswith("measured in this run"))
        return _sig(out)
    line["roofline"] = roof(full.get("roofline"))
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else _sig({"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                                     "host_cores": cb.get("host_cores"), "seconds": cb.get("seconds"),
                                                     "sample": cb.get("sample_short") or cb["sample"][:200]})
    ss = full.get("steady_state_200")
    line["steady_state_200"] = None if not ss else _sig({"ms_per_iteration": ss["ms_per_iteration"], "value": ss["value"]})
    line["process_group"] = _sig(full.get("process_group"))
    for k in ("per_rank_ms_per_step", "rank_imbalance", "spectra_gather_ms", "spectrum_checksum", "setup_s",
              "device_warmup_iterations", "columns_converged_during_run"):
        line[k] = _sig(full.get(k), 5 if k == "per_rank_ms_per_step" else 9 if k == "spectrum_checksum" else 6)
    gr = cfg.get("graph_replays") or {}
    line["graphs"] = {"replays": int(gr.get("nine_iterations", 0)) + int(gr.get("decades", 0)), "builds": gr.get("builds")}
    sec = {}
    for name, r in (full.get("secondary") or {}).items():
        if "error" in r:
            sec[name] = {"error": str(r["error"])[:120]}
            continue
        rr = r.get("roofline") or {}
        sec[name] = _sig({"value": r["value"], "ms_per_step": r["ms_per_step"], "frac": rr.get("frac"), "kernel": rr.get("kernel"),
                          "kernel_ms": rr.get("avg_launch_ms"),
                          "whole_step_frac": (rr.get("whole_step") or {}).get("frac")})
    line["secondary"] = sec or None
    if full.get("env_knobs"):
        line["env_knobs"] = full["env_knobs"]
    line["detail"] = detail_path
    return line


def emit(full, detail_path):
    """side file first, then the one line -- the last thing this process writes to stdout"""
    written = None
    if detail_path:
        try:
            tmp = detail_path + ".tmp%d" % os.getpid()
            with open(tmp, "w") as f:
                json.dump(full, f, indent=1)
            os.replace(tmp, detail_path)
            written = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT + os.sep) else detail_path
        except OSError as e:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (detail_path, e))
    text = json.dumps(compact_line(full, written), separators=(",", ":"))
    if len(text) > LINE_LIMIT:       # never happens with the fields above; if it does, the secondaries go first
        slim = compact_line(dict(full, secondary=None), written)
        slim["secondary_dropped_for_size"] = sorted(full.get("secondary") or {})
        text = json.dumps(slim, separators=(",", ":"))
    sys.stdout.flush()
    sys.stderr.flush()
    print(text, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="headline workload (default c2, the configuration the metric is quoted on, at every N)")
    ap.add_argument("--columns-per-gpu", type=int, default=None)
    ap.add_argument("--secondary", default="default",
                    help="'default' (N = 1: c3,c4,c5; N > 1: c4), 'all' (N = 1: c3,c4,c5,c5conv,d64,d64s), 'none', or a "
                         "comma-separated list of workloads measured in the same run into the line's `secondary` block")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="side file with the full record of the run (the line on stdout is its compact form); '' = none")
    ap.add_argument("--full-line", action="store_true",
                    help="print the full record instead of the compact line (the A/B scripts under tools/ read per-kernel "
                         "times from it; never what the driver runs)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0,
                    help="CPU work the oracle's timed sample is sized for (whole decades of iterations)")
    ap.add_argument("--init-timeout", type=float, default=180.0,
                    help="N > 1: seconds the process group's set-up, roll call and first collective may take each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--allow-debug-knobs", action="store_true")
    ap.add_argument("--live-counters", default="headline", choices=["off", "headline", "all"],
                    help="N = 1 only: measure the HBM traffic (and, for mixing workloads, the vector instructions) per launch "
                         "in this run -- child passes of this command under rocprofv3 --pmc -- for the headline workload, or "
                         "for every workload of the line ('all': about a minute more per species workload)")
    ap.add_argument("--device-warmup-ms", type=float, default=25.0,
                    help="untimed decades of iterations are added to the warm-up until the device has been busy this long "
                         "(0: exactly --warmup iterations, so that two runs do the same iterations)")
    ap.add_argument("--phase", default="radiative", choices=["radiative", "convection"],
                    help="which loop a step is taken from: radiation_loop (default, the headline number) or "
                         "convection_loop (convective adjustment + sweeps + temperature step, all on the device)")
    args = ap.parse_args()
    knobs = env_knobs(args.allow_debug_knobs)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    wname = args.workload or "c2"
    w = WORKLOADS[wname]
    heavy = bool(w.get("nspecies")) and w["nbin"] >= 5000
    steps = args.steps if args.steps is not None else (20 if heavy else 200)
    warmup = args.warmup if args.warmup is not None else (10 if heavy else 20)
    ncol = args.columns_per_gpu or w.get("columns_per_gpu", 1)
    if args.secondary in ("default", "all"):
        # (config 4 with the SAME shape at every N: one GPU's share of the 512-column sweep, 64 columns in one batch)
        # ('all' adds the regime HELIOS users run: the reference's default grid as a 64-column batch, premixed and with
        # species.dat's list, and config 5's steps taken from the convection loop)
        n1 = ["c3", "c4", "c5"] + (["c5conv", "d64", "d64s"] if args.secondary == "all" else [])
        secondary = (n1 if world == 1 else ["c4"]) if wname == "c2" and args.phase == "radiative" else []
    else:
        secondary = [x for x in args.secondary.split(",") if x and x != "none"]
    for x in secondary:
        if x not in WORKLOADS:
            raise SystemExit("unknown secondary workload %r" % x)

    if world == 1 and args.live_counters != "off" and under_a_profiler(os.environ):
        # `rocprofv3 ... -- python3 bench.py`: no profiler children inside a profiled run (they would inherit its preloaded tool
        # library and counter configuration); the committed per-launch figures are quoted instead (`traffic_live`: false)
        sys.stderr.write("bench.py: running under a profiler -- HBM traffic is quoted from profiles/traffic.json, not counted in this run\n")
        args.live_counters = "off"
    if world == 1 and args.live_counters != "off":
        # before this process touches the GPU: the profiler's children have the device to themselves
        for x in [wname] + (secondary if args.live_counters == "all" else []):
            species = bool(WORKLOADS[x].get("nspecies"))
            got = live_counters(x, ("FETCH_SIZE", "WRITE_SIZE") + (("SQ_INSTS_VALU",) if species else ()),
                                ncol if x == wname else WORKLOADS[x].get("columns_per_gpu", 1))
            if got is not None:
                LIVE_COUNTERS[x] = got
    # test hook for single-GPU machines: HELIOS_BENCH_BACKEND=gloo runs all ranks on GPU 0 with CPU-side collectives,
    # so that the multi-rank logic can be exercised where RCCL cannot (one GPU cannot host two RCCL ranks)
    backend = os.environ.get("HELIOS_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else 0
    coll_device = "cuda" if backend == "nccl" else "cpu"
    # before the first GPU call of this rank: onto the host cores of its GPU's NUMA node (plain sysfs reads and
    # sched_setaffinity; at N = 1 the CPU baseline keeps all cores)
    from helios_amd import parallel
    affinity = parallel.bind_to_gpu_numa_node(device_index, apply=world > 1)
    import torch  # first, so that its bundled HIP runtime is the one libhelios_hip.so binds to
    dist = None
    init_times = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        init_times = parallel.init_process_group_checked(dist, backend, rank, world, device=torch.device("cuda", device_index),
                                                         timeout_s=args.init_timeout)
    ranks = Ranks(dist, world, rank, coll_device)

    from helios_amd.device import Context
    seed = 20240 + 2
    ctx = Context(device_index)
    head = measure(ctx, ranks, wname, ncol, steps, warmup, seed, phase=args.phase, profile_steps=args.profile_steps,
                   steady_state=True, device_warmup_ms=args.device_warmup_ms)
    line = {"metric": "rad-conv iterations/sec x (lambda-bins x layers)", "value": head.pop("value"),
            "unit": head.pop("unit"), "n_gpus": world, "steps": head.pop("steps"), "warmup": head.pop("warmup"),
            "device_warmup_iterations": head.pop("device_warmup_iterations"), "ms_per_step": head.pop("ms_per_step"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    line.update(head)
    line["env_knobs"] = knobs
    # what the collective layer itself reports: the backend of the process group and the number of ranks it connected
    # ("nccl" is RCCL on ROCm; the gloo hook of the single-GPU tests says so here)
    def rccl_version():
        try:
            v = torch.cuda.nccl.version()
            return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception as e:      # (the line must not depend on this accessor)
            return "unavailable: %s" % type(e).__name__
    # (per rank: the NUMA node its process was bound to before the first GPU call, -1 = not bound, and the cores it kept)
    numa = ranks.gather_scalars(float(affinity["numa_node"]) if affinity["bound"] else -1.0)
    cpus = ranks.gather_scalars(float(affinity["cpus"] or 0))
    line["process_group"] = None if dist is None else {
        "backend": str(dist.get_backend()), "world_size": int(dist.get_world_size()),
        "rccl_world": int(dist.get_world_size()) if str(dist.get_backend()) == "nccl" else None,
        "rccl_version": rccl_version() if str(dist.get_backend()) == "nccl" else None,
        "ranks_seen_by_all_gather": len(ranks.gather_scalars(float(rank))),
        "numa_node_per_rank": [int(v) for v in numa], "cpus_per_rank": [int(v) for v in cpus],
        "init_s": init_times["init_s"], "roll_call_s": init_times["roll_call_s"],
        "first_collective_s": init_times["first_collective_s"]}
    line["host_affinity"] = affinity
    # the other configurations, measured live in this run on every rank (weak scaling like the headline)
    sec = {}
    for x in secondary:
        wx = WORKLOADS[x]
        hx_ = bool(wx.get("nspecies")) and wx["nbin"] >= 5000
        try:
            r = measure(ctx, ranks, x, wx.get("columns_per_gpu", 1), 20 if hx_ else 100, 10, seed,
                        phase=wx.get("phase", "radiative"), profile_steps=args.profile_steps,
                        device_warmup_ms=args.device_warmup_ms)
        except Exception as e:   # the headline above is measured and stands; a secondary that could not run says why
            r = {"error": "%s: %s" % (type(e).__name__, e)}   # (measure() lets the ranks agree on a failed set-up first)
        sec[x] = r
    line["secondary"] = sec or None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(w, seed, args.cpu_baseline_seconds)
    elif rank == 0:
        line["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and args.full_line:
        print(json.dumps(line), flush=True)
    elif rank == 0:
        emit(line, args.detail or None)


if __name__ == "__main__":
    main()
