#!/usr/bin/env python3
"""Headline benchmark: rad-conv iterations/s x (lambda-bins x layers) on BASELINE.json's config 2
(single column per GPU, 10 000 bins x 100 layers x 20 Gauss points, premixed corr-k table,
isotropic scattering, fp64), synthetic tables per SURVEY.md section 8(d).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the radiation_loop body (reference source/computation.py:851-984): interface
temperatures, Planck interpolation, every-10th-iteration opacity/transmission refresh, 3*scat+1
two-stream sweeps, flux integration, temperature step -- all through libhelios_hip.so (hx_rt_step).
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU, columns sharded
(weak scaling: one column per GPU), no collective on the iteration path; the output spectra are
gathered over RCCL once after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (nbin, nlayer, ny, ntemp, npress)
    "c2": dict(nbin=10000, nlayer=100, ny=20, ntemp=30, npress=20,
               desc="BASELINE config 2: single column, 10 000 bins x 100 layers, premixed corr-k, isotropic scattering"),
    "c1": dict(nbin=300, nlayer=50, ny=20, ntemp=30, npress=20,
               desc="BASELINE config 1 shape: 300 bins x 50 layers, premixed"),
    "c3": dict(nbin=10000, nlayer=100, ny=20, ntemp=30, npress=20, nspecies=20,
               desc="BASELINE config 3: on-the-fly mixing, 20 species random overlap, 10 000 bins x 100 layers"),
    "c5": dict(nbin=30000, nlayer=200, ny=20, ntemp=12, npress=10, clouds=2, albedo=0.1, scat_corr=1,
               desc="BASELINE config 5 shape, one column: 30 000 bins x 200 layers, non-iso scattering (I2S "
                    "correction), two cloud decks, surface albedo; premixed table (the mixing cost is config 3's)"),
    "c3small": dict(nbin=1000, nlayer=100, ny=20, ntemp=12, npress=10, nspecies=20,
                    desc="config 3 shape at 1000 bins (quick check)"),
}


def build_case(w, seed):
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
    c.iso, c.scat, c.dir_beam, c.clouds, c.scat_corr, c.geom_zenith_corr = 0, 1, 0, 0, 0, 0
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
    c.opac_k = syn.ktable(rng, c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y)
    c.opac_scat_cross = syn.rayleigh_table(c.opac_wave, c.ntemp, c.npress)
    c.opac_meanmass = syn.meanmass_table(c.ntemp, c.npress)
    c.p_lay, c.p_int = syn.pressure_levels(1e9, 1e-1, c.nlayer)
    T_eff = c.f_factor ** 0.25 * (c.R_star / c.a) ** 0.5 * c.T_star
    c.T_lay = np.ones(c.nlayer + 1) * max(T_eff, 500.0)       # host_functions.py:164-176
    c.surf_albedo = np.zeros(c.nbin)
    c.starflux = np.zeros(c.nbin)
    c.c_p_lay = np.full(c.nlayer, 3.5 * pc.R_UNIV)
    if w.get("albedo"):
        c.surf_albedo = np.full(c.nbin, float(w["albedo"]))
    if w.get("scat_corr"):
        c.scat_corr, c.g_0 = 1, 0.1
    if w.get("clouds"):
        c.clouds = 1            # two synthetic decks (bases at 0.1 bar and 1 mbar), SURVEY.md 8(d) config 5
        c.update(syn.cloud_arrays(c.nbin, c.nlayer, c.opac_wave, np.asarray(c.p_lay), np.asarray(c.p_int), rng))
    c.species = None
    if w.get("nspecies"):
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
    return c


def make_batch(ctx, c, ncol):
    from helios_amd.rt import batch_from_case
    sp = c.species
    rt = batch_from_case(ctx, c, ncol=ncol, nspecies=len(sp) if sp else 0)
    if sp:
        from helios_amd import synthetic as syn
        for k, s_ in enumerate(sp):
            tab = s_["pretab"]
            if isinstance(tab, str):
                tab = syn.ktable(np.random.default_rng(s_["table_seed"]), c.nbin, c.ny, c.ktemp, c.kpress, c.gauss_y)
            rt.set_species(k, tab, s_["scat"], s_["weight"], is_h2o=0, is_cia=0, in_mu=1)
            del tab
        vl = np.array([np.full(c.nlayer, s_["vmr"]) for s_ in sp])
        vi = np.array([np.full(c.nlayer + 1, s_["vmr"]) for s_ in sp])
        rt.set_column_vmr(-1, vl, vi)
    return rt


def cpu_baseline(w, seed):
    """the CPU oracle (oracle/helios_oracle.c, OpenMP over bins) on a bounded sample of the same
    workload: the first `nb` bins, 20-60 iterations from iteration 0 (one opacity refresh per 10), sized
    for roughly 10-30 s; the thread count is the fastest of a short probe."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import oracle
    nb = min(w["nbin"], 5000)
    ws = dict(w)
    ws["nbin"] = nb
    c = build_case(ws, seed)
    cc = cases.Case(c)
    cc.z_lay = np.zeros(c.nlayer)
    cc.T_int = np.zeros(c.nlayer + 1)
    for k in ("F_add_heat_lay", "F_add_heat_sum", "F_smooth", "F_smooth_sum"):
        cc[k] = np.zeros(c.nlayer)
    for nm, n in (("lay", c.nlayer), ("int", c.nlayer + 1)):
        for p in ("abs_cross_all_clouds_", "scat_cross_all_clouds_", "g_0_all_clouds_"):
            cc[p + nm] = np.zeros(n * nb)
    cc.delta_colmass = (c.p_int[:-1] - c.p_int[1:]) / c.g
    cc.delta_col_upper = (c.p_lay - c.p_int[1:]) / c.g
    cc.delta_col_lower = (c.p_int[:-1] - c.p_lay) / c.g
    # a coarse Planck table keeps the (untimed) set-up short; the timed part does not depend on it
    cc.plancktable_dim, cc.plancktable_step = 800, 20
    s = cases.alloc_state(cc)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    oracle.port.set_num_threads(min(ncpu, 32))
    cases.setup_planck(oracle.port, cc, s)
    # the visible core count can exceed what the container may actually use (CPU quota): time one
    # flux sweep at a few thread counts and keep the fastest
    cases.interpolate_temperatures_and_planck(oracle.port, cc, s)
    cases.refresh_premixed(oracle.port, cc, s)
    best = (1e30, 1)
    for th in sorted(set(t for t in (4, 8, 16, 32, 64, 128, ncpu) if t <= ncpu)):
        oracle.port.set_num_threads(th)
        cases.flux_sweeps(oracle.port, cc, s, 1)
        t1 = time.perf_counter()
        cases.flux_sweeps(oracle.port, cc, s, 2)
        el = time.perf_counter() - t1
        if el < best[0]:
            best = (el, th)
    threads = best[1]
    oracle.port.set_num_threads(threads)
    # size the timed run for roughly 10-30 s of CPU work: one flux sweep took best[0]/2 s on nb bins
    per_iter = 2.2 * best[0]                         # 4 sweeps + refresh share + the rest
    n_it = int(min(60, max(20, 10 * round(15.0 / max(per_iter, 1e-3) / 10))))
    s = cases.alloc_state(cc)
    cc.T_lay = c.T_lay.copy()
    cases.setup_planck(oracle.port, cc, s)
    t0 = time.perf_counter()
    cases.radiation_iterations(oracle.port, cc, s, n_it)
    dt = time.perf_counter() - t0
    return dict(value=n_it / dt * nb * c.nlayer, unit="bin*layer*iterations/s", cores=threads, kind="port",
                sample="%d of %d bins x %d layers x %d Gauss points, %d iterations from iteration 0 "
                       "(one opacity refresh per 10), oracle/helios_oracle.c with OpenMP over bins (%d of %d visible "
                       "cores: fastest of a thread-count probe), %.1f s"
                       % (nb, w["nbin"], c.nlayer, c.ny, n_it, threads, ncpu, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--columns-per-gpu", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--phase", default="radiative", choices=["radiative", "convection"],
                    help="which loop a step is taken from: radiation_loop (default, the headline number) or "
                         "convection_loop (convective adjustment + sweeps + temperature step, all on the device)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch  # first, so that its bundled HIP runtime is the one libhelios_hip.so binds to
    dist = None
    # test hook for single-GPU machines: HELIOS_BENCH_BACKEND=gloo runs all ranks on GPU 0 with CPU-side collectives,
    # so that the multi-rank logic can be exercised where RCCL cannot (one GPU cannot host two RCCL ranks)
    backend = os.environ.get("HELIOS_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else 0
    coll_device = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from helios_amd.device import Context
    from helios_amd.rt import batch_from_case

    w = WORKLOADS[args.workload]
    seed = 20240 + 2
    c = build_case(w, seed)
    ctx = Context(device_index)
    ncol = args.columns_per_gpu
    rt = make_batch(ctx, c, ncol)
    # every column of a sweep differs (here: internal temperature -> different T-P trajectories)
    if ncol > 1 or world > 1:
        for i in range(ncol):
            gi = rank * ncol + i
            rt.set_temperatures(i, c.T_lay * (1.0 + 0.01 * gi))
    rt.build_planck_table(1)
    ctx.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    run = rt.run
    if args.phase == "convection":
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
    # warm-up: W untimed steps starting at iteration 0 (includes the first refresh)
    run(0, args.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    run(args.warmup, args.steps)
    ev_ms = ctx.timer_stop_ms()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    done = [int(rt.get("done", i)[0]) for i in range(ncol)]

    # per-kernel timing of the dominant kernel: a separate short pass with HIP events on the stream
    roofline = None
    rt_prof = {}
    tm = rt.traffic_model()
    if args.profile_steps > 0:
        it0 = args.warmup + args.steps
        it0 += (-it0) % 10 + 1                      # start right after a refresh boundary
        run(args.warmup + args.steps, it0 - (args.warmup + args.steps))
        ctx.synchronize()
        rt.profile(True)
        run(it0, min(args.profile_steps, 9))     # E-iterations only
        rt.profile(False)
        flux_ms, nflux = rt.profile_read("rt_flux")
        if c.species:   # one refresh under the profiler: where the on-the-fly mixing time goes
            rt.profile(True)
            rt.refresh()
            rt.profile(False)
        rt_prof = {k: rt.profile_read(k)[0] for k in ("refresh_total", "add_to_mixed_opac",
                                                      "opac_species_interpol", "mixed_scat", "rt_coef")}
        if nflux:
            achieved = tm["step_algorithmic"] / (flux_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get(args.workload, {}).get("rt_flux_hbm_bytes_per_launch")
                if traffic is not None:
                    traffic *= ncol                 # measured with one column per launch; a launch covers all columns
            roofline = dict(bound="hbm", kernel="k_rt_flux", achieved=achieved, peak=8000.0, unit="GB/s",
                            frac=achieved / 8000.0, traffic=traffic,
                            algorithmic_bytes_per_launch=tm["step_algorithmic"],
                            actual_bytes_model_per_launch=tm["step_actual"], avg_launch_ms=flux_ms,
                            launches_timed=nflux)

    # the path's only exchange: gather the emission spectra of all columns once, after the run
    gather_ms = None
    spec = np.stack([rt.get("F_up_band", i)[-c.nbin:] for i in range(ncol)])
    if dist is not None:
        tg = time.perf_counter()
        mine = torch.from_numpy(spec).to(coll_device)
        out = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(out, mine)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        spec = torch.cat(out).cpu().numpy()

    total_cols = ncol * world
    value = args.steps / dt * c.nbin * c.nlayer * total_cols
    line = {
        "metric": "rad-conv iterations/sec x (lambda-bins x layers)",
        "value": value, "unit": "bin*layer*iterations/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["desc"], "nbin": c.nbin, "nlayer": c.nlayer, "ny": c.ny,
                   "opacity_table_TP_grid": [c.ntemp, c.npress], "columns_per_gpu": ncol,
                   "columns_total": total_cols, "refresh_every": 10, "sweeps_per_iteration": 4, "loop": args.phase,
                   "parallelism": "columns sharded, %d per GPU" % ncol},
        "iterations_per_s_per_column": args.steps / dt,
        "stream_event_ms_per_step": ev_ms / args.steps,
        "columns_converged_during_run": int(sum(done)),
        "spectra_gather_ms": gather_ms,
        "spectrum_checksum": float(np.sum(spec)),
        "roofline": roofline,
    }
    if roofline is not None and c.species:
        roofline["refresh_kernels_ms"] = {k: rt_prof[k] for k in rt_prof}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not c.species:
        line["cpu_baseline"] = cpu_baseline(w, seed)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line))
    rt.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
