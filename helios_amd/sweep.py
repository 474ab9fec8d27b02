"""Parameter sweeps: many independent atmosphere columns through ONE device-resident batch per GPU.

The reference runs one column per process (helios.py).  Columns of a sweep share the wavelength grid, the opacity
tables and the physics switches and differ in planet / star / orbit / internal temperature / albedo / start profile --
exactly what `RTBatch` keeps per column -- so a sweep is one batch whose kernels loop over the columns
(BASELINE config 4).  With several GPUs the columns are block-partitioned over the ranks
(`parallel.shard_columns`); the iteration path has no collective, the emission spectra are gathered once at the end.

    python sweep.py -sweep "internal_temperature=100,300,1000;f_factor=0.25,0.5" [helios.py options ...]

Every column gets the full set of output files under `<output>/<name>_<k>/`, exactly as a single run would write them.
"""
import itertools

import numpy as np

from . import additional_heating as add_heat
from . import computation as comp
from . import host_functions as hsfunc
from . import quantities as quant_mod
from . import read as read_mod
from . import write as write_mod

# options that may vary between the columns of one batch (everything else defines the batch itself)
# (directory_with_fastchem_files: every column reads its own chemistry -- a sweep over metallicity or C/O -- and keeps its own
# (T, P) mixing-ratio tables on the device, hx_rt_set_column_vmr_table)
PER_COLUMN_OPTIONS = ("internal_temperature", "f_factor", "stellar_zenith_angle", "surface_albedo", "surface_gravity",
                      "orbital_distance", "radius_planet", "radius_star", "temperature_star",
                      "radiative_equilibrium_criterion", "directory_with_fastchem_files", "name")


def expand_sweep(spec):
    """'key=v1,v2;key2=w1,w2' -> list of {key: value} dicts, the cartesian product in the order given"""
    axes = []
    for part in [p for p in str(spec).split(";") if p.strip()]:
        key, values = part.split("=", 1)
        key = key.strip().lstrip("-")
        if key not in PER_COLUMN_OPTIONS:
            raise ValueError("option '%s' cannot vary inside one batch; allowed: %s" % (key, ", ".join(PER_COLUMN_OPTIONS)))
        axes.append([(key, v.strip()) for v in values.split(",") if v.strip()])
    return [dict(combo) for combo in itertools.product(*axes)] if axes else [{}]


def _prepare_column(base_argv, overrides, shared):
    """the read -> grid -> start-profile part of helios.py:35-83 for one column; tables are read once and shared"""
    reader = read_mod.Read()
    keeper = quant_mod.Store()
    argv = list(base_argv)
    for k, v in overrides.items():
        argv += ["-" + k, str(v)]
    reader.read_param_file_and_command_line(keeper, reader.cloud, argv)
    reader.check_run_configuration(keeper)
    if "tables" not in shared:
        if keeper.opacity_mixing in ("premixed", "synthetic"):
            reader.load_premixed_opacity_table(keeper)
        else:
            reader.read_species_file(keeper)
            reader.read_species_opacities(keeper)
            reader.read_species_scat_cross_sections(keeper)
        names = ("opacity_mixing", "opac_k", "opac_scat_cross", "opac_meanmass", "opac_wave", "opac_interwave",
                 "opac_deltawave", "gauss_y", "ktemp", "kpress", "nbin", "ny", "ntemp", "npress")
        shared["tables"] = {n: getattr(keeper, n) for n in names if hasattr(keeper, n)}
        shared["species_template"] = keeper.species_list
    else:
        for n, v in shared["tables"].items():
            setattr(keeper, n, v)
    if keeper.opacity_mixing == "on-the-fly":
        import copy
        keeper.species_list = [copy.copy(sp) for sp in shared["species_template"]]   # tables shared, VMR profiles own
        reader.read_species_mixing_ratios(keeper)
    reader.read_kappa_table_or_use_constant_kappa(keeper)
    reader.read_or_fill_surf_albedo_array(keeper)
    keeper.dimensions()
    reader.read_star(keeper)
    hsfunc.planet_param(keeper, reader)
    hsfunc.set_up_numerical_parameters(keeper)
    hsfunc.construct_grid(keeper)
    hsfunc.initial_temp(keeper, reader)
    if keeper.approx_f == 1 and keeper.planet_type == "rocky":      # helios.py:70-71
        hsfunc.approx_f_from_formula(keeper, reader)
    hsfunc.calc_F_intern(keeper)
    add_heat.load_heating_terms_or_not(keeper)
    reader.cloud.cloud_pre_processing(keeper)
    keeper.create_zero_arrays()
    keeper.convert_input_list_to_array()
    return keeper, reader


def _batch_signature(q):
    # (the species list and where each species' mixing ratio comes from: a batch shares the opacity tables and hands the
    # device one (T, P) chemistry table per column for the FastChem species -- which species those are must agree)
    chem = tuple((str(getattr(sp, "name", "")), str(getattr(sp, "source_for_vmr", "")))
                 for sp in (getattr(q, "species_list", None) or [])) if str(q.opacity_mixing) == "on-the-fly" else ()
    return (int(q.nbin), int(q.ny), int(q.nlayer), int(q.scat), int(q.dir_beam), int(q.clouds), int(q.scat_corr),
            int(q.smooth), int(q.convection), str(q.opacity_mixing), float(q.g_0), float(q.epsi), str(q.planet_type),
            int(q.iso), int(q.singlewalk), str(q.flux_calc_method), chem)


def _radiation_loop(computer, quants, rt):
    """Compute.radiation_loop for every column of the batch at once: a column that satisfies its criterion is frozen
    on the device; the host paces the batch in chunks that end at refresh boundaries and criterion relaxations"""
    q0 = quants[0]
    ncol = len(quants)
    done = np.zeros(ncol, bool)
    iters = np.zeros(ncol, np.int64)
    if q0.singlewalk == 1:
        # post-processing run type (computation.py:983-984, Compute.radiation_loop): ONE pass over the given T-P profiles --
        # refresh, 1000*scat+1 sweeps inside one launch of the flux kernel, quadrature -- and no temperature step
        if q0.opacity_mixing == "on-the-fly":
            for q in quants:
                computer._push_vmr(q)
        rt.step(0, step_temperature=False)
        computer.report_diagnostics(q0)
        for q in quants:
            q.iter_value = np.int32(0)
        return iters
    it = 0
    while not done.all():
        # once per loop, as Compute.radiation_loop does: constant and file-given profiles never change, a FastChem species'
        # profile is re-interpolated on the device at every refresh from ITS COLUMN's (T, P) table (make_rt_batch)
        if q0.opacity_mixing == "on-the-fly" and it == 0:
            for c in np.nonzero(~done)[0]:
                computer._push_vmr(quants[c])
        nxt = min(it + (10 - it % 10), int(q0.max_nr_iterations) + 1)
        for r in q0.crit_relaxation_numbers:
            if it < r < nxt:
                nxt = int(r)
        nxt = min(nxt, it + 1 + (100 - it % 100) % 100)      # the 100-iteration surface-temperature check
        rt.run(it, nxt - it)
        computer.report_diagnostics(quants[0])
        it = nxt
        for c in np.nonzero(~done)[0]:
            if int(rt.get("done", c)[0]):
                done[c] = True
                iters[c] = int(rt.get("iters_done", c)[0])
        if (it - 1) % 100 == 0:     # computation.py:946-952: a surface hotter than the Planck table ends the radiative loop
            for c in np.nonzero(~done)[0]:
                q = quants[c]
                if not rt.get("T_lay", c)[int(q.nlayer)] < q.plancktable_dim * q.plancktable_step - 2:
                    if q.iso == 0:
                        q.convection = 1
                    done[c] = True
                    iters[c] = it
                    rt.set_state(int(c), "done", np.ones(1, np.int32))
        if it in q0.crit_relaxation_numbers:
            for c in np.nonzero(~done)[0]:
                hsfunc.relax_radiative_convergence_criterion(quants[c])
                rt.set_convergence_limit(int(c), quants[c].rad_convergence_limit)
        if it > q0.max_nr_iterations:
            for c in np.nonzero(~done)[0]:      # give up on the rest; they are written with an abort marker
                quants[c].aborted = True
                iters[c] = it
                done[c] = True
            rt.set_state(-1, "done", np.ones(1, np.int32))
    for c, q in enumerate(quants):
        q.iter_value = np.int32(iters[c])
    return iters


def _convection_loop(computer, quants, rt):
    """Compute._convection_loop_fused for the batch: columns without a super-adiabatic layer stay frozen, the others
    run the device-side convection loop until their own loop condition turns false"""
    q0 = quants[0]
    ncol = len(quants)
    active = np.zeros(ncol, bool)
    iters = np.zeros(ncol, np.int64)
    from_table = any(computer._kappa_from_table(q) for q in quants)
    if from_table:    # kappa / c_p follow the profile the radiation loop left behind (computation.py:1037), as in a single run
        rt.kappa_cp_refresh()
    for c, q in enumerate(quants):
        if not (q.singlewalk == 0 and q.convection == 1) or getattr(q, "aborted", False):
            continue
        for n in ("T_lay", "F_net", "F_up_tot", "F_down_tot"):
            setattr(q, n, rt.get(n, c))
        if from_table:
            for n in ("kappa_lay", "kappa_int", "c_p_lay"):
                setattr(q, n, rt.get(n, c))
        q.p_lay, q.p_int = np.asarray(q.p_lay, float), np.asarray(q.p_int, float)
        hsfunc.conv_check(q)
        hsfunc.mark_convective_layers(q, stitching=0)
        q.iter_value = np.int32(0)      # the convection loop counts from zero, also when it has nothing to do
        if sum(q.conv_unstable) > 0:
            active[c] = True
            for name, v, dt in (("kappa_lay", q.kappa_lay, np.float64), ("kappa_int", q.kappa_int, np.float64),
                                ("c_p_lay", q.c_p_lay, np.float64), ("conv_layer", q.conv_layer, np.int32),
                                ("conv_unstable", q.conv_unstable, np.int32)):
                rt.set_state(c, name, np.asarray(v, dt))
            rt.set_state(c, "dampara", np.array([-1.0 if q.input_dampara == "automatic" else float(q.input_dampara)]))
            rt.set_state(c, "done", np.zeros(1, np.int32))
            rt.set_convergence_limit(c, q.rad_convergence_limit)
    if not active.any():
        return iters
    running = active.copy()
    it = 0
    while running.any():
        nxt = min(it + (10 - it % 10), int(q0.max_nr_iterations) + 1)
        for r in q0.crit_relaxation_numbers:
            if it < r < nxt:
                nxt = int(r)
        # (tabulated chemistry follows the temperatures on the device, ahead of the adjustment and for the adjusted profile:
        # k_rt_mmm_from_vmr and the refresh, computation.py:1030-1036, :1056-1061 -- no host step per decade)
        rt.conv_run(it, nxt - it)
        computer.report_diagnostics(quants[0])
        it = nxt
        for c in np.nonzero(running)[0]:
            if int(rt.get("done", c)[0]):
                running[c] = False
                iters[c] = int(rt.get("iters_done", c)[0])
        if it in q0.crit_relaxation_numbers:
            for c in np.nonzero(running)[0]:
                hsfunc.relax_radiative_convergence_criterion(quants[c])
                rt.set_convergence_limit(int(c), quants[c].rad_convergence_limit)
        if it > q0.max_nr_iterations:
            for c in np.nonzero(running)[0]:
                quants[c].aborted = True
                iters[c] = it
            rt.set_state(-1, "done", np.ones(1, np.int32))
            break
    for c in np.nonzero(active)[0]:
        q = quants[c]
        q.iter_value = np.int32(iters[c])
        q.conv_layer, q.conv_unstable = rt.get("conv_layer", c), rt.get("conv_unstable", c)
        q.marked_red = rt.get("marked_red", c)
        for n in ("kappa_lay", "kappa_int", "c_p_lay"):      # what the output files report, as in a single run
            setattr(q, n, rt.get(n, c))
    return iters


def _finish_column(computer, q, reader, writer):
    """post-loop diagnostics and output files of one column (helios.py:88-126), on the Store's own device arrays"""
    tables = {n: getattr(q, n, None) for n in ("opac_k", "opac_scat_cross", "opac_meanmass")}
    for n in tables:                  # the look-up tables live in the batch; the diagnostics do not touch them
        setattr(q, n, np.zeros(1))
    q.copy_host_to_device()
    for n, v in tables.items():
        setattr(q, n, v)
    q.allocate_on_device()
    computer.sync_store_from_rt(q)
    if getattr(q, "conv_layer", None) is not None:
        q.dev_conv_layer.set(np.asarray(q.conv_layer, np.int32))
    q.dev_F_smooth_sum.set(q.rt.get("F_smooth_sum", q.rt_col))
    computer.integrate_optdepth_transmission(q)
    computer.calculate_contribution_function(q)
    if q.convection == 1:
        computer.interpolate_entropy(q)
        computer.interpolate_phase_state(q)
    computer.calculate_mean_opacities(q)
    computer.integrate_beamflux(q)
    q.copy_device_to_host()
    if q.conv_unstable is None:
        q.conv_unstable = np.zeros(int(q.nlayer) + 1, np.int32)
    hsfunc.calculate_conv_flux(q)
    hsfunc.calc_F_ratio(q)
    if getattr(q, "aborted", False):
        writer.write_abort_file(q, reader)
    writer.write_all(q, reader)
    for name in [n for n in vars(q) if n.startswith("dev_")]:       # give the device memory back before the next column
        arr = getattr(q, name)
        if hasattr(arr, "free"):
            arr.free()
        setattr(q, name, None)


_SWEEP_SEQUENCE = 0      # number of run_sweep calls with a shared work list in this process


def _run_columns(base_argv, overrides_list, cols, computer, writer, shared, columns, timing, write_output):
    """prepare the columns `cols`, run them as device batches (one per batch signature) to the end of both loops, write
    their files and append them to `columns`"""
    import time
    fresh = []
    for k in cols:
        ov = dict(overrides_list[k])
        ov.setdefault("name", "%s_%d" % (_base_name(base_argv), k))
        q, reader = _prepare_column(base_argv, ov, shared)
        q._ctx = computer.ctx
        q.sweep_index = k
        fresh.append((q, reader))
    groups = {}
    for q, reader in fresh:
        if not computer._fused_supported(q):
            raise IOError("sweeps run on the fused device path: flux calculation method 'iteration' or 'matrix' and at most 1024 "
                          "layers (2048 isothermal ones); run this configuration column by column with helios.py")
        groups.setdefault(_batch_signature(q), []).append((q, reader))
    for members in groups.values():
        quants = [q for q, _ in members]
        t0 = time.perf_counter()
        rt = computer.make_rt_batch(quants)
        computer.ctx.synchronize()
        t1 = time.perf_counter()
        _radiation_loop(computer, quants, rt)
        _convection_loop(computer, quants, rt)
        computer.ctx.synchronize()
        t2 = time.perf_counter()
        timing["batch"] += t1 - t0
        timing["loops"] += t2 - t1
        for q, reader in members:
            if write_output:
                _finish_column(computer, q, reader, writer)
            else:
                q.T_lay = rt.get("T_lay", q.rt_col)
                q.F_up_band = rt.get("F_up_band", q.rt_col)
        timing["finish"] += time.perf_counter() - t2
        for q in quants:
            q.rt = None
        rt.close()
    columns += fresh


def run_sweep(base_argv, overrides_list, dist=None, coll_device="cpu", write_output=True):
    """run every column of `overrides_list` (this rank's share if `dist` is initialised); returns
    (columns of this rank as Stores, emission spectra of ALL columns [ncol_total, nbin] in sweep order)"""
    import os
    from .parallel import column_list, gather_spectra
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if world > len(overrides_list):
        raise ValueError("%d ranks for %d columns: a sweep needs at least one column per rank" % (world, len(overrides_list)))
    # neighbours in a sweep (similar planets) converge after similar numbers of iterations: dealing the columns out in
    # turn spreads the long-running ones over the ranks; HELIOS_SWEEP_PARTITION=block keeps contiguous blocks
    # HELIOS_SWEEP_PARTITION=dynamic[:chunk]: a shared work list instead -- every rank claims `chunk` columns (default:
    # an eighth of its even share), runs them as one batch, retires them and claims again until the list is empty
    mode = os.environ.get("HELIOS_SWEEP_PARTITION", "cyclic")
    computer = comp.Compute()
    writer = write_mod.Write()
    shared = {}
    columns = []
    import time
    timing = dict(batch=0.0, loops=0.0, finish=0.0)
    if mode.startswith("dynamic"):
        from .parallel import WorkList
        chunk = int(mode.split(":")[1]) if ":" in mode else max(1, len(overrides_list) // (8 * world))
        # one counter per sweep: every rank calls run_sweep the same number of times, and a restarted worker group
        # (torchrun --max-restarts) may find the previous attempt's keys still in the agent's store
        global _SWEEP_SEQUENCE
        _SWEEP_SEQUENCE += 1
        key = "%s/restart%s/sweep%d" % (WorkList.KEY, os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), _SWEEP_SEQUENCE)
        work = WorkList(len(overrides_list), chunk, dist, key=key)
        claims = iter(work.claim, [])
        mine_cols = work.claimed
    else:
        claims = iter([column_list(len(overrides_list), rank, world, mode)])
        mine_cols = None
    for claimed in claims:
        _run_columns(base_argv, overrides_list, claimed, computer, writer, shared, columns, timing, write_output)
    if mine_cols is None:
        mine_cols = [q.sweep_index for q, _ in columns]
    if rank == 0:
        print("\nSweep timing [s]: batch set-up %.2f, iteration loops %.2f, diagnostics + output files %.2f"
              % (timing["batch"], timing["loops"], timing["finish"]))
    if world > 1:
        # columns converge after different numbers of iterations: how unevenly the ranks were loaded
        mine = dict(rank=rank, columns=len(mine_cols), loops=timing["loops"],
                    iterations=int(sum(int(q.iter_value or 0) for q, _ in columns)))
        every = [None] * world
        dist.all_gather_object(every, mine)
        if rank == 0:
            loops = np.array([e["loops"] for e in every])
            print("Load over %d ranks: loop time min %.2f / mean %.2f / max %.2f s (imbalance max/mean = %.2f); "
                  "iterations per rank %s" % (world, loops.min(), loops.mean(), loops.max(),
                                              loops.max() / max(loops.mean(), 1e-30),
                                              [e["iterations"] for e in every]))
    X = int(columns[0][0].nbin) if columns else 0
    local = np.array([np.asarray(q.F_up_band)[-X:] for q, _ in columns]).reshape(len(columns), X)
    spectra = gather_spectra(local, dist, coll_device, columns=mine_cols)
    if spectra.shape[0] != len(overrides_list):
        raise RuntimeError("sweep: %d columns were run, %d were asked for (partition '%s')"
                           % (spectra.shape[0], len(overrides_list), mode))
    return [q for q, _ in columns], spectra


def wavelength_grid(base_argv, overrides):
    """the wavelength grid of a sweep, for a rank that ran none of its columns"""
    q, _ = _prepare_column(base_argv, dict(overrides), {})
    return q.opac_wave


def _base_name(argv):
    argv = list(argv)
    return argv[argv.index("-name") + 1] if "-name" in argv else "sweep"
