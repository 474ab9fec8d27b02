"""bench.py's contract on a real GPU: one JSON line with the agreed fields, single rank and -- through the
HELIOS_BENCH_BACKEND=gloo hook (two ranks sharing GPU 0, CPU-side collectives) -- the multi-rank path the driver
launches with torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
          "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _last_json(out, detail=True):
    """the ONE line on stdout -- compact (bench.LINE_LIMIT), the last thing printed -- and, for the assertions on what the run
    measured in full, the side file it names"""
    import bench
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    assert out.rstrip().endswith(lines[0]) and len(lines[0]) < bench.LINE_LIMIT, len(lines[0])
    line = json.loads(lines[0])
    for f in FIELDS:
        assert f in line, f
    if not detail:
        return line
    path = line["detail"] if os.path.isabs(line["detail"]) else os.path.join(ROOT, line["detail"])
    with open(path) as f:
        full = json.load(f)
    assert full["value"] == line["value"] and full["ms_per_step"] == line["ms_per_step"]      # the same run
    return full


def test_the_drivers_command_prints_one_short_line_last(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` as the driver runs it (default secondaries, live counters, CPU
    baseline): what a reader of the last 2 000 characters of stdout sees ends in one line that parses and holds the
    contract's fields with `roofline` and `cpu_baseline`"""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--detail",
                        str(tmp_path / "detail.json")], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout[-4000:].rstrip().splitlines()[-1]
    line = json.loads(last)
    assert len(last) < 4000
    for f in FIELDS:
        assert f in line, f
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["config"]["nbin"] == 10000
    r = line["roofline"]
    assert r["kernel"] == "k_rt_flux" and r["bound"] == "hbm" and 0.2 < r["frac"] < 1 and r["traffic"] > r["algorithmic_bytes_per_launch"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["seconds"] < 60
    assert sorted(line["secondary"]) == ["c3", "c4", "c5"]
    for x in line["secondary"].values():
        assert x["value"] > 0 and x["kernel"] == "k_rt_mix_species" and 0 < x["frac"] < 1
    assert abs(line["value"] - 10000 * 100 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    with open(str(tmp_path / "detail.json")) as f:
        full = json.load(f)
    assert len(full["spectrum_checksum_per_column"]) == 1 and "kernels_ms" in full["roofline"]


def test_bench_single_rank_contract():
    p = subprocess.run([sys.executable, "bench.py", "--workload", "c1", "--steps", "40", "--warmup", "10"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 1 and line["steps"] == 40 and line["warmup"] == 10 and line["value"] > 0
    assert line["dtype"] == "f64" and line["scaling"] == "weak" and line["vs_baseline"] is None
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert str(cb["cores"]) in cb["thread_probe_iterations_per_s"] and cb["host_cores"] >= cb["cores"]
    ss = line["steady_state_200"]
    assert ss["iterations"] == 200 and ss["from_iteration"] == 0 and ss["refreshes"] == 20 and ss["value"] > 0
    assert line["env_knobs"] == {} and line["secondary"] is None and line["process_group"] is None
    assert "scaling_series_n1" not in line and "single_gpu_same_workload" not in line
    # a small grid keeps the front-to-back launch order and non-temporal state stores
    assert line["config"]["flux_launch_policy"] == {"back_and_forth": False, "state_cached_mib": 0.0}


def test_headline_batch_keeps_its_state_in_the_infinity_cache():
    """BASELINE config 2's batch (one column: 333 MB of up-flux state, 49 MB of node and band arrays) switches the
    back-and-forth launch order with 240 MiB of cached state on by itself; four columns per batch (the arrays of the
    kernels in between would displace the state) do not"""
    common = ["--steps", "10", "--warmup", "10", "--no-cpu-baseline", "--secondary", "none", "--live-counters", "off",
              "--profile-steps", "0"]
    p = subprocess.run([sys.executable, "bench.py"] + common, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["config"]["nbin"] == 10000 and line["config"]["columns_per_gpu"] == 1
    assert line["config"]["flux_launch_policy"] == {"back_and_forth": True, "state_cached_mib": 240.0}
    p = subprocess.run([sys.executable, "bench.py", "--columns-per-gpu", "4"] + common, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _last_json(p.stdout)["config"]["flux_launch_policy"] == {"back_and_forth": False, "state_cached_mib": 0.0}


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the process starts a child torch.distributed.run
    with two ranks before touching a GPU and relays rank 0's line (here: both ranks on GPU 0 through the gloo hook); the
    line is the N = 1 workload on every rank, with a secondary workload measured in the same run"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HELIOS_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "10", "--workload", "c1",
                        "--secondary", "c4small"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    one = subprocess.run([sys.executable, "bench.py", "--workload", "c1", "--steps", "20", "--warmup", "10",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    one = _last_json(one.stdout)
    assert line["n_gpus"] == 2 and one["n_gpus"] == 1
    assert line["config"]["workload"] == one["config"]["workload"]          # one workload for the whole series
    assert line["config"]["columns_per_gpu"] == one["config"]["columns_per_gpu"] == 1
    assert line["config"]["columns_total"] == 2 and line["spectra_gather_ms"] is not None
    assert line["env_knobs"] == {"HELIOS_BENCH_BACKEND": "gloo"}
    sec = line["secondary"]["c4small"]
    assert sec["config"]["columns_total"] == 12 and sec["config"]["species"] == 7 and sec["ms_per_step"] > 0
    assert len(sec["per_rank_ms_per_step"]) == 2 and sec["roofline"] is not None


def test_rccl_gathers_the_spectra_with_one_rank():
    """RCCL itself (backend "nccl") on the one GPU of this box: a process group of one rank, the spectra gather of
    helios_amd/parallel.py through it on the device"""
    code = ("import os, numpy as np, torch, torch.distributed as dist\n"
            "from helios_amd.parallel import gather_spectra\n"
            "torch.cuda.set_device(0)\n"
            "dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', world_size=1, rank=0,"
            " device_id=torch.device('cuda', 0))\n"
            "a = np.arange(12.0).reshape(3, 4)\n"
            "out = gather_spectra(a, dist, device='cuda', columns=[2, 0, 1])\n"
            "assert np.array_equal(out, a[[1, 2, 0]]), out\n"
            "out = gather_spectra(a, dist, device='cuda')\n"
            "assert np.array_equal(out, a)\n"
            "t = torch.arange(12.0, dtype=torch.float64, device='cuda').reshape(3, 4)\n"      # a tensor that lives on the GPU already
            "out = gather_spectra(t, dist, device='cuda', columns=[2, 0, 1])\n"
            "assert np.array_equal(out, a[[1, 2, 0]]), out\n"
            "print('backend', dist.get_backend())\n"
            "dist.destroy_process_group()\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "backend nccl" in p.stdout


def test_checked_process_group_init_over_rccl_with_one_rank():
    """bench.py's set-up of its ranks on the real backend, as far as one GPU allows: init with a deadline, the roll call on a
    gloo side group (loopback interface), the first all-reduce on RCCL that must count every rank"""
    code = ("import os, torch, torch.distributed as dist\n"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29537', RANK='0', WORLD_SIZE='1')\n"
            "from helios_amd.parallel import init_process_group_checked, bind_to_gpu_numa_node\n"
            "print('affinity', bind_to_gpu_numa_node(0, apply=False))\n"
            "torch.cuda.set_device(0)\n"
            "t = init_process_group_checked(dist, 'nccl', 0, 1, device=torch.device('cuda', 0), timeout_s=120.0)\n"
            "assert dist.get_backend() == 'nccl' and set(t) == {'init_s', 'roll_call_s', 'first_collective_s'}, t\n"
            "print('times', t, os.environ.get('GLOO_SOCKET_IFNAME'))\n"
            "dist.destroy_process_group()\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "times" in p.stdout and " lo" in p.stdout.splitlines()[-1] and "roll call incomplete" not in p.stderr
    assert "[Gloo]" not in p.stdout        # (the side group's connection messages go to stderr: nothing in front of a bench line)


def test_bench_two_ranks_share_the_columns():
    env = dict(os.environ, HELIOS_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "2", "--steps",
                        "20", "--warmup", "10", "--workload", "c1"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["config"]["columns_total"] == 2 and line["value"] > 0
    assert line["cpu_baseline"] is None and line["spectra_gather_ms"] is not None


def test_bench_species_workload_carries_roofline_and_cpu_baseline():
    """on-the-fly mixing (config 3 shape, reduced): the line names the mixing kernel, prices it against HBM and has a
    CPU baseline too"""
    p = subprocess.run([sys.executable, "bench.py", "--workload", "c3small", "--steps", "20", "--warmup", "10"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    r = line["roofline"]
    assert r["kernel"] == "k_rt_mix_species" and r["bound"] == "hbm" and r["whole_step"]["frac"] > 0
    assert r["t_only_ms_per_refresh"] > r["e_only_ms_per_iteration"] > 0
    assert line["config"]["species"] == 22
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and "species" in cb["sample"]


def test_bench_secondary_workload_under_a_launcher():
    """the driver's launch (torch.distributed.run around bench.py) with the sharded sweep as the secondary workload: the
    reduced config-4 shape, two ranks on one GPU through the gloo hook"""
    env = dict(os.environ, HELIOS_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29519", "bench.py", "--gpus", "2", "--steps",
                        "10", "--warmup", "10", "--workload", "c1", "--secondary", "c4small"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    sec = line["secondary"]["c4small"]
    assert line["n_gpus"] == 2 and sec["config"]["columns_total"] == 12 and sec["config"]["species"] == 7
    assert len(sec["per_rank_ms_per_step"]) == 2 and 0 <= sec["rank_imbalance"] < 1
    import bench
    assert bench.WORKLOADS["c4"]["columns_per_gpu"] * 8 == 512


def test_bench_eight_ranks_rehearsal_on_one_gpu():
    """the 8-GPU launch of the driver, rehearsed on this box's one GPU through the gloo hook: eight ranks, each with its own
    context, column and start profile; per-rank arrays of length 8, the process group echoed into the line, the sharded
    sweep as secondary workload -- and the spectra, gathered from device tensors, in global column order: the same eight
    columns as ONE batch in one process give the same per-column checksums in the same order"""
    env = dict(os.environ, HELIOS_BENCH_BACKEND="gloo")
    common = ["--steps", "10", "--warmup", "10", "--workload", "c2small", "--profile-steps", "0", "--device-warmup-ms", "0",
              "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "8", "--secondary",
                        "c4small"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    # the N = 8 line itself: within the size limit, with what the first real 8-GPU run will be read for
    short = _last_json(p.stdout, detail=False)
    assert short["n_gpus"] == 8 and len(short["per_rank_ms_per_step"]) == 8 and 0 <= short["rank_imbalance"] < 1
    assert short["spectra_gather_ms"] > 0 and short["secondary"]["c4small"]["value"] > 0
    spg = short["process_group"]
    assert spg["world_size"] == 8 and len(spg["numa_node_per_rank"]) == 8 and len(spg["cpus_per_rank"]) == 8
    assert spg["init_s"] > 0 and spg["roll_call_s"] >= 0 and spg["first_collective_s"] >= 0
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 8 and line["config"]["columns_total"] == 8 and line["config"]["columns_per_gpu"] == 1
    assert len(line["per_rank_ms_per_step"]) == 8 and all(v > 0 for v in line["per_rank_ms_per_step"])
    pg = line["process_group"]
    assert pg["backend"] == "gloo" and pg["world_size"] == 8 and pg["ranks_seen_by_all_gather"] == 8 and pg["rccl_world"] is None
    cs = line["spectrum_checksum_per_column"]
    assert len(cs) == 8 and len(set(cs)) == 8
    assert abs(sum(cs) - line["spectrum_checksum"]) <= 1e-9 * abs(line["spectrum_checksum"])
    sec = line["secondary"]["c4small"]
    assert sec["config"]["columns_total"] == 48 and len(sec["per_rank_ms_per_step"]) == 8 and sec["spectra_gather_ms"] is not None
    assert len(sec["spectrum_checksum_per_column"]) == 48
    one = subprocess.run([sys.executable, "bench.py", "--columns-per-gpu", "8", "--secondary", "none"] + common, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    one = _last_json(one.stdout)
    assert one["config"]["columns_total"] == 8 and one["process_group"] is None
    for a, b in zip(cs, one["spectrum_checksum_per_column"]):        # column k of the batch = the column of rank k
        assert abs(a - b) <= 1e-9 * abs(b), (cs, one["spectrum_checksum_per_column"])


def test_sweep_work_list_is_exhausted_by_eight_ranks(tmp_path):
    """HELIOS_SWEEP_PARTITION=dynamic:1 with eight ranks (gloo hook: all on GPU 0) and twelve columns: the ranks claim
    columns one at a time until the list is exhausted -- every column is run exactly once, by whichever rank got there
    first -- and the spectra arrive in sweep order and equal the one-process sweep"""
    import numpy as np
    sys.path.insert(0, ROOT)
    import sweep
    base = ["-parameter_file", "/nonexistent", "-opacity_mixing", "synthetic", "-synthetic", "20 6 5 11",
            "-number_of_layers", "16", "-maximum_number_of_iterations", "20000", "-name", "wl8",
            "-radiative_equilibrium_criterion", "1e-4", "-convective_adjustment", "no"]
    spec = "internal_temperature=150,600,1100;f_factor=0.25,0.4,0.5,0.6"
    cols, spectra = sweep.main(["-sweep", spec] + base + ["-output_directory", str(tmp_path) + "/one/"])
    assert len(cols) == 12
    env = dict(os.environ, HELIOS_BENCH_BACKEND="gloo", HELIOS_SWEEP_PARTITION="dynamic:1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", "29543", "sweep.py", "-sweep", spec] + base +
                       ["-output_directory", str(tmp_path) + "/eight/"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1500)
    assert p.returncode == 0, p.stderr[-6000:]
    assert "Load over 8 ranks" in p.stdout
    z = np.load(os.path.join(str(tmp_path), "eight", "wl8_sweep_spectra.npz"))
    np.testing.assert_allclose(z["F_up_TOA"], spectra, rtol=1e-12)
    for k in range(12):
        a = open(os.path.join(str(tmp_path), "one", "wl8_%d" % k, "wl8_%d_tp.dat" % k)).read()
        b = open(os.path.join(str(tmp_path), "eight", "wl8_%d" % k, "wl8_%d_tp.dat" % k)).read()
        assert a == b, k
