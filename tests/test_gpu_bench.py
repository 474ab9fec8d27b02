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


def _last_json(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_rank_contract():
    p = subprocess.run([sys.executable, "bench.py", "--workload", "c1", "--steps", "40", "--warmup", "10"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    for f in FIELDS:
        assert f in line, f
    assert line["n_gpus"] == 1 and line["steps"] == 40 and line["warmup"] == 10 and line["value"] > 0
    assert line["dtype"] == "f64" and line["scaling"] == "weak" and line["vs_baseline"] is None
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


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


def test_bench_multi_rank_default_is_the_sharded_sweep():
    """N > 1 without --workload: config 4 (columns of the parameter sweep sharded over the ranks); here the reduced
    shape, two ranks on one GPU through the gloo hook"""
    env = dict(os.environ, HELIOS_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29519", "bench.py", "--gpus", "2", "--steps",
                        "10", "--warmup", "10", "--workload", "c4small"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["config"]["columns_total"] == 12 and line["config"]["species"] == 7
    assert len(line["per_rank_ms_per_step"]) == 2 and 0 <= line["rank_imbalance"] < 1
    import bench
    assert bench.WORKLOADS["c4"]["columns_per_gpu"] * 8 == 512
