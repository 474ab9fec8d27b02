"""bench.py's command line, as far as it can be exercised without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_the_result_altering_debug_knobs():
    env = dict(os.environ, HELIOS_RT_DEBUG_SKIP="1")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "1"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0 and "HELIOS_RT_DEBUG_SKIP" in p.stderr and "--allow-debug-knobs" in p.stderr


def test_bench_series_is_one_workload_and_holds_no_stored_numbers():
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "profiles/r0" not in src                       # no committed measurement is copied into a live line
    assert bench.WORKLOADS["c5"]["nspecies"] == 20 and bench.WORKLOADS["c5"]["dir_beam"] == 1
    assert bench.WORKLOADS["c5"]["nbin"] == 30000 and bench.WORKLOADS["c5"]["nlayer"] == 200


def test_bench_mismatched_world_size_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def _canned(name):
    import json
    with open(os.path.join(ROOT, "profiles", name)) as f:     # (the rehearsal's file begins with gloo's connection chatter)
        return json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][-1])


def test_compact_line_stays_short_at_one_and_eight_ranks():
    """the line the driver parses is built by bench.compact_line() from the full record of a run: canned records -- round 5's
    N = 1 run with six secondaries (the 20 KB line the driver could not parse) and its eight-rank rehearsal, widened to
    what a real N = 8 run carries -- come out below bench.LINE_LIMIT, round-trip through json, and keep the contract's
    fields with `roofline` and `cpu_baseline`"""
    import json
    import bench
    n1 = _canned("r05_bench_n1.json")
    assert len(json.dumps(n1)) > 15000
    n8 = _canned("r05_bench_8ranks_one_gpu_gloo.json")
    n8["secondary"]["c4"] = dict(n1["secondary"]["c4"], per_rank_ms_per_step=[247.123456789] * 8)
    n8["process_group"].update(numa_node_per_rank=[0, 0, 0, 0, 1, 1, 1, 1], cpus_per_rank=[48] * 8, init_s=1.234567891,
                               roll_call_s=0.0123456789, first_collective_s=0.123456789, rccl_version="2.26.6")
    n8["roofline"] = n1["roofline"]
    n1["secondary"]["broken"] = {"error": "RuntimeError: " + "x" * 500}
    for full, n in ((n1, 1), (n8, 8)):
        line = bench.compact_line(full, "bench_detail.json")
        text = json.dumps(line, separators=(",", ":"))
        assert len(text) < bench.LINE_LIMIT, (n, len(text))
        back = json.loads(text)
        assert back == json.loads(json.dumps(line))
        for f in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert f in back, f
        assert back["n_gpus"] == n and back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
        assert back["config"]["nbin"] == full["config"]["nbin"] and "workload" in back["config"]
        r = back["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
        assert r["traffic"] is None or r["traffic"] > 0
        for x in back["secondary"].values():
            assert set(x) <= {"value", "ms_per_step", "frac", "kernel", "kernel_ms", "whole_step_frac", "error"}
    assert bench.compact_line(n1)["cpu_baseline"]["kind"] == "port" and bench.compact_line(n8)["cpu_baseline"] is None
    assert len(bench.compact_line(n8)["per_rank_ms_per_step"]) == 8


def test_emit_prints_the_line_last_and_writes_the_side_file(tmp_path, capsys):
    import json
    import bench
    full = _canned("r05_bench_n1.json")
    path = str(tmp_path / "detail.json")
    bench.emit(full, path)
    out = capsys.readouterr().out
    assert out.endswith("\n") and out.count("\n") == 1 and len(out) < bench.LINE_LIMIT
    assert json.loads(out)["detail"] == path
    with open(path) as f:
        assert json.load(f) == json.loads(json.dumps(full))


def test_default_secondaries_are_the_three_baseline_configs():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'n1 = ["c3", "c4", "c5"] + (["c5conv", "d64", "d64s"] if args.secondary == "all" else [])' in src


def test_bench_knows_when_it_runs_under_a_profiler():
    """`rocprofv3 ... -- python3 bench.py` must not start profiler children of its own (the live HBM counters): the preloaded
    tool library and the ROCPROF_* variables of the outer run give it away"""
    import bench
    assert not bench.under_a_profiler({"PATH": "/usr/bin", "LD_PRELOAD": "/usr/lib/libjemalloc.so"})
    assert bench.under_a_profiler({"LD_PRELOAD": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so:/opt/rocm/lib/librocprofiler-sdk.so"})
    assert bench.under_a_profiler({"ROCPROF_OUTPUT_PATH": "/tmp/x"}) and bench.under_a_profiler({"ROCPROFILER_LIBRARY_CTOR": "1"})
