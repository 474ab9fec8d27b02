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
