#!/bin/bash
# the matrix method as three scans (k_rt_flux<.., true>): its tests, then the same-box A/B against the
# per-stage kernels of round 4 (HELIOS_RT_MATRIX=stage) at config 2's size, and the kernel's own line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O && cd $R
timeout 1200 python3 -m pytest tests -m gpu -q -k "matrix or fused or stage" > $O/pytest_matrix.log 2>&1; echo "pytest rc=$?" >> $O/pytest_matrix.log
grep -v "Energy budget\|^$" $O/pytest_matrix.log | tail -40
{
for pass in 1 2; do
  echo "# scans (default)"; python3 tools/time_matrix_method.py 10000 100 2>/dev/null | grep MATRIX_METHOD
  echo "# per-stage kernels (HELIOS_RT_MATRIX=stage)"; HELIOS_RT_MATRIX=stage python3 tools/time_matrix_method.py 10000 100 2>/dev/null | grep MATRIX_METHOD
done
} > $O/matrix_method_timing.txt 2>&1
cat $O/matrix_method_timing.txt
python3 bench.py --workload c2matrix --steps 100 --warmup 20 --no-cpu-baseline --secondary none > $O/c2matrix_bench.json 2> $O/c2matrix.err
python3 - <<PY
import json
l=json.loads([x for x in open("$O/c2matrix_bench.json") if x.startswith("{")][-1])
print("c2matrix", l["value"], l["ms_per_step"], l["roofline"])
PY
rocprofv3 --kernel-trace --stats -d $O/prof_c2matrix -o run -- python3 bench.py --workload c2matrix --steps 50 --warmup 10 --no-cpu-baseline --profile-steps 0 --secondary none --live-counters off > $O/prof_c2matrix.log 2>&1
python3 tools/rocpd_summary.py $(find $O/prof_c2matrix -name "*.db" | head -1) > $O/c2matrix_kernel_stats.txt 2>&1
rm -rf $O/prof_c2matrix
head -12 $O/c2matrix_kernel_stats.txt
# where a whole run's wall clock goes (config 2 and config 3 through helios.py to equilibrium)
python3 tools/whole_run_timeline.py --out $O/whole_run_timeline.json > $O/timeline.log 2>&1; tail -5 $O/timeline.log
