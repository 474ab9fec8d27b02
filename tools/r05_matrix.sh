#!/bin/bash
# the matrix method as two scans (k_rt_matrix_prep + k_rt_flux<.., true>): its tests, then the same-box A/B against the
# per-stage kernels of round 4 (HELIOS_RT_MATRIX=stage) at config 2's size
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
