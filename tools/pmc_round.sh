#!/bin/bash
# HBM traffic per kernel launch from the PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes with
# --kernel-trace only (gpurun refuses other trace domains with --pmc).  Run through gpurun; results in gpurun_out/<tag>/.
export TMPDIR=/tmp
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O && cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o c2 -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --profile-steps 0 > $O/pmc_$C.log 2>&1
done
python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) c2 > $O/pmc_traffic.txt 2>&1
cp profiles/traffic.json $O/traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -30 $O/pmc_traffic.txt
