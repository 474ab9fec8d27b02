#!/bin/bash
# Regenerates the measured evidence kept under profiles/ (run on the GPU box through gpurun; outputs land
# in gpurun_out/ and are copied into profiles/ by hand afterwards).  Usage: bash tools/profile_round.sh r01
export TMPDIR=/tmp
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O && cd $R
python3 bench.py > $O/c2_bench.json 2> $O/c2_bench.err
rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --profile-steps 0 > $O/prof.log 2>&1
python3 tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/c2_kernel_stats.txt 2>&1
rm -rf $O/prof
python3 bench.py --columns-per-gpu 4 --steps 50 --no-cpu-baseline > $O/c2_bench_4columns.json 2>> $O/c2_bench.err
python3 bench.py --columns-per-gpu 64 --steps 20 --warmup 10 --no-cpu-baseline > $O/c4_bench_64columns.json 2>> $O/c2_bench.err
python3 bench.py --workload c1 --steps 500 --no-cpu-baseline > $O/c1_bench.json 2>> $O/c2_bench.err
python3 bench.py --workload c3 --steps 20 --warmup 10 --no-cpu-baseline > $O/c3_bench.json 2>> $O/c2_bench.err
rocprofv3 --kernel-trace --stats -d $O/prof3 -o run -- python3 bench.py --workload c3 --steps 20 --warmup 10 --no-cpu-baseline --profile-steps 0 > $O/prof3.log 2>&1
python3 tools/rocpd_summary.py $(find $O/prof3 -name "*.db" | head -1) > $O/c3_kernel_stats.txt 2>&1
rm -rf $O/prof3
python3 bench.py --phase convection --steps 100 --no-cpu-baseline > $O/c2_bench_convection_loop.json 2>> $O/c2_bench.err
python3 bench.py --workload c5 --steps 30 --warmup 10 --no-cpu-baseline > $O/c5_bench.json 2>> $O/c2_bench.err
tail -c 600 $O/c2_bench.json
