export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof && cd $R
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o run -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/bench_prof.log 2>&1
python3 tools/rocpd_summary.py $(find $R/gpurun_out/prof -name "*.db" | head -1) > $R/gpurun_out/kernel_stats.txt 2>&1
head -20 $R/gpurun_out/kernel_stats.txt
python3 bench.py --no-cpu-baseline 2>&1 | tail -1
