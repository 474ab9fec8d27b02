#!/bin/bash
# Regenerates the measured evidence kept under profiles/ (run on the GPU box through gpurun; outputs land in
# gpurun_out/<tag>/ and are copied into profiles/ afterwards).  Usage: bash tools/profile_round.sh r02
# rocprofv3: --kernel-trace [--stats] only, counters in their own passes, the program itself after `--`.
export TMPDIR=/tmp
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O && cd $R
stats() {   # stats <workload> <bench args...>
  local W=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o run -- python3 bench.py --workload $W "$@" --no-cpu-baseline --profile-steps 0 --secondary none --live-counters off > $O/prof_$W.log 2>&1
  python3 tools/rocpd_summary.py $(find $O/prof_$W -name "*.db" | head -1) > $O/${W}_kernel_stats.txt 2>&1
  rm -rf $O/prof_$W
}
traffic() { # traffic <workload> <bench args...>: FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU in separate passes
  local W=$1; shift
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    rocprofv3 --pmc $C --kernel-trace -d $O/pmc_${W}_$C -o run -- python3 bench.py --workload $W "$@" --no-cpu-baseline --profile-steps 0 --secondary none --live-counters off > $O/pmc_${W}_$C.log 2>&1
  done
  python3 tools/pmc_traffic.py $(find $O/pmc_${W}_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_${W}_WRITE_SIZE -name "*.db" | head -1) $W \
      $(find $O/pmc_${W}_SQ_INSTS_VALU -name "*.db" | head -1) > $O/${W}_pmc_traffic.txt 2>&1
  rm -rf $O/pmc_${W}_FETCH_SIZE $O/pmc_${W}_WRITE_SIZE $O/pmc_${W}_SQ_INSTS_VALU
}
# counters first: bench.py quotes profiles/traffic.json in its roofline object
traffic c2 --steps 20 --warmup 10
traffic c3 --steps 10 --warmup 10
traffic c5 --steps 10 --warmup 10
cp profiles/traffic.json $O/traffic.json
# the driver's line: config 2 as the headline (its HBM traffic counted live, in child passes under rocprofv3), config 3, config 4
# at 8 columns and config 5 (20 species on the fly + clouds + beam) live in `secondary`
# (exactly the driver's command: the compact line on stdout, the full record beside it)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_n1_detail.json > $O/bench_n1.json 2> $O/bench.err
stats c2 --steps 50 --warmup 10
python3 bench.py --workload c3 --live-counters all --full-line > $O/c3_bench.json 2>> $O/bench.err
stats c3 --steps 20 --warmup 10
stats c5 --steps 20 --warmup 10
# (config 4 at 64 columns per GPU, the default grid as a 64-column batch: live in bench_n1.json's `secondary` since round 5)
python3 bench.py --workload c2matrix --steps 100 --warmup 20 --no-cpu-baseline --secondary none --full-line > $O/c2matrix_bench.json 2>> $O/bench.err
stats c2matrix --steps 50 --warmup 10
{ for pass in 1 2; do
  echo "# three scans inside k_rt_flux<.., true> (default)"; python3 tools/time_matrix_method.py 10000 100 2>/dev/null | grep MATRIX_METHOD
  echo "# per-stage kernels (HELIOS_RT_MATRIX=stage)"; HELIOS_RT_MATRIX=stage python3 tools/time_matrix_method.py 10000 100 2>/dev/null | grep MATRIX_METHOD
done; } > $O/matrix_method_timing.txt 2>&1
python3 tools/whole_run_timeline.py --out $O/whole_run_timeline.json > $O/timeline.log 2>&1
python3 bench.py --workload c1 --steps 500 --no-cpu-baseline --full-line > $O/c1_bench.json 2>> $O/bench.err
python3 bench.py --columns-per-gpu 4 --steps 50 --no-cpu-baseline --secondary none --full-line > $O/c2_bench_4columns.json 2>> $O/bench.err
python3 bench.py --phase convection --steps 100 --no-cpu-baseline --full-line > $O/c2_bench_convection_loop.json 2>> $O/bench.err
python3 bench.py --workload c5 --steps 20 --warmup 10 --full-line > $O/c5_bench.json 2>> $O/bench.err
python3 bench.py --workload c5premixed --steps 30 --warmup 10 --no-cpu-baseline --full-line > $O/c5premixed_bench.json 2>> $O/bench.err
python3 bench.py --workload c2beam --steps 50 --warmup 10 --no-cpu-baseline --secondary none --full-line > $O/c2beam_bench.json 2>> $O/bench.err
python3 bench.py --phase convection --workload c5 --steps 20 --warmup 10 --no-cpu-baseline --secondary none --live-counters off --full-line > $O/c5_bench_convection_loop.json 2>> $O/bench.err
# eight ranks on the one GPU of this box through the gloo hook: the multi-rank path incl. the config-4 share as `secondary`
HELIOS_BENCH_BACKEND=gloo python3 bench.py --gpus 8 --workload c2small --steps 20 --warmup 10 --secondary c4small --no-cpu-baseline --detail $O/bench_8ranks_one_gpu_gloo_detail.json > $O/bench_8ranks_one_gpu_gloo.json 2>> $O/bench.err
for K in generic ktable dominated; do for S in lean q32 bitonic rank; do HELIOS_RO_SORT=$S python3 tools/ro_bench.py --kind $K --reps 3; done; done > $O/ro_bench.txt 2>&1
# shader counters of the mixing kernel: in the species loop of config 3 and on problems that all take the network
bash tools/pmc_sq.sh $TAG c3 > /dev/null 2>&1
bash tools/pmc_cmd.sh ${TAG}_ro python3 tools/ro_bench.py --kind ktable > /dev/null 2>&1
cp $R/gpurun_out/${TAG}_ro/pmc_sq.txt $O/ro_bench_pmc_sq.txt
tail -c 4000 $O/bench_n1.json
