#!/bin/bash
# round 5, first GPU call: the GPU suite, the mixing kernel's occupancy sweep, the Planck table and coefficient kernels'
# new timings, the driver's line with the new secondaries.  Outputs under gpurun_out/r05a/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O && cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
# occupancy sweep of k_rt_mix_species at config 3: unused dynamic LDS leaves 16 / 12 / 10 / 8 / 6 / 4 wavefronts per CU
{
echo "# k_rt_mix_species at config 3 (2.01 M points x 20 absorbers), ms per launch against wavefronts per CU"
echo "# (HELIOS_RT_MIX_EXTRA_LDS: unused dynamic LDS on top of the kernel's 10 192 B; allocation granule 1280 B; two passes, alternating)"
for pass in 1 2; do
for cfg in 16:0 12:2608 10:5168 8:10288 6:15408 4:25648; do
  W=${cfg%%:*}; X=${cfg##*:}
  HELIOS_RT_MIX_EXTRA_LDS=$X python3 bench.py --workload c3 --steps 10 --warmup 10 --no-cpu-baseline --secondary none --live-counters off 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']
print('waves/CU %2d  waves/SIMD %.2f  extra LDS %6d B:  k_rt_mix_species %.3f ms   (ms/step %.3f)' % ($W, $W/4.0, $X, r['kernels_ms']['add_to_mixed_opac'], l['ms_per_step']))"
done
done
} > $O/mix_occupancy.txt 2>&1
cat $O/mix_occupancy.txt
stats() {   # stats <workload> <bench args...>
  local W=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o run -- python3 bench.py --workload $W "$@" --no-cpu-baseline --profile-steps 0 --secondary none --live-counters off > $O/prof_$W.log 2>&1
  python3 tools/rocpd_summary.py $(find $O/prof_$W -name "*.db" | head -1) > $O/${W}_kernel_stats.txt 2>&1
  rm -rf $O/prof_$W
}
stats c2 --steps 50 --warmup 10
stats c5 --steps 20 --warmup 10
grep -E "plancktable|k_rt_coef|k_rt_table_outer" $O/c2_kernel_stats.txt $O/c5_kernel_stats.txt
( time python3 bench.py > $O/bench_n1.json 2> $O/bench.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_n1.json") if x.startswith("{")][-1])
print("headline", l["value"], l["ms_per_step"], "setup_s", l.get("setup_s"))
for k,v in (l.get("secondary") or {}).items():
    print(k, v.get("error") or (v["value"], v["ms_per_step"], "setup_s %.1f" % v["setup_s"], (v.get("roofline") or {}).get("kernels_ms")))
PY
