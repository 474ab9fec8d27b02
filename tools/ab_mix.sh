#!/bin/bash
# same-box A/B of the mixing kernel: tools/ab_mix.sh WORKLOAD REPS lib1.so lib2.so ...  (alternating; ms per launch of k_rt_mix_species)
W=$1; REPS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $REPS); do
  for lib in "$@"; do
    echo -n "$(basename $lib) $W: "
    HELIOS_HIP_LIB=$(readlink -f $lib) python3 $R/bench.py --workload $W --steps 10 --warmup 10 --no-cpu-baseline --secondary none --live-counters off --full-line 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']
print('mix %.3f ms  step %.3f ms  refresh %.3f ms' % (k['add_to_mixed_opac'], d['ms_per_step'], k['refresh_total']))"
  done
done
