#!/bin/bash
# k_rt_flux under the profiling knobs, all on ONE box: tools/flux_knobs.sh reps "skip[:nsweep]" ...
# HELIOS_RT_DEBUG_SKIP bits: 1 no quadrature, 2 no U-tile stores, 4 stores ahead of the quadrature (the round-1 order)
REPS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $REPS); do
  for v in "$@"; do
    skip=${v%%:*}; ns=${v#*:}; [ "$ns" = "$v" ] && ns=""
    echo -n "skip=$skip nsweep=${ns:-4}: "
    env HELIOS_RT_DEBUG_SKIP=$skip ${ns:+HELIOS_RT_DEBUG_NSWEEP=$ns} python3 $R/tools/step_profile.py 2>&1 | tail -1 | cut -c1-140
  done
done
