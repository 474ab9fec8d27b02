#!/bin/bash
# k_rt_flux variants on ONE box: tools/flux_k.sh reps "K[:PREFETCH]" ...   (HELIOS_RT_K, HELIOS_RT_PREFETCH)
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=${1:-2}; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    k=${v%%:*}; pf=${v#*:}; [ "$pf" = "$v" ] && pf=0
    echo -n "k=$k prefetch=$pf: "; HELIOS_RT_K=$k HELIOS_RT_PREFETCH=$pf python3 $R/tools/step_profile.py 2>&1 | tail -1 | cut -c1-140
  done
done
