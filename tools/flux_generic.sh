#!/bin/bash
# k_rt_flux with compile-time scans (default) against the generic runtime-k kernel, ONE box: tools/flux_generic.sh reps [K]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq ${1:-2}); do
  for gen in 0 1; do
    echo -n "k=${2:-16} generic=$gen: "; env HELIOS_RT_K=${2:-16} HELIOS_RT_GENERIC_SCANS=$gen python3 $R/tools/step_profile.py 2>&1 | tail -1 | cut -c1-140
  done
done
