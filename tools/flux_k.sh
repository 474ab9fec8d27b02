#!/bin/bash
# k_rt_flux with k = 16 / 32 lanes per spectral point on ONE box: tools/flux_k.sh reps
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq ${1:-2}); do
  for k in 16 32; do
    echo -n "k=$k: "; HELIOS_RT_K=$k python3 $R/tools/step_profile.py 2>&1 | tail -1 | cut -c1-140
  done
done
