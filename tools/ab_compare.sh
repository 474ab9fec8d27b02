#!/bin/bash
# A/B/... of several builds of libhelios_hip.so on ONE box (box-to-box spread is +-3 %):
#   tools/ab_compare.sh reps lib1.so lib2.so ...
# Alternates the libraries `reps` times and prints the fused-step profile of each run.
REPS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/helios_amd/libhelios_hip.so /tmp/_orig.so
for i in $(seq $REPS); do
  for lib in "$@"; do
    cp $lib $R/helios_amd/libhelios_hip.so
    echo -n "$(basename $lib): "; python3 $R/tools/step_profile.py 2>&1 | tail -1 | cut -c1-110
  done
done
cp /tmp/_orig.so $R/helios_amd/libhelios_hip.so
