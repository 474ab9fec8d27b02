#!/bin/bash
# A/B/... of several builds of libhelios_hip.so on ONE box (box-to-box spread is +-3 %):
#   tools/ab_compare.sh reps lib1.so lib2.so ...
# Alternates the libraries `reps` times and prints the fused-step profile of each run.  The variant is selected through
# HELIOS_HIP_LIB (helios_amd/_lib.py): the in-tree library is never overwritten.
REPS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $REPS); do
  for lib in "$@"; do
    echo -n "$(basename $lib): "; HELIOS_HIP_LIB=$(readlink -f $lib) python3 $R/tools/step_profile.py 2>&1 | tail -1 | cut -c1-110
  done
done
