#!/bin/bash
# ro_bench through several builds, alternating:  scratch/ab_ro.sh "libs" "kinds"
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=${1:-"head new"}; KINDS=${2:-"generic ktable"}
cp $R/helios_amd/libhelios_hip.so /tmp/_orig.so
for i in 1 2; do
  for lib in $LIBS; do
    cp $R/ab/$lib.so $R/helios_amd/libhelios_hip.so
    for kind in $KINDS; do
      echo -n "$lib: "; python3 $R/tools/ro_bench.py --kind $kind 2>&1 | tail -1 | cut -c1-120
    done
  done
done
cp /tmp/_orig.so $R/helios_amd/libhelios_hip.so
