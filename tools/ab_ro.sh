#!/bin/bash
# ro_bench through several builds of the library, alternating on one box:  bash tools/ab_ro.sh "libs" "kinds"
# (variants are ab/<name>.so, selected through HELIOS_HIP_LIB: the in-tree library is never touched)
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=${1:-"head new"}; KINDS=${2:-"generic ktable"}
for i in 1 2; do
  for lib in $LIBS; do
    for kind in $KINDS; do
      echo -n "$lib: "; HELIOS_HIP_LIB=$R/ab/$lib.so python3 $R/tools/ro_bench.py --kind $kind 2>&1 | tail -1 | cut -c1-120
    done
  done
done
