#!/bin/bash
# A variant build of libhelios_hip.so for same-box A/B runs: tools/build_variant.sh NAME [extra compiler flags ...]
# -> ab/NAME.so (objects under ab/obj_NAME/; the in-tree library and its objects are not touched).  Select it at run time
# with HELIOS_HIP_LIB=ab/NAME.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
O=$R/ab/obj_$NAME
mkdir -p $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $*"
pids=()
for f in context stage_interp stage_trans stage_flux stage_matrix stage_mixing stage_post rt_fused; do
  /opt/rocm/bin/hipcc $FLAGS -c $R/helios_amd/csrc/$f.hip -o $O/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $R/ab/$NAME.so $O/*.o
echo "built ab/$NAME.so"
