#!/bin/bash
# Shader-side PMC evidence for the dominant kernels (instruction mix, issue/wait cycles, LDS conflicts, occupancy).
# One rocprofv3 pass per counter group, --kernel-trace only (gpurun refuses other trace domains with --pmc).
# Run through gpurun; the per-kernel table lands in gpurun_out/<tag>/pmc_sq.txt.
export TMPDIR=/tmp
TAG=${1:-r01}
WL=${2:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O && cd $R
GROUPS_=(
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
  "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64"
  "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32"
  "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
  "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_CYCLES SQ_THREAD_CYCLES_VALU"
)
DBS=()
i=0
for G in "${GROUPS_[@]}"; do
  rocprofv3 --pmc $G --kernel-trace -d $O/sq_$i -o $WL -- python3 bench.py --workload $WL --steps 10 --warmup 5 --no-cpu-baseline --profile-steps 0 --secondary none --live-counters off > $O/sq_$i.log 2>&1
  DB=$(find $O/sq_$i -name "*.db" | head -1)
  [ -n "$DB" ] && DBS+=($DB) || { echo "group $i failed: $G"; tail -5 $O/sq_$i.log; }
  i=$((i+1))
done
python3 tools/pmc_sq.py "${DBS[@]}" > $O/pmc_sq_$WL.txt 2>&1
for j in $(seq 0 $((i-1))); do rm -rf $O/sq_$j; done
cat $O/pmc_sq_$WL.txt
