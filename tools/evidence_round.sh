#!/bin/bash
# Whole-run parity evidence at size (run on the GPU box through gpurun; JSONs land in gpurun_out/<tag>/ and are copied into
# profiles/ afterwards).  Usage: bash tools/evidence_round.sh r04 [a|b]     a: config 2, config 5's physics at 6000 x 100
# (~17 min);  b: config 5 to equilibrium on the library, config 1, config 3, radiation + convection loop at 2000 x 100 (~60 min)
TAG=${1:-r04}; PART=${2:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O && cd $R
run() { local name=$1; shift; timeout $1 python3 "${@:2}" --out $O/$name.json > $O/$name.log 2>&1; echo "$name rc=$?"; tail -c 300 $O/$name.log; echo; }
if [ "$PART" = a ]; then
  run ${TAG}_trajectory_c2 1500 tests/loop_to_convergence_on_gpu.py --workload c2
  run ${TAG}_trajectory_c5flags_6000x100 1200 tests/loop_to_convergence_on_gpu.py --workload c5premixed --nbin 6000 --nlayer 100
else
  run ${TAG}_c5_equilibrium 2400 tests/c5_equilibrium_on_gpu.py --T-intern 1000
  run ${TAG}_trajectory_c1 600 tests/loop_to_convergence_on_gpu.py --workload c1
  run ${TAG}_trajectory_c3 2400 tests/loop_to_convergence_on_gpu.py --workload c3
  run ${TAG}_trajectory_radconv_2000x100 2400 tests/loop_to_convergence_on_gpu.py --workload c2 --nbin 2000 --convection --T-intern 1000 --max-iterations 40000
fi
