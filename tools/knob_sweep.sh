#!/bin/bash
# tuning knobs that date from earlier kernels, re-measured on one box (round 5)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { # workload steps ENV...
local W=$1 S=$2; shift 2
env "$@" python3 bench.py --workload $W --steps $S --warmup 10 --no-cpu-baseline --secondary none --live-counters off 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']; k=r['kernels_ms']
print('%-9s %-28s ms/step %.4f  flux %.4f  coef %.3f  nodes %.4f totals %.4f %.4f' % ('$W', '$*', l['ms_per_step'], k.get('rt_flux', k.get('matrix_solve', 0)), k.get('rt_coef',0), k.get('rt_nodes',0), k.get('rt_totals_a',0), k.get('rt_totals_b',0)))"
}
for i in 1 2; do
  run c2 100 X=0
  run c2 100 HELIOS_RT_COEF_TPB=2
  run c2 100 HELIOS_RT_COEF_TPB=8
  run c2 100 HELIOS_RT_NCHUNK=104
  run c2 100 HELIOS_RT_NCHUNK=417
  run c2matrix 100 X=0
  run c2matrix 100 HELIOS_RT_MAXTHREADS=128
  run c2matrix 100 HELIOS_RT_MAXTHREADS=256
  run c2matrix 100 HELIOS_RT_K=32
  run c5premixed 30 X=0
  run c5premixed 30 HELIOS_RT_COEF_TPB=4
  run c5premixed 30 HELIOS_RT_COEF_TPB=2
done
