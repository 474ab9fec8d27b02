#!/bin/bash
# sweep of the up-flux state share kept in the Infinity Cache (DESIGN_HISTORY.md, round 4)
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out/r04z
run() { # S MB workload steps extra
HELIOS_RT_SERPENTINE=$1 HELIOS_RT_STATE_CACHE_MB=$2 python3 bench.py --workload $3 --steps $4 --warmup 20 --no-cpu-baseline --secondary none --live-counters off $5 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']
print('%-8s S=$1 MB=%-4s'%('$3$5','$2'), 'ms/step %.4f'%l['ms_per_step'], {k:round(v,4) for k,v in r['kernels_ms'].items() if k in ('rt_flux','rt_coef','rt_nodes','rt_totals_a','rt_totals_b','refresh_total')})"
}
for i in 1 2; do
for cfg in 0:0 1:220 1:240 1:256 1:280; do run ${cfg%%:*} ${cfg##*:} c2 100; done
for cfg in 0:0 1:240; do run ${cfg%%:*} ${cfg##*:} c2 40 "--columns-per-gpu 4"; done
for cfg in 0:0 1:240; do run ${cfg%%:*} ${cfg##*:} c1 500; done
for cfg in 0:0 1:240; do run ${cfg%%:*} ${cfg##*:} c5 20; done
for cfg in 0:0 1:240; do run ${cfg%%:*} ${cfg##*:} c3 40; done
done
