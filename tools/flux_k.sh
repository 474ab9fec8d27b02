#!/bin/bash
# k_rt_flux tilings for the direct-beam configurations, same box: lanes per spectral point forced through HELIOS_RT_K
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { # workload K steps
HELIOS_RT_K=$2 python3 bench.py --workload $1 --steps $3 --warmup 10 --no-cpu-baseline --secondary none --live-counters off 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']
print('%-10s HELIOS_RT_K=%-4s ms/step %.4f  rt_flux %.4f ms  rt_coef %.3f  e-only %.4f' % ('$1', '$2' or 'auto', l['ms_per_step'], r['kernels_ms'].get('rt_flux',0), r['kernels_ms'].get('rt_coef',0), r['e_only_ms_per_iteration']))"
}
for i in 1 2; do
  for K in "" 16 32; do run c2beam "$K" 50; done
  for K in "" 32 64; do run c5 "$K" 20; done
  for K in "" 32 64; do run c5premixed "$K" 30; done
done
