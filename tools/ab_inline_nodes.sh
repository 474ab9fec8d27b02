#!/bin/bash
# k_rt_nodes' work in k_rt_flux's prologue (measurement build ab/inline_nodes.so: make INLINE_NODES=1) against the product's
# four kernels per iteration, alternating on one box, config 2: E-only ms per iteration (nine refresh-free iterations)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { # label lib inline
HELIOS_HIP_LIB=$2 HELIOS_RT_INLINE_NODES=$3 python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --secondary none --live-counters off 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']
print('%-28s ms/step %.4f  E-only %.4f ms  ' % ('$1', l['ms_per_step'], r['e_only_ms_per_iteration']), {k:round(v,4) for k,v in r['kernels_ms'].items() if k in ('rt_flux','rt_nodes','rt_totals_a','rt_totals_b')}, 'checksum %.17g' % l['spectrum_checksum'])"
}
for i in 1 2 3; do
  run "product (4 kernels)" $R/helios_amd/libhelios_hip.so 0
  run "variant, switch off" $R/ab/inline_nodes.so 0
  run "variant, nodes in prologue" $R/ab/inline_nodes.so 1
done
