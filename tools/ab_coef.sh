#!/bin/bash
# (round 4 experiment, kept for the record: ab/contract.so was the library with `#pragma clang fp contract(fast)` inside slab_coeffs*,
#  E_factor and the row loop of k_rt_coef -- the HX_COEF_CONTRACT macro of commit "Random overlap: the abscissae ... contraction measured
#  and not kept"; the macro is no longer in the tree)
# coefficient kernel with / without FMA contraction, alternating on one box:  scratch/ab_coef.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp helios_amd/libhelios_hip.so /tmp/_orig.so
for i in 1 2; do
  for lib in new contract; do
    cp ab/$lib.so helios_amd/libhelios_hip.so
    for W in c2 c5; do
      python3 bench.py --workload $W --steps 20 --warmup 10 --no-cpu-baseline --secondary none 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']
print('$lib $W: ms/step %.4f  refresh %.3f ms  rt_coef %.4f ms  e-only %.4f' % (l['ms_per_step'], r['t_only_ms_per_refresh'], r['kernels_ms'].get('rt_coef',0), r['e_only_ms_per_iteration']))"
    done
  done
done
# which parity tests hold with the contracted coefficient kernel
cp ab/contract.so helios_amd/libhelios_hip.so
timeout 1500 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_reference.py tests/test_gpu_fullsize.py tests/test_gpu_onthefly.py -q -m gpu 2>&1 | tail -25
cp /tmp/_orig.so helios_amd/libhelios_hip.so
