cd ${GRAFT_REPO_ROOT:-.}
run() { env "$@" python3 bench.py --workload c2matrix --steps 100 --warmup 20 --no-cpu-baseline --secondary none --live-counters off 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']; k=r['kernels_ms']
print('%-60s ms/step %.4f  solve %.4f  coef %.3f  checksum %.15g' % ('$*', l['ms_per_step'], k.get('matrix_solve',0), k.get('rt_coef',0), l['spectrum_checksum']))"; }
for i in 1 2 3; do
  run X=newton_rcp
  run HELIOS_HIP_LIB=$PWD/ab/matrix_div.so
  run X=newton_rcp HELIOS_RT_K=32
  run HELIOS_HIP_LIB=$PWD/ab/matrix_div.so HELIOS_RT_K=32
done
timeout 600 python3 -m pytest tests -m gpu -q -k "matrix" 2>&1 | grep -v "Energy budget\|^$" | tail -3
