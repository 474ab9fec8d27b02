#!/bin/bash
# config-5 shape, one line per setting of HELIOS_RT_COEF_TPB given as arguments (default: library default)
for tpb in "${@:-}"; do
${tpb:+env HELIOS_RT_COEF_TPB=$tpb} python3 bench.py --workload c5 --steps 40 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c5 tpb=${tpb:-default}', round(d['ms_per_step'],4), round(r['avg_launch_ms'],4), {k:round(v,4) for k,v in r['kernels_ms'].items()})"
done
