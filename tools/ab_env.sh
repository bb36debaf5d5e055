#!/bin/bash
# A/B of one library environment switch on the bench workloads: bash tools/ab_env.sh O3D_NO_DYNPRUNE  (GPU box)
V=$1; VAL=${2:-1}
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', round(d['value']), round(d['ms_per_step'],4), 'c2', round(d['c2']['value']), 'c4', round(d['c4_one_gpu']['value']), 'match', round(d['kernels']['k_match_g8']['avg_ms']*1e3,1), 'pair', round(d['kernels']['fused_pair']['avg_ms']*1e3,1))"; }
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | show x "default   "  || exit 1
  env $V=$VAL timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | show x "$V=$VAL" || exit 1
done
