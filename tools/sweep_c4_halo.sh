#!/bin/bash
# C4-map figures (one GPU) against the halo knobs, the halo-bin edge follows the map density
for cfg in "1.5 0.4" "1.2 0.4" "1.8 0.4" "1.5 0.3" "1.5 0.5"; do
  set -- $cfg
  O3D_HALO_RATIO=$1 O3D_HALO_RHO=$2 timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('halo ratio $1 rho $2:', 'c3', round(d['value']), 'c2', round(d['c2']['value']), 'c4', round(d['c4_one_gpu']['value']), 'slice', round(d['c4_one_gpu']['one_rank_slice']['iter_per_s']), 'table MB', round(d['c4_one_gpu']['table_MB']))" || exit 1
done
