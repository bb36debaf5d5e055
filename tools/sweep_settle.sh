#!/bin/bash
# bench figures against O3D_SETTLE (relative change of the trimmed limit below which the fused iterations start)
for v in 0.05 0.08 0.12 0.2; do
O3D_SETTLE=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('settle $v:', round(d['value']), round(d['ms_per_step'],4), 'stalls', d['band_stalls_last_step'], 'c2', round(d['c2']['value']), 'c4', round(d['c4_one_gpu']['value']))" || exit 1
done
