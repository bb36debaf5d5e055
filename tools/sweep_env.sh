for cfg in "2 0.05" "3 0.05" "4 0.05" "2 0.1" "2 0.2" "3 0.1" "3 0.2" "1 0.05"; do
  set -- $cfg
  O3D_KAHEAD=$1 O3D_SETTLE=$2 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('kahead $1 settle $2:', round(d['value']), round(d['ms_per_step'],4))
" || exit 1
done
