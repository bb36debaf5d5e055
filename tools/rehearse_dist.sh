#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a one-GPU box: (1) one rank on a real RCCL communicator, (2) two ranks sharing the
# GPU over the gloo host-staged transport (RCCL refuses two ranks on one device).  usage: bash tools/rehearse_dist.sh
set -o pipefail
mkdir -p gpurun_out
show() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), 'iter/s', round(d['ms_per_step'],3), 'ms', d['config']['dist_loop'])" $1; }
O3D_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/dist1.json 2> gpurun_out/dist1.err || exit 1
show gpurun_out/dist1.json
O3D_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/dist2.json 2> gpurun_out/dist2.err || exit 1
show gpurun_out/dist2.json
