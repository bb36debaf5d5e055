#!/bin/bash
# Kernel timeline of one warm C4-map registration on one GPU (200 k reading against the 20 M-point map):
# bash tools/collect_c4_profile.sh  (through gpurun, from the repo root) -> gpurun_out/profiles_c4/timeline.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_c4
rm -rf $OUT && mkdir -p $OUT
SEED=1238 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/timeline -- python3 tools/tools_reg1.py 200000 20000000 4 > $OUT/timeline_run.log 2>&1 || exit 1
python3 tools/tools_timeline.py $OUT/timeline > $OUT/timeline.txt
tail -3 $OUT/timeline.txt
