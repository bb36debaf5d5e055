#!/bin/bash
# Collects the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench command (C3 headline, no secondary objects), FETCH_SIZE / WRITE_SIZE PMC passes
#   (separate runs, counters + kernel trace only, as the MI355X guide prescribes), and the FETCH_SIZE calibration.
# usage: bash tools/collect_profiles.sh [r03]
set -o pipefail
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT && mkdir -p $OUT
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/bench_write.log 2>&1 || exit 1
[ -x tools/tools_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/tools_calib tools/tools_calib.hip
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -- ./tools/tools_calib > $OUT/calib.log 2>&1 || exit 1
# timeline of one warm C3 registration (gaps between kernels)
SEED=1237 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/timeline -- python3 tools/tools_reg1.py 200000 5000000 4 > $OUT/timeline_run.log 2>&1 || exit 1
python3 tools/tools_timeline.py $OUT/timeline > $OUT/timeline.txt
# the plain bench line of this build (all secondary objects, CPU baseline)
timeout -k 10 600 python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err || exit 1
python3 tools/summarise_profiles.py $OUT $TAG c3
