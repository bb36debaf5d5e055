#!/bin/bash
# Collects the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench command, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as the
#   MI355X guide prescribes), and the FETCH_SIZE calibration on known byte counts.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r01
rm -rf $OUT && mkdir -p $OUT
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/bench_write.log 2>&1
[ -x tools/tools_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/tools_calib tools/tools_calib.hip
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -- ./tools/tools_calib > $OUT/calib.log 2>&1
# next-row kernel (normals / covariances): kernel trace + stats of 1 M points, k = 10
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/normals -- python tools/tools_normals.py 1000000 10 > $OUT/normals.log 2>&1
# timeline of one warm registration (gaps between kernels)
rocprofv3 --kernel-trace --output-format csv -d $OUT/timeline -- python tools/tools_reg1.py 100000 1000000 4 > $OUT/timeline_run.log 2>&1
python tools/tools_timeline.py $OUT/timeline > $OUT/timeline.txt
# target-side preparation (crop + fp64->fp32 + build), 5 M points; and the plain bench line of this build
python tools/tools_target_prep.py 5000000 > $OUT/target_prep.log 2>&1
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
# duration of the two search kernels against the reading size (fixed cost of a launch vs cost per point)
python tools/tools_scaling.py > $OUT/scaling.txt 2>&1
python tools/summarise_profiles.py $OUT
