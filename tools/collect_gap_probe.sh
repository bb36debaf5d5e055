#!/bin/bash
# Where does the ~6 us gap in front of the select-based iterations' search kernel come from?  Timeline of a select-based-only
# registration (fused path off) with everything submitted ahead (O3D_KAHEAD=40) against the default lookahead of 2.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gap_probe
rm -rf $OUT && mkdir -p $OUT
for k in ${KS:-2 40}; do
  DBGF=${DBGF:-1} O3D_KAHEAD=$k SEED=1237 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/k$k -- python3 tools/tools_reg1.py 200000 5000000 3 > $OUT/run_k$k.log 2>&1 || exit 1
  python3 tools/tools_timeline.py $OUT/k$k > $OUT/timeline_k$k.txt
  echo "lookahead $k: gaps in front of k_match_g8:"; grep k_match_g8 $OUT/timeline_k$k.txt | awk '{printf "%s ", $5}'; echo; tail -1 $OUT/timeline_k$k.txt
done
