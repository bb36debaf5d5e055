#!/bin/bash
# Hardware-counter passes for the fused iteration kernel (one registration workload, counters only: no other trace domain).
# Usage on the GPU box: bash tools/collect_counters.sh ; then python tools/summarise_counters.py gpurun_out/counters_r01
set -e
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/counters_r01
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
P() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/tools_reg1.py 100000 1000000 3 > $OUT/$name.log 2>&1; echo "pass $name done"; }
P sq_time SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY &&
P sq_insts SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
# A TA/TCP pass (TA_BUSY, TCP_PENDING_STALL_CYCLES, TCP_TCC_READ_REQ_LATENCY ...) never returned on this pool (killed
# after 7 silent minutes): not part of the default collection.
