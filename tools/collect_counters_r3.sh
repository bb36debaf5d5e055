#!/bin/bash
# Round 3: hardware-counter passes (counters + kernel trace only) on one registration workload, summarised PER DISPATCH for the
# search kernel (iteration 0, 1, 2, ... of the last registration) and as means for the persistent tail kernel.
# Usage on the GPU box: bash tools/collect_counters_r3.sh 200000 5000000
set -e
N=${1:-200000}; M=${2:-5000000}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/counters_r03_${N}_${M}
rm -rf $OUT; mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
export SEED=1237
P() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/tools_reg1.py $N $M 3 > $OUT/$name.log 2>&1; echo "pass $name done"; }
P sq_time SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS &&
P sq_insts SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE &&
P tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum &&
P fetch FETCH_SIZE &&
P write WRITE_SIZE
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/tools_reg1.py $N $M 3 > $OUT/trace.log 2>&1
echo "trace done"
python3 tools/summarise_dispatches.py $OUT > $OUT/summary.txt
