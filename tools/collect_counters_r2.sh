#!/bin/bash
# Hardware-counter passes (counters + kernel trace only: no other trace domain) on one registration workload.
# Usage on the GPU box: bash tools/collect_counters_r2.sh 200000 5000000 ; summary: tools/summarise_counters.py <dir> <kernel...>
set -e
N=${1:-200000}; M=${2:-5000000}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/counters_r02_${N}_${M}
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
