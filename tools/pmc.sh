#!/bin/bash
# usage: tools/pmc.sh <case> <tag>   -> gpurun_out/pmc_<tag>_*.csv   (one rocprofv3 run per counter group)
set -e
CASE=$1; TAG=$2
export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$n -- python3 tools/run_case.py $CASE 9 > gpurun_out/pmc_${TAG}_$n.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
if [ -z "$SQONLY" ]; then
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
fi
find gpurun_out -name '*.db' -delete; find gpurun_out -name '*agent_info.csv' -delete
