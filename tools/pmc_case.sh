#!/bin/bash
# usage: tools/pmc_case.sh <case> <tag> [NPT]: SQ counters per mil kernel -> gpurun_out/pmc_<tag>.txt
export TMPDIR=/tmp
CASE=$1; TAG=$2; export NPT=${3:-256}
rm -rf gpurun_out/pmcd_$TAG; : > gpurun_out/pmc_$TAG.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcd_$TAG/$i -- python3 tools/run_case.py $CASE 9 > gpurun_out/pmcd_$TAG.log 2>&1
  python3 tools/pmc_kernels.py gpurun_out/pmcd_$TAG/$i >> gpurun_out/pmc_$TAG.txt
done
rm -rf gpurun_out/pmcd_$TAG
