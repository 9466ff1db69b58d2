#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <case> [case...]: one SQ counter pass per run_case case -> gpurun_out/pmcq_<tag>.txt
export TMPDIR=/tmp
TAG=$1; shift
: > gpurun_out/pmcq_$TAG.txt
for CASE in "$@"; do
  rm -rf gpurun_out/pmcqd
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmcqd -- python3 tools/run_case.py $CASE 9 > gpurun_out/pmcqd.log 2>&1 || exit 1
  echo "== $CASE" >> gpurun_out/pmcq_$TAG.txt
  python3 tools/pmc_kernels.py gpurun_out/pmcqd >> gpurun_out/pmcq_$TAG.txt
done
rm -rf gpurun_out/pmcqd
cat gpurun_out/pmcq_$TAG.txt
