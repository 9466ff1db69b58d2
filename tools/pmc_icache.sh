#!/bin/bash
# usage: tools/pmc_icache.sh <tag> <case> [case...]: instruction-cache and scalar-cache counters per run_case case
export TMPDIR=/tmp
TAG=$1; shift
: > gpurun_out/pmci_$TAG.txt
for CASE in "$@"; do
  for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    rm -rf gpurun_out/pmcid
    timeout -s KILL 120 rocprofv3 --pmc $SET --output-format csv -d gpurun_out/pmcid -- python3 tools/run_case.py $CASE 9 > gpurun_out/pmcid.log 2>&1 || { tail -3 gpurun_out/pmcid.log; continue; }
    echo "== $CASE" >> gpurun_out/pmci_$TAG.txt
    python3 tools/pmc_kernels.py gpurun_out/pmcid >> gpurun_out/pmci_$TAG.txt
  done
done
rm -rf gpurun_out/pmcid
cat gpurun_out/pmci_$TAG.txt
