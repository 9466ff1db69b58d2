#!/bin/bash
# usage: tools/pmc_traffic_case.sh <tag> <case> [case...]: FETCH_SIZE and WRITE_SIZE passes (one rocprofv3 run each) per run_case case
export TMPDIR=/tmp
TAG=$1; shift
: > gpurun_out/traffic_$TAG.txt
for CASE in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmctd
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmctd -- python3 tools/run_case.py $CASE 9 > gpurun_out/pmctd.log 2>&1 || exit 1
    echo "== $CASE $C" >> gpurun_out/traffic_$TAG.txt
    python3 tools/pmc_kernels.py gpurun_out/pmctd >> gpurun_out/traffic_$TAG.txt
  done
done
rm -rf gpurun_out/pmctd
cat gpurun_out/traffic_$TAG.txt
