#!/bin/bash
# usage: tools/prof_script.sh <tag> <python script> [args...] -> gpurun_out/kt_<tag>.txt : per-kernel stats under rocprofv3
export TMPDIR=/tmp
TAG=$1; shift
rm -rf gpurun_out/kt_$TAG
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$TAG -- python3 "$@" > gpurun_out/kt_$TAG.log 2>&1
f=$(find gpurun_out/kt_$TAG -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' > gpurun_out/kt_$TAG.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in [r for r in rows if 'mil::' in r['Name']]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f}")
PY
rm -rf gpurun_out/kt_$TAG
