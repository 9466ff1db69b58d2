#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench command, then the
# HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, one rocprofv3 run each) for the two kernels
# bench.py prices.  Summaries land in gpurun_out/; copy the ones to keep into profiles/.
set -e
export TMPDIR=/tmp
R=${1:-r01}
O=gpurun_out/prof_$R
mkdir -p $O
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu > $O/bench_under_rocprof.json 2> $O/stats.err
# the same command without the side measurements: every launch of its kernel is the timed step's workload, so the kernel's
# average here is the one bench.py's roofline leg must agree with
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -- python3 bench.py --steps 30 --warmup 5 --no-cpu --no-extra > $O/bench_step_under_rocprof.json 2> $O/stats_step.err
for c in FETCH_SIZE WRITE_SIZE; do
  for k in headline dual l4096 k1 accumulate; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${k}_$c -- python3 tools/run_case.py $k 9 > $O/pmc_${k}_$c.log 2>&1
  done
done
for k in headline dual l4096 k1; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc_${k}_SQ -- python3 tools/run_case.py $k 9 > $O/pmc_${k}_SQ.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${k}_GRBM -- python3 tools/run_case.py $k 9 > $O/pmc_${k}_GRBM.log 2>&1
done
# per-workload kernel times: bench.py runs many workloads in one process and k_tiled / k_shade serve several of them, so its
# kernel_stats cannot be read per workload; every run_case case alone under --kernel-trace --stats
for k in headline dual l4096 k1 tonemap fxaa accumulate; do
  NPT=256; [ $k = l4096 ] && NPT=4096
  NPT=$NPT timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $O/case_$k -- python3 tools/run_case.py $k 12 > $O/case_$k.log 2>&1
done
find $O -name '*.db' -delete
python3 tools/prof_summary.py $O $R
# keep only the condensed artefacts (gpurun copies back at most 64 MiB)
for f in $(find $O/stats -name '*kernel_stats.csv'); do cp $f $O/kernel_stats_$R.csv; done
rm -rf $O/stats $O/stats_step $O/pmc_*/ $O/case_*/ $O/*.log
