export TMPDIR=/tmp
for c in headline hl_noshadow hl_nopoints hl_nodir; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_q_$c -- python3 tools/run_case.py $c 9 > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for c in ['headline','hl_noshadow','hl_nopoints','hl_nodir']:
    agg=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/pmc_q_{c}/**/*counter_collection.csv',recursive=True):
        for row in csv.DictReader(open(f)):
            if 'k_tiled' in row['Kernel_Name']: agg[row['Counter_Name']].append(float(row['Counter_Value']))
    print(c, {k: round(sum(v[3:])/len(v[3:])/1e6,2) for k,v in sorted(agg.items())})
PY
rm -rf gpurun_out/pmc_q_*
