"""Summarises rocprofv3 --pmc CSV output: mean counter value per kernel (skipping the first 3 dispatches)."""
import csv, glob, os, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_{tag}_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if 'mil' not in k: continue
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[3:] if len(v) > 3 else v
        print(f'   {c:28s} {sum(v)/len(v):18.1f}  (n={len(v)})')
