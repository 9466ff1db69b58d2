"""Sums rocprofv3 --pmc csv output per kernel name. usage: pmc_kernels.py <dir> -> prints counters per mil kernel (per launch averages)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mil::' not in k: continue
        k = k.split('(')[0][-40:]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); calls[k].add(r['Dispatch_Id'])
for k in acc:
    n = len(calls[k])
    print(k, 'launches', n, ' '.join(f'{c}={v / n:.4g}' for c, v in sorted(acc[k].items())))
