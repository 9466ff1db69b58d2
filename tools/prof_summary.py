"""Condenses a tools/profile_round.sh output directory into small text/JSON summaries."""
import collections, csv, glob, json, os, sys
O, R = sys.argv[1], sys.argv[2]
out = []
# --- kernel stats of the timed step alone (bench.py --no-extra)
for f in glob.glob(f'{O}/stats_step/**/*kernel_stats.csv', recursive=True):
    out.append(f'# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu --no-extra   (the default step only: every launch is the headline workload)')
    out.append(f'{"kernel":70s} {"calls":>6s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}')
    for r in csv.DictReader(open(f)):
        if 'mil::' not in r['Name']: continue
        out.append(f'{r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:10.2f} {float(r["MinNs"])/1e3:10.2f} {float(r["MaxNs"])/1e3:10.2f} {float(r["Percentage"]):6.2f}')
    try:
        b = json.loads(open(f'{O}/bench_step_under_rocprof.json').read().strip().splitlines()[-1])
        out.append(f'bench.py line of that run: ms_per_step {b["ms_per_step"]}, roofline.ms {b["roofline"]["ms"]} ({b["roofline"]["kernel"]}), value {b["value"]} {b["unit"]}')
    except Exception as e:
        out.append(f'(bench line of that run not readable: {e})')
    out.append('')
# --- kernel stats (rocprofv3 --kernel-trace --stats)
for f in glob.glob(f'{O}/stats/**/*kernel_stats.csv', recursive=True):
    out.append(f'# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu   (with the side measurements: one kernel name serves several workloads and the two-frames-in-flight leg overlaps launches; {os.path.basename(f)})')
    rows = list(csv.DictReader(open(f)))
    out.append(f'{"kernel":70s} {"calls":>6s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}')
    for r in rows:
        if 'mil::' not in r['Name'] and float(r['Percentage']) < 2.0: continue
        out.append(f'{r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:10.2f} {float(r["MinNs"])/1e3:10.2f} {float(r["MaxNs"])/1e3:10.2f} {float(r["Percentage"]):6.2f}')
# --- the same kernels per workload: one tools/run_case.py case per rocprofv3 run
out.append('')
out.append('# rocprofv3 --kernel-trace --stats -- python3 tools/run_case.py <case> 12   (one workload per run: 4K Scene A; headline / dual: 2 dir + PCF 2048^2 +')
out.append('# 256 point + ambient, hdr / hdr + sRGB8; l4096: the same with 4096 point lights; k1: 1 directional light; tonemap / fxaa / accumulate: those passes alone)')
out.append(f'{"case":12s} {"kernel":62s} {"calls":>6s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s}')
for f in sorted(glob.glob(f'{O}/case_*/**/*kernel_stats.csv', recursive=True)):
    case = f.split('/case_')[1].split('/')[0]
    for r in csv.DictReader(open(f)):
        if 'mil::' not in r['Name'] or int(r['Calls']) < 9: continue
        out.append(f'{case:12s} {r["Name"][:62]:62s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:10.2f} {float(r["MinNs"])/1e3:10.2f} {float(r["MaxNs"])/1e3:10.2f}')
# --- PMC
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{O}/pmc_*/**/*counter_collection.csv', recursive=True):
    case = f.split('/pmc_')[1].split('_')[0]
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if 'mil::' not in k: continue
        agg[(case, k)][row['Counter_Name']].append(float(row['Counter_Value']))
traffic = {}
out.append('')
out.append('# rocprofv3 --pmc passes (tools/run_case.py <case> 9), mean over dispatches 4..9')
for (case, k), cs in sorted(agg.items()):
    out.append(f'[{case}] {k}')
    m = {}
    for c, v in sorted(cs.items()):
        v = v[3:] if len(v) > 3 else v
        m[c] = sum(v) / len(v)
        out.append(f'   {c:24s} {m[c]:18.1f}')
    if 'SQ_ACTIVE_INST_VALU' in m and 'GRBM_GUI_ACTIVE' in m and 'SQ_INSTS_VALU' in m:
        # SQ counters are in quad-cycles summed over the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0
        out.append(f'   -> kernel {cyc / 2.4e3:.1f} us at 2.4 GHz; VALU pipe busy {m["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024) * 100:.0f} % of SIMD-cycles; '
                   f'{m["SQ_ACTIVE_INST_VALU"] * 4 / m["SQ_INSTS_VALU"]:.2f} cycles per VALU wave-instruction; '
                   f'{m["SQ_INSTS_VALU"] / 1e6:.1f} M VALU wave-instructions')
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KiB) reads exactly half of a coalesced stream's bytes on
        # gfx950 -> x2; WRITE_SIZE (KiB) is exact for 16-B-per-lane streaming stores.
        rd, wr = 2 * m['FETCH_SIZE'] * 1024, m['WRITE_SIZE'] * 1024
        out.append(f'   -> HBM bytes/launch: read {rd/1e6:.1f} MB (2 x FETCH_SIZE), write {wr/1e6:.1f} MB, total {(rd+wr)/1e6:.1f} MB')
        kn = 'k_tiled' if 'k_tiled' in k else ('k_stream' if 'k_stream' in k else ('k_shade' if 'k_shade' in k else k.split('::')[-1]))
        wl = {'headline': 'headline', 'dual': 'headline_dual', 'l4096': 'valu_4k_4096', 'k1': 'k1_1light', 'accumulate': 'accumulate_4k_layer'}[case]
        traffic[f'{wl}:{kn}'] = dict(hbm_bytes_per_launch=int(rd + wr), fetch_size_kib=m['FETCH_SIZE'], write_size_kib=m['WRITE_SIZE'],
                                     note='read = 2 x FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md HBM), write = WRITE_SIZE KiB')
open(f'{O}/summary_{R}.txt', 'w').write('\n'.join(out) + '\n')
json.dump(traffic, open(f'{O}/pmc_traffic.json', 'w'), indent=1)
print('\n'.join(out))
