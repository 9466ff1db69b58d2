"""Reads a rocprofv3 --kernel-trace csv (kernel_trace.csv) of bench.py and prints, for the timed steps, how the lighting
launch of step i + 1 and the tonemap of step i lie against each other.  usage: overlap_trace.py <dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if 'mil::' in r['Kernel_Name']), key=lambda e: e[0])
lit = [e for e in ev if 'k_tiled' in e[2] or 'k_shade' in e[2]]
tm = [e for e in ev if 'k_tonemap' in e[2]]
print('lighting launches', len(lit), 'tonemap launches', len(tm))
# steady-state window: the middle third of the lighting launches
a, b = len(lit) // 3, 2 * len(lit) // 3
win = lit[a:b]
span = (win[-1][1] - win[0][0]) / 1e3
print(f'window of {len(win)} lighting launches: {span / len(win):.1f} us per step; lighting avg {sum(e[1] - e[0] for e in win) / len(win) / 1e3:.1f} us; gaps between lighting launches avg {sum(win[i + 1][0] - win[i][1] for i in range(len(win) - 1)) / (len(win) - 1) / 1e3:.1f} us')
t0, t1 = win[0][0], win[-1][1]
tw = [e for e in tm if e[0] >= t0 and e[1] <= t1]
if tw:
    ov = 0
    for s, e, _ in tw:
        for ls, le, _ in win:
            ov += max(0, min(e, le) - max(s, ls))
    print(f'tonemap avg {sum(e[1] - e[0] for e in tw) / len(tw) / 1e3:.1f} us, of which inside a lighting launch {ov / len(tw) / 1e3:.1f} us')
