"""Per-kernel breakdown of the LAST evaluation in a rocprofv3 kernel-trace CSV, in launch order, grouped by stage.
usage: python tools/step_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import OrderedDict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
builds = [i for i, r in enumerate(rows) if 'build_kernel' in r[2]]
i0 = builds[-1]
agg = OrderedDict()
t0 = rows[i0][0]
tend = t0
for s, e, n in rows[i0:]:
    short = n.replace('(anonymous namespace)::', '').replace('void ', '')
    short = short[:short.find('(')] if '(' in short else short
    a = agg.setdefault(short, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    tend = max(tend, e)
for k, (c, d) in agg.items():
    print(f"{k:44s} n={c:4d} total {d:9.1f} us  avg {d / c:8.1f}")
print(f"span {(tend - t0) / 1e3:.1f} us, sum of kernels {sum(d for _, d in agg.values()):.1f} us")
