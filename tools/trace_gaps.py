"""Inter-kernel gaps of one stream from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv).

usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]
Prints per kernel name: count, mean duration, mean gap to the PREVIOUS kernel's end (start_i - end_{i-1}).
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
agg = defaultdict(lambda: [0, 0, 0, 0])
tot_gap = tot_dur = 0
for (s0, e0, _), (s1, e1, name) in zip(rows[:-1], rows[1:]):
    gap = s1 - e0
    short = name.split("(")[0][-60:]
    if "tile_gemm" in name or "leaf" in name or "syrk" in name:
        short = name[name.find("::", 10) + 2:name.find(">(") + 1][:60]
    a = agg[short]
    a[0] += 1
    a[1] += e1 - s1
    if gap < 200000:          # ignore step boundaries (host sync)
        a[2] += gap
        a[3] += 1
        tot_gap += gap
    tot_dur += e1 - s1
print(f"{'kernel':62s} {'n':>6s} {'dur_us':>9s} {'gap_us':>8s}")
for k, (n, d, g, ng) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:62s} {n:6d} {d / n / 1e3:9.2f} {g / max(ng, 1) / 1e3:8.2f}")
print(f"total kernel time {tot_dur / 1e6:.3f} ms, total gaps {tot_gap / 1e6:.3f} ms over {len(rows)} kernels")
