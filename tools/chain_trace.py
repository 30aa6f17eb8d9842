"""Prints the durations (us) and grid sizes of the Cholesky chain launches of the LAST evaluation in a rocprofv3
kernel-trace CSV: usage  python tools/chain_trace.py <kernel_trace.csv> [first_panel] [n_panels]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', ''))))
rows.sort()
zs = [i for i, r in enumerate(rows) if 'zero_stats' in r[2]]
i0 = zs[-1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
npan = int(sys.argv[3]) if len(sys.argv) > 3 else 3
pan = -1
line = []
for s, e, n, g in rows[i0 + 1:]:
    if 'tile_gemm<' in n and ', 2,' in n.split('(')[0]:
        break
    if 'leaf' in n:
        pan += 1
        if line and first <= pan - 1 < first + npan:
            print(' '.join(line))
        line = []
    nm = ('step' if 'chain_step' in n else 'leaffill' if 'leaf_fill' in n else 'leaf' if 'leaf_k' in n
          else 'rect' if 'syrk_rect' in n else n[n.find('tile_gemm'):n.find('>(') + 1].replace('tile_gemm', 'g'))
    line.append(f"{nm}:{(e - s) / 1e3:.1f}({int(g) // 256})")
t0 = rows[i0][0]
print('potrf span (ms):', (max(e for s, e, n, g in rows[i0:] if 'leaf' in n or 'chain' in n) - t0) / 1e6)
