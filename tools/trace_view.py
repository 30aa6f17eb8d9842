"""Prints the kernels of the LAST evaluation in a rocprofv3 --kernel-trace CSV as a timeline (us), with the queue of each.

    python tools/trace_view.py <kernel_trace.csv> [first row] [rows]
"""
import csv
import re
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        r['n'] = re.sub(r'\(.*', '', n).replace('void ', '')
    rows.sort(key=lambda r: r['s'])
    return rows


def main():
    rows = load(sys.argv[1])
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
    starts = [i for i, r in enumerate(rows) if r['n'].startswith('build_kernel')]
    ev = rows[starts[-1]:]
    t0 = ev[0]['s']
    busy = {}
    for r in ev:
        busy.setdefault(r['n'], [0, 0.0])
        busy[r['n']][0] += 1
        busy[r['n']][1] += (r['e'] - r['s']) / 1e3
    for r in ev[first:first + count]:
        print('%8.1f %8.1f %7.1f q%s %-44s grid=%s' % ((r['s'] - t0) / 1e3, (r['e'] - t0) / 1e3, (r['e'] - r['s']) / 1e3,
                                                      r['Queue_Id'], r['n'][:44], r.get('Grid_Size_X')))
    print('--- last evaluation: %.1f us wall, per kernel name (launches, summed us):' % ((max(r['e'] for r in ev) - t0) / 1e3))
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1][1]):
        print('%-50s %5d %10.1f' % (k[:50], v[0], v[1]))


if __name__ == '__main__':
    main()
