"""Summarise a rocprofv3 --pmc ... --kernel-trace CSV directory: per kernel name, summed counters and durations."""
import csv, collections, glob, os, sys
d = sys.argv[1]
cc = list(csv.DictReader(open(max(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime))))
kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(max(glob.glob(d + '/*/*kernel_trace.csv'), key=os.path.getmtime)))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
for r in cc:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:44]
    agg[name][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id'])
        k = kt[r['Dispatch_Id']]
        agg[name]['dur_ns'] += float(k['End_Timestamp']) - float(k['Start_Timestamp'])
        agg[name]['n'] += 1
for name, a in sorted(agg.items(), key=lambda x: -x[1]['dur_ns'])[:12]:
    extra = ' '.join('%s=%.4g' % (k, v) for k, v in a.items() if k not in ('dur_ns', 'n'))
    print('%-46s n=%5d dur_ms %8.3f  %s' % (name, a['n'], a['dur_ns'] / 1e6, extra))
