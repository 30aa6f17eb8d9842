import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
n=float(sys.argv[2]) if len(sys.argv)>2 else 1
for x in csv.DictReader(open(f)):
    print('%-60s calls %6s total_ms/eval %8.3f avg_us %9.2f pct %6s' % (x['Name'].replace('(anonymous namespace)::','')[:60], x['Calls'], float(x['TotalDurationNs'])/1e6/n, float(x['AverageNs'])/1e3, x['Percentage']))
